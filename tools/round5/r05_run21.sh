# soak of the general-shape kernels on random shapes (odd strides, radix 3 / 5 lengths, tiny batches) against the oracle
set -x
mkdir -p gpurun_out/r05o
(time timeout 1200 python tests/generic_soak.py 40 400 24) > gpurun_out/r05o/soak.txt 2>&1
tail -12 gpurun_out/r05o/soak.txt
