# four-wave build of the general-shape kernel at deeper batches (no spills in the pair-load form) + larger soaks (many clips per configuration: a wave walks several chunks)
set -x
mkdir -p gpurun_out/r05p
(time timeout 1200 python tools/gpu_generic_rate.py 8192 "product,L4 shallow,L8 shallow,L4 u2f2e8,L8 u2f2e8,L4 u2f2e16,L8 u2f2e16,L4 u1f2e16,L8 u1f2e16") > gpurun_out/r05p/generic_rate.txt 2>&1
grep -v "tuned\|0\.4[0-9] ns" gpurun_out/r05p/generic_rate.txt
(time timeout 900 python tests/generic_soak.py 440 2000 24) > gpurun_out/r05p/soak_small.txt 2>&1
tail -3 gpurun_out/r05p/soak_small.txt
(time timeout 900 python tests/generic_soak.py 2440 60 3000) > gpurun_out/r05p/soak_large.txt 2>&1
tail -3 gpurun_out/r05p/soak_large.txt
