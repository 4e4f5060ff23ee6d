# deep soaks on the final library: every clip of 8 full batches per model against the oracle, exact mode and fast mode
set -x
mkdir -p gpurun_out/r05v
sha256sum ei-keyword-spotting_amd/libkws_mi355x.so > gpurun_out/r05v/lib_sha256.txt
(time timeout 1200 python tests/deep_soak.py 8 65536) > gpurun_out/r05v/deep_soak.txt 2>&1
tail -12 gpurun_out/r05v/deep_soak.txt
(time timeout 1200 python tests/deep_soak.py 8 65536 fast) > gpurun_out/r05v/fast_soak.txt 2>&1
tail -8 gpurun_out/r05v/fast_soak.txt
