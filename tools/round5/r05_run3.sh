set -x
mkdir -p gpurun_out/r05c
timeout 600 python tools/ab_rate.py r4,dev 2 cfg5_dscnn_mfcc40_f32.kwsm > gpurun_out/r05c/ab_rate_cfg5.txt 2>&1
cat gpurun_out/r05c/ab_rate_cfg5.txt
(time timeout 1500 python -m pytest tests -m gpu -x -q) > gpurun_out/r05c/pytest_gpu.txt 2>&1
tail -15 gpurun_out/r05c/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05c/smoke.txt 2>&1
tail -6 gpurun_out/r05c/smoke.txt
