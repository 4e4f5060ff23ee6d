# general-shape DSP kernel: pair loads one sub-batch ahead; tests, rates, phase clocks
set -x
mkdir -p gpurun_out/r05k
(time timeout 900 python -m pytest tests/test_gpu_generic_dsp.py -m gpu -q -x) > gpurun_out/r05k/pytest_generic.txt 2>&1
tail -4 gpurun_out/r05k/pytest_generic.txt
(time timeout 1200 python tools/gpu_generic_rate.py 8192) > gpurun_out/r05k/generic_rate.txt 2>&1
cat gpurun_out/r05k/generic_rate.txt
