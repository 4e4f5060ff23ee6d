# general-shape DSP kernel: product = two builds (deep batches at two waves per SIMD, shallow at four) chosen by resident waves; tests, rates, phase clocks
set -x
mkdir -p gpurun_out/r05j
(time timeout 900 python -m pytest tests/test_gpu_generic_dsp.py -m gpu -q -x) > gpurun_out/r05j/pytest_generic.txt 2>&1
tail -4 gpurun_out/r05j/pytest_generic.txt
(time timeout 1200 python tools/gpu_generic_rate.py 8192) > gpurun_out/r05j/generic_rate.txt 2>&1
cat gpurun_out/r05j/generic_rate.txt
