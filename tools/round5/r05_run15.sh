# general-shape DSP kernel: MEL/DCT buffers over the dead work buffers, scalar wave index; builds for 3 and 4 waves per SIMD at several batch depths
set -x
mkdir -p gpurun_out/r05h
(time timeout 900 python -m pytest tests/test_gpu_generic_dsp.py -m gpu -q -x) > gpurun_out/r05h/pytest_generic.txt 2>&1
tail -4 gpurun_out/r05h/pytest_generic.txt
(time timeout 1200 python tools/gpu_generic_rate.py 8192) > gpurun_out/r05h/generic_rate.txt 2>&1
cat gpurun_out/r05h/generic_rate.txt
