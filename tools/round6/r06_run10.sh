#!/bin/bash
# round 6, tenth GPU call: general plans whose spectral stage fits the tuned kernel run it over chunks of frames: tests, rates
set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_generic_dsp.py tests/test_gpu_mfcc_layouts.py -x -q > gpurun_out/r06j_generic_tests.txt 2>&1
tail -5 gpurun_out/r06j_generic_tests.txt
python tools/gpu_generic_rate.py 8192 > gpurun_out/r06j_generic_rate.txt 2>&1
cat gpurun_out/r06j_generic_rate.txt | tail -30
