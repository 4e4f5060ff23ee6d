#!/bin/bash
# round 6, sixth GPU call: same-box A/B of round 5's final library against the current one (fast mode, four graphs), then the fast-mode tests on the current one
set -u
mkdir -p gpurun_out
python tools/ab_rate.py r05,cur 3 cfg2_mfcc40_f32.kwsm,l476_no_yes.kwsm,l476_no_yes_f32.kwsm,cfg2_mfcc40_int8.kwsm fast > gpurun_out/r06f_ab.txt 2>&1
cat gpurun_out/r06f_ab.txt
timeout 1500 python -m pytest tests/test_gpu_fast_families.py tests/test_gpu_fast_mode.py -x -q > gpurun_out/r06f_fast_tests.txt 2>&1
tail -3 gpurun_out/r06f_fast_tests.txt
