#!/bin/bash
# round 6: the library with both builds of the fast kernel (two and three waves per SIMD) -- fast-mode tests, then same-box rate against round 6's first final (c7e4a922)
set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_fast_families.py tests/test_gpu_fast_mode.py -x -q > gpurun_out/r06aa_fast_tests.txt 2>&1
tail -3 gpurun_out/r06aa_fast_tests.txt
timeout 900 python tools/ab_rate.py base,new,newdev+KWS_DEV_FAST_WPS=3,newdev+KWS_DEV_FAST_WPS=2 2 cfg2_mfcc40_f32.kwsm,l476_no_yes_f32.kwsm,l476_no_yes.kwsm,cfg5_dscnn_mfcc40_f32.kwsm > gpurun_out/r06aa_ab.txt 2>&1
cat gpurun_out/r06aa_ab.txt
