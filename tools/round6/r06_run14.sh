#!/bin/bash
# round 6: the three-waves-per-SIMD build of the fast kernel (KWS_FAST_WPS = 3: 168 registers, 11 waves per CU, first block's fragments from device memory)
# -- fast against exact mode on 2 048 clips + specials, then same-box A/B against the product library
set -u
mkdir -p gpurun_out
KWS_LIB=ab_tmp/libkws_wps3.so timeout 600 python tools/gpu_fast_check.py 2048 65536 cfg2_mfcc40_f32.kwsm,l476_no_yes_f32.kwsm,l476_no_yes.kwsm > gpurun_out/r06t_wps3_check.txt 2>&1
tail -30 gpurun_out/r06t_wps3_check.txt
timeout 900 python tools/ab_rate.py base,wps3 3 > gpurun_out/r06t_ab.txt 2>&1
cat gpurun_out/r06t_ab.txt
