#!/bin/bash
# round 6: the three-waves-per-SIMD build -- its LDS split as the plan reports it, and the clocks inside the first convolution block
set -u
mkdir -p gpurun_out
{
KWS_DEV_FAST_REPORT=1 KWS_LIB=ab_tmp/libkws_wps3dev.so python tools/gpu_fast_phase_profile.py models/cfg2_mfcc40_f32.kwsm 4096 2>&1 | grep "fast plan"
KWS_DEV_FAST_REPORT=1 KWS_LIB=ab_tmp/libkws_wps3dev.so python tools/gpu_fast_phase_profile.py models/l476_no_yes_f32.kwsm 4096 2>&1 | grep "fast plan"
echo "== subprof3 (11 waves)"; KWS_LIB=ab_tmp/libkws_subprof3.so python tools/gpu_fast_subphase.py 50 11 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r06x_report.txt 2>&1
cat gpurun_out/r06x_report.txt
