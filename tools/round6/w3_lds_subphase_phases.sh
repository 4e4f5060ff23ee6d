#!/bin/bash
# round 6: the three-waves-per-SIMD build -- LDS split, clocks inside the first convolution block, phases, check, same-box rate
set -u
mkdir -p gpurun_out
TAG=${1:-y}
{
KWS_DEV_FAST_REPORT=1 KWS_LIB=ab_tmp/libkws_wps3dev.so python tools/gpu_fast_phase_profile.py models/cfg2_mfcc40_f32.kwsm 4096 2>&1 | grep "fast plan"
echo "== subprof3 (11 waves)"; KWS_LIB=ab_tmp/libkws_subprof3.so python tools/gpu_fast_subphase.py 50 11 2>&1 | grep -v amdgpu.ids
for M in models/cfg2_mfcc40_f32.kwsm models/l476_no_yes_f32.kwsm; do
    echo "=== wps3 (168 registers): $M"; KWS_DEV_FAST_WAVES=11 KWS_LIB=ab_tmp/libkws_wps3dev.so python tools/gpu_fast_phase_profile.py $M 65536 2>/dev/null | grep -v "amdgpu.ids\|block 7\|column-0"
done
KWS_LIB=ab_tmp/libkws_wps3dev.so timeout 600 python tools/gpu_fast_check.py 2048 65536 cfg2_mfcc40_f32.kwsm,l476_no_yes_f32.kwsm 2>&1 | grep -v "special\|amdgpu.ids"
timeout 900 python tools/ab_rate.py basedev,wps3dev 2 cfg2_mfcc40_f32.kwsm,l476_no_yes_f32.kwsm 2>&1
} > gpurun_out/r06${TAG}_wps3.txt 2>&1
cat gpurun_out/r06${TAG}_wps3.txt
