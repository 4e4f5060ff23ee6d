#!/bin/bash
# round 6: the three-waves-per-SIMD build, step by step -- phase clocks of wave 0 and the rate against the product library (same box)
set -u
mkdir -p gpurun_out
TAG=${1:-v}
{
for M in models/cfg2_mfcc40_f32.kwsm models/l476_no_yes_f32.kwsm; do
  for W in 8 12; do
    echo "=== wps3 (168 registers), $W waves: $M"; KWS_DEV_FAST_WAVES=$W KWS_LIB=ab_tmp/libkws_wps3dev.so python tools/gpu_fast_phase_profile.py $M 65536 2>/dev/null | grep -v amdgpu.ids
  done
done
} > gpurun_out/r06${TAG}_wps3_phases.txt
cat gpurun_out/r06${TAG}_wps3_phases.txt
KWS_LIB=ab_tmp/libkws_wps3dev.so timeout 600 python tools/gpu_fast_check.py 2048 65536 cfg2_mfcc40_f32.kwsm,l476_no_yes_f32.kwsm 2>&1 | grep -v special > gpurun_out/r06${TAG}_wps3_check.txt
cat gpurun_out/r06${TAG}_wps3_check.txt
timeout 900 python tools/ab_rate.py basedev,wps3dev+KWS_DEV_FAST_WAVES=8,wps3dev 2 cfg2_mfcc40_f32.kwsm,l476_no_yes_f32.kwsm > gpurun_out/r06${TAG}_ab.txt 2>&1
cat gpurun_out/r06${TAG}_ab.txt
