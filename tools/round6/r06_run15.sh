#!/bin/bash
# round 6: where the three-waves-per-SIMD build loses -- phase clocks of wave 0 (product registers against 168) and the rate over waves per workgroup
set -u
mkdir -p gpurun_out
{
for M in models/cfg2_mfcc40_f32.kwsm models/l476_no_yes_f32.kwsm; do
  echo "=== base (256 registers), 8 waves: $M"; KWS_LIB=ab_tmp/libkws_basedev.so python tools/gpu_fast_phase_profile.py $M 65536 2>/dev/null | grep -v amdgpu.ids
  for W in 8 11; do
    echo "=== wps3 (168 registers), $W waves: $M"; KWS_DEV_FAST_WAVES=$W KWS_LIB=ab_tmp/libkws_wps3dev.so python tools/gpu_fast_phase_profile.py $M 65536 2>/dev/null | grep -v amdgpu.ids
  done
done
} > gpurun_out/r06u_wps3_phases.txt
cat gpurun_out/r06u_wps3_phases.txt
timeout 900 python tools/ab_rate.py basedev,wps3dev+KWS_DEV_FAST_WAVES=8,wps3dev+KWS_DEV_FAST_WAVES=9,wps3dev+KWS_DEV_FAST_WAVES=10,wps3dev+KWS_DEV_FAST_WAVES=11,wps3dev+KWS_DEV_FAST_WAVES=12 2 cfg2_mfcc40_f32.kwsm,l476_no_yes_f32.kwsm > gpurun_out/r06u_ab.txt 2>&1
cat gpurun_out/r06u_ab.txt
