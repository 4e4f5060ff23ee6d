#!/bin/bash
# round 6: the library with the three-waves-per-SIMD forms chosen by plan -- whole GPU suite, continuous mode under both builds, same-box rates
set -u
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r06ac_gpu_suite.txt 2>&1
tail -4 gpurun_out/r06ac_gpu_suite.txt
{
for W in 2 3; do echo "== continuous mode, fused plan laid out for $W waves per SIMD"; KWS_DEV_FAST_WPS=$W KWS_LIB=ab_tmp/libkws_newdev.so python tools/gpu_streams_rate.py cfg2_mfcc40_f32.kwsm l476_no_yes_f32.kwsm 2>&1 | grep "fast\]: 65536"; done
timeout 900 python tools/ab_rate.py base,new 3 cfg2_mfcc40_f32.kwsm,l476_no_yes_f32.kwsm,l476_no_yes.kwsm,cfg5_dscnn_mfcc40_f32.kwsm,cfg2_mfcc40_int8.kwsm 2>&1
} > gpurun_out/r06ac_rates.txt 2>&1
cat gpurun_out/r06ac_rates.txt
