#!/bin/bash
# round 6: clips by ticket in the three-waves-per-SIMD build -- check against the exact mode, then same-box rates (forced builds of the development library)
set -u
mkdir -p gpurun_out
{
KWS_DEV_FAST_WPS=3 KWS_LIB=ab_tmp/libkws_newdev.so timeout 600 python tools/gpu_fast_check.py 2048 65536 cfg2_mfcc40_f32.kwsm,l476_no_yes_f32.kwsm,cfg5_dscnn_mfcc40_f32.kwsm 2>&1 | grep -v "special\|amdgpu.ids"
timeout 900 python tools/ab_rate.py base,new,newdev+KWS_DEV_FAST_WPS=3,newdev+KWS_DEV_FAST_WPS=2 3 cfg2_mfcc40_f32.kwsm,l476_no_yes_f32.kwsm,l476_no_yes.kwsm,cfg5_dscnn_mfcc40_f32.kwsm 2>&1
} > gpurun_out/r06ab_tickets.txt 2>&1
cat gpurun_out/r06ab_tickets.txt
