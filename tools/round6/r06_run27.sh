#!/bin/bash
# round 6: the int8-fused forms of the fast kernel at three waves per SIMD (development library, forced): check and same-box rate
set -u
mkdir -p gpurun_out
{
KWS_DEV_FAST_REPORT=1 KWS_DEV_FAST_WPS_Q=3 KWS_LIB=ab_tmp/libkws_newdev.so timeout 600 python tools/gpu_fast_check.py 2048 65536 l476_no_yes.kwsm,cfg2_mfcc40_int8.kwsm 2>&1 | grep -v "special\|amdgpu.ids"
timeout 900 python tools/ab_rate.py newdev,newdev+KWS_DEV_FAST_WPS_Q=3 3 l476_no_yes.kwsm,cfg2_mfcc40_int8.kwsm,cfg5_dscnn_mfcc40_int8.kwsm 2>&1
} > gpurun_out/r06ag_q3.txt 2>&1
cat gpurun_out/r06ag_q3.txt
