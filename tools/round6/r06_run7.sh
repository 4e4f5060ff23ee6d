#!/bin/bash
# round 6, seventh GPU call: which of the round's additions to kws_fast_kernel costs the 1 - 2 % of r06f_ab.txt (same-box A/B of isolation variants)
set -u
mkdir -p gpurun_out
python tools/ab_rate.py r05,cur,vA,vB,vC 3 cfg2_mfcc40_f32.kwsm,l476_no_yes.kwsm fast > gpurun_out/r06g_ab.txt 2>&1
cat gpurun_out/r06g_ab.txt
