#!/bin/bash
# round 6, fourth GPU call: same-box A/B of the kws_fast.hip variants (bpermute batch, two-step operand prefetch), the diagnostic of the random MFCC configuration
set -u
mkdir -p gpurun_out
python tools/ab_rate.py base,bperm,pf2,both 3 cfg2_mfcc40_f32.kwsm,l476_no_yes_f32.kwsm fast > gpurun_out/r06d_ab.txt 2>&1
cat gpurun_out/r06d_ab.txt
python tools/round6/diag_seed15.py 15 3 7 > gpurun_out/r06d_diag.txt 2>&1
cat gpurun_out/r06d_diag.txt | tail -20
