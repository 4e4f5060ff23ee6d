#!/bin/bash
set -u
mkdir -p gpurun_out
python tools/ab_rate.py r05,c2932f6d,cur9,cur10 3 cfg2_mfcc40_f32.kwsm,l476_no_yes.kwsm,l476_no_yes_f32.kwsm,cfg2_mfcc40_int8.kwsm fast > gpurun_out/r06r_ab.txt 2>&1
cat gpurun_out/r06r_ab.txt
