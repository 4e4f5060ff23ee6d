#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 900 python tools/ab_rate.py base,new,new3,new4 3 cfg2_mfcc40_f32.kwsm,l476_no_yes_f32.kwsm > gpurun_out/r06ai_ab.txt 2>&1; cat gpurun_out/r06ai_ab.txt
