#!/bin/bash
# round 6: two product libraries under ab_tmp/ on the same box -- friendly batches (tools/ab_rate.py) and input families (tools/gpu_family_steps.py)
# usage: w3_family_ab.sh <lib a> <lib b> [tag]
set -u
A=${1:-new7}; B=${2:-new9}; TAG=${3:-ap}
mkdir -p gpurun_out
{
timeout 900 python tools/ab_rate.py $A,$B 3 cfg2_mfcc40_f32.kwsm,l476_no_yes_f32.kwsm,cfg5_dscnn_mfcc40_f32.kwsm 2>&1
for rep in 1 2; do for fam in bursts word_silence amp_sweep; do for L in $A $B; do echo -n "$L: "; KWS_LIB=$(pwd)/ab_tmp/libkws_$L.so python tools/gpu_family_steps.py $fam 200 2>&1 | grep "ms per step"; done; done; done
} > gpurun_out/r06${TAG}_family_ab.txt 2>&1
cat gpurun_out/r06${TAG}_family_ab.txt
