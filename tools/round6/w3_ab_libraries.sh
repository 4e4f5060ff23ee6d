#!/bin/bash
# round 6: same-box rate of product libraries kept under ab_tmp/ (tools/ab_rate.py); usage: w3_ab_libraries.sh <names, comma separated> [models] [tag]
set -u
mkdir -p gpurun_out
NAMES=${1:-base,new4,new5}; MODELS=${2:-cfg2_mfcc40_f32.kwsm,l476_no_yes_f32.kwsm}; TAG=${3:-ak}
timeout 900 python tools/ab_rate.py $NAMES 3 $MODELS > gpurun_out/r06${TAG}_ab.txt 2>&1; cat gpurun_out/r06${TAG}_ab.txt
