#!/bin/bash
# round 6: SQ counters of the three-waves-per-SIMD build against the product library, headline graph and its 49x13 twin (one --pmc pass per set)
set -u
mkdir -p gpurun_out
for L in base wps3; do
  for M in cfg2_mfcc40_f32 l476_no_yes_f32; do
    KWS_LIB=$(pwd)/ab_tmp/libkws_$L.so tools/pmc_sets.sh r06_pmc_${L}_$M python bench.py --model models/$M.kwsm --mode fast --steps 3 --warmup 1 --no-cpu-baseline --no-also > gpurun_out/r06_pmc_${L}_$M.txt 2>&1
  done
done
grep -A24 "kws_fast_kernel" gpurun_out/r06_pmc_*_*.txt | grep -v "^--" | head -120
