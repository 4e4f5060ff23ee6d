#!/bin/bash
# round 6: shader clock and socket power under the product library and under the three-waves-per-SIMD build (is the difference between cycles and time power?)
set -u
mkdir -p gpurun_out
{
for L in base wps3 base wps3; do
  for M in cfg2_mfcc40_f32.kwsm l476_no_yes_f32.kwsm; do
    echo "##### $L $M"
    KWS_LIB=$(pwd)/ab_tmp/libkws_$L.so python tools/gpu_clock_watch.py $M fast 6 2>&1 | grep -v amdgpu.ids | sed -e 's/GPU\[0\]\t\t: //g' -e 's/=* Power Consumption =* | //' | cut -c1-400 | awk 'NR<=4 || NR%3==0'
  done
done
} > gpurun_out/r06_clock_power_wps3.txt 2>&1
grep "#####\|ms per step" gpurun_out/r06_clock_power_wps3.txt
grep -o "sclk clock level: [0-9]: ([0-9]*Mhz)\|Power (W): [0-9.]*" gpurun_out/r06_clock_power_wps3.txt | sort | uniq -c | sort -rn | head -30
