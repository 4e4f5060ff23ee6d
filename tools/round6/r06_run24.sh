#!/bin/bash
# round 6: SQ counters and phase clocks of the headline graph on the library with the three-waves-per-SIMD form (clips by ticket)
set -u
mkdir -p gpurun_out
KWS_LIB=$(pwd)/ab_tmp/libkws_new.so tools/pmc_sets.sh r06_pmc_new_cfg2_mfcc40_f32 python bench.py --model models/cfg2_mfcc40_f32.kwsm --mode fast --steps 3 --warmup 1 --no-cpu-baseline --no-also > gpurun_out/r06_pmc_new_cfg2_mfcc40_f32.txt 2>&1
KWS_LIB=ab_tmp/libkws_newdev.so python tools/gpu_fast_phase_profile.py models/cfg2_mfcc40_f32.kwsm 65536 2>/dev/null | grep -v "amdgpu.ids" > gpurun_out/r06ad_phases.txt
cat gpurun_out/r06ad_phases.txt
