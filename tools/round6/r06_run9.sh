#!/bin/bash
# round 6, ninth GPU call: why kws_fast_kernel takes 10 % longer on word_silence / amp_sweep clips than on the bench's (phase clocks per input family)
set -u
mkdir -p gpurun_out
M=models/cfg2_mfcc40_f32.kwsm
for fam in "" word_silence amp_sweep word_background; do
  python tools/gpu_fast_phase_profile.py $M 65536 $fam 2>/dev/null | grep -v amdgpu.ids
done > gpurun_out/r06i_phase_by_family.txt
cat gpurun_out/r06i_phase_by_family.txt
