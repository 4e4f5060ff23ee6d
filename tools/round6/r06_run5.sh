#!/bin/bash
# round 6, fifth GPU call: fast-mode tests on the library with the lane-local pivot for short windows; what the tiers behind the fast kernel cost on two input families
# (rocprofv3 kernel trace of tools/gpu_family_steps.py); the bench line
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_fast_families.py tests/test_gpu_fast_mode.py -x -q -s > gpurun_out/r06e_fast_tests.txt 2>&1
tail -5 gpurun_out/r06e_fast_tests.txt
for fam in word_silence amp_sweep; do
  python tools/gpu_family_steps.py $fam 50 > gpurun_out/r06e_family_$fam.txt 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r06e_trace_$fam -o t -- python tools/gpu_family_steps.py $fam 50 > gpurun_out/r06e_trace_$fam.log 2>&1
  db=$(find gpurun_out/r06e_trace_$fam -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocprof_summary.py "$db" gpurun_out/r06e_family_${fam}_kernel_stats.md "r06e: python tools/gpu_family_steps.py $fam 50 (65536 clips per step, KWS_MODE_FAST, cfg2_mfcc40_f32.kwsm)"
  find gpurun_out/r06e_trace_$fam -name "*.db" -delete
  cat gpurun_out/r06e_family_$fam.txt | tail -1; head -16 gpurun_out/r06e_family_${fam}_kernel_stats.md
done
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06e_bench_driver_flags.json 2> gpurun_out/r06e_bench_driver_flags.err
cp bench_detail.json gpurun_out/r06e_bench_detail.json
python -c "
import json; j=json.load(open('gpurun_out/r06e_bench_driver_flags.json')); print(j['value'], j['ms_per_step'], j['also_inputs'])"
