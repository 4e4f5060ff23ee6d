#!/bin/bash
# round 6, eighth GPU call: A/B of round 5's library against the current one (pivot = the clip's first silent frame), fast-mode tests, guard-off study, bench
set -u
mkdir -p gpurun_out
python tools/ab_rate.py r05,cur3 3 cfg2_mfcc40_f32.kwsm,l476_no_yes.kwsm,l476_no_yes_f32.kwsm,cfg2_mfcc40_int8.kwsm fast > gpurun_out/r06k_ab.txt 2>&1
cat gpurun_out/r06k_ab.txt
timeout 1500 python -m pytest tests/test_gpu_fast_families.py tests/test_gpu_fast_mode.py -x -q -s > gpurun_out/r06k_fast_tests.txt 2>&1
tail -3 gpurun_out/r06k_fast_tests.txt; grep "worst logit" gpurun_out/r06k_fast_tests.txt | cut -c1-330
python tools/gpu_guard_study.py 2048 gpurun_out/r06k_guard_study.npz > gpurun_out/r06k_guard_study.txt 2> gpurun_out/r06k_guard_study.err
cat gpurun_out/r06k_guard_study.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06k_bench_driver_flags.json 2> gpurun_out/r06k_bench_driver_flags.err
cp bench_detail.json gpurun_out/r06k_bench_detail.json
python -c "
import json; j=json.load(open('gpurun_out/r06k_bench_driver_flags.json')); print(j['value'], j['ms_per_step'], j['also_inputs'])"
