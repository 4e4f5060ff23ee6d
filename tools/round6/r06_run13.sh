#!/bin/bash
# round 6: fast-mode tests, guard-off study and bench on the library with the test for near-constant columns (float32 forms) and the rms-level constants
set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_fast_families.py tests/test_gpu_fast_mode.py tests/test_gpu_generic_dsp.py -x -q -s > gpurun_out/r06s_fast_tests.txt 2>&1
tail -3 gpurun_out/r06s_fast_tests.txt; grep "worst logit" gpurun_out/r06s_fast_tests.txt | cut -c1-330
python tools/gpu_guard_study.py 2048 gpurun_out/r06s_guard_study.npz > gpurun_out/r06s_guard_study.txt 2> gpurun_out/r06s_guard_study.err
cat gpurun_out/r06s_guard_study.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06s_bench_driver_flags.json 2> gpurun_out/r06s_bench_driver_flags.err
cp bench_detail.json gpurun_out/r06s_bench_detail.json
python -c "
import json; j=json.load(open('gpurun_out/r06s_bench_driver_flags.json')); print(j['value'], j['ms_per_step'], len(json.dumps(j))); [print(r) for r in j['also_inputs']]; [print(r) for r in j['also']]"
