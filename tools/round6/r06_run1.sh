#!/bin/bash
# round 6, first GPU call: the compact bench line with the driver's flags, rocprofv3 evidence for configs[3] / configs[4], the guard-off study
set -u
mkdir -p gpurun_out
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06a_bench_driver_flags.json 2> gpurun_out/r06a_bench_driver_flags.err
cp bench_detail.json gpurun_out/r06a_bench_detail.json
wc -c gpurun_out/r06a_bench_driver_flags.json
bash tools/profile_round.sh r06a_int8 l476_no_yes.kwsm exact 0 > gpurun_out/r06a_int8.log 2>&1
bash tools/profile_round.sh r06a_cfg5 cfg5_dscnn_mfcc40_f32.kwsm fast 0 > gpurun_out/r06a_cfg5.log 2>&1
python tools/gpu_guard_study.py 2048 gpurun_out/r06a_guard_study.npz > gpurun_out/r06a_guard_study.txt 2> gpurun_out/r06a_guard_study.err
tail -40 gpurun_out/r06a_guard_study.txt
timeout 900 python -m pytest tests/test_gpu_bench.py -x -q 2>&1 | tail -5
cat gpurun_out/r06a_bench_driver_flags.json
