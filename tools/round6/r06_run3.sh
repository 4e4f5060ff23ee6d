#!/bin/bash
# round 6, second GPU call: the re-fitted guard (per-column DCT constants, the reference's row for digitally silent frames): families + fast-mode tests,
# the guard-off study on the new library, the bench line with the driver's flags
set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_fast_families.py tests/test_gpu_fast_mode.py -x -q -s > gpurun_out/r06c_fast_tests.txt 2>&1
tail -15 gpurun_out/r06c_fast_tests.txt
python tools/gpu_guard_study.py 2048 gpurun_out/r06c_guard_study.npz > gpurun_out/r06c_guard_study.txt 2> gpurun_out/r06c_guard_study.err
cat gpurun_out/r06c_guard_study.txt; tail -3 gpurun_out/r06c_guard_study.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06c_bench_driver_flags.json 2> gpurun_out/r06c_bench_driver_flags.err
cp bench_detail.json gpurun_out/r06c_bench_detail.json
cat gpurun_out/r06c_bench_driver_flags.json
