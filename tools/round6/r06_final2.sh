#!/bin/bash
# round 6 final, part 2 (profiles/r06*_pmc hold the final library's counters): the bench lines with the counters attached -- the driver's flags, then the defaults --,
# the seed sweep of the fast mode against the exact mode (tools/gpu_fast_sweep.py 128 8: 8.57 M clips per float model)
set -x
mkdir -p gpurun_out/r06
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06/bench_driver_flags.json 2> gpurun_out/r06/bench_driver_flags.err
cp bench_detail.json gpurun_out/r06/bench_detail_driver_flags.json
wc -c gpurun_out/r06/bench_driver_flags.json
python bench.py > gpurun_out/r06/bench_default.json 2> gpurun_out/r06/bench_default.err
cp bench_detail.json gpurun_out/r06/bench_detail_default.json
cat gpurun_out/r06/bench_default.json
(time timeout 2400 python tools/gpu_fast_sweep.py 128 8) > gpurun_out/r06/fast_sweep.txt 2>&1
tail -6 gpurun_out/r06/fast_sweep.txt
