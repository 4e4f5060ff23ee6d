#!/bin/bash
# round 6: bench line (driver's flags) on the library with the three-waves-per-SIMD forms
set -u
mkdir -p gpurun_out
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06af_bench_driver_flags.json 2> gpurun_out/r06af_bench_driver_flags.err
cp bench_detail.json gpurun_out/r06af_bench_detail.json
python -c "
import json; j=json.load(open('gpurun_out/r06af_bench_driver_flags.json')); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['kernel'], len(json.dumps(j))); [print(r) for r in j['also_inputs']]; [print(r) for r in j['also']]"
