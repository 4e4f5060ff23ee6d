#!/bin/bash
# round 6: the new test of the three-waves-per-SIMD forms, the bench tests
set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_fast_mode.py::test_three_waves_per_simd_forms tests/test_gpu_bench.py -x -q > gpurun_out/r06ae_tests.txt 2>&1
tail -25 gpurun_out/r06ae_tests.txt
