#!/bin/bash
# round 6: the ticket counter that cleans up after itself -- the test of the three-wave forms, fast-mode tests, a captured call replayed, same-box rate
set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_fast_mode.py tests/test_gpu_fast_families.py -x -q > gpurun_out/r06ah_tests.txt 2>&1
tail -3 gpurun_out/r06ah_tests.txt
timeout 300 python tools/gpu_graph_replay.py > gpurun_out/r06ah_graph_replay.txt 2>&1; tail -5 gpurun_out/r06ah_graph_replay.txt
timeout 900 python tools/ab_rate.py base,new,new2 3 cfg2_mfcc40_f32.kwsm,l476_no_yes_f32.kwsm > gpurun_out/r06ah_ab.txt 2>&1; cat gpurun_out/r06ah_ab.txt
