# the split of a convolution block's image: reads batched instead of one per conditional block -- sub-phase clocks (scratch builds with clock reads), same-box A/B, fast-mode tests
set -x
mkdir -p gpurun_out/r05x
KWS_LIB=$GRAFT_REPO_ROOT/ab_tmp/libkws_subprof.so python tools/gpu_fast_subphase.py 50 > gpurun_out/r05x/subphase_before.txt 2>&1
KWS_LIB=$GRAFT_REPO_ROOT/ab_tmp/libkws_subprof2.so python tools/gpu_fast_subphase.py 50 > gpurun_out/r05x/subphase_after.txt 2>&1
tail -2 gpurun_out/r05x/subphase_before.txt gpurun_out/r05x/subphase_after.txt
python tools/ab_rate.py final1,split2 3 cfg2_mfcc40_f32.kwsm,l476_no_yes_f32.kwsm,cfg5_dscnn_mfcc40_f32.kwsm fast > gpurun_out/r05x/ab_fast.txt 2>&1
cat gpurun_out/r05x/ab_fast.txt
(time timeout 900 python -m pytest tests/test_gpu_fast_mode.py tests/test_gpu_fast_families.py -m gpu -q -x) > gpurun_out/r05x/pytest_fast.txt 2>&1
tail -4 gpurun_out/r05x/pytest_fast.txt
