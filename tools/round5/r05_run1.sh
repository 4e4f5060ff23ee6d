set -x
mkdir -p gpurun_out/r05a
timeout 600 python tools/gpu_fast_check.py 2048 0 cfg2_mfcc40_f32.kwsm,l476_no_yes_f32.kwsm,cfg5_dscnn_mfcc40_f32.kwsm > gpurun_out/r05a/fast_check.txt 2>&1
tail -30 gpurun_out/r05a/fast_check.txt
timeout 900 python tools/ab_rate.py r4,dev+KWS_DEV_FAST_F32_CONV,dev,dev+KWS_DEV_FAST_B_GLOBAL 2 cfg2_mfcc40_f32.kwsm,l476_no_yes_f32.kwsm,cfg5_dscnn_mfcc40_f32.kwsm > gpurun_out/r05a/ab_rate.txt 2>&1
cat gpurun_out/r05a/ab_rate.txt
timeout 600 python tools/gpu_fast_phase_profile.py > gpurun_out/r05a/fast_phase.txt 2>&1
tail -25 gpurun_out/r05a/fast_phase.txt
timeout 900 python -m pytest tests/test_gpu_fast_mode.py tests/test_gpu_fast_families.py -m gpu -x -q > gpurun_out/r05a/pytest_fast.txt 2>&1
tail -15 gpurun_out/r05a/pytest_fast.txt
