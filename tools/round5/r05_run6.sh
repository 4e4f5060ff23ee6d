set -x
mkdir -p gpurun_out/r05f
timeout 900 python tools/ab_rate.py dev,wps3,wps3+KWS_DEV_FAST_B_GLOBAL,wps3+KWS_DEV_FAST_WAVES=10,wps3+KWS_DEV_FAST_WAVES=8 2 cfg2_mfcc40_f32.kwsm,l476_no_yes_f32.kwsm > gpurun_out/r05f/ab_wps3.txt 2>&1
cat gpurun_out/r05f/ab_wps3.txt
timeout 600 python -m pytest tests/test_gpu_generic_dsp.py -m gpu -x -q > gpurun_out/r05f/pytest_generic.txt 2>&1
tail -3 gpurun_out/r05f/pytest_generic.txt
