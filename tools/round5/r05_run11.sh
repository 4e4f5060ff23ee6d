set -x
mkdir -p gpurun_out/r05i
timeout 900 python tools/ab_rate.py prev,dev 2 cfg5_dscnn_mfcc40_f32.kwsm,cfg2_mfcc40_f32.kwsm > gpurun_out/r05i/ab_fromcep.txt 2>&1
cat gpurun_out/r05i/ab_fromcep.txt
timeout 600 python tools/gpu_streams_rate.py cfg2_mfcc40_f32.kwsm > gpurun_out/r05i/streams.txt 2>&1
grep fast gpurun_out/r05i/streams.txt
timeout 900 python -m pytest tests/test_gpu_fast_mode.py tests/test_gpu_fast_families.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r05i/pytest_sel.txt 2>&1
tail -3 gpurun_out/r05i/pytest_sel.txt
