# same-box A/B: the library of profiles/r05_pmc (base) against the current one (plan reciprocals, 24-bit index multiplies), fast and exact mode; generic tests + rates
set -x
mkdir -p gpurun_out/r05n
(time timeout 900 python -m pytest tests/test_gpu_generic_dsp.py tests/test_gpu_fast_mode.py -m gpu -q -x) > gpurun_out/r05n/pytest_some.txt 2>&1
tail -4 gpurun_out/r05n/pytest_some.txt
python tools/ab_rate.py base,new 3 cfg2_mfcc40_f32.kwsm,l476_no_yes_f32.kwsm,cfg5_dscnn_mfcc40_f32.kwsm fast > gpurun_out/r05n/ab_fast.txt 2>&1
cat gpurun_out/r05n/ab_fast.txt
python tools/ab_rate.py base,new 2 cfg2_mfcc40_f32.kwsm,cfg5_dscnn_mfcc40_f32.kwsm,cfg5_dscnn_mfcc40_int8.kwsm exact > gpurun_out/r05n/ab_exact.txt 2>&1
cat gpurun_out/r05n/ab_exact.txt
(time timeout 1200 python tools/gpu_generic_rate.py 8192 "product,auto,L4 shallow,L8 shallow") > gpurun_out/r05n/generic_rate.txt 2>&1
grep -v "tuned\|0\.4[0-9] ns" gpurun_out/r05n/generic_rate.txt
