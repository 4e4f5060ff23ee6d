# exact-mode float network: epilogue constants requested before the accumulation loop -- same-box A/B against the final library, float parity tests
set -x
mkdir -p gpurun_out/r05u
python tools/ab_rate.py final1,epi 3 cfg2_mfcc40_f32.kwsm,l476_no_yes_f32.kwsm,cfg5_dscnn_mfcc40_f32.kwsm exact > gpurun_out/r05u/ab_exact.txt 2>&1
cat gpurun_out/r05u/ab_exact.txt
(time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "f32 or float or graph") > gpurun_out/r05u/pytest_f32.txt 2>&1
tail -4 gpurun_out/r05u/pytest_f32.txt
