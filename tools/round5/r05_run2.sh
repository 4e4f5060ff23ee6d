set -x
mkdir -p gpurun_out/r05b
timeout 600 python tools/ab_rate.py r4,dev 2 cfg2_mfcc40_f32.kwsm,l476_no_yes_f32.kwsm,cfg5_dscnn_mfcc40_f32.kwsm > gpurun_out/r05b/ab_rate.txt 2>&1
cat gpurun_out/r05b/ab_rate.txt
timeout 600 python tools/gpu_fast_phase_profile.py > gpurun_out/r05b/fast_phase.txt 2>&1
tail -25 gpurun_out/r05b/fast_phase.txt
timeout 900 python -m pytest tests/test_boundary_hooks.py tests/test_gpu_fast_mode.py tests/test_gpu_generic_dsp.py tests/test_gpu_mfcc_layouts.py tests/test_c_demo.py -m gpu -x -q > gpurun_out/r05b/pytest_sel.txt 2>&1
tail -15 gpurun_out/r05b/pytest_sel.txt
(time timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05b/bench_driver_flags.json 2> gpurun_out/r05b/bench_driver_flags.err)
tail -3 gpurun_out/r05b/bench_driver_flags.err
python - <<'PY'
import json
j = json.load(open("gpurun_out/r05b/bench_driver_flags.json"))
print(j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["bound"])
for x in j.get("also", []): print(x["kwsm_file"], x["mode"], x["value"], x["ms_per_step"], x["fast_fallback_rate"], x["fast_exact_rate"])
for x in j.get("also_inputs", []): print(x["family"], x["mode"], x["value"], x["ms_per_step"], x["fast_fallback_rate"], x["fast_exact_rate"])
print(j.get("cpu_baseline"))
PY
