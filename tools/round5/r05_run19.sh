# reciprocals from the plans instead of in-kernel divisions (fast kernel, float network), general-shape kernel as of run18: full GPU suite, smoke, bench lines
set -x
mkdir -p gpurun_out/r05m
(time timeout 1800 python -m pytest tests -m gpu -q) > gpurun_out/r05m/pytest_gpu.txt 2>&1
tail -4 gpurun_out/r05m/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05m/smoke.txt 2>&1
tail -4 gpurun_out/r05m/smoke.txt
(time python bench.py > gpurun_out/r05m/bench.json 2> gpurun_out/r05m/bench.err)
(time python bench.py --steps 20 --warmup 5 > gpurun_out/r05m/bench_driver_flags.json 2> gpurun_out/r05m/bench_driver_flags.err)
python - <<'PY'
import json
for f in ("gpurun_out/r05m/bench.json", "gpurun_out/r05m/bench_driver_flags.json"):
    j = json.load(open(f))
    print(f, j["value"], j["ms_per_step"], j["roofline"]["frac"])
    for a in j.get("also", []):
        print("   ", a.get("model"), a.get("mode"), a.get("value"), a.get("ms_per_step"))
PY
