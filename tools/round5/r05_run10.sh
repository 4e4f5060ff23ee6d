set -x
mkdir -p gpurun_out/r05r
(time timeout 1800 python -m pytest tests -m gpu -q) > gpurun_out/r05r/pytest_gpu.txt 2>&1
tail -4 gpurun_out/r05r/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05r/smoke.txt 2>&1
tail -4 gpurun_out/r05r/smoke.txt
bash tools/profile_round.sh r05r > gpurun_out/r05r_profile_round.log 2>&1
(time python bench.py --steps 20 --warmup 5 > gpurun_out/r05r/bench_driver_flags.json 2> gpurun_out/r05r/bench_driver_flags.err)
python - <<'PY'
import json
for f in ("gpurun_out/r05r/bench.json", "gpurun_out/r05r/bench_driver_flags.json"):
    j = json.load(open(f))
    print(f, j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["bound"], j["roofline"]["traffic"])
PY
