set -x
bash tools/profile_round.sh r05p > gpurun_out/r05p_profile_round.log 2>&1
tail -5 gpurun_out/r05p_profile_round.log
(time python bench.py --steps 20 --warmup 5 > gpurun_out/r05p/bench_driver_flags.json 2> gpurun_out/r05p/bench_driver_flags.err)
python - <<'PY'
import json
for f in ("gpurun_out/r05p/bench.json", "gpurun_out/r05p/bench_driver_flags.json"):
    j = json.load(open(f))
    print(f, j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["bound"], j["roofline"].get("compute"), j["roofline"]["traffic"])
PY
