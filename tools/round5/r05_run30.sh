# second final (part 2): the bench lines once profiles/r05_pmc holds the final library's counters
set -x
mkdir -p gpurun_out/r05z
sha256sum ei-keyword-spotting_amd/libkws_mi355x.so
(time python bench.py > gpurun_out/r05z/bench.json 2> gpurun_out/r05z/bench.err)
(time python bench.py --steps 20 --warmup 5 > gpurun_out/r05z/bench_driver_flags.json 2> gpurun_out/r05z/bench_driver_flags.err)
python - <<'PY'
import json
for f in ("gpurun_out/r05z/bench.json", "gpurun_out/r05z/bench_driver_flags.json"):
    j = json.load(open(f))
    print(f, j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["bound"], j["roofline"]["traffic"], (j["roofline"].get("compute") or {}).get("valu_active_frac"))
PY
