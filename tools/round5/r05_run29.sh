# second final (part 1), after the batched split: full GPU suite, smoke, tools/profile_round.sh (bench, kernel traces, SQ counter passes, HBM traffic)
set -x
mkdir -p gpurun_out/r05y
sha256sum ei-keyword-spotting_amd/libkws_mi355x.so > gpurun_out/r05y/lib_sha256.txt
(time timeout 1800 python -m pytest tests -m gpu -q) > gpurun_out/r05y/pytest_gpu.txt 2>&1
tail -4 gpurun_out/r05y/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05y/smoke.txt 2>&1
tail -4 gpurun_out/r05y/smoke.txt
bash tools/profile_round.sh r05y > gpurun_out/r05y_profile_round.log 2>&1
python - <<'PY'
import json
j = json.load(open("gpurun_out/r05y/bench.json"))
print(j["value"], j["ms_per_step"], j["roofline"]["frac"])
PY
