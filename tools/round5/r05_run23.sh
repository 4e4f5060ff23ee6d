# final (part 1): full GPU suite, smoke, tools/profile_round.sh (bench, kernel traces, SQ counter passes, HBM traffic), general-shape rates, code objects
set -x
mkdir -p gpurun_out/r05s
sha256sum ei-keyword-spotting_amd/libkws_mi355x.so > gpurun_out/r05s/lib_sha256.txt
(time timeout 1800 python -m pytest tests -m gpu -q) > gpurun_out/r05s/pytest_gpu.txt 2>&1
tail -4 gpurun_out/r05s/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05s/smoke.txt 2>&1
tail -4 gpurun_out/r05s/smoke.txt
bash tools/profile_round.sh r05s > gpurun_out/r05s_profile_round.log 2>&1
(time timeout 600 python tools/gpu_generic_rate.py 8192 "product,auto,auto, sample wait,L4 deep,L8 deep,L4 shallow,L8 shallow,L4 shallow, by sample,L8 shallow, by sample") > gpurun_out/r05s/generic_rate.txt 2>&1
grep -v "tuned\|0\.4[0-9] ns" gpurun_out/r05s/generic_rate.txt | head -60
python - <<'PY'
import json
j = json.load(open("gpurun_out/r05s/bench.json"))
print(j["value"], j["ms_per_step"], j["roofline"]["frac"], j.get("also_dsp"))
PY
