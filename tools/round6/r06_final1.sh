#!/bin/bash
# round 6 final, part 1: full GPU suite + smoke on the final library, rocprofv3 evidence (kernel traces, SQ counters, HBM traffic) for the headline,
# BASELINE configs[3] (int8, exact) and the configs[4] shape (fp32, fast), the guard-off study.  Part 2 (r06_final2.sh) runs once the traffic files sit under profiles/.
set -x
mkdir -p gpurun_out/r06
sha256sum ei-keyword-spotting_amd/libkws_mi355x.so > gpurun_out/r06/lib_sha256.txt
(time timeout 2400 python -m pytest tests -m gpu -q) > gpurun_out/r06/pytest_gpu.txt 2>&1
tail -4 gpurun_out/r06/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06/smoke.txt 2>&1
tail -4 gpurun_out/r06/smoke.txt
bash tools/profile_round.sh r06 > gpurun_out/r06_profile_round.log 2>&1
bash tools/profile_round.sh r06_int8 l476_no_yes.kwsm exact 0 > gpurun_out/r06_int8.log 2>&1
bash tools/profile_round.sh r06_cfg5 cfg5_dscnn_mfcc40_f32.kwsm fast 0 > gpurun_out/r06_cfg5.log 2>&1
python tools/gpu_guard_study.py 2048 gpurun_out/r06/guard_study.npz > gpurun_out/r06/guard_study.txt 2> gpurun_out/r06/guard_study.err
tail -32 gpurun_out/r06/guard_study.txt
python tools/codeobj_meta.py ei-keyword-spotting_amd/libkws_mi355x.so gpurun_out/r06/codeobj.md > /dev/null 2>&1
python - <<'PY'
import json
j = json.load(open("gpurun_out/r06/bench.json"))
print(j["value"], j["ms_per_step"], j["roofline"]["frac"])
PY
