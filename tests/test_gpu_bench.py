"""-m gpu: bench.py's N > 1 path on the 1-GPU box: the RCCL communicator of the C ABI (kws_comm_* / kws_allgather_scores) at
world size 1, inside bench.py's own timed region, and directly."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from kws_testlib import MODELS, ROOT

pytestmark = pytest.mark.gpu


def test_bench_force_collective_on_one_gpu():
    env = dict(os.environ, MASTER_PORT=str(35500 + os.getpid() % 2000))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-collective", "--steps", "5", "--warmup", "2",
                          "--batch", "4096", "--no-cpu-baseline", "--no-also"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout                                      # banners of gloo / RCCL are kept off stdout
    assert len(lines[0]) < 6000
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["config"]["mode"] == "fast" and j["config"]["clips_per_gpu"] == 4096
    assert j["collective"]["ranks"] == 1 and j["collective"]["allgather_ms_per_step"] > 0
    # round 4: what RCCL itself reports for the communicator (ncclCommCount, ncclGetVersion), every rank's own rate
    assert j["collective"]["ranks_seen_by_rccl"] == 1 and j["collective"]["rccl_version"] >= 20000
    assert len(j["collective"]["per_rank_clips_per_s"]) == 1 and j["collective"]["rank_time_skew_max_over_min"] == 1.0
    assert "RCCL" in j["config"]["collective"]
    assert abs(j["checksum"] - 4096.0) < 0.05                               # the gathered scores: 4096 softmax rows
    assert j["value"] > 1e5 and j["roofline"]["kernel"] in ("kws_fast_kernel", "kws_fast_kernel_w3")


def test_bench_line_with_the_drivers_flags_is_compact_and_complete():
    """The driver's own command (`python bench.py --gpus 1 --steps 20 --warmup 5`, everything on: the CPU baseline leg, the other modes / models /
    input families): ONE stdout line under 6 KB (round 5's 27 KB line was not parsed by the driver) that carries the headline, its roofline
    object, the CPU baseline, the exact-mode row of the headline graph and BASELINE configs[3] (int8_exact); the rest is in bench_detail.json."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5"], stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0]) < 6000, (len(lines), [len(ln) for ln in lines])
    j = json.loads(lines[0])
    assert j["steps"] == 20 and j["warmup"] == 5 and j["n_gpus"] == 1 and j["unit"] == "clips/s" and j["value"] > 1e6
    assert abs(j["value"] - 65536 / (j["ms_per_step"] * 1e-3)) < 1e-3 * j["value"]
    rf = j["roofline"]
    assert rf["bound"] == "hbm" and rf["kernel"] == "kws_fast_kernel_w3" and rf["peak"] == 8000.0 and rf["algorithmic_bytes_per_launch"] == 65536 * 32016
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4 and 0 < rf["hot_path_ms"] <= j["ms_per_step"] * 1.02
    cb = j["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    rows = {(r["kwsm"], r["mode"]): r for r in j["also"]}
    assert ("cfg2_mfcc40_f32", "exact") in rows and ("cfg5_dscnn_mfcc40_f32", "fast") in rows
    assert j["int8_exact"]["kwsm"] == "l476_no_yes" and j["int8_exact"]["mode"] == "exact" and j["int8_exact"]["roofline"]["frac"] > 0
    assert {r["family"] for r in j["also_inputs"]} >= {"word_silence", "amp_sweep", "bursts"}
    detail = json.load(open(os.path.join(ROOT, j["detail"])))
    assert detail["value"] == j["value"] and "also_dsp" in detail and len(detail["also_inputs"]) == 2 * len(j["also_inputs"])


def test_allgather_scores_c_abi_world_size_one():
    sys.path.insert(0, ROOT)
    import torch
    from __graft_entry__ import load_package
    pkg = load_package()
    gm = pkg.Model(os.path.join(MODELS, "l476_no_yes.kwsm"), device=0)
    B = 300
    pcm = torch.empty((B, 16000), dtype=torch.int16, device="cuda:0")
    pkg.synth_clips_device(3, 0, B, 16000, pcm.data_ptr())
    s = torch.zeros((B, 4), dtype=torch.float32, device="cuda:0")
    allg = torch.zeros((B, 4), dtype=torch.float32, device="cuda:0")
    comm = pkg.Comm(pkg.Comm.unique_id(), 1, 0, 0)
    stream = torch.cuda.current_stream().cuda_stream
    gm.run_classifier_batch_device(pcm.data_ptr(), B, s.data_ptr(), None, None, stream)
    comm.allgather_scores(s.data_ptr(), allg.data_ptr(), B, 4, stream)     # same stream: no synchronisation in between
    comm.wait(stream)                                                      # deadline-guarded wait (kws_comm_wait)
    torch.cuda.synchronize()
    assert torch.equal(allg, s) and float(s.sum()) > 0
    assert comm.ranks_seen_by_rccl == 1 and comm.rccl_version // 10000 == 2
    with pytest.raises(pkg.KwsError):
        pkg.Comm(b"\0" * 16, 1, 0, 0)                                       # id too short
    comm.close()
    gm.close()


def test_comm_create_times_out_when_a_rank_is_missing():
    """a communicator for two ranks of which only one shows up: kws_comm_create must give up at the deadline (KWS_COMM_TIMEOUT_MS) with
    KWS_ERROR_HIP instead of waiting for ever.  In a process of its own: the abandoned helper thread stays
    inside RCCL."""
    code = ("import sys, time; sys.path.insert(0, %r); import torch; from __graft_entry__ import load_package; pkg = load_package(); t0 = time.time()\n"
            "try:\n    pkg.Comm(pkg.Comm.unique_id(), 2, 0, 0); print('CREATED')\n"
            "except pkg.KwsError as e:\n    print('TIMEOUT %%.1f %%s' %% (time.time() - t0, e))\n"
            "import os; sys.stdout.flush(); os._exit(0)\n") % ROOT
    env = dict(os.environ, KWS_COMM_TIMEOUT_MS="4000")
    out = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120, env=env)
    ln = [x for x in out.stdout.splitlines() if x.startswith(("TIMEOUT", "CREATED"))]
    assert ln and ln[0].startswith("TIMEOUT"), (out.stdout[-500:], out.stderr[-1500:])
    assert 3.0 <= float(ln[0].split()[1]) <= 30.0 and "not every rank joined" in ln[0]
