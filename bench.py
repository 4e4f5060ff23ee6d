#!/usr/bin/env python3
"""bench.py -- clips/sec of the run_classifier() hot path (MFCC + CNN) on N MI355X.

Headline workload = BASELINE.json configs[1] as worded: batch 65 536 synthetic 1 s @ 16 kHz clips, 40-band MFCC (49x40),
2-Conv CNN, fp32, 1x MI355X.  The reference ships no such model, so the graph is generated with
seeded weights (tools/synth_model.py + tools/dequantize_model.py -> models/cfg2_mfcc40_f32.kwsm); the DSP block is the
reference's own code path for that configuration (bit-exact, tests/golden/mfcc40_l476.npz).  At N = 1 the same run also
times the model the reference DOES ship (49x13 MFCC, int8), its fp32 twin and the int8 form of the headline graph (the two
readings of BASELINE configs[3]); they are reported under "also" in the same JSON line.

One "step" = one pass of the hot path over one batch of B synthetic 1 s @ 16 kHz int16 clips per GPU, the clips
already resident in HBM (generated on the device by kws_synth_clips_device).  N > 1: one process per GPU
(torch.distributed / RCCL), clips sharded contiguously, no data-path collective except the all-gather of the
per-clip scores (16 B/clip) which is inside the timed region.  Weak scaling: per-GPU batch fixed.

Prints ONE JSON line (rank 0).  See DESIGN.md "Measurement" for how roofline / cpu_baseline are defined.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK_GBS = 8000.0                        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


SHIPPED_MODEL = os.path.join(ROOT, "models", "l476_no_yes.kwsm")            # BASELINE configs[3]: what the reference ships
DEFAULT_MODEL = os.path.join(ROOT, "models", "cfg2_mfcc40_f32.kwsm")       # BASELINE configs[1] as worded
ALSO_MODELS = [SHIPPED_MODEL] + [os.path.join(ROOT, "models", n) for n in ("l476_no_yes_f32.kwsm", "cfg2_mfcc40_int8.kwsm",
                                                                       "cfg5_dscnn_mfcc40_int8.kwsm", "cfg5_dscnn_mfcc40_f32.kwsm")]
WORKLOADS = {
    "cfg2_mfcc40_f32.kwsm": "BASELINE configs[1]: 40-band MFCC (49x40: 40 mel, 40 cepstra, fft 256, CMVN 101) + 2-Conv CNN, fp32; "
                            "graph with seeded synthetic weights (the reference ships no such model)",
    "cfg2_mfcc40_int8.kwsm": "BASELINE configs[3] read as the configs[1] graph quantised: 49x40 MFCC + int8 2-Conv CNN, seeded synthetic weights",
    "l476_no_yes.kwsm": "BASELINE configs[3] read as the reference's own int8 model: the impulse the reference ships, 4-class no/noise/unknown/yes (MFCC 49x13: 32 mel, "
                        "fft 256, CMVN 101; int8 2-Conv CNN)",
    "l476_no_yes_f32.kwsm": "de-quantised fp32 twin of the shipped impulse (MFCC 49x13 + fp32 2-Conv CNN)",
    "cfg5_dscnn_mfcc40_int8.kwsm": "BASELINE configs[4] shape: 49x40 MFCC + 7-block depthwise-separable CNN, 12 labels, int8, synthetic weights",
    "cfg5_dscnn_mfcc40_f32.kwsm": "BASELINE configs[4] shape: 49x40 MFCC + 7-block depthwise-separable CNN, 12 labels, fp32, synthetic weights",
}


def cpu_worker(kind, n_clips, seconds, model_path=DEFAULT_MODEL):
    """Child process: run the CPU path over n_clips synthetic clips again and again for ~`seconds`; prints clips/s."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from kws_testlib import Oracle, OracleModel, Reference
    o = Oracle()
    clips = o.synth(0, 0, n_clips)
    if kind == "reference" and os.path.samefile(model_path, SHIPPED_MODEL):
        run = Reference().time_run                       # the reference's run_classifier() with its compiled-in model
    elif kind == "reference":
        ref, blob = Reference(), open(model_path, "rb").read()
        run = lambda c, it: ref.time_graph(blob, c, it)  # reference extract_mfcc_features + reference op registrations  # noqa: E731
    else:
        run = OracleModel(o, model_path).time_run
    run(clips[:4], 1)
    done, spent = 0, 0.0
    while spent < seconds:
        spent += run(clips, 1)
        done += n_clips
    print(done / spent)


def usable_cores():
    """CPUs this process may really use: affinity mask capped by the cgroup CPU quota (containers)."""
    n = len(os.sched_getaffinity(0))
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", None)):
        try:
            if parse:
                quota, period = parse(open(path).read())
            else:
                quota = open(path).read().strip()
                period = open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()
            if quota not in ("max", "-1"):
                n = max(1, min(n, int(float(quota) / float(period))))
            break
        except Exception:
            continue
    return n


def cpu_baseline(seconds=8.0, model_path=DEFAULT_MODEL):
    """The reference SDK (oracle/_ref, compiled from the unmodified sources) on the host cores, one PROCESS per core
    (the reference keeps state in globals: non-reentrant), for a bounded time.  The shipped model runs through the
    reference's run_classifier(); any other model file through the reference's extract_mfcc_features() + the reference's
    TFLite-Micro op registrations driven by the model file (oracle/ref_driver.cpp eiref_time_graph_classifier)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from kws_testlib import have_reference
    kind = "reference" if have_reference() else "port"
    cores = usable_cores()
    n_clips = 64
    cmd = [sys.executable, os.path.abspath(__file__), "--model", model_path, "--cpu-worker", kind, str(n_clips)]
    single = subprocess.run(cmd + ["2.0"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    try:
        single = float(single.stdout.strip().splitlines()[-1])
    except Exception:
        single = float("nan")
    t0 = time.time()
    procs = [subprocess.Popen(cmd + [str(seconds)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
             for _ in range(cores)]
    rates = []
    for p in procs:
        out, _ = p.communicate()
        try:
            rates.append(float(out.strip().splitlines()[-1]))
        except Exception:
            pass
    wall = time.time() - t0
    return {"value": round(sum(rates), 1), "unit": "clips/s", "cores": len(rates), "kind": kind,
            "per_core": round(sum(rates) / max(1, len(rates)), 1), "single_process": round(single, 1),
            "kwsm_file": os.path.basename(model_path),
            "sample": "%d concurrent processes, each looping the reference's MFCC + network over %d seed-0 synthetic clips "
                      "for %.0f s (%.1f s wall incl. start-up)" % (len(rates), n_clips, seconds, wall)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=65536, help="clips per GPU per step")
    ap.add_argument("--model", default=DEFAULT_MODEL)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the extra workloads timed at N = 1")
    ap.add_argument("--force-collective", action="store_true",
                    help="initialise RCCL and all-gather the scores even with one rank (smoke test of the N > 1 path on a 1-GPU box)")
    ap.add_argument("--cpu-worker", nargs=3, metavar=("KIND", "N_CLIPS", "SECONDS"))
    a = ap.parse_args()
    if a.cpu_worker:
        cpu_worker(a.cpu_worker[0], int(a.cpu_worker[1]), float(a.cpu_worker[2]), a.model)
        return

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(model_path=a.model)                     # before the GPU is touched: children never see a HIP context

    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_package
    pkg = load_package()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or a.force_collective
    saved_stdout = None
    if use_dist:
        # RCCL prints a version banner on stdout when it initialises; this process's stdout carries exactly one JSON line,
        # so fd 1 points at stderr while the process group exists
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        if "MASTER_ADDR" not in os.environ:                        # plain `python bench.py --force-collective`
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29531"), RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=dev)
    B = a.batch
    n = 16000
    # this rank's shard of the global batch: clips [rank*B, (rank+1)*B), resident in HBM before timing
    pcm = torch.empty((B, n), dtype=torch.int16, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    pkg.synth_clips_device(0, rank * B, B, n, pcm.data_ptr(), stream)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(model_path, steps, warmup, collective):
        """K timed steps of the hot path over the resident batch with the given model; returns the result fields."""
        model = pkg.Model(model_path, device=local_rank)
        assert model.clip_samples == n
        C, F = model.n_labels, model.n_features
        feats = torch.empty((B, F), dtype=torch.float32, device=dev)
        is_float = model.is_float
        q = None if is_float else torch.empty((B, F), dtype=torch.int8, device=dev)
        scores = torch.empty((B, C), dtype=torch.float32, device=dev)
        gathered = torch.empty((world * B, C), dtype=torch.float32, device=dev) if collective else scores
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]

        def step(k=None):
            if k is not None:
                ev[k][0].record()
            model.extract_mfcc_batch_device(pcm.data_ptr(), B, feats.data_ptr(), None if is_float else q.data_ptr(), stream)   # extract_mfcc_features
            if k is not None:
                ev[k][1].record()
            if is_float:
                model.run_inference_batch_device(feats.data_ptr(), B, scores.data_ptr(), stream)      # the float network
            else:
                model.nn_batch_device(q.data_ptr(), B, scores.data_ptr(), stream)                     # the int8 network
            if k is not None:
                ev[k][2].record()
            if collective:
                dist.all_gather_into_tensor(gathered, scores)

        for _ in range(warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for k in range(steps):
            step(k)
        fence()
        dt = time.perf_counter() - t0
        if collective:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        res = {"model": os.path.basename(model_path), "dt": dt, "is_float": is_float, "nn_kernel": model.nn_kernel,
               "ms_mfcc": sum(e[0].elapsed_time(e[1]) for e in ev) / steps,
               "ms_nn": sum(e[1].elapsed_time(e[2]) for e in ev) / steps,
               "checksum": float(gathered.double().sum().item()), "labels": C}
        model.close()
        return res

    r = measure(a.model, a.steps, a.warmup, use_dist)
    also = []
    if world == 1 and not a.no_also:
        for mp in ALSO_MODELS:
            if not os.path.samefile(mp, a.model):
                also.append(measure(mp, a.steps, a.warmup, False))

    if use_dist:
        dist.destroy_process_group()
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)                             # the banner may still sit in the C library's buffer
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    if rank == 0:
        dt, ms_mfcc, ms_nn, is_float = r["dt"], r["ms_mfcc"], r["ms_nn"], r["is_float"]
        algo_bytes = 16000 * 2 + r["labels"] * 4          # int16 PCM in + C float scores out, per clip
        achieved = algo_bytes * B / (ms_mfcc * 1e-3) / 1e9
        # HBM traffic of the dominant kernel: rocprofv3 PMC passes of this same command (profiles/r01_pmc/), per launch
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc", "traffic.json")))
            if pmc.get("batch") == B and pmc.get("model") == r["model"]:
                traffic = pmc["kernels"]["kws_mfcc_kernel"]["traffic_bytes"]
        except Exception:
            pass

        # what actually bounds the kernel: rocprofv3 SQ counters of this same command (profiles/r01h_pmc_util/), headline model only
        valu = None
        try:
            if traffic is not None:
                u = json.load(open(os.path.join(ROOT, "profiles", "r01h_pmc_util", "summary.json")))["kernels"]["kws_mfcc_kernel"]
                valu = {"VALUBusy_pct": round(u["VALUBusy"], 1), "VALUUtilization_pct": round(u["VALUUtilization"], 1),
                        "valu_instructions_per_clip": round(u["VALU_instructions_per_clip"]),
                        "source": "profiles/r01h_pmc_util/summary.json (rocprofv3 --pmc, separate passes)"}
        except Exception:
            pass

        def workload(name):
            return WORKLOADS.get(name, name) + "; %d clips of 1 s @ 16 kHz int16 per GPU resident in HBM" % B

        def parity(fl):
            return ("MFCC features + logits bit-exact, scores <= 1e-6 vs the reference's float kernels" if fl
                    else "bit-exact vs reference") + " (tests/test_gpu_parity.py)"
        out = {
            "metric": "1s@16kHz clips/sec (MFCC+CNN)", "value": round(world * B * a.steps / dt, 1), "unit": "clips/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32+f64 (MFCC) / %s (CNN)" % ("f32" if is_float else "i8"), "data": "synthetic",
            "config": {"workload": workload(r["model"]), "clips_per_gpu": B, "global_batch": world * B, "kwsm_file": r["model"],
                       "parity": parity(is_float), "collective": "all_gather(scores) over RCCL" if use_dist else "none"},
            "roofline": {"bound": "hbm", "kernel": "kws_mfcc_kernel", "achieved": round(achieved, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": traffic, "traffic_unit": "bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_pmc)",
                         "algorithmic_bytes_per_launch": algo_bytes * B, "algorithmic_bytes_per_clip": algo_bytes,
                         "kernel_ms": {"kws_mfcc_kernel": round(ms_mfcc, 4), r["nn_kernel"]: round(ms_nn, 4)},
                         "note": "the kernel is VALU-issue-bound (order-constrained fp32/fp64 arithmetic of the reference), not HBM-bound; see valu",
                         "valu": valu},
            "checksum": r["checksum"],
        }
        if also:
            out["also"] = [{"kwsm_file": x["model"], "workload": workload(x["model"]), "value": round(B * a.steps / x["dt"], 1),
                            "unit": "clips/s", "ms_per_step": round(x["dt"] / a.steps * 1e3, 4),
                            "dtype": "f32+f64 (MFCC) / %s (CNN)" % ("f32" if x["is_float"] else "i8"), "parity": parity(x["is_float"]),
                            "kernel_ms": {"kws_mfcc_kernel": round(x["ms_mfcc"], 4), x["nn_kernel"]: round(x["ms_nn"], 4)},
                            "hbm_frac_mfcc_kernel": round((16000 * 2 + x["labels"] * 4) * B / (x["ms_mfcc"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
                           for x in also]
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out))


if __name__ == "__main__":
    main()
