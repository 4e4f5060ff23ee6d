#!/usr/bin/env python3
"""bench.py -- clips/sec of the run_classifier() hot path (MFCC + int8 CNN) on N MI355X.

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
ALGO_BYTES_PER_CLIP = 16000 * 2 + 4 * 4      # int16 PCM in + C=4 float scores out = 32 016 B
HBM_PEAK_GBS = 8000.0                        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


DEFAULT_MODEL = os.path.join(ROOT, "models", "l476_no_yes.kwsm")


def cpu_worker(kind, n_clips, seconds, model_path=DEFAULT_MODEL):
    """Child process: run the CPU path over n_clips synthetic clips again and again for ~`seconds`; prints clips/s."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from kws_testlib import MODELS, Oracle, OracleModel, Reference
    o = Oracle()
    clips = o.synth(0, 0, n_clips)
    runner = Reference() if kind == "reference" else OracleModel(o, model_path)
    runner.time_run(clips[:4], 1)
    done, spent = 0, 0.0
    while spent < seconds:
        spent += runner.time_run(clips, 1)
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
    (the reference keeps state in globals: non-reentrant), for a bounded time."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from kws_testlib import have_reference
    # oracle/_ref is the reference built around ITS shipped model; any other model file is timed on the C restatement
    kind = "reference" if have_reference() and os.path.samefile(model_path, DEFAULT_MODEL) else "port"
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
            "sample": "%d concurrent processes, each looping run_classifier() over %d seed-0 synthetic clips for %.0f s "
                      "(%.1f s wall incl. start-up)" % (len(rates), n_clips, seconds, wall)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=65536, help="clips per GPU per step")
    ap.add_argument("--model", default=DEFAULT_MODEL)
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    model = pkg.Model(a.model, device=local_rank)
    B, n, C, F = a.batch, model.clip_samples, model.n_labels, model.n_features

    # this rank's shard of the global batch: clips [rank*B, (rank+1)*B), resident in HBM before timing
    pcm = torch.empty((B, n), dtype=torch.int16, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    pkg.synth_clips_device(0, rank * B, B, n, pcm.data_ptr(), stream)
    feats = torch.empty((B, F), dtype=torch.float32, device=dev)
    is_float = model.is_float
    q = None if is_float else torch.empty((B, F), dtype=torch.int8, device=dev)
    scores = torch.empty((B, C), dtype=torch.float32, device=dev)
    gathered = torch.empty((world * B, C), dtype=torch.float32, device=dev) if world > 1 else scores

    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(a.steps)]

    def step(k=None):
        if k is not None:
            ev[k][0].record()
        model.extract_mfcc_batch_device(pcm.data_ptr(), B, feats.data_ptr(), None if is_float else q.data_ptr(), stream)   # extract_mfcc_features
        if k is not None:
            ev[k][1].record()
        if is_float:
            model.run_inference_batch_device(feats.data_ptr(), B, scores.data_ptr(), stream)          # the float network
        else:
            model.nn_batch_device(q.data_ptr(), B, scores.data_ptr(), stream)                         # the int8 network
        if k is not None:
            ev[k][2].record()
        if world > 1:
            dist.all_gather_into_tensor(gathered, scores)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for k in range(a.steps):
        step(k)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        ms_mfcc = sum(e[0].elapsed_time(e[1]) for e in ev) / a.steps
        ms_nn = sum(e[1].elapsed_time(e[2]) for e in ev) / a.steps
        achieved = ALGO_BYTES_PER_CLIP * B / (ms_mfcc * 1e-3) / 1e9
        checksum = float(gathered.double().sum().item())
        # HBM traffic of the dominant kernel: rocprofv3 PMC passes of this same command (profiles/r01_pmc/), per launch
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc", "traffic.json")))
            if pmc.get("batch") == B:
                traffic = pmc["kernels"]["kws_mfcc_kernel"]["traffic_bytes"]
        except Exception:
            pass
        out = {
            "metric": "1s@16kHz clips/sec (MFCC+CNN)", "value": round(world * B * a.steps / dt, 1), "unit": "clips/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32+f64 (MFCC) / %s (CNN)" % ("f32" if is_float else "i8"), "data": "synthetic",
            "config": {"workload": "%s 4-class no/noise/unknown/yes impulse (MFCC 49x13: 32 mel, fft 256, CMVN 101; "
                                   "%s 2-Conv CNN), %d clips of 1 s @ 16 kHz int16 per GPU resident in HBM"
                                   % ("de-quantised fp32 twin of the shipped" if is_float else "shipped", "fp32" if is_float else "int8", B),
                       "clips_per_gpu": B, "global_batch": world * B, "model": os.path.basename(a.model),
                       "parity": ("features+logits bit-exact, scores <= 1e-6 vs reference float kernels" if is_float
                                  else "bit-exact vs reference") + " (tests/test_gpu_parity.py)",
                       "collective": "all_gather(scores) over RCCL" if world > 1 else "none"},
            "roofline": {"bound": "hbm", "kernel": "kws_mfcc_kernel", "achieved": round(achieved, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": traffic, "traffic_unit": "bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_pmc)",
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_CLIP * B, "algorithmic_bytes_per_clip": ALGO_BYTES_PER_CLIP,
                         "kernel_ms": {"kws_mfcc_kernel": round(ms_mfcc, 4), model.nn_kernel: round(ms_nn, 4)}},
            "checksum": checksum,
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
