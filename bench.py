#!/usr/bin/env python3
"""bench.py -- clips/sec of the run_classifier() hot path (MFCC + CNN) on N MI355X.

Headline workload = BASELINE.json configs[1] as worded: batch 65 536 synthetic 1 s @ 16 kHz clips, 40-band MFCC (49x40),
2-Conv CNN, fp32, 1x MI355X.  The reference ships no such model, so the graph is generated with
seeded weights (tools/synth_model.py + tools/dequantize_model.py -> models/cfg2_mfcc40_f32.kwsm); its DSP block is the
reference's own code path for that configuration.

Two arithmetic modes of the same library are timed (include/kws/kws.h):
  * KWS_MODE_FAST  -- the headline `value`: scores within 1e-4 of the reference's (the tolerance BASELINE.json's north_star
    states; tests/test_gpu_fast_mode.py holds every one of 65 536 clips to it, tests/test_gpu_fast_families.py nine adversarial
    input families), MFCC + network in ONE kernel launch; clips whose cmvnw would amplify the re-ordering past that are re-run by
    the exact kernels inside the same call (their share is reported as config.fast_fallback_rate);
  * KWS_MODE_EXACT -- MFCC features bit-identical to the reference's, scores within 1e-6; reported under "modes".
At N = 1 the same run also times the model the reference DOES ship (49x13 MFCC, int8) and the other BASELINE configurations;
they are reported under "also".  BASELINE configs[3] ("int8 ... bit-exact check, batch 65 536, 1 GPU") is the line "int8_exact": the
reference's own int8 impulse in KWS_MODE_EXACT -- bit-identical to the reference end to end -- with a roofline object of its own.  The
KWS_MODE_FAST lines of int8 graphs are NOT that configuration (an int8 input value may sit one step away): they say so.

One "step" = one pass of the hot path over one batch of B synthetic 1 s @ 16 kHz int16 clips per GPU, the clips already
resident in HBM (generated on the device by kws_synth_clips_device).  N > 1: one process per GPU, clips sharded contiguously
(rank r owns clips [r*B, (r+1)*B)), no data-path collective except the all-gather of the per-clip scores (16 B/clip) over RCCL
(kws_allgather_scores, the library's C ABI on librccl), which is inside the timed region and also timed on its own.
`python bench.py --gpus N` without a launcher starts the N ranks itself (torch.distributed.run, 127.0.0.1).  Default clips
per GPU: 65 536 (configs[1]); 131 072 at N = 8, i.e. BASELINE configs[2]'s 1 M clips over 8 GPUs.

Prints ONE JSON line (rank 0).  See DESIGN.md "Measurement" for how roofline / cpu_baseline are defined.
"""
import argparse
import hashlib
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
    "cfg2_mfcc40_int8.kwsm": "the configs[1] graph quantised: 49x40 MFCC + int8 2-Conv CNN, seeded synthetic weights",
    "l476_no_yes.kwsm": "the impulse the reference ships, 4-class no/noise/unknown/yes (MFCC 49x13: 32 mel, fft 256, CMVN 101; int8 2-Conv CNN)",
    "l476_no_yes_f32.kwsm": "de-quantised fp32 twin of the shipped impulse (MFCC 49x13 + fp32 2-Conv CNN)",
    "cfg5_dscnn_mfcc40_int8.kwsm": "BASELINE configs[4] shape: 49x40 MFCC + 7-block depthwise-separable CNN, 12 labels, int8, synthetic weights",
    "cfg5_dscnn_mfcc40_f32.kwsm": "BASELINE configs[4] shape: 49x40 MFCC + 7-block depthwise-separable CNN, 12 labels, fp32, synthetic weights",
}
WORKLOADS_SHORT = {
    "cfg2_mfcc40_f32.kwsm": "BASELINE configs[1]: 49x40 MFCC + 2-Conv CNN fp32, synthetic weights",
    "cfg2_mfcc40_int8.kwsm": "configs[1] graph quantised to int8",
    "l476_no_yes.kwsm": "BASELINE configs[3]: the shipped int8 impulse (49x13 MFCC + 2-Conv CNN)",
    "l476_no_yes_f32.kwsm": "fp32 twin of the shipped impulse",
    "cfg5_dscnn_mfcc40_int8.kwsm": "BASELINE configs[4] shape, int8",
    "cfg5_dscnn_mfcc40_f32.kwsm": "BASELINE configs[4] shape, fp32",
}
CLIP_LEN = 16000
# also_inputs: the headline graph on inputs that are not the bench's noise-floored synthetic clips (tests/kws_families.py)
INPUT_FAMILIES = ("word_noise_gain", "word_background", "word_silence", "amp_sweep", "bursts", "quiet_noise")
N_BASE = 2048


def free_port():
    """a TCP port nobody is bound to right now on 127.0.0.1 (asked of the kernel, not derived from the pid)"""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def default_batch(world):
    """clips per GPU: configs[1]'s 65 536; at 8 GPUs configs[2]'s 1 M clips / 8"""
    return 131072 if world == 8 else 65536


def lib_sha256():
    p = os.path.join(ROOT, "ei-keyword-spotting_amd", "libkws_mi355x.so")
    try:
        return hashlib.sha256(open(p, "rb").read()).hexdigest()
    except OSError:
        return None


# ---------------------------------------------------------------------------------------------------------------------
#  CPU baseline (N = 1 only): the reference itself, compiled from its own sources (oracle/_ref), on the host cores
# ---------------------------------------------------------------------------------------------------------------------
def cpu_worker(kind, n_clips, seconds, model_path=DEFAULT_MODEL, first_clip=0):
    """Child process: run the CPU path over n_clips DISTINCT synthetic clips (again if time remains) for ~`seconds`."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from kws_testlib import Oracle, OracleModel, Reference
    o = Oracle()
    clips = o.synth(0, first_clip, n_clips)
    if kind == "reference" and os.path.samefile(model_path, SHIPPED_MODEL):
        run = Reference().time_run                       # the reference's run_classifier() with its compiled-in model
    elif kind == "reference":
        ref, blob = Reference(), open(model_path, "rb").read()
        run = lambda c, it: ref.time_graph(blob, c, it)  # reference extract_mfcc_features + reference op registrations  # noqa: E731
    else:
        run = OracleModel(o, model_path).time_run
    run(clips[:4], 1)
    done, spent, chunk = 0, 0.0, 250
    while spent < seconds:
        for i in range(0, n_clips, chunk):
            spent += run(clips[i:i + chunk], 1)
            done += len(clips[i:i + chunk])
            if spent >= seconds and done >= n_clips:
                break
    print(done / spent, done)


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


def cpu_model_name():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(seconds=10.0, model_path=DEFAULT_MODEL, n_clips=2000):
    """The reference SDK (oracle/_ref, compiled from the unmodified sources) on the host cores, one PROCESS per core (the
    reference keeps state in globals: non-reentrant), each over its own 2 000 distinct seed-0 clips, for a bounded time.  The
    shipped model runs through the reference's run_classifier(); any other model file through the reference's
    extract_mfcc_features() + the reference's TFLite-Micro op registrations driven by the model file."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from kws_testlib import have_reference
    kind = "reference" if have_reference() else "port"
    cores = usable_cores()
    cmd = [sys.executable, os.path.abspath(__file__), "--model", model_path, "--cpu-worker", kind, str(n_clips)]
    t0 = time.time()
    procs = [subprocess.Popen(cmd + [str(seconds), str(k * n_clips)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
             for k in range(cores)]
    rates, clips = [], 0
    for p in procs:
        out, _ = p.communicate()
        try:
            r, d = out.strip().splitlines()[-1].split()
            rates.append(float(r))
            clips += int(d)
        except Exception:
            pass
    wall = time.time() - t0
    return {"value": round(sum(rates), 1), "unit": "clips/s", "cores": len(rates), "kind": kind, "cpu": cpu_model_name(),
            "per_core": round(sum(rates) / max(1, len(rates)), 1), "kwsm_file": os.path.basename(model_path),
            "sample_short": "%d processes x %d distinct synthetic clips, >= %.0f s each, %d clips" % (len(rates), n_clips, seconds, clips),
            "sample": "%d concurrent processes (one per usable core), each running the reference's MFCC + network over its own %d "
                      "distinct seed-0 synthetic clips for >= %.0f s: %d clips in all (%.1f s wall incl. start-up)"
                      % (len(rates), n_clips, seconds, clips, wall)}


# ---------------------------------------------------------------------------------------------------------------------
#  the timed region.  A backend owns one rank's resident batch and knows how to run one step on it; GpuBackend is the product
#  (libkws_mi355x.so through its C ABI), CpuOracleBackend exists for the world-size-2 gloo test and --dry-run-cpu only.
# ---------------------------------------------------------------------------------------------------------------------
DSP_B = 8192
DSP_SHAPES = (("fft 512, 49 frames of 20 ms, 32 filters, 13 cepstra", dict(fft_length=512)),
              ("fft 512, 2 s windows (99 frames)", dict(fft_length=512, raw_samples=32000, blocks=((8, 3, 1), (4, 3, 1)))),
              ("fft 128, 49 frames, cmvnw window 51", dict(fft_length=128, win_size=51)),
              ("fft 256, frames every 10 ms (98 frames: the tuned spectral kernel over two chunks + the general cmvnw)", dict(frame_stride=0.01, win_size=31, blocks=((8, 3, 1), (4, 3, 1)))),
              ("fft 1024, 50 ms frames, 36 filters, 17 cepstra", dict(fft_length=1024, num_filters=36, ncep=17, frame_length=0.05, frame_stride=0.025, win_size=21,
                                                                   blocks=((8, 3, 1), (4, 3, 1)))))


def also_dsp(backend, pkg, calls=20):
    """ms per call of extract_mfcc_features on general-shape DSP configurations (models with synthetic weights from tools/synth_model.py: the
    network is not run)."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from synth_model import synth_model_blob
    rows = []
    for name, kw in DSP_SHAPES:
        gm = pkg.Model(blob=synth_model_blob(seed=3, **dict(dict(blocks=((8, 3, 7), (4, 3, 7)), n_labels=3), **kw)), device=backend.local_rank)
        ns = gm.clip_samples
        pcm = torch.empty((DSP_B, ns), dtype=torch.int16, device=backend.dev)
        pkg.synth_clips_device(0, 0, DSP_B, ns, pcm.data_ptr(), backend.stream)
        ft = torch.zeros((DSP_B, gm.n_features), dtype=torch.float32, device=backend.dev)
        for _ in range(10):                                           # (the handle measures its chunk length on its first large calls)
            gm.extract_mfcc_batch_device(pcm.data_ptr(), DSP_B, ft.data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(calls):
            gm.extract_mfcc_batch_device(pcm.data_ptr(), DSP_B, ft.data_ptr())
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / calls * 1e3
        rows.append({"dsp": name, "kernel": gm.mfcc_kernel, "frames": gm.n_frames, "clips_per_call": DSP_B, "ms_per_call": round(ms, 4),
                     "clips_per_s": round(DSP_B / (ms * 1e-3), 1), "ns_per_frame": round(ms * 1e6 / (DSP_B * gm.n_frames), 3)})
        gm.close()
        del pcm, ft
    return rows


class GpuBackend:
    def __init__(self, pkg, local_rank, rank, world, B, use_comm):
        import torch
        self.torch, self.pkg = torch, pkg
        self.rank, self.world, self.B = rank, world, B
        torch.cuda.set_device(local_rank)
        self.dev = torch.device("cuda", local_rank)
        self.local_rank = local_rank
        self.pcm = torch.empty((B, CLIP_LEN), dtype=torch.int16, device=self.dev)
        self.stream = torch.cuda.current_stream().cuda_stream
        pkg.synth_clips_device(0, rank * B, B, CLIP_LEN, self.pcm.data_ptr(), self.stream)     # this rank's shard
        self.comm = None
        self.use_comm = use_comm
        self.model = None
        self.pcm_override = None                  # another resident batch (the input families of also_inputs)

    def make_comm(self, unique_id):
        self.comm = self.pkg.Comm(unique_id, self.world, self.rank, self.local_rank)

    def load(self, model_path, mode):
        torch = self.torch
        self.close_model()
        m = self.model = self.pkg.Model(model_path, device=self.local_rank)
        assert m.clip_samples == CLIP_LEN
        m.set_mode(self.pkg.MODE_FAST if mode == "fast" else self.pkg.MODE_EXACT)
        self.scores = torch.empty((self.B, m.n_labels), dtype=torch.float32, device=self.dev)
        self.gathered = torch.empty((self.world * self.B, m.n_labels), dtype=torch.float32, device=self.dev) if self.use_comm else self.scores
        tol = m.fast_tolerance() if mode == "fast" else None
        if tol and tol["dev_overrides"]:
            raise SystemExit("bench.py: a KWS_DEV_FAST_* development switch is set in the environment: KWS_MODE_FAST would be outside its tolerance; refusing to measure")
        return {"labels": m.n_labels, "is_float": m.is_float, "nn_kernel": m.nn_kernel, "fused": bool(m.fast_is_fused and mode == "fast"),
                "entry_tier": tol["entry_tier"] if tol else None, "fused_wps": tol.get("fused_waves_per_simd") if tol else None,
                "guard": {k: tol[k] for k in ("k_sigma", "score_tol", "total_gain", "uniform_feature_tol", "calibrated")} if tol else None}

    def events(self, steps):
        self.ev = [[self.torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]

    def step(self, k=None):
        m = self.model
        if k is not None:
            self.ev[k][0].record()
        pcm = self.pcm if self.pcm_override is None else self.pcm_override
        m.run_classifier_batch_device(pcm.data_ptr(), self.B, self.scores.data_ptr(), None, None, self.stream)   # the hot path
        if k is not None:
            self.ev[k][1].record()
        if self.use_comm:
            self.comm.allgather_scores(self.scores.data_ptr(), self.gathered.data_ptr(), self.B, m.n_labels, self.stream)
        if k is not None:
            self.ev[k][2].record()

    def sync(self):
        self.torch.cuda.synchronize()

    def prewarm(self, seconds):
        """un-counted steps until `seconds` of wall time with the GPU busy have passed: the clocks ramp (the driver's 5 warm-up steps are
        9 ms behind a 10 s CPU-baseline leg with the GPU idle) before the counted warm-up starts; returns how many steps that took"""
        n, t0 = 0, time.perf_counter()
        pcm = self.pcm if self.pcm_override is None else self.pcm_override
        while time.perf_counter() - t0 < seconds:
            for _ in range(8):      # the hot path only: a time-based count differs between ranks, so no collective in here
                self.model.run_classifier_batch_device(pcm.data_ptr(), self.B, self.scores.data_ptr(), None, None, self.stream)
            self.sync()
            n += 8
        return n

    def phase_ms(self, steps):
        return (sum(e[0].elapsed_time(e[1]) for e in self.ev) / steps, sum(e[1].elapsed_time(e[2]) for e in self.ev) / steps)

    def checksum(self):
        return float(self.gathered.double().sum().item())

    def checksum_class0(self):
        # the probability mass of class 0 over every clip: unlike the plain sum (= the number of softmax rows) it depends on the scores
        return float(self.gathered[:, 0].double().sum().item())

    def fallback(self):
        return self.model.fast_fallback_count()

    def exact_count(self):
        return self.model.fast_exact_count()

    def comm_info(self):
        """what RCCL itself says about the communicator (not what the launcher said)"""
        if self.comm is None:
            return None
        self.comm.wait(self.stream)                                   # deadline-guarded: a missing / failed rank is an error here, not a hang
        return {"ranks_seen_by_rccl": self.comm.ranks_seen_by_rccl, "rccl_version": self.comm.rccl_version}

    def close_model(self):
        if self.model is not None:
            self.model.close()
            self.model = None


class CpuOracleBackend:
    """Test double: the oracle on the host in place of the GPU library, torch.distributed (gloo) in place of RCCL.  Used by
    tests/test_distributed_cpu.py and --dry-run-cpu to exercise the sharding / gather / timing code that bench.py runs."""

    def __init__(self, rank, world, B):
        import torch
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from kws_testlib import Oracle
        self.torch = torch
        self.rank, self.world, self.B = rank, world, B
        self.pcm_override = None
        self.oracle = Oracle()
        self.pcm = self.oracle.synth(0, rank * B, B)                 # this rank's shard, same generator as the device's
        self.use_comm = world > 1
        self.t = [0.0, 0.0]

    def make_comm(self, unique_id):
        pass

    def load(self, model_path, mode):
        from kws_testlib import OracleModel
        self.model = OracleModel(self.oracle, model_path)
        self.scores = self.torch.zeros((self.B, self.model.n_labels), dtype=self.torch.float32)
        self.gathered = self.torch.zeros((self.world * self.B, self.model.n_labels), dtype=self.torch.float32) if self.use_comm else self.scores
        return {"labels": self.model.n_labels, "is_float": bool(self.oracle.L.kwso_model_is_float(self.model.h)),
                "nn_kernel": "oracle (CPU test double)", "fused": False, "entry_tier": None, "fused_wps": None, "guard": None}

    def events(self, steps):
        self.t = [0.0, 0.0]

    def step(self, k=None):
        import torch.distributed as dist
        t0 = time.perf_counter()
        self.scores.copy_(self.torch.from_numpy(self.model.run_batch(self.pcm)))
        t1 = time.perf_counter()
        if self.use_comm:
            dist.all_gather_into_tensor(self.gathered, self.scores)
        if k is not None:
            self.t[0] += (t1 - t0) * 1e3
            self.t[1] += (time.perf_counter() - t1) * 1e3

    def sync(self):
        pass

    def prewarm(self, seconds):
        return 0

    def phase_ms(self, steps):
        return self.t[0] / steps, self.t[1] / steps

    def checksum(self):
        return float(self.gathered.double().sum().item())

    def checksum_class0(self):
        return float(self.gathered[:, 0].double().sum().item())

    def fallback(self):
        return 0

    def exact_count(self):
        return 0

    def comm_info(self):
        import torch.distributed as dist
        return {"ranks_seen_by_rccl": None, "rccl_version": None, "gloo_world_size": dist.get_world_size()} if self.use_comm else None

    def close_model(self):
        pass


def timed_steps(backend, steps, warmup, barrier, max_over_ranks, own=None):
    """W untimed steps, then exactly K steps bracketed by barrier + device synchronisation on both sides; the MAX over ranks.
    own (a list): receives this rank's own time for its K steps, up to its own synchronisation and before the closing barrier -- the
    per-rank figure behind the skew reported for N > 1."""
    backend.events(steps)
    for _ in range(warmup):
        backend.step()
    backend.sync(); barrier(); backend.sync()
    t0 = time.perf_counter()
    for k in range(steps):
        backend.step(k)
    backend.sync()
    if own is not None:
        own.append(time.perf_counter() - t0)
    barrier(); backend.sync()
    return max_over_ranks(time.perf_counter() - t0)


PREWARM_S = 0.3       # un-counted, time-based (clock ramp); the W warm-up and K timed steps that follow are exactly the ones asked for


def measure(backend, model_path, mode, steps, warmup, barrier, max_over_ranks, gather_ranks=None, pcm=None):
    info = backend.load(model_path, mode)
    if pcm is not None:
        backend.pcm_override = pcm
    info["prewarm_steps"] = backend.prewarm(PREWARM_S)
    own = []
    dt = timed_steps(backend, steps, warmup, barrier, max_over_ranks, own)
    ms_path, ms_gather = backend.phase_ms(steps)
    res = dict(info, model=os.path.basename(model_path), mode=mode, dt=dt, ms_path=ms_path, ms_gather=ms_gather, checksum=backend.checksum(), checksum0=backend.checksum_class0(),
               fallback=backend.fallback() if mode == "fast" else 0, exact_count=backend.exact_count() if mode == "fast" else 0,
               comm=backend.comm_info(), rank_dt=gather_ranks(own[0]) if gather_ranks else [own[0]])
    backend.close_model()
    backend.pcm_override = None
    return res


def pmc_for(kernels, model, batch, mode="fast"):
    """HBM traffic / SQ counters of the hot path's kernels (`kernels`: the dominant one first) from rocprofv3 PMC passes of this same
    command (tools/profile_round.sh), used only when they were taken with the library that is running now (SHA-256 of libkws_mi355x.so
    recorded next to them).  -> (directory, {traffic_bytes: summed over `kernels`, per_kernel}, SQ figures of the dominant kernel) or None"""
    sha = lib_sha256()
    best = None
    prof = os.path.join(ROOT, "profiles")
    for d in sorted(os.listdir(prof)) if os.path.isdir(prof) else []:
        f = os.path.join(prof, d, "traffic.json")
        try:
            j = json.load(open(f))
        except Exception:
            continue
        ks = j.get("kernels", {})
        if j.get("lib_sha256") == sha and j.get("model") == model and j.get("batch") == batch and j.get("mode", "fast") == mode and kernels[0] in ks:
            per = {k: ks[k]["traffic_bytes"] for k in kernels if k in ks}
            best = (os.path.join("profiles", d), {"traffic_bytes": sum(per.values()), "per_kernel": per}, j.get("sq", {}).get(kernels[0]))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=600, help="timed steps (default: > 1 s of GPU time in the timed region)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=0, help="clips per GPU per step (default: 65536; 131072 at 8 GPUs = 1 M clips)")
    ap.add_argument("--model", default=DEFAULT_MODEL)
    ap.add_argument("--mode", choices=["fast", "exact"], default="fast", help="which arithmetic mode is the headline value")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the extra workloads / the other mode timed at N = 1")
    ap.add_argument("--force-collective", action="store_true",
                    help="create the RCCL communicator and all-gather the scores even with one rank (the N > 1 path on a 1-GPU box)")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="no GPU: the oracle stands in for the library and gloo for RCCL (exercises sharding, gather and timing code)")
    ap.add_argument("--cpu-worker", nargs=4, metavar=("KIND", "N_CLIPS", "SECONDS", "FIRST_CLIP"))
    a = ap.parse_args()
    if a.cpu_worker:
        cpu_worker(a.cpu_worker[0], int(a.cpu_worker[1]), float(a.cpu_worker[2]), a.model, int(a.cpu_worker[3]))
        return

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: start the N ranks ourselves, one per GPU, and pass rank 0's JSON line through
        port = os.environ.get("MASTER_PORT") or str(free_port())
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % a.gpus, "--master-addr", "127.0.0.1",
               "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    B = a.batch if a.batch > 0 else (64 if a.dry_run_cpu else default_batch(world))
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and not a.dry_run_cpu:
        cpu = cpu_baseline(model_path=a.model)                     # before the GPU is touched: children never see a HIP context

    import torch
    import torch.distributed as dist
    use_comm = world > 1 or a.force_collective
    # control plane (barrier, max over ranks, the communicator's id): gloo over 127.0.0.1; data plane: RCCL through the C ABI
    saved_stdout = None
    if use_comm:
        # gloo and RCCL print connection / version banners on stdout when they initialise; this process's stdout carries exactly
        # one JSON line, so fd 1 points at stderr while the process group and the communicator exist
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    def barrier():
        if use_comm:
            dist.barrier()

    def max_over_ranks(dt):
        if not use_comm:
            return dt
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def gather_ranks(dt):
        """every rank's own time for its K steps (control plane: gloo)"""
        if not use_comm:
            return [dt]
        ts = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(ts, torch.tensor([dt], dtype=torch.float64))
        return [float(t.item()) for t in ts]

    if a.dry_run_cpu:
        backend = CpuOracleBackend(rank, world, B)
    else:
        sys.path.insert(0, ROOT)
        from __graft_entry__ import load_package
        pkg = load_package()
        backend = GpuBackend(pkg, local_rank, rank, world, B, use_comm)
        if use_comm:
            ids = [pkg.Comm.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            backend.make_comm(ids[0])

    r = measure(backend, a.model, a.mode, a.steps, a.warmup, barrier, max_over_ranks, gather_ranks)
    others, also, int8_exact, inputs, dsp_shapes = [], [], None, [], []
    if world == 1 and not a.no_also and not a.dry_run_cpu:
        side_steps = max(20, a.steps // 8)
        others.append(measure(backend, a.model, "exact" if a.mode == "fast" else "fast", side_steps, a.warmup, barrier, max_over_ranks))
        for mp_ in ALSO_MODELS:
            if not os.path.samefile(mp_, a.model):
                for md in ("fast", "exact"):
                    also.append(measure(backend, mp_, md, side_steps, min(a.warmup, 5), barrier, max_over_ranks))
        # BASELINE configs[3]: the reference's own int8 impulse, bit-exact (KWS_MODE_EXACT), with as many timed steps as the headline
        int8_exact = measure(backend, SHIPPED_MODEL, "exact", max(side_steps, a.steps // 2), min(a.warmup, 5), barrier, max_over_ranks)
        # the headline graph OFF the synthetic distribution: input families of tests/kws_families.py, N_BASE distinct
        # clips each, tiled to the batch on the device.  How much of a batch the fast tiers keep depends on the input; the worst case
        # (every clip handed on) is the exact mode's rate.
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import numpy as np
        import kws_families
        for fam in INPUT_FAMILIES:
            base = torch.from_numpy(np.ascontiguousarray(kws_families.family(fam, N_BASE, seed=5))).to(backend.dev)
            fam_pcm = base.repeat((B + N_BASE - 1) // N_BASE, 1)[:B].contiguous()
            del base
            for md in ("fast", "exact"):
                x = measure(backend, a.model, md, side_steps, min(a.warmup, 5), barrier, max_over_ranks, pcm=fam_pcm)
                x["family"] = fam
                inputs.append(x)
            del fam_pcm

        # DSP configurations OUTSIDE the tuned shape (fft 512 / 128 / 1024, 2 s windows): extract_mfcc_features on the general-shape kernels
        # (kws_spectral_lds_kernel + kws_cmvn_lds_kernel, bit-exact, exact mode only) -- not a BASELINE configuration, reported
        # because SURVEY 8(f)2's ingestion accepts such models
        try:
            dsp_shapes = also_dsp(backend, pkg)
        except Exception as e:                                        # a side measurement must never cost the bench line
            dsp_shapes = [{"error": "%s: %s" % (type(e).__name__, e)}]

    if not a.dry_run_cpu and backend.comm is not None:
        backend.comm.close()
    if use_comm:
        dist.destroy_process_group()
    if saved_stdout is not None:
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)                             # a banner may still sit in the C library's buffer
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    if rank != 0:
        return

    def workload(name, mode="fast", is_float=True):
        # BASELINE configs[3] is the BIT-EXACT int8 configuration: only the KWS_MODE_EXACT lines of int8 graphs are it
        tag = "" if is_float else ("BASELINE configs[3] (bit-exact int8): " if mode == "exact" else "NOT a BASELINE configuration (KWS_MODE_FAST on an int8 graph is not bit-exact; "
                                                                                                   "configs[3] is the int8_exact line): ")
        return tag + WORKLOADS.get(name, name) + "; %d clips of 1 s @ 16 kHz int16 per GPU resident in HBM" % B

    def parity(x):
        if x["mode"] == "fast":
            return ("scores within 1e-4 of the reference's (KWS_MODE_FAST: which clips the fast tiers keep follows from the loaded graph's calibrated logit gain, "
                    "DESIGN.md 4.4.1; the others are finished by the exact kernels inside the call)" if x["is_float"] else
                    "KWS_MODE_FAST on an int8 graph is NOT bit-exact (MFCC within tolerance, the network bit-exact from the int8 tensor on, an input value may move "
                    "one step at a rounding boundary): this line is not BASELINE configs[3] -- see int8_exact") + " -- tests/test_gpu_fast_mode.py, tests/test_gpu_fast_families.py"
        return ("MFCC features + logits bit-exact, scores <= 1e-6 vs the reference's float kernels" if x["is_float"]
                else "bit-exact vs reference") + " -- tests/test_gpu_parity.py"

    def dominant(x):
        # (float32 graphs whose batch calls enter through the three-waves-per-SIMD build run kws_fast_kernel_w3: the second compilation of the same source)
        fast = "kws_fast_kernel_w3" if (x.get("fused") and x.get("fused_wps") == 3) else "kws_fast_kernel"
        return fast if x["mode"] == "fast" and x.get("entry_tier", 0) in (None, 0, 1) else "kws_mfcc8_kernel"

    def path_kernels(x):
        """the kernels of x's hot path that move data, the dominant one first (bench.py sums their counter traffic)"""
        if x["mode"] == "fast":
            # a float graph whose gain leaves the fast MFCC no room (entry tier >= 2) takes the exact features + the fused network
            return [dominant(x)] if dominant(x).startswith("kws_fast_kernel") else ["kws_mfcc8_kernel", "kws_fast_kernel"]
        return ["kws_mfcc8_kernel", x["nn_kernel"].split("<")[0].split(" ")[0]]

    def dtype(x):
        # the fused float network's contractions run on v_mfma_f32_16x16x32_f16 with every f32 operand carried as two f16 halves (22 bits) and three
        # products per pair, accumulated in f32: fp32-grade (scores within 6e-7 of a float64 evaluation, tools/split_operand_study.py), not an f16 network
        if x["mode"] == "fast":
            cnn = ("f32 (contractions: f32 operands as split f16 pairs on the matrix cores, f32 accumulate)" if x.get("fused") else "f32") if x["is_float"] else "i8"
            return "f32 (MFCC, KissFFT-order FFT) / %s (CNN)" % cnn
        return "f32+f64 (MFCC) / %s (CNN)" % ("f32" if x["is_float"] else "i8")

    algo_bytes = CLIP_LEN * 2 + r["labels"] * 4          # int16 PCM in + C float scores out, per clip
    achieved = algo_bytes * B / (r["ms_path"] * 1e-3) / 1e9
    pmc = pmc_for(path_kernels(r), r["model"], B, r["mode"]) if not a.dry_run_cpu else None
    compute = None
    if pmc and pmc[2] and pmc[2].get("compute"):
        compute = dict(pmc[2]["compute"])
    hbm_frac = achieved / HBM_PEAK_GBS
    # which pipe is closest to its ceiling (the judged figure stays the HBM fraction: SURVEY 8(d))
    binds = "hbm"
    if compute:
        cands = {"hbm": hbm_frac, "valu": max(compute.get("valu_issue_frac") or 0.0, compute.get("valu_active_frac") or 0.0),
                 "mfma": compute.get("mfma_busy_frac") or 0.0, "lds": compute.get("lds_busy_frac") or 0.0}
        binds = max(cands, key=cands.get)
    roof = {"bound": binds, "kernel": dominant(r), "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": pmc[1]["traffic_bytes"] if pmc else None,
            "traffic_unit": "bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE; only reported when the PMC passes under profiles/ were "
                            "taken with the library that is running: SHA-256 match)",
            "traffic_source": pmc[0] if pmc else None,
            "algorithmic_bytes_per_launch": algo_bytes * B, "algorithmic_bytes_per_clip": algo_bytes,
            "hot_path_ms": round(r["ms_path"], 4),
            "hot_path_ms_note": "HIP events on the launch stream around the hot-path call of every timed step (fast mode, fused graph: one "
                                "kws_fast_kernel launch + three empty-list launches of the exact kernels; exact mode: kws_mfcc8_kernel + the network kernel)",
            "bound_note": "`bound` names the pipe closest to its ceiling among HBM (achieved / peak), vector-ALU issue, matrix pipe and LDS (roofline.compute, from the "
                          "SQ counter passes of this library under profiles/; 'hbm' when no SHA-matched counters exist).  achieved / peak / frac are the HBM figures "
                          "SURVEY 8(d) asks for whatever binds: ~1 MFLOP per 32 KB clip is above the fp32 ridge, and at two waves per SIMD no pipe is full",
            "compute": compute,
            "valu": pmc[2] if pmc else None}
    out = {
        "metric": "1s@16kHz clips/sec (MFCC+CNN)", "value": round(world * B * a.steps / r["dt"], 1), "unit": "clips/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(r["dt"] / a.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype(r),
        "dtype_short": ("f32" if r["is_float"] else "f32 (MFCC) / i8 (CNN)") if r["mode"] == "fast" else ("f32+f64 (MFCC) / %s (CNN)" % ("f32" if r["is_float"] else "i8")),
        "data": "synthetic" + (" (DRY RUN ON CPU: oracle + gloo stand in for the GPU library + RCCL; not a measurement)" if a.dry_run_cpu else ""),
        "config": {"workload": workload(r["model"], r["mode"], r["is_float"]), "mode": r["mode"], "clips_per_gpu": B, "global_batch": world * B, "kwsm_file": r["model"],
                   "workload_short": WORKLOADS_SHORT.get(r["model"], r["model"]) + "; %d clips of 1 s @ 16 kHz per GPU in HBM" % B,
                   "parity": parity(r), "network_fused_into_mfcc_kernel": r["fused"], "clips_handed_on_by_the_first_fast_tier_last_step": r["fallback"],
                   "clips_finished_by_exact_kernels_last_step": r["exact_count"],
                   "fast_fallback_rate": round(r["fallback"] / float(B), 6), "fast_entry_tier": r["entry_tier"], "fast_guard": r["guard"],
                   "fast_fallback_note": "share of this workload's clips the entry tier of KWS_MODE_FAST handed on (the guard derived from the loaded graph's logit gain, "
                                         "DESIGN.md 4.4.1); other input families: profiles/r04_fast_families.txt (tests/test_gpu_fast_families.py)",
                   "collective": ("all_gather(scores) over RCCL (kws_allgather_scores, %d ranks)" % world) if use_comm else "none",
                   "lib_sha256": lib_sha256()},
        "collective": dict({"allgather_ms_per_step": round(r["ms_gather"], 4), "inside_timed_region": True, "ranks": world,
                            "per_rank_clips_per_s": [round(B * a.steps / t, 1) for t in r["rank_dt"]],
                            "rank_time_skew_max_over_min": round(max(r["rank_dt"]) / min(r["rank_dt"]), 5)}, **(r["comm"] or {})) if use_comm else None,
        "roofline": roof,
        "checksum": r["checksum"],
        "checksum_class0": r["checksum0"],
    }

    def line(x, steps):
        px = pmc_for(path_kernels(x), x["model"], B, x["mode"]) if not a.dry_run_cpu else None
        return {"traffic": px[1]["traffic_bytes"] if px else None, "traffic_per_kernel": px[1]["per_kernel"] if px else None, "traffic_source": px[0] if px else None,
                "kwsm_file": x["model"], "mode": x["mode"], "workload": workload(x["model"], x["mode"], x["is_float"]), "value": round(B * steps / x["dt"], 1),
                "fast_fallback_rate": round(x["fallback"] / float(B), 6), "fast_exact_rate": round(x["exact_count"] / float(B), 6), "fast_entry_tier": x["entry_tier"],
                "unit": "clips/s", "ms_per_step": round(x["dt"] / steps * 1e3, 4), "steps": steps, "dtype": dtype(x), "parity": parity(x),
                "network_kernel": "fused into kws_fast_kernel" if x["fused"] else x["nn_kernel"],
                "hbm_frac": round((CLIP_LEN * 2 + x["labels"] * 4) * B / (x["ms_path"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
    if others:
        out["modes"] = [line(r, a.steps)] + [line(x, side_steps) for x in others]
    if also:
        out["also"] = [line(x, side_steps) for x in also]
    if inputs:
        fam_note = {"word_noise_gain": "mix_audio-shaped (dataset-curation.py:129-135): a word of 0.2 .. 0.9 s at a random volume over a window of a background track at a random volume",
                    "word_background": "the same mix at the reference's default volumes (word 1.0, background 0.1)",
                    "word_silence": "a word followed by digital silence (what the reference's script makes of every file shorter than 1 s when no background is mixed in)",
                    "amp_sweep": "tone groups under an envelope, no noise floor, peak amplitude swept 1 .. 32767 LSB",
                    "bursts": "digital silence with 1 .. 6 bursts shorter than one frame",
                    "quiet_noise": "stationary noise of 1 .. 50 LSB and nothing else (a quiet room between words): every cepstral column is near-constant by nature"}
        out["also_inputs"] = [dict(line(x, side_steps), family=x["family"], family_is=fam_note.get(x["family"], ""),
                                   distinct_clips=N_BASE, tiled_to=B) for x in inputs]
        out["also_inputs_note"] = ("the headline graph (%s) on %d distinct clips of each input family of tests/kws_families.py, tiled to the batch; fast_fallback_rate = share the "
                                   "first fast tier handed on, fast_exact_rate = share finished by the exact kernels.  Worst case of KWS_MODE_FAST = every clip handed "
                                   "on = the exact mode's rate plus the fast tiers' attempt." % (r["model"], N_BASE))
    if dsp_shapes:
        out["also_dsp"] = dsp_shapes
        out["also_dsp_note"] = ("extract_mfcc_features (MFCC + cmvnw -> feature matrix, bit-exact) of synthetic-weight models whose DSP block is outside the tuned shape, "
                                "%d clips per call resident in HBM, on the general-shape kernels; ns_per_frame = time / (clips x frames); for scale: the tuned "
                                "fft-256 x 49-frame shape runs at 0.74 ns per frame through the same entry point" % DSP_B)
    if int8_exact is not None:
        x, st = int8_exact, max(side_steps, a.steps // 2)
        ab = CLIP_LEN * 2 + x["labels"] * 4
        ach = ab * B / (x["ms_path"] * 1e-3) / 1e9
        out["int8_exact"] = dict(line(x, st), configs="BASELINE configs[3]: int8-quantised weights/activations, bit-exact (the reference's own impulse; every MFCC feature, "
                                                     "int8 tensor and score identical to the reference's), batch 65 536, 1x MI355X",
                                 roofline={"bound": "hbm", "kernel": "kws_mfcc8_kernel", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                           "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": None, "traffic_source": None, "algorithmic_bytes_per_launch": ab * B,
                                           "hot_path_ms": round(x["ms_path"], 4),
                                           "hot_path_ms_note": "HIP events around the hot-path call: kws_mfcc8_kernel (MFCC -> int8 tensor) + kws_nn_mfma_kernel"})
        out["int8_exact"]["roofline"].update(traffic=out["int8_exact"]["traffic"], traffic_source=out["int8_exact"]["traffic_source"],
                                             traffic_per_kernel=out["int8_exact"]["traffic_per_kernel"])
    if cpu is not None:
        out["cpu_baseline"] = cpu
    emit(out, a)


DETAIL_FILE = "bench_detail.json"
LINE_LIMIT = 6000                # bytes: the driver keeps a bounded tail of stdout (round 5's 27 KB line was not parsed)


def compact_line(out):
    """The ONE stdout line: numbers and short names only (every prose field of `out` stays in bench_detail.json / stderr)."""
    def short_row(x, extra=()):
        row = {"kwsm": x["kwsm_file"].replace(".kwsm", ""), "mode": x["mode"], "value": x["value"], "ms_per_step": x["ms_per_step"],
               "hbm_frac": x["hbm_frac"], "fallback": x["fast_fallback_rate"]}
        if x.get("traffic"):
            row["traffic"] = x["traffic"]
        for k in extra:
            row[k] = x[k]
        return row

    def short_roof(rf):
        c = rf.get("compute") or None
        return {"bound": "hbm", "limited_by": rf.get("bound"), "kernel": rf.get("kernel"), "achieved": rf.get("achieved"), "peak": rf.get("peak"),
                "unit": rf.get("unit"), "frac": rf.get("frac"), "traffic": rf.get("traffic"), "traffic_source": rf.get("traffic_source"),
                "algorithmic_bytes_per_launch": rf.get("algorithmic_bytes_per_launch"), "hot_path_ms": rf.get("hot_path_ms"),
                "compute": {k: c.get(k) for k in ("valu_issue_frac", "valu_active_frac", "mfma_busy_frac", "lds_busy_frac")} if c else None}

    cfg = out["config"]
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline")}
    line["dtype"] = out["dtype_short"]
    line["data"] = out["data"] if len(out["data"]) < 40 else "synthetic (dry run on CPU: not a measurement)"
    line["config"] = {"workload": cfg["workload_short"], "mode": cfg["mode"], "clips_per_gpu": cfg["clips_per_gpu"], "global_batch": cfg["global_batch"],
                      "kwsm_file": cfg["kwsm_file"], "fast_fallback_rate": cfg["fast_fallback_rate"], "collective": cfg["collective"].split(" (")[0],
                      "lib_sha256": cfg["lib_sha256"]}
    line["roofline"] = short_roof(out["roofline"])
    if out.get("cpu_baseline"):
        cb = out["cpu_baseline"]
        line["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"], "cpu": cb["cpu"],
                                "sample": cb["sample_short"]}
    col = out.get("collective")
    line["collective"] = ({k: col.get(k) for k in ("allgather_ms_per_step", "inside_timed_region", "ranks", "ranks_seen_by_rccl", "rccl_version", "rank_time_skew_max_over_min",
                                                      "per_rank_clips_per_s", "gloo_world_size")
                           if k in col} if col else None)
    rows = [short_row(x) for x in out.get("modes", [])[1:]] + [short_row(x) for x in out.get("also", [])]
    if rows:
        line["also"] = rows
    if out.get("int8_exact"):
        x = out["int8_exact"]
        line["int8_exact"] = dict(short_row(x), roofline={k: v for k, v in short_roof(x["roofline"]).items() if k not in ("limited_by", "compute") or v})
    if out.get("also_inputs"):
        line["also_inputs"] = [{"family": x["family"], "mode": x["mode"], "value": x["value"], "handed_on": x["fast_fallback_rate"], "exact": x["fast_exact_rate"]}
                               for x in out["also_inputs"] if x["mode"] == "fast"]
    line["checksum"], line["checksum_class0"] = out["checksum"], out["checksum_class0"]
    line["detail"] = DETAIL_FILE
    return line


def emit(out, a):
    """bench_detail.json (next to this script) and stderr get everything; stdout gets one line of at most LINE_LIMIT bytes."""
    detail = json.dumps(out, indent=1)
    try:
        with open(os.path.join(ROOT, DETAIL_FILE), "w") as f:
            f.write(detail + "\n")
    except OSError as e:                                            # a read-only tree must not cost the bench line
        print("bench.py: could not write %s: %s" % (DETAIL_FILE, e), file=sys.stderr)
    print(json.dumps(out), file=sys.stderr)
    sys.stderr.flush()
    line = compact_line(out)
    text = json.dumps(line, separators=(",", ":"))
    for drop in ("also_inputs", "also"):                           # never reached with the shipped model list; the headline must survive
        if len(text) >= LINE_LIMIT and drop in line:
            del line[drop]
            text = json.dumps(line, separators=(",", ":"))
    assert len(text) < LINE_LIMIT, "bench.py: the stdout line is %d bytes" % len(text)
    print(text)
    sys.stdout.flush()


if __name__ == "__main__":
    main()
