"""World-size-2 gloo test (CPU) of the multi-GPU plumbing: contiguous clip sharding + all-gather of the scores.
The per-rank compute is the C oracle here (no GPU in this container); on the GPUs the same helpers are fed by the
HIP path in bench.py."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from kws_testlib import MODELS, ROOT


def _worker(rank, world, port, per_rank, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from __graft_entry__ import load_package
    from kws_testlib import Oracle, OracleModel
    pkg = load_package()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = Oracle()
    m = OracleModel(o, os.path.join(MODELS, "l476_no_yes.kwsm"))
    first = pkg.shard_first_clip(rank, per_rank)
    local = torch.from_numpy(m.run_batch(o.synth(0, first, per_rank)))
    alls = pkg.all_gather_scores(local, world)
    if rank == 0:
        ret.put(alls.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_equals_single_process(oracle, l476):
    world, per_rank = 2, 6
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, per_rank, ret)) for r in range(world)]
    for p in procs:
        p.start()
    got = ret.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = l476.run_batch(oracle.synth(0, 0, world * per_rank))
    assert got.shape == (world * per_rank, 4)
    assert (got == want).all()


def _bench_worker(rank, world, port, per_rank, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), GLOO_SOCKET_IFNAME="lo")
    sys.path.insert(0, ROOT)
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def max_over_ranks(dt):
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    backend = bench.CpuOracleBackend(rank, world, per_rank)
    res = bench.measure(backend, os.path.join(MODELS, "l476_no_yes.kwsm"), "exact", 2, 1, dist.barrier, max_over_ranks)
    if rank == 0:
        ret.put((backend.gathered.numpy().copy(), res["dt"], res["ms_gather"]))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_timed_region_shards_and_gathers(oracle, l476):
    """The code bench.py times -- measure() / timed_steps() with its sharding (rank r owns clips [r*B, (r+1)*B)) and the gather --
    at world size 2 under gloo, the oracle standing in for the GPU library: the gathered scores are the single-process ones."""
    world, per_rank = 2, 5
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, per_rank, ret)) for r in range(world)]
    for p in procs:
        p.start()
    got, dt, ms_gather = ret.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = l476.run_batch(oracle.synth(0, 0, world * per_rank))
    assert got.shape == want.shape and (got == want).all()
    assert dt > 0 and ms_gather >= 0


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher starts two ranks itself and prints exactly one JSON line with n_gpus = 2
    (--dry-run-cpu: no GPU here); its checksum is the sum of the scores of the 2 x B clips of the global batch."""
    import json
    import subprocess
    from bench import default_batch
    assert default_batch(1) == 65536 and default_batch(8) * 8 == 1 << 20
    env = dict(os.environ, MASTER_PORT=str(33500 + os.getpid() % 2000))
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-cpu", "--steps", "2", "--warmup", "1",
                          "--batch", "4", "--model", os.path.join(MODELS, "l476_no_yes.kwsm"), "--mode", "exact"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    # round 6: the driver keeps a bounded tail of stdout (round 5's 27 KB line went unparsed): the line stays under 6 KB, the prose and the
    # side measurements go to bench_detail.json / stderr
    from bench import DETAIL_FILE, LINE_LIMIT
    assert LINE_LIMIT <= 6000 and len(lines[0]) < 6000, len(lines[0])
    j = json.loads(lines[0])
    assert j["detail"] == DETAIL_FILE and json.load(open(os.path.join(ROOT, DETAIL_FILE)))["value"] == j["value"]
    assert j["roofline"]["bound"] in ("hbm", "mfma") and j["roofline"]["unit"] == "GB/s" and j["roofline"]["peak"] == 8000.0
    assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 8 and j["collective"]["ranks"] == 2
    assert abs(j["checksum"] - 8.0) < 0.1                       # 8 softmax rows
    # round 4: the line carries every rank's own rate and the spread between the ranks, and what the communicator itself reports
    # (the dry run has no RCCL: the gloo group's size stands in, ranks_seen_by_rccl is null)
    c = j["collective"]
    assert len(c["per_rank_clips_per_s"]) == 2 and all(v > 0 for v in c["per_rank_clips_per_s"]) and c["rank_time_skew_max_over_min"] >= 1.0
    assert c["ranks_seen_by_rccl"] is None and c["gloo_world_size"] == 2


def test_rendezvous_port_is_asked_of_the_kernel():
    """bench.py --gpus N without a launcher takes MASTER_PORT when set, otherwise a port the kernel reports free (round 3's pid-derived
    port could collide)"""
    import socket
    from bench import free_port
    p = free_port()
    assert 1024 < p < 65536
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", p))                               # still free
