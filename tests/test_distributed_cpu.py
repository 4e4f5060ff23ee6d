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
