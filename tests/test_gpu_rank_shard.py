"""-m gpu: the per-rank workload of BASELINE configs[2] / configs[4] on the one GPU there is.  With 1 M
clips split over 8 GPUs, rank 7 owns clips [7 * 131 072, 8 * 131 072): a 131 072-clip batch (4 GiB of PCM resident in HBM) whose
clip numbers start at 917 504.  That launch -- grid, 32-bit clip offsets, the generator's first_clip -- is what `bench.py --gpus 8`
runs on every rank and had never been executed.  Size-independent properties + a strided oracle sample, both arithmetic modes:
  (a) the device generator equals the host generator for this shard,   (b) duplicated clips give identical rows,
  (c) permuting the batch permutes the scores bit for bit,            (d) a strided sample equals the oracle (exact mode: features
  bit for bit and int8 scores bit for bit / float scores within 1e-6; fast mode: float scores within 1e-4, int8 network exact on
  the GPU's own tensor),                                              (e) rows are softmaxes.
What stays unmeasured is only the N > 1 collective itself (no multi-GPU box here); RCCL at world size 1 runs in test_gpu_bench.py."""
import os

import numpy as np
import pytest

from kws_testlib import MODELS, ROOT, OracleModel, bits

pytestmark = pytest.mark.gpu

RANK, PER_RANK = 7, 131072


@pytest.fixture(scope="module")
def pkg():
    import sys
    sys.path.insert(0, ROOT)
    import torch  # noqa: F401
    from __graft_entry__ import load_package
    return load_package()


@pytest.fixture(scope="module")
def shard(pkg, oracle):
    import torch
    B = PER_RANK
    first = pkg.shard_first_clip(RANK, B)
    assert first == 917504
    pcm = torch.empty((B, 16000), dtype=torch.int16, device="cuda:0")
    pkg.synth_clips_device(0, first, B, 16000, pcm.data_ptr())
    torch.cuda.synchronize()
    idx = np.arange(5, B, 4099)
    host = np.stack([oracle.synth(0, first + int(i), 1)[0] for i in idx])
    assert (pcm[torch.from_numpy(idx).cuda()].cpu().numpy() == host).all()                     # (a)
    pcm[1::8192] = pcm[0::8192]                                                                # (b) duplicates
    host = pcm[torch.from_numpy(idx).cuda()].cpu().numpy()
    yield pcm, idx, host
    del pcm


@pytest.mark.parametrize("name", ["cfg2_mfcc40_f32.kwsm", "cfg5_dscnn_mfcc40_int8.kwsm"])
@pytest.mark.parametrize("mode", ["exact", "fast"])
def test_rank7_shard_of_the_one_million_clip_split(name, mode, pkg, oracle, shard):
    import torch
    pcm, idx, host = shard
    B = PER_RANK
    path = os.path.join(MODELS, name)
    gm = pkg.Model(path, device=0)
    om = OracleModel(oracle, path)
    gm.set_mode(pkg.MODE_FAST if mode == "fast" else pkg.MODE_EXACT)
    C, F = gm.n_labels, gm.n_features
    scores = torch.empty((B, C), dtype=torch.float32, device="cuda:0")
    gm.run_classifier_batch_device(pcm.data_ptr(), B, scores.data_ptr())                       # what bench.py's step launches
    torch.cuda.synchronize()
    s = scores.cpu().numpy()
    assert (s[1::8192] == s[0::8192]).all()                                                    # (b)
    perm = torch.randperm(B, device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(3))
    pcm2 = pcm[perm].contiguous()
    scores2 = torch.empty_like(scores)
    gm.run_classifier_batch_device(pcm2.data_ptr(), B, scores2.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(scores2, scores[perm])                                                  # (c)
    del pcm2
    so, fo, qo = om.run_batch(host, want_features=True)
    # features / int8 tensor of the sampled clips (a second, small call on the same handle and mode)
    sub = torch.from_numpy(host).to("cuda:0")
    n = len(idx)
    ss = torch.empty((n, C), dtype=torch.float32, device="cuda:0")
    ff = torch.empty((n, F), dtype=torch.float32, device="cuda:0")
    qq = None if gm.is_float else torch.empty((n, F), dtype=torch.int8, device="cuda:0")
    gm.run_classifier_batch_device(sub.data_ptr(), n, ss.data_ptr(), ff.data_ptr(), qq.data_ptr() if qq is not None else None)
    torch.cuda.synchronize()
    assert (bits(ss.cpu().numpy()) == bits(s[idx])).all()                                      # the big launch and the small one agree
    f = ff.cpu().numpy()
    if mode == "exact":
        assert (bits(f) == bits(fo)).all()                                                     # (d)
        if gm.is_float:
            assert np.abs(s[idx] - so).max() <= 1e-6
        else:
            assert (qq.cpu().numpy() == qo).all() and (bits(s[idx]) == bits(so)).all()
    else:
        assert np.abs(f - fo).max() <= 2e-3
        if gm.is_float:
            assert np.abs(s[idx] - so).max() <= 1e-4
        else:
            q = qq.cpu().numpy()
            assert np.abs(q.astype(np.int32) - qo.astype(np.int32)).max() <= 1 and (q != qo).mean() <= 1e-3
            for k in range(n):
                assert (bits(om.dequantize(om.nn_invoke(q[k]))) == bits(s[idx][k])).all(), k
        assert gm.fast_fallback_count() <= n // 50
    assert (s >= 0).all() and (np.abs(s.sum(1) - 1.0) <= (1e-5 if gm.is_float else 4 / 256)).all()   # (e)
    gm.close()
