"""-m gpu: KWS_MODE_FAST off the friendly distribution.  Nine input families that put cmvnw columns
anywhere between "constant" and "lively", plus the reference's own data shape (word + background recording at dataset-curation.py's
default volumes) for scale (tests/kws_families.py), x 8 192 clips go through the fast kernel AND the C oracle, clip
by clip.  Bar: float32 scores within 1e-4 (north_star); int8 graphs: the network is exact from the GPU's own int8 tensor on, flip
rate reported.  Per family the test prints max |score - oracle|, max |feature - oracle| and the fallback rate.

The guard that decides which clips go back to the exact kernels is read from the library (kws_fast_guard) and re-evaluated here
on the oracle's cepstra: clips well inside it (margin < 0.9) MUST have been handed back, clips well outside it (margin > 1.1)
are the fast kernel's and must meet the bar; the families are built so that hundreds of clips sit at 0.5x .. 2x the guard."""
import multiprocessing as mp
import os

import numpy as np
import pytest

from kws_families import FAMILIES, background_track, column_conditioning, family, word_waveforms
from kws_testlib import MODELS, ROOT, Oracle, OracleModel, bits

pytestmark = pytest.mark.gpu

FAST_SCORE_TOL = 1e-4          # north_star: "per-class scores match the reference C path within 1e-4 fp32"
N_PER_FAMILY = 8192


@pytest.fixture(scope="module")
def pkg():
    import sys
    sys.path.insert(0, ROOT)
    import torch  # noqa: F401
    from __graft_entry__ import load_package
    return load_package()


_W = {}


def _oracle_worker(args):
    path, pcm = args
    if path not in _W:
        o = _W.setdefault("oracle", Oracle())
        _W[path] = OracleModel(o, path)
    om, o = _W[path], _W["oracle"]
    s, f, q = om.run_batch(pcm, want_features=True)
    cep = np.stack([o.mfcc_nocmvn(p, om.cfg) for p in pcm])
    sdw, mw = column_conditioning(cep, om.cfg.win_size, full=True)
    return s, f, q, sdw.astype(np.float32), mw.astype(np.float32)


@pytest.fixture(scope="module")
def pool():
    with mp.get_context("spawn").Pool(len(os.sched_getaffinity(0))) as p:
        yield p


def oracle_clips(pool, path, pcm, chunk=128):
    parts = pool.map(_oracle_worker, [(path, pcm[i:i + chunk]) for i in range(0, len(pcm), chunk)])
    return [np.concatenate([p[k] for p in parts]) for k in range(5)]


def family_pcm(pkg, name, n, seed):
    """host int16 [n][16000]; word_silence is made on the GPU by kws_mix_audio_device (word shorter than the window, no background:
    the zero padding of dataset-curation.py:114-116)"""
    import torch
    if name not in ("word_silence", "word_background"):
        return family(name, n, seed)
    w, ln = word_waveforms(n, seed)
    words = torch.from_numpy(w).to("cuda:0")
    lens = torch.from_numpy(ln).to("cuda:0")
    out = torch.zeros((n, 16000), dtype=torch.int16, device="cuda:0")
    if name == "word_silence":
        pkg.mix_audio_device(words.data_ptr(), lens.data_ptr(), 16000, None, 0, None, 1.0, 0.0, n, 16000, out.data_ptr())
    else:                                            # the reference's defaults: word_vol 1.0, bg_vol 0.1 (dataset-curation.py:167-181)
        track = torch.from_numpy(background_track(seed)).to("cuda:0")
        start = torch.from_numpy(np.random.default_rng([seed, 5]).integers(0, track.numel() - 16000 + 1, n).astype(np.int32)).to("cuda:0")
        pkg.mix_audio_device(words.data_ptr(), lens.data_ptr(), 16000, track.data_ptr(), track.numel(), start.data_ptr(), 1.0, 0.1, n, 16000, out.data_ptr())
    torch.cuda.synchronize()
    host = out.cpu().numpy()
    if name == "word_silence":
        assert (np.abs(host).max(axis=1) > 0).mean() > 0.9 and all((host[i, ln[i]:] == 0).all() for i in range(0, n, 97))   # word, then digital silence
    else:
        assert all(np.abs(host[i, ln[i]:].astype(np.int32)).max() > 20 for i in range(0, n, 97))                          # word, then the background
    return host


def run_device(pkg, gm, mode, pcm_t):
    import torch
    n = pcm_t.shape[0]
    gm.set_mode(mode)
    s = torch.zeros((n, gm.n_labels), dtype=torch.float32, device="cuda:0")
    f = torch.zeros((n, gm.n_features), dtype=torch.float32, device="cuda:0")
    q = None if gm.is_float else torch.zeros((n, gm.n_features), dtype=torch.int8, device="cuda:0")
    gm.run_classifier_batch_device(pcm_t.data_ptr(), n, s.data_ptr(), f.data_ptr(), q.data_ptr() if q is not None else None)
    torch.cuda.synchronize()
    return s.cpu().numpy(), f.cpu().numpy(), (q.cpu().numpy() if q is not None else None)


def guard_margin(gm, sdw, mw):
    """per clip: min over cmvnw windows of deviation / (abs_thr[c] + rel_thr[c] |mean|) -- below 1 the clip is handed back"""
    a, rel = gm.fast_guard()
    thr = a[None, None, :] + rel[None, None, :] * np.abs(mw)
    return (sdw / thr).reshape(len(sdw), -1).min(axis=1)


@pytest.mark.parametrize("name", ["cfg2_mfcc40_f32.kwsm", "l476_no_yes_f32.kwsm", "l476_no_yes.kwsm"])
def test_fast_mode_on_adversarial_input_families(name, pkg, pool):
    import torch
    path = os.path.join(MODELS, name)
    gm = pkg.Model(path, device=0)
    om = OracleModel(Oracle(), path)
    n = N_PER_FAMILY
    near = np.zeros(4, int)                                   # clips at 0.5-0.9, 0.9-1.1, 1.1-2, 2-4 x the guard, all families
    worst = 0.0
    print()
    for fam in FAMILIES:
        host = family_pcm(pkg, fam, n, seed=11)
        pcm = torch.from_numpy(host).to("cuda:0")
        s1, f1, q1 = run_device(pkg, gm, pkg.MODE_FAST, pcm)
        nfb = gm.fast_fallback_count()
        if gm.fast_is_fused:                                   # the form bench.py times: scores only, features never leave the chip
            gm.set_mode(pkg.MODE_FAST)
            s2 = torch.zeros((n, gm.n_labels), dtype=torch.float32, device="cuda:0")
            gm.run_classifier_batch_device(pcm.data_ptr(), n, s2.data_ptr())
            torch.cuda.synchronize()
            assert (s2.cpu().numpy() == s1).all(), fam
        s0, f0, q0 = run_device(pkg, gm, pkg.MODE_EXACT, pcm)
        so, fo, qo, sdw, mw = oracle_clips(pool, path, host)
        assert (bits(f0) == bits(fo)).all(), fam               # the exact kernels stay bit-exact on these inputs too
        margin = guard_margin(gm, sdw, mw)
        handed_back = (bits(f1) == bits(f0)).all(axis=1)       # a re-run clip carries the exact kernels' bits
        ds = np.abs(s1 - so).max(axis=1)
        df = np.abs(f1 - fo).max(axis=1)
        near += np.histogram(margin, [0.5, 0.9, 1.1, 2.0, 4.0])[0]
        line = "%-22s %-13s handed back %5d / %d (%5.1f %%)  max |score - oracle| %.3g  max |feature - oracle| %.3g" % (
            name, fam, nfb, n, 100.0 * nfb / n, ds.max(), df[~handed_back].max() if (~handed_back).any() else 0.0)
        assert not np.isnan(s1).any(), fam
        # the guard does what it says: well inside it -> handed back (results are the exact mode's); well outside -> kept
        assert handed_back[margin < 0.9].all(), fam
        assert nfb == n or not (margin > 1.1).any() or (~handed_back[margin > 1.1]).mean() > 0.99, fam
        assert (bits(s1[handed_back]) == bits(s0[handed_back])).all(), fam
        if gm.is_float:
            assert ds.max() <= FAST_SCORE_TOL, line
            worst = max(worst, float(ds.max()))
        else:
            # exact from the int8 tensor on: the oracle's network on the GPU's tensor gives the GPU's scores
            flips = (q1 != qo).sum(axis=1)
            changed = (s1 != so).any(axis=1)
            assert np.abs(q1.astype(np.int32) - qo.astype(np.int32)).max() <= 1, fam
            assert not changed[flips == 0].any(), fam
            for i in np.nonzero(changed)[0][:32]:
                assert (bits(om.dequantize(om.nn_invoke(q1[i]))) == bits(s1[i])).all(), (fam, i)
            line += "  int8 flips / clip %.4f  clips with a changed score %d" % (flips.mean(), int(changed.sum()))
            assert flips.mean() <= 0.1 and changed.mean() <= 0.02, line
        print(line)
    print("%s: clips at 0.5-0.9 / 0.9-1.1 / 1.1-2 / 2-4 x the guard: %s; worst score error %.3g" % (name, near.tolist(), worst))
    assert (near >= 100).all()                                 # the guard's neighbourhood was really probed
    gm.close()
