"""-m gpu: KWS_MODE_FAST off the friendly distribution.  Nine input families that put cmvnw columns
anywhere between "constant" and "lively", plus the reference's own data shape (word + background recording at dataset-curation.py's
default volumes) for scale (tests/kws_families.py), x 8 192 clips go through the fast kernel AND the C oracle, clip
by clip.  Bar: float32 scores within 1e-4 (north_star); int8 graphs: the network is exact from the GPU's own int8 tensor on, flip
rate reported.  Per family the test prints max |score - oracle|, max |feature - oracle| and the fallback rate.

The guards that decide which clips leave the fast kernel (tier 1 -> exact cepstra + fast cmvnw / network) and which of those end in the
exact kernels (tier 2 -> exact) follow from the LOADED MODEL since round 4 (kws.h: kws_fast_guard / kws_fast_gain /
kws_fast_tolerance_info): per clip the kernels estimate the variance V of the error of a logit difference from the
windows' deviations, the graph's calibrated gain per cepstral column and the clip's log-mel level, and keep the clip iff
k_sigma sqrt(V) stays below the score tolerance through the clip's own softmax.  The rule is re-evaluated here from the oracle's
cepstra, log-mel energies and scores: clips well inside a guard (margin < 0.9) MUST have been handed on, clips well outside it
(margin > 1.1) must not, every clip must meet the 1e-4 score bar wherever it ended, and -- float32 graphs -- the LOGITS of every clip a
fast tier kept must sit within k_sigma sqrt(V) of the oracle's (the claim the guard makes); the families are built so that hundreds of
clips sit at 0.5x .. 2x either guard."""
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
    z = np.zeros_like(s)
    if o.L.kwso_model_is_float(om.h):
        for i in range(len(pcm)):
            _, taps = om.nn_invoke_f32(f[i], taps=True)
            z[i] = [t for t in taps if len(t) == om.n_labels][-2]            # the tensor SOFTMAX reads
    # the clip's log-mel level (what the fast kernel's rounding errors scale with): mean over the frames of |mean over the filters|
    # ... and which frames are digitally silent (a frame energy of exactly 0, which zero handling turns into FLT_EPSILON).  Round 6: tier 1
    # writes the reference's own cepstral row into such frames, so the rule's level is the LIVE frames' and its absolute / per-level terms
    # are scaled by sqrt(live frames / frames): lvl = [level over all frames, level over the live frames], sil = the number of silent frames
    lvl, sil = np.zeros((len(pcm), 2), np.float32), np.zeros(len(pcm), np.int32)
    for i, p in enumerate(pcm):
        mel, en = o.mfe(p, om.cfg)
        per_frame = np.abs(np.log(mel.astype(np.float64)).mean(axis=1))
        dead = en == np.float32(1.1920929e-7)
        lvl[i, 0] = per_frame.mean()
        lvl[i, 1] = per_frame[~dead].mean() if (~dead).any() else 0.0
        sil[i] = int(dead.sum())
    return s, f, q, sdw.astype(np.float32), mw.astype(np.float32), z, lvl, sil


@pytest.fixture(scope="module")
def pool():
    with mp.get_context("spawn").Pool(len(os.sched_getaffinity(0))) as p:
        yield p


def oracle_clips(pool, path, pcm, chunk=128):
    parts = pool.map(_oracle_worker, [(path, pcm[i:i + chunk]) for i in range(0, len(pcm), chunk)])
    return [np.concatenate([p[k] for p in parts]) for k in range(8)]


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
    z = torch.zeros((n, gm.n_labels), dtype=torch.float32, device="cuda:0")
    q = None if gm.is_float else torch.zeros((n, gm.n_features), dtype=torch.int8, device="cuda:0")
    if gm.is_float:
        gm.set_logits_tap(z.data_ptr())
    gm.run_classifier_batch_device(pcm_t.data_ptr(), n, s.data_ptr(), f.data_ptr(), q.data_ptr() if q is not None else None)
    torch.cuda.synchronize()
    if gm.is_float:
        gm.set_logits_tap(None)
    return s.cpu().numpy(), f.cpu().numpy(), (q.cpu().numpy() if q is not None else None), z.cpu().numpy()


def guard_variance(gm, sdw, mw, lvl, sil, tier):
    """The guard's variance estimate V of that tier, re-evaluated from the oracle's windows (kws.h): (lo, hi) per clip -- column 0's
    window-mean term is dropped when the kernel replayed its means in the reference's order, a decision taken inside the kernel: lo assumes
    it did, hi that it did not.  The other columns take the alternative coefficient for clips with digitally silent frames; in tier 1 those
    frames' rows are the reference's own (kws_fast_tolerance::silent_rows_exact): the level is the live frames', the absolute and per-level
    terms are scaled by sqrt(live frames / frames).  lvl [clips][2] = level over all / over the live frames, sil [clips] = silent frames."""
    coef = gm.fast_guard(tier).astype(np.float64)                     # [4][columns]: abs, per level, per |mean|, the alternative per |mean|
    tol = gm.fast_tolerance()
    nfr = sdw.shape[1]
    sil = np.asarray(sil)
    exact_rows = bool(tol["silent_rows_exact"]) and tier == 1
    live = (nfr - sil).astype(np.float64)
    scale = np.sqrt(live / nfr)[:, None, None] if exact_rows else 1.0
    level = (lvl[:, 1] if exact_rows else lvl[:, 0]).astype(np.float64)[:, None, None] if tier == 1 else 0.0
    rd = 1.0 / (sdw.astype(np.float64) + 1.1920929e-7)
    base = (coef[0][None, None, :] + coef[1][None, None, :] * level) * scale
    # a lane's first window decides per column block whether the column is near-constant (deviation below systematic_ratio x |mean|): the window means
    # then round systematically and the column takes the alternative coefficient, like every column of a clip with silent frames.  The kernel compares in
    # fp32 on its own running sums: lo takes the threshold 20 % lower, hi 25 % higher
    ncol = sdw.shape[2]
    cr = 13 if (ncol <= 16 and nfr <= 52) else 17
    first = np.minimum(np.arange(nfr) // cr * cr, nfr - 1)                 # the row whose window a row's lane looked at
    ratio = sdw.astype(np.float64)[:, first, :] / np.maximum(np.abs(mw.astype(np.float64)[:, first, :]), 1e-300)
    v = []
    for rel0, thr in ((coef[3][0], 0.8 * tol["systematic_ratio"]), (coef[2][0], 1.25 * tol["systematic_ratio"])):
        sysm = (ratio < thr) | (sil > 0)[:, None, None]                         # [clips][rows][columns]
        rel = np.where(sysm, coef[3][None, None, :], coef[2][None, None, :]).astype(np.float64)
        # column 0: its means replayed in the reference's order (lo, and every clip with silent frames) -> coef[3][0]; not replayed (hi) -> coef[2][0]
        rel[:, :, 0] = np.where((sil > 0)[:, None], coef[3][0], rel0)
        b = (base + rel * np.abs(mw)) * rd
        v.append((b * b).reshape(len(sdw), -1).sum(axis=1) + tol["sigma_net"] ** 2)
    return v[0], v[1]


def guard_margin(gm, sdw, mw, lvl, sil, tier, pq):
    """per clip (lo, hi): 1 / sqrt(V max(g_c1 P^2, g_c2)) -- below 1 the tier hands the clip on"""
    tol = gm.fast_tolerance()
    vlo, vhi = guard_variance(gm, sdw, mw, lvl, sil, tier)
    w = np.maximum(tol["g_c1"] * pq.astype(np.float64) ** 2, tol["g_c2"])
    with np.errstate(divide="ignore"):
        return 1.0 / np.sqrt(vhi * w), 1.0 / np.sqrt(vlo * w)


@pytest.mark.parametrize("name", ["cfg2_mfcc40_f32.kwsm", "l476_no_yes_f32.kwsm", "l476_no_yes.kwsm"])
def test_fast_mode_on_adversarial_input_families(name, pkg, pool):
    import torch
    path = os.path.join(MODELS, name)
    gm = pkg.Model(path, device=0)
    om = OracleModel(Oracle(), path)
    n = N_PER_FAMILY
    near1, near2 = np.zeros(4, int), np.zeros(4, int)          # clips at 0.5-0.9, 0.9-1.1, 1.1-2, 2-4 x each tier's guard, all families
    worst = 0.0
    worst_sigma = {}
    print()
    for fam in FAMILIES:
        host = family_pcm(pkg, fam, n, seed=11)
        pcm = torch.from_numpy(host).to("cuda:0")
        s1, f1, q1, z1 = run_device(pkg, gm, pkg.MODE_FAST, pcm)
        n_t2, n_ex = gm.fast_fallback_count(), gm.fast_exact_count()
        if gm.fast_is_fused:                                   # the form bench.py times: scores only, features never leave the chip
            gm.set_mode(pkg.MODE_FAST)
            s2 = torch.zeros((n, gm.n_labels), dtype=torch.float32, device="cuda:0")
            gm.run_classifier_batch_device(pcm.data_ptr(), n, s2.data_ptr())
            torch.cuda.synchronize()
            # (without the feature matrix the one launch knows the clip's own scores and may keep a saturated clip that the feature-emitting
            # launch -- which must assume the largest p (1 - p) there is -- hands on: fewer clips leave, the scores agree within the bar)
            assert np.abs(s2.cpu().numpy() - s1).max() <= FAST_SCORE_TOL, fam
            n_t2s, n_exs = gm.fast_fallback_count(), gm.fast_exact_count()
            assert n_t2s <= n_t2, (fam, n_t2s, n_t2)
        s0, f0, q0, z0 = run_device(pkg, gm, pkg.MODE_EXACT, pcm)
        so, fo, qo, sdw, mw, zo, lvl, sil = oracle_clips(pool, path, host)
        assert (bits(f0) == bits(fo)).all(), fam               # the exact kernels stay bit-exact on these inputs too
        # P of the rule: 1/4 -- the call above asked for the feature matrix, and the feature-emitting launch's list decides for features and
        # scores (kws.h); the scores-only call (fused float32 graphs: the clip's own largest p (1 - p)) was held to the bar above
        pq = np.full(n, 0.25)
        (m1lo, m1hi), (m2lo, m2hi) = guard_margin(gm, sdw, mw, lvl, sil, 1, pq), guard_margin(gm, sdw, mw, lvl, sil, 2, pq)
        exact = (bits(f1) == bits(f0)).all(axis=1)             # a clip the exact kernels finished carries their bits
        ds = np.abs(s1 - so).max(axis=1)
        df = np.abs(f1 - fo).max(axis=1)
        near1 += np.histogram(m1hi, [0.5, 0.9, 1.1, 2.0, 4.0])[0]
        near2 += np.histogram(m2hi, [0.5, 0.9, 1.1, 2.0, 4.0])[0]
        line = "%-22s %-15s second tier %5d (%5.1f %%), exact kernels %5d (%5.1f %%) of %d  max |score - oracle| %.3g  max |feature - oracle| %.3g" % (
            name, fam, n_t2, 100.0 * n_t2 / n, n_ex, 100.0 * n_ex / n, n, ds.max(), df[~exact].max() if (~exact).any() else 0.0)
        assert not np.isnan(s1).any(), fam
        # the guards do what they say: a clip well inside the first tier's goes on to the second, one well inside the second tier's
        # is finished by the exact kernels (and has their bits), one well outside stays where it is
        assert (m1hi < 0.9).sum() <= n_t2 <= (m1lo < 1.1).sum(), (fam, n_t2, int((m1hi < 0.9).sum()), int((m1lo < 1.1).sum()))
        in_t2 = m1lo < 1.1                                      # a clip can only reach the exact kernels through the second tier
        assert ((m2hi < 0.9) & (m1hi < 0.9)).sum() <= n_ex <= ((m2lo < 1.1) & in_t2).sum(), (fam, n_ex, int(((m2hi < 0.9) & (m1hi < 0.9)).sum()), int(((m2lo < 1.1) & in_t2).sum()))
        assert exact[(m2hi < 0.9) & (m1hi < 0.9)].all() and not exact[(m2lo > 1.1) | (m1lo > 1.1)].any(), fam
        assert (bits(s1[exact]) == bits(s0[exact])).all(), fam
        if gm.is_float:
            # the claim the guard makes, held at the LOGITS (a saturated score hides its logit): every clip a fast tier kept has its
            # logit differences within k_sigma sqrt(V) of the oracle's -- V re-evaluated from the oracle -- and within the linearisation cap
            tol = gm.fast_tolerance()
            dz = z1 - zo
            dzp = np.abs(dz[:, :, None] - dz[:, None, :]).reshape(n, -1).max(axis=1)
            assert (bits(z1[exact]) == bits(z0[exact])).all(), fam
            v1 = guard_variance(gm, sdw, mw, lvl, sil, 1)[1]
            v2 = guard_variance(gm, sdw, mw, lvl, sil, 2)[1]
            kept1, kept2 = m1lo > 1.1, (~exact) & (m1hi < 0.9)         # surely stayed in tier 1 / surely finished by tier 2
            for kept, v, what in ((kept1, v1, "tier 1"), (kept2, v2, "tier 2")):
                if kept.any():
                    r = dzp[kept] / np.sqrt(v[kept])
                    worst_sigma[what] = max(worst_sigma.get(what, 0.0), float(r.max()))
                    assert r.max() <= tol["k_sigma"], (fam, what, float(r.max()))
            assert dzp[~exact].max(initial=0.0) <= tol["logit_cap"], (fam, float(dzp[~exact].max()))
            line += "  max |dlogit| %.3g" % dzp.max()
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
    print("%s: clips at 0.5-0.9 / 0.9-1.1 / 1.1-2 / 2-4 x the guard: first tier %s, second tier %s; worst score error %.3g; worst logit error in "
          "units of the guard's sigma: %s (k_sigma = %.2g)" % (name, near1.tolist(), near2.tolist(), worst, worst_sigma, gm.fast_tolerance()["k_sigma"]))
    assert (near1 >= 100).all() and (near2 >= 100).all()       # both guards' neighbourhoods were really probed
    gm.close()
