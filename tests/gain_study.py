#!/usr/bin/env python3
"""Opt-in study (not collected by pytest; runs on the GPU box): what KWS_MODE_FAST's arithmetic moves in features, LOGITS and scores
of the committed float32 models, guard switched off, clip by clip against the C oracle -- the data behind the gain-derived guard of
DESIGN.md 4.4.1.  Per clip it keeps the logit error, the score error, max p (1 - p) and, per cepstral column,
sum_r 1 / dev^2, sum_r mean^2 / dev^2 (the two moments an error model in (E + kappa |mean|) / dev needs) and sum_r (feature error)^2.

    KWS_DEV_FAST_GUARD_SCALE=0 python tests/gain_study.py gpurun_out/gain_study.npz [n_synth] [n_per_family] [model,model,...]
"""
import multiprocessing as mp
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from kws_families import FAMILIES, column_conditioning  # noqa: E402
from kws_testlib import MODELS, Oracle, OracleModel  # noqa: E402

_W = {}


def _worker(args):
    path, pcm = args
    if path not in _W:
        o = _W.setdefault("oracle", Oracle())
        _W[path] = OracleModel(o, path)
    om, o = _W[path], _W["oracle"]
    s, f, _ = om.run_batch(pcm, want_features=True)
    z = np.zeros_like(s)
    for i in range(len(pcm)):
        _, taps = om.nn_invoke_f32(f[i], taps=True)
        z[i] = [t for t in taps if len(t) == om.n_labels][-2]            # the tensor SOFTMAX reads
    cep = np.stack([o.mfcc_nocmvn(p, om.cfg) for p in pcm])
    sd, mean = column_conditioning(cep, om.cfg.win_size, full=True)
    return s, f, z, sd.astype(np.float32), mean.astype(np.float32)


def main():
    out = sys.argv[1]
    n_synth = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    n_fam = int(sys.argv[3]) if len(sys.argv) > 3 else 512
    models = sys.argv[4].split(",") if len(sys.argv) > 4 else ["cfg2_mfcc40_f32.kwsm", "l476_no_yes_f32.kwsm", "cfg5_dscnn_mfcc40_f32.kwsm"]
    import torch
    from __graft_entry__ import load_package
    from test_gpu_fast_families import family_pcm
    pkg = load_package()
    oracle = Oracle()
    res = {}
    with mp.get_context("spawn").Pool(len(os.sched_getaffinity(0))) as pool:
        sets = [("synth", oracle.synth(0, 0, n_synth))] + [(fam, family_pcm(pkg, fam, n_fam, seed=23)) for fam in FAMILIES]
        for name in models:
            path = os.path.join(MODELS, name)
            gm = pkg.Model(path, device=0)
            ncep = gm.n_features // gm.n_frames
            for fam, host in sets:
                n = len(host)
                pcm = torch.from_numpy(host).to("cuda:0")
                s = torch.zeros((n, gm.n_labels), dtype=torch.float32, device="cuda:0")
                f = torch.zeros((n, gm.n_features), dtype=torch.float32, device="cuda:0")
                z = torch.zeros((n, gm.n_labels), dtype=torch.float32, device="cuda:0")
                gm.set_logits_tap(z.data_ptr())
                gm.set_mode(pkg.MODE_EXACT)
                gm.run_classifier_batch_device(pcm.data_ptr(), n, s.data_ptr(), f.data_ptr())
                torch.cuda.synchronize()
                z0, s0 = z.cpu().numpy().copy(), s.cpu().numpy().copy()
                gm.set_mode(pkg.MODE_FAST)
                gm.run_classifier_batch_device(pcm.data_ptr(), n, s.data_ptr(), f.data_ptr())
                torch.cuda.synchronize()
                handed = gm.fast_fallback_count()
                f1 = f.cpu().numpy().copy()
                z.zero_()
                gm.run_classifier_batch_device(pcm.data_ptr(), n, s.data_ptr())          # the fused form: scores (and logits) only
                torch.cuda.synchronize()
                z1, s1 = z.cpu().numpy().copy(), s.cpu().numpy().copy()
                gm.set_logits_tap(None)
                parts = pool.map(_worker, [(path, host[i:i + 64]) for i in range(0, n, 64)])
                so, fo, zo, sd, mean = [np.concatenate([p[k] for p in parts]) for k in range(5)]
                assert (z0.view(np.uint32) == zo.view(np.uint32)).all(), "exact-mode logits differ from the oracle's (%s, %s)" % (name, fam)
                dz = z1 - zo
                pair = (dz[:, :, None] - dz[:, None, :])
                dzp = np.abs(pair).reshape(n, -1).max(axis=1)                              # largest error of a logit DIFFERENCE
                ds = np.abs(s1 - so).max(axis=1)
                pq = (so * (1 - so)).max(axis=1)
                df = (f1 - fo).reshape(n, gm.n_frames, ncep).astype(np.float64)
                rd = 1.0 / (sd.astype(np.float64) + 1.1920929e-7)
                key = "%s/%s/" % (name, fam)
                res[key + "dz_pair"] = dzp.astype(np.float32)
                res[key + "dz"] = dz.astype(np.float32)
                res[key + "ds"] = ds.astype(np.float32)
                res[key + "pq"] = pq.astype(np.float32)
                res[key + "A"] = (rd ** 2).sum(axis=1).astype(np.float32)                  # [n][ncep]
                res[key + "B"] = ((mean * rd) ** 2).sum(axis=1).astype(np.float32)
                res[key + "D"] = (df ** 2).sum(axis=1).astype(np.float32)
                res[key + "E2"] = ((df * sd) ** 2).mean(axis=1).astype(np.float32)         # mean square error of the (mean-free) cepstra per column
                res[key + "mindev"] = sd.min(axis=1).astype(np.float32)
                print("%-28s %-15s n %5d handed back %5d  max |dlogit pair| %.3g  rms %.3g  max |dscore| %.3g  max |dfeat| %.3g  rms dfeat %.3g"
                      % (name, fam, n, handed, dzp.max(), np.sqrt((dzp ** 2).mean()), ds.max(), np.abs(df).max(), np.sqrt((df ** 2).mean())), flush=True)
            gm.close()
    np.savez_compressed(out, **res)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
