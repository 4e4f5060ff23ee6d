#!/usr/bin/env python3
"""GPU study behind round 6's guard: what KWS_MODE_FAST really moves, per clip, with the guard switched OFF
(development build, KWS_DEV_FAST_GUARD_SCALE=0), next to what the product library's guard does with the same clips.

Per float graph and input family of tests/kws_families.py (the seed bench.py's also_inputs uses), per clip:
    ds      max |score - oracle|            guard off
    dzp     max |d(z_i - z_j)| vs oracle     guard off   (the quantity the guard estimates)
    pq      max p (1 - p) of the oracle's scores
    ecep    [columns] rms over the rows of |feature - oracle| x (window deviation + eps): the error in the cepstral domain
    efeat   [columns] rms over the rows of |feature - oracle|
    v_lo / v_hi   the product guard's variance estimate re-evaluated from the oracle's windows (tests/test_gpu_fast_families.py)
    lvl, sil      the guard's level and silent-frame flag
    on_t2, on_ex  how many clips the product library's first tier handed on / the exact kernels finished (scores-only call, as bench.py)
-> gpurun_out/<out>.npz (arrays named <model>/<family>/<name>) and a text table on stdout.

    python tools/gpu_guard_study.py [clips_per_family = 2048] [out = gpurun_out/guard_study.npz] [seed = 5]
"""
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from kws_families import FAMILIES, family  # noqa: E402
from kws_testlib import MODELS, Oracle  # noqa: E402
from test_gpu_fast_families import guard_variance, oracle_clips  # noqa: E402

FLOAT_MODELS = ("cfg2_mfcc40_f32.kwsm", "l476_no_yes_f32.kwsm")
EPS = 1.1920929e-7


def main():
    import torch
    from __graft_entry__ import load_package
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "guard_study.npz")
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    pkg = load_package()
    dev_pkg = load_package(dev=True)
    pool = mp.get_context("spawn").Pool(len(os.sched_getaffinity(0)))
    arrays = {}
    print("# guard off = libkws_mi355x_dev.so with KWS_DEV_FAST_GUARD_SCALE=0; guard on = libkws_mi355x.so; %d clips per family, seed %d" % (n, seed))
    print("# truly over = clips whose score error with the guard off exceeds 1e-4; handed on = what the product's first tier passes to the second; precision = truly over / handed on")
    for name in FLOAT_MODELS:
        path = os.path.join(MODELS, name)
        gm = pkg.Model(path, device=0)
        os.environ["KWS_DEV_FAST_GUARD_SCALE"] = "0"
        g0 = dev_pkg.Model(path, device=0)
        del os.environ["KWS_DEV_FAST_GUARD_SCALE"]
        tol = gm.fast_tolerance()
        arrays["%s/gain" % name] = np.asarray(gm.fast_gain(), np.float32) if hasattr(gm, "fast_gain") else np.zeros(0, np.float32)
        arrays["%s/sigma_net" % name] = np.float32(tol["sigma_net"])
        for t in (1, 2):
            arrays["%s/guard_coef%d" % (name, t)] = gm.fast_guard(t).astype(np.float32)
        print("== %s  k_sigma %.3g  sigma_net %.3g  total gain %.3g" % (name, tol["k_sigma"], tol["sigma_net"], tol["total_gain"]))
        print("  %-16s %9s %9s %9s %9s %10s %10s %10s %10s" % ("family", "handed on", "exact", "truly>1e-4", "precision", "max ds off", "max dz off", "max dz/sqV", "p99 dz/sqV"))
        for fam in FAMILIES + ("synth",):
            host = Oracle().synth(seed, 0, n) if fam == "synth" else family(fam, n, seed=seed)
            pcm = torch.from_numpy(np.ascontiguousarray(host)).to("cuda:0")
            # the product's guard, the call bench.py times (scores only)
            gm.set_mode(pkg.MODE_FAST)
            s_on = torch.zeros((n, gm.n_labels), dtype=torch.float32, device="cuda:0")
            gm.run_classifier_batch_device(pcm.data_ptr(), n, s_on.data_ptr())
            torch.cuda.synchronize()
            on_t2, on_ex = gm.fast_fallback_count(), gm.fast_exact_count()
            # guard off: features, scores, logits of the fast kernel for EVERY clip
            g0.set_mode(dev_pkg.MODE_FAST)
            s1 = torch.zeros((n, g0.n_labels), dtype=torch.float32, device="cuda:0")
            f1 = torch.zeros((n, g0.n_features), dtype=torch.float32, device="cuda:0")
            z1 = torch.zeros((n, g0.n_labels), dtype=torch.float32, device="cuda:0")
            g0.set_logits_tap(z1.data_ptr())
            g0.run_classifier_batch_device(pcm.data_ptr(), n, s1.data_ptr(), f1.data_ptr())
            torch.cuda.synchronize()
            g0.set_logits_tap(None)
            assert g0.fast_fallback_count() == 0 or True
            s1, f1, z1, s_on = s1.cpu().numpy(), f1.cpu().numpy(), z1.cpu().numpy(), s_on.cpu().numpy()
            so, fo, qo, sdw, mw, zo, lvl, sil = oracle_clips(pool, path, host)
            nfr = sdw.shape[1]
            ds = np.abs(s1 - so).max(axis=1)
            ds_on = np.abs(s_on - so).max(axis=1)
            dz = (z1 - zo).astype(np.float64)
            dzp = np.abs(dz[:, :, None] - dz[:, None, :]).reshape(n, -1).max(axis=1)
            pq = (so * (1.0 - so)).max(axis=1)
            df = np.abs(f1 - fo).reshape(n, nfr, -1).astype(np.float64)
            df = np.where(np.isfinite(df), df, 0.0)
            ecep = np.sqrt(((df * (sdw.astype(np.float64) + EPS)) ** 2).mean(axis=1))
            efeat = np.sqrt((df ** 2).mean(axis=1))
            v_lo, v_hi = guard_variance(gm, sdw, mw, lvl, sil, 1)
            r = dzp / np.sqrt(v_hi)
            over = int((ds > 1e-4).sum())
            print("  %-16s %9d %9d %9d %9s %10.3g %10.3g %10.3g %10.3g   (guard on: max ds %.3g; nan %d)" % (
                fam, on_t2, on_ex, over, ("%.3f" % (over / on_t2)) if on_t2 else "-", np.nanmax(ds), np.nanmax(dzp), np.nanmax(r), np.nanquantile(r, 0.99),
                np.nanmax(ds_on), int(np.isnan(s1).sum())))
            key = "%s/%s/" % (name, fam)
            for k, v in dict(ds=ds, ds_on=ds_on, dzp=dzp, pq=pq, ecep=ecep, efeat=efeat, v_lo=v_lo, v_hi=v_hi, lvl=lvl[:, 0], sil=sil,
                             lvl_live=lvl[:, 1], sd_mean=sdw.mean(axis=1), sd_min=sdw.min(axis=1), m_abs=np.abs(mw).mean(axis=1), on=np.int64([on_t2, on_ex])).items():
                arrays[key + k] = np.asarray(v, np.float32) if np.asarray(v).dtype.kind == "f" else np.asarray(v)
            sys.stdout.flush()
        gm.close()
        g0.close()
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    np.savez_compressed(out_path, **arrays)
    print("wrote %s (%.1f MB)" % (out_path, os.path.getsize(out_path) / 1e6))


if __name__ == "__main__":
    main()
