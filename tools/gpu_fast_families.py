#!/usr/bin/env python3
"""GPU study behind the KWS_MODE_FAST guard: the adversarial input families of tests/kws_families.py through the
fast kernel, the exact kernels and the C oracle.  Per family and model: fallback rate, max |score - oracle|, max |feature - oracle|
over the clips the fast kernel kept, and the same split by how well conditioned the clip's worst cmvnw column is
(deviation / max(1, |mean|), the quantity the guard is stated in).

    python tools/gpu_fast_families.py [clips_per_family] [model,model,...] [out.json]
"""
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from kws_families import FAMILIES, column_conditioning, family  # noqa: E402
from kws_testlib import MODELS, Oracle, OracleModel  # noqa: E402

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
    rel = (sdw / np.maximum(1.0, np.abs(mw))).reshape(len(pcm), -1).min(axis=1)
    level = np.float32([np.abs(np.log(o.mfe(p, om.cfg)[0].astype(np.float64))).max() for p in pcm])     # the guard's level: largest |log-mel energy|
    return s, f, q, rel, sdw.reshape(len(pcm), -1).min(axis=1), sdw.astype(np.float32), mw.astype(np.float32), level


def oracle_clips(pool, path, pcm, chunk=128):
    jobs = [(path, pcm[i:i + chunk]) for i in range(0, len(pcm), chunk)]
    parts = pool.map(_oracle_worker, jobs)
    return [np.concatenate([p[k] for p in parts]) for k in range(8)]


def main():
    import torch
    from __graft_entry__ import load_package
    pkg = load_package()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    models = sys.argv[2].split(",") if len(sys.argv) > 2 else ["cfg2_mfcc40_f32.kwsm", "l476_no_yes_f32.kwsm", "l476_no_yes.kwsm"]
    out_path = sys.argv[3] if len(sys.argv) > 3 else None
    dev = torch.device("cuda:0")
    report = {}
    pool = mp.get_context("spawn").Pool(len(os.sched_getaffinity(0)))
    edges = [0.0, 1e-3, 2e-3, 4e-3, 1e-2, 3e-2, 1e-1, 1e9]
    for name in models:
        path = os.path.join(MODELS, name)
        gm = pkg.Model(path, device=0)
        report[name] = {}
        print("== %s (fused network: %s)" % (name, bool(gm.fast_is_fused)))
        for fam in FAMILIES + ("synth",):
            t0 = time.time()
            host = Oracle().synth(9, 0, n) if fam == "synth" else family(fam, n, seed=3)
            pcm = torch.from_numpy(host).to(dev)

            def run(mode):
                gm.set_mode(mode)
                s = torch.zeros((n, gm.n_labels), dtype=torch.float32, device=dev)
                f = torch.zeros((n, gm.n_features), dtype=torch.float32, device=dev)
                q = None if gm.is_float else torch.zeros((n, gm.n_features), dtype=torch.int8, device=dev)
                gm.run_classifier_batch_device(pcm.data_ptr(), n, s.data_ptr(), f.data_ptr(), q.data_ptr() if q is not None else None)
                torch.cuda.synchronize()
                return s.cpu().numpy(), f.cpu().numpy(), (q.cpu().numpy() if q is not None else None)
            s1, f1, q1 = run(pkg.MODE_FAST)
            nfb = gm.fast_fallback_count()
            s0, f0, q0 = run(pkg.MODE_EXACT)
            so, fo, qo, rel, sd, sdw, mw, level = oracle_clips(pool, path, host)
            exact_ok = bool((f0.view(np.uint32) == fo.view(np.uint32)).all())
            kept = (f1.view(np.uint32) != f0.view(np.uint32)).any(axis=1)          # a re-run clip has the exact kernels' bits
            ds = np.abs(s1 - so).max(axis=1)
            df = np.abs(f1 - fo).max(axis=1)
            row = dict(n=n, fallback=int(nfb), fallback_rate=nfb / n, exact_bit_identical=exact_ok,
                       max_dscore=float(ds.max()), max_dfeature=float(df.max()), nan=int(np.isnan(s1).sum()),
                       clips_over_1e4=int((ds > 1e-4).sum()))
            if q1 is not None:
                row["int8_flips_per_clip"] = float((q1 != qo).sum(axis=1).mean())
                row["clips_changed"] = int((s1 != so).any(axis=1).sum())
            bins = []
            for lo, hi in zip(edges[:-1], edges[1:]):
                m = (rel >= lo) & (rel < hi)
                mk = m & kept
                bins.append(dict(lo=lo, hi=hi, clips=int(m.sum()), kept=int(mk.sum()),
                                 max_dscore=float(ds[mk].max()) if mk.any() else 0.0, max_dfeature=float(df[mk].max()) if mk.any() else 0.0))
            row["by_conditioning"] = bins
            report[name][fam] = row
            row["level_quantiles"] = [float(v) for v in np.quantile(level, [0, 0.5, 1])]
            print("  %-14s level min / median / max %.1f / %.1f / %.1f" % ((fam,) + tuple(row["level_quantiles"])))
            print("  %-14s fallback %5d/%d  max|ds| %.3g  max|df| %.3g  >1e-4: %d  exact==oracle: %s  (%.1fs)"
                  % (fam, nfb, n, ds.max(), df.max(), row["clips_over_1e4"], exact_ok, time.time() - t0))
            print("      " + "  ".join("[%g,%g): %d kept %d ds %.2g df %.2g" % (b["lo"], b["hi"], b["clips"], b["kept"], b["max_dscore"], b["max_dfeature"])
                                       for b in bins if b["clips"]))
            # what the fast arithmetic moved in the cepstra, per column: |dfeature| x the window's deviation, over the kept clips'
            # well-defined windows (deviation > 1e-4: below that the reference's own rounding decides the feature)
            nr, nc = gm.n_frames, gm.n_features // gm.n_frames
            ec = np.abs(f1 - fo).reshape(n, nr, nc) * sdw
            ec = np.where((sdw > 1e-4) & kept[:, None, None] & np.isfinite(ec), ec, 0.0)
            row["ecol_max"] = [float(v) for v in ec.max(axis=(0, 1))]
            row["ecol_q999"] = [float(v) for v in np.quantile(ec.reshape(-1, nc), 0.999, axis=0)]
            print("      E[col] max  : " + " ".join("%.1e" % v for v in row["ecol_max"]))
            # the absolute part alone: windows of small deviation (the relative part |feature| x 1e-6 x deviation is negligible there)
            NFh = gm.n_filters // 2
            ea = np.abs(f1 - fo).reshape(n, nr, nc) * sdw
            sel = (sdw > 1e-3) & (sdw < 0.05) & kept[:, None, None] & np.isfinite(ea)
            def cls(lo, hi, rel_to_mean=False, per_level=False):
                e = ea[:, :, lo:hi][sel[:, :, lo:hi]]
                if rel_to_mean:
                    e = e / np.maximum(1.0, np.abs(mw[:, :, lo:hi][sel[:, :, lo:hi]]))
                if per_level:
                    e = e / np.broadcast_to(level[:, None, None], ea[:, :, lo:hi].shape)[sel[:, :, lo:hi]]
                return (float(e.max()), float(np.quantile(e, 0.999)), int(e.size)) if e.size else (0.0, 0.0, 0)
            row["eabs"] = dict(c0=cls(0, 1), c0_per_mean=cls(0, 1, True), dct=cls(1, min(nc, NFh + 1)), dct_per_level=cls(1, min(nc, NFh + 1), per_level=True), stale=cls(NFh + 1, nc), stale_per_mean=cls(NFh + 1, nc, True))
            print("      E_abs (windows with 1e-3 < deviation < 0.05) max / 99.9%% / count:  " + "  ".join("%s %.2g / %.2g / %d" % ((k,) + v) for k, v in row["eabs"].items()))
            # second tier candidate: the exact kernels' cepstra (bit-identical to the reference) through the fast cmvnw + network
            gm.set_mode(pkg.MODE_EXACT)
            cep = torch.zeros((n, gm.n_features), dtype=torch.float32, device=dev)
            gm.mfcc_batch_device(pcm.data_ptr(), n, cep.data_ptr())
            gm.set_mode(pkg.MODE_FAST)
            s2 = torch.zeros((n, gm.n_labels), dtype=torch.float32, device=dev)
            f2 = torch.zeros((n, gm.n_features), dtype=torch.float32, device=dev)
            gm.cmvn_inference_batch_device(cep.data_ptr(), n, s2.data_ptr(), f2.data_ptr())
            torch.cuda.synchronize()
            nfb2 = gm.fast_fallback_count()
            s2, f2 = s2.cpu().numpy(), f2.cpu().numpy()
            kept2 = (f2.view(np.uint32) != f0.view(np.uint32)).any(axis=1)
            ds2, df2 = np.abs(s2 - so).max(axis=1), np.abs(f2 - fo).max(axis=1)
            b2 = []
            for lo, hi in zip(edges[:-1], edges[1:]):
                mk = (rel >= lo) & (rel < hi) & kept2
                b2.append("[%g,%g): kept %d ds %.2g df %.2g" % (lo, hi, int(mk.sum()), ds2[mk].max() if mk.any() else 0.0, df2[mk].max() if mk.any() else 0.0))
            row["tier2"] = dict(fallback=int(nfb2), max_dscore=float(np.nanmax(ds2)), max_dfeature=float(np.nanmax(df2)), bins=b2)
            print("      exact cepstra -> fast cmvnw + network: fallback %d  max|ds| %.3g  max|df| %.3g   %s" % (nfb2, np.nanmax(ds2), np.nanmax(df2), "  ".join(b2)))
            # worst kept clip: which column / row, and how large the cepstral error must have been (|dfeature| x deviation)
            if kept.any():
                k = int(np.argmax(np.where(kept, df, -1)))
                d = np.abs(f1[k] - fo[k]).reshape(gm.n_frames, -1)
                r, c = np.unravel_index(np.argmax(d), d.shape)
                print("      worst kept clip %d: row %d col %d  |df| %.3g  rel-sd(clip min) %.3g  sd(clip min) %.3g" % (k, r, c, d[r, c], rel[k], sd[k]))
        gm.close()
    if out_path:
        os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
        with open(out_path, "w") as fh:
            json.dump(report, fh, indent=1)


if __name__ == "__main__":
    main()
