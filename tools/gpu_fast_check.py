#!/usr/bin/env python3
"""GPU development check of KWS_MODE_FAST: fast vs exact mode on the same clips (features, scores, fallback list) and the
rate of both on a full batch.   python tools/gpu_fast_check.py [n_parity_clips] [batch]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package  # noqa: E402
from kws_testlib import MODELS, Oracle, special_clips  # noqa: E402

pkg = load_package()
n_par = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
models = sys.argv[3].split(",") if len(sys.argv) > 3 else ["cfg2_mfcc40_f32.kwsm", "l476_no_yes_f32.kwsm", "l476_no_yes.kwsm", "cfg2_mfcc40_int8.kwsm",
                                                           "cfg5_dscnn_mfcc40_f32.kwsm"]
o = Oracle()
sp = special_clips()
host = np.concatenate([o.synth(11, 0, n_par), np.stack(list(sp.values()))])
B = host.shape[0]
dev = torch.device("cuda:0")
pcm = torch.from_numpy(host).to(dev)
big = torch.empty((max(batch, 1), 16000), dtype=torch.int16, device=dev)
pkg.synth_clips_device(0, 0, max(batch, 1), 16000, big.data_ptr())
torch.cuda.synchronize()


def run(model, mode, pcm_t, want_f=True):
    n = pcm_t.shape[0]
    model.set_mode(mode)
    s = torch.zeros((n, model.n_labels), dtype=torch.float32, device=dev)
    f = torch.zeros((n, model.n_features), dtype=torch.float32, device=dev) if want_f else None
    q = torch.zeros((n, model.n_features), dtype=torch.int8, device=dev) if (want_f and not model.is_float) else None
    model.run_classifier_batch_device(pcm_t.data_ptr(), n, s.data_ptr(), f.data_ptr() if want_f else None, q.data_ptr() if q is not None else None)
    torch.cuda.synchronize()
    return s.cpu().numpy(), (f.cpu().numpy() if want_f else None), (q.cpu().numpy() if q is not None else None)


def rate(model, mode, steps=10):
    model.set_mode(mode)
    s = torch.zeros((batch, model.n_labels), dtype=torch.float32, device=dev)
    for _ in range(2):
        model.run_classifier_batch_device(big.data_ptr(), batch, s.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        model.run_classifier_batch_device(big.data_ptr(), batch, s.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return dt * 1e3, float(s.double().sum().item())


for name in models:
    m = pkg.Model(os.path.join(MODELS, name), device=0)
    try:
        m.set_mode(pkg.MODE_FAST)
    except pkg.KwsError as e:
        print("%s: fast mode unavailable: %s" % (name, e))
        m.close()
        continue
    s0, f0, q0 = run(m, pkg.MODE_EXACT, pcm)
    s1, f1, q1 = run(m, pkg.MODE_FAST, pcm)
    nfb = m.fast_fallback_count()
    s2, _, _ = run(m, pkg.MODE_FAST, pcm, want_f=False)
    df = np.abs(f1 - f0).max(axis=1)
    ds = np.abs(s1 - s0).max(axis=1)
    print("== %s  fused=%s  %d clips (+%d special): fallback %d" % (name, m.fast_is_fused, n_par, len(sp), nfb))
    print("   synthetic: max |dfeature| %.3g  max |dscore| %.3g  (scores-only call vs with features: %.3g)  nan: %d" %
          (df[:n_par].max(), ds[:n_par].max(), np.abs(s2 - s1).max(), int(np.isnan(s1).sum())))
    badc = np.nonzero(df > 1e-3)[0]
    if badc.size:
        print("   clips with |dfeature| > 1e-3:", badc[:40], "n =", badc.size)
        k = badc[0]
        d = np.abs(f1[k] - f0[k]).reshape(m.n_frames, -1)
        print("   clip %d: rows with diffs %s cols with diffs %s" % (k, np.nonzero(d.max(axis=1) > 1e-3)[0], np.nonzero(d.max(axis=0) > 1e-3)[0]))
        print("   fast ", f1[k].reshape(m.n_frames, -1)[np.nonzero(d.max(axis=1) > 1e-3)[0][0]][:8])
        print("   exact", f0[k].reshape(m.n_frames, -1)[np.nonzero(d.max(axis=1) > 1e-3)[0][0]][:8])
    if q0 is not None:
        print("   int8 input flips per clip: %.4f, clips whose scores changed: %d" % ((q1[:n_par] != q0[:n_par]).sum() / n_par, int((ds[:n_par] > 0).sum())))
    for i, k in enumerate(sp):
        print("   special %-22s |dfeature| %.3g |dscore| %.3g" % (k, df[n_par + i], ds[n_par + i]))
    if batch <= 0:
        m.close()
        continue
    t_e, c_e = rate(m, pkg.MODE_EXACT)
    t_f, c_f = rate(m, pkg.MODE_FAST)
    print("   batch %d: exact %.3f ms (%.2f M clips/s)  fast %.3f ms (%.2f M clips/s)  checksum exact %.6f fast %.6f  fallback %d" %
          (batch, t_e, batch / t_e / 1e3, t_f, batch / t_f / 1e3, c_e, c_f, m.fast_fallback_count()))
    m.close()
