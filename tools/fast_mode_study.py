#!/usr/bin/env python3
"""Sensitivity study behind KWS_MODE_FAST: how far do the scores move when the MFCC block is
evaluated in plain fp32 in ANY operation order (library FFT, re^2+im^2 power, matrix DCT, running-sum cmvnw) instead of
replaying the reference's operation order?  CPU only (numpy emulation of the fast arithmetic against the C oracle).

    python tools/fast_mode_study.py [n_clips] [model.kwsm]
"""
import os
import sys

import numpy as np
import scipy.fft

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from kws_testlib import MODELS, Oracle, OracleModel, special_clips  # noqa: E402

f32 = np.float32


def fast_log(a):
    """numpy::log (SDK/dsp/numpy.hpp:1350-1371), vectorised; fmaf emulated through float64 (exact for these operands)"""
    a = np.asarray(a, f32)
    g = a.view(np.int32).astype(np.int64)
    e = ((g - 0x3f2aaaab) & 0xff800000).astype(np.int64)
    e = np.where(e >= 2 ** 31, e - 2 ** 32, e)
    m = (g - e).astype(np.int32).view(f32)
    i = (e.astype(f32) * f32(1.19209290e-7)).astype(f32)
    f = (m - f32(1.0)).astype(f32)
    s = (f * f).astype(f32)
    fma = lambda x, y, z: (np.float64(x) * np.float64(y) + np.float64(z)).astype(f32)  # noqa: E731
    r = fma(f32(0.230836749), f, f32(-0.279208571))
    t = fma(f32(0.331826031), f, f32(-0.498910338))
    r = fma(r, s, t)
    r = fma(r, s, f)
    return fma(i, f32(0.693147182), r)


def pad_map(rows, pad):
    """numpy::pad_1d_symmetric row order (SDK/dsp/numpy.hpp:479-541)"""
    idx = np.arange(-pad, rows + pad)
    period = 2 * rows
    m = np.mod(idx, period)
    return np.where(m < rows, m, period - 1 - m)


def fast_mfcc(pcm, cfg, fb, dct_mode="matrix", oracle=None):
    """features [B][frames*ncep] with fp32 arithmetic in an arbitrary order"""
    B, n = pcm.shape
    x = (pcm.astype(f32) * f32(1.0 / 32768.0)).astype(f32)
    prev = np.roll(x, 1, axis=1)
    y = (x - (f32(cfg.pre_cof) * prev).astype(f32)).astype(f32)
    flen = int(round(cfg.sampling_frequency * cfg.frame_length))
    stride = int(round(cfg.sampling_frequency * cfg.frame_stride))
    nfr = int(np.floor((n - flen) / stride))
    N, nfft, ncep = cfg.num_filters, cfg.fft_length, cfg.num_cepstral
    idx = (np.arange(nfr) * stride)[:, None] + np.arange(min(flen, nfft))[None, :]
    fr = y[:, idx]                                                     # [B][frames][256]
    X = scipy.fft.rfft(fr.astype(f32), n=nfft, axis=-1)                # complex64
    p = ((X.real * X.real + X.imag * X.imag).astype(f32) * f32(1.0 / nfft)).astype(f32)
    energy = p.sum(axis=-1, dtype=f32)
    energy = np.where(energy == 0, f32(np.finfo(f32).eps), energy)
    mel = (p @ fb.astype(f32)).astype(f32)
    mel = np.where(mel == 0, f32(np.finfo(f32).eps), mel)
    lm = fast_log(mel)
    if dct_mode == "exact":
        out = np.empty_like(lm)
        for b in range(B):
            for r in range(nfr):
                out[b, r] = oracle.dct2_ortho(lm[b, r])
    else:
        k = np.arange(N // 2 + 1)[:, None]
        j = np.arange(N)[None, :]
        D = np.cos(np.pi * k * (2 * j + 1) / (2 * N)) * 2.0
        D[0] *= np.sqrt(1.0 / (4 * N))
        D[1:] *= np.sqrt(1.0 / (2 * N))
        lo = (lm @ D.T.astype(f32)).astype(f32)                        # outputs 0..N/2
        hi = ((lm[..., N // 2 + 1:] * f32(2.0)) * f32(np.sqrt(f32(1.0 / (2 * N))))).astype(f32)   # stale inputs (fast-dct-fft.cpp:71)
        out = np.concatenate([lo, hi], axis=-1)
    cep = out[..., :ncep].copy()
    cep[..., 0] = fast_log(energy)
    # cmvnw with running sums over the padded row map, shifted by the column's first value
    win, pad = cfg.win_size, (cfg.win_size - 1) // 2
    pm = pad_map(nfr, pad)
    piv = cep[:, :1, :]
    d = (cep - piv).astype(f32)
    dp = d[:, pm, :]                                                   # padded
    S = np.zeros_like(d)
    Q = np.zeros_like(d)
    s = dp[:, :win].sum(axis=1, dtype=f32)
    q = (dp[:, :win] * dp[:, :win]).sum(axis=1, dtype=f32)
    for r in range(nfr):
        S[:, r], Q[:, r] = s, q
        if r + 1 < nfr:
            a, bb = dp[:, r + win], dp[:, r]
            s = ((s + a).astype(f32) - bb).astype(f32)
            q = ((q + a * a).astype(f32) - bb * bb).astype(f32)
    mean = (S / f32(win)).astype(f32)
    var = np.maximum((Q / f32(win)).astype(f32) - mean * mean, f32(0))
    sd = np.sqrt(var).astype(f32)
    feat = ((d - mean) / (sd + f32(np.finfo(f32).eps))).astype(f32)
    cond = sd / (np.abs(mean + piv) + f32(1e-30))                      # conditioning of each window
    return feat.reshape(B, -1), sd, cond


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(MODELS, "cfg2_mfcc40_f32.kwsm")
    o = Oracle()
    om = OracleModel(o, path)
    cfg = om.cfg
    fb = o.filterbanks(cfg)
    sp = special_clips()
    pcm = np.concatenate([o.synth(7, 0, n), np.stack(list(sp.values()))])
    s_ref, f_ref, q_ref = om.run_batch(pcm, want_features=True)
    is_float = bool(o.L.kwso_model_is_float(om.h))
    for dct_mode in ("matrix",):
        feat, sd, cond = fast_mfcc(pcm, cfg, fb, dct_mode, o)
        if is_float:
            s = np.stack([om.nn_invoke_f32(f) for f in feat])
        else:
            s = np.stack([om.run_inference(f) for f in feat])
        df = np.abs(feat - f_ref).max(axis=1)
        ds = np.abs(s - s_ref).max(axis=1)
        print("== %s, dct=%s, %d synthetic clips" % (os.path.basename(path), dct_mode, n))
        print("   synthetic: max |dfeature| %.3g (median of per-clip max %.3g), max |dscore| %.3g" % (df[:n].max(), np.median(df[:n]), ds[:n].max()))
        print("   min window std %.3g, min std/|mean| %.3g" % (sd[:n].min(), cond[:n].min()))
        if not is_float:
            qf = np.stack([om.quantize_input(f) for f in feat])
            print("   int8 input flips: %.3g per clip (of %d), clips with any score change: %d of %d" %
                  ((qf[:n] != q_ref[:n]).sum() / n, feat.shape[1], int((ds[:n] > 0).sum()), n))
        for i, name in enumerate(sp):
            k = n + i
            print("   special %-22s |dfeature| %.3g |dscore| %.3g  min std %.3g min std/|mean| %.3g" % (name, df[k], ds[k], sd[k].min(), cond[k].min()))


if __name__ == "__main__":
    main()
