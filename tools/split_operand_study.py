#!/usr/bin/env python3
"""CPU study behind KwsFastBlock::hconv (csrc/kws_fast.hip: fast_conv_tiles_h): how far the scores of a float graph move when its CONV_2D
contractions run on 16-bit matrix instructions with SPLIT operands instead of fp32 ones.  The graph is evaluated in float64 (the yardstick),
then with the contraction of the first / of every CONV_2D block replaced by
    h3     half hi + lo of both operands (scaled by powers of two), products hi hi + hi lo + lo hi          <- what the kernel does
    h3ftz  the same with subnormal halves flushed to zero (whatever the matrix pipe does with them)
    h4     + lo lo
    x3/x4/x6  bfloat16 splits with 3 / 4 / 6 products
on the oracle's features of synthetic clips.   python tools/split_operand_study.py [model.kwsm] [n_clips]
Result (cfg2_mfcc40, 512 clips): oracle fp32 vs float64 2.2e-6 in a score (the reference's own summation order); h3 5.4e-7; x3 2.1e-5."""
import os, sys, numpy as np
sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo/tests")
import eon_import, synth_model
from kws_testlib import Oracle, OracleModel, MODELS

def bf16_rn(x):
    x = np.asarray(x, np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7fff + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)

def split(x, n):
    parts = []
    r = np.asarray(x, np.float32).copy()
    for i in range(n):
        p = bf16_rn(r)
        parts.append(p)
        r = (r - p).astype(np.float32)
    return parts

def conv_variant(g, w, mode):
    # g [n][w][t][c] float, w [o][t][c]
    if mode == "f64":
        return np.einsum("nwtc,otc->nwo", g.astype(np.float64), w.astype(np.float64))
    if mode in ("h3", "h4", "h3ftz"):
        def hsplit(x, n, ftz):
            parts = []; r = np.asarray(x, np.float32).copy()
            for i in range(n):
                p = r.astype(np.float16)
                if ftz: p = np.where(np.abs(p) < 6.104e-5, np.float16(0), p)
                p = p.astype(np.float32); parts.append(p); r = (r - p).astype(np.float32)
            return parts
        ftz = mode == "h3ftz"
        sg, sw = 256.0, 4096.0
        assert np.abs(g).max() * sg < 65000 and np.abs(w).max() * sw < 65000, (np.abs(g).max(), np.abs(w).max())
        gs, ws = hsplit(g * np.float32(sg), 2, ftz), hsplit(w * np.float32(sw), 2, ftz)
        pairs = [(0,0),(0,1),(1,0)] + ([(1,1)] if mode == "h4" else [])
        acc = 0.0
        for i, j in pairs:
            acc = acc + np.einsum("nwtc,otc->nwo", gs[i].astype(np.float64), ws[j].astype(np.float64))
        return (acc.astype(np.float32) * np.float32(1.0 / (sg * sw))).astype(np.float64)
    nsp = {"x3": 2, "x4": 2, "x6": 3}[mode]
    gs, ws = split(g, nsp), split(w, nsp)
    pairs = {"x3": [(0,0),(0,1),(1,0)], "x4": [(0,0),(0,1),(1,0),(1,1)], "x6": [(0,0),(0,1),(1,0),(0,2),(2,0),(1,1)]}[mode]
    acc = 0.0
    for i, j in pairs:   # products of bf16 are exact in fp32; accumulation in MFMA ~fp32: model with f64 sum then round to f32
        acc = acc + np.einsum("nwtc,otc->nwo", gs[i].astype(np.float64), ws[j].astype(np.float64))
    return acc.astype(np.float32).astype(np.float64)

def forward(tensors, nodes, t_in, x, mode_by_conv):
    const = synth_model.dequantised_constants(tensors)
    const = {k: np.float32(v).astype(np.float64) for k, v in const.items()}
    act = {t_in: np.asarray(x, np.float64)}
    ci = 0
    def fused(v, a): return v if a == 0 else np.maximum(v, 0.0) if a == 1 else np.clip(v, 0.0, 6.0)
    def windows(v, size, out_w, stride, pad_left):
        n, w, c = v.shape
        idx = np.arange(out_w)[:, None] * stride - pad_left + np.arange(size)[None, :]
        ok = (idx >= 0) & (idx < w)
        return v[:, np.clip(idx, 0, w - 1), :], ok
    logits = None
    for nd in nodes:
        o, p = nd["out"][0], nd["p"]; dims = tensors[o]["dims"]; a = act.get(nd["in"][0])
        if nd["op"] == 0:
            n = a.shape[0]; act[o] = a.reshape(n, -1) if len(dims) == 2 else a.reshape(n, -1, dims[-1])
        elif nd["op"] == 1:
            w = const[nd["in"][1]]; b = const[nd["in"][2]]
            n, in_w, in_c = a.shape; taps = w.shape[2]
            out_w = in_w if p[0] == 1 else in_w - taps + 1
            pad_left = max(0, (out_w - 1) + taps - in_w) // 2
            g, ok = windows(a, taps, out_w, 1, pad_left)
            g = np.where(ok[None, :, :, None], g, 0.0)
            v = conv_variant(np.float32(g), np.float32(w[:, 0]), mode_by_conv[ci]) + b
            ci += 1
            act[o] = fused(v, p[3])
        elif nd["op"] == 2:
            act[o] = fused(a + const[nd["in"][1]], p[0])
        elif nd["op"] == 3:
            n, in_w, c = a.shape; size, stride = p[4], p[2]; out_w = dims[1]
            pad_left = max(0, (out_w - 1) * stride + size - in_w) // 2 if p[0] == 1 else 0
            g, ok = windows(a, size, out_w, stride, pad_left)
            act[o] = fused(np.where(ok[None, :, :, None], g, -np.inf).max(axis=2), p[5])
        elif nd["op"] == 4:
            act[o] = fused(a @ const[nd["in"][1]].T + const[nd["in"][2]], p[0]); logits = act[o]
        elif nd["op"] == 5:
            e = np.exp((a - a.max(axis=1, keepdims=True)) * nd["beta"]); act[o] = e / e.sum(axis=1, keepdims=True); scores = act[o]
        else:
            raise ValueError(nd["op"])
    return logits, scores

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2_mfcc40_int8.kwsm"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
blob = open(os.path.join(MODELS, name), "rb").read()
tensors, nodes, t_in, t_out, meta = eon_import.parse_blob(blob)
oracle = Oracle()
pcm = oracle.synth(7, 0, N)
om = OracleModel(oracle, os.path.join(MODELS, name.replace("int8", "f32")))
s_ref, f_ref, _ = om.run_batch(pcm, want_features=True)
x = f_ref.reshape(N, -1, meta["dsp"]["num_cepstral"]) if False else f_ref
nconv = sum(1 for nd in nodes if nd["op"] == 1)
l64, s64 = forward(tensors, nodes, t_in, x, ["f64"] * nconv)
print("oracle fp32 vs f64 forward: max |dscore| %.3g" % np.abs(s_ref - s64).max(), " logit std %.3g, |logit| max %.3g" % (l64.std(), np.abs(l64).max()))
for mode in ("h3", "h3ftz", "h4", "x3", "x4", "x6"):
    for which in ("conv1", "all"):
        modes = [mode if (which == "all" or i == 0) else "f64" for i in range(nconv)]
        l, s = forward(tensors, nodes, t_in, x, modes)
        dl = l - l64
        dd = (dl[:, :, None] - dl[:, None, :])
        print("%s %-5s: logit err max %.3g rms %.3g ; logit-diff err max %.3g ; score err max %.3g" % (mode, which, np.abs(dl).max(), np.sqrt((dl**2).mean()), np.abs(dd).max(), np.abs(s - s64).max()))
