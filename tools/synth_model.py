#!/usr/bin/env python3
"""Synthetic members of the Edge Impulse 1-D CNN graph family, as .kwsm blobs (tools/eon_import.py layout).

The reference ships two int8 models, both 32 mel filters / 13 cepstra.  BASELINE.json's other configurations
("40-band MFCC (49x40) + 2-Conv CNN, fp32", ...) have NO model file in the reference, so their
models are generated here: same graph (RESHAPE/CONV_2D/ADD/MAX_POOL_2D/FULLY_CONNECTED/SOFTMAX), seeded random int8
weights and quantisation parameters; tools/dequantize_model.py turns one into its float32 twin.  The parity tests use
the same generator for other widths / taps / pools / label counts.

    python tools/synth_model.py models/cfg2_mfcc40_int8.kwsm --seed 40 --num-filters 40 --ncep 40 --low 300 --high 0
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import eon_import  # noqa: E402

def dequantised_constants(tensors):
    """float64 value of every constant tensor of an int8 graph: (q - zero_point[ch]) * scale[ch] (what tools/dequantize_model.py stores)"""
    out = {}
    for i, t in enumerate(tensors):
        if not t["const"] or not t["scale"]:
            continue
        q = np.frombuffer(t["data"], np.int8 if t["type"] == 9 else np.int32).astype(np.float64).reshape(t["dims"])
        scale, zero = np.float64(t["scale"]), np.float64(t["zero"])
        if len(scale) > 1:
            shape = [1] * len(t["dims"])
            shape[t["qdim"]] = len(scale)
            out[i] = (q - zero.reshape(shape)) * scale.reshape(shape)
        else:
            out[i] = (q - zero[0]) * scale[0]
    return out


def float_forward(tensors, nodes, t_in, x, stop_before_op=None, on_output=None):
    """The graph's float32 twin evaluated in numpy for a batch x [n][features] -- the arithmetic of the reference's float kernels
    (TFL/kernels/internal/reference/conv.h:28-99, depthwiseconv_float.h:25, add.h:179-215, pooling.h:189-237, fully_connected.h:26-60)
    without their rounding order: used to CALIBRATE a synthetic head, never as a checker.  Activations are kept as [n][time][channel];
    RESHAPE only renames axes in this graph family.  Returns {tensor id: value}; stops before the first node whose op is stop_before_op."""
    act = {t_in: np.asarray(x, np.float64)}

    def fused(v, a):                                    # TfLiteFusedActivation: 0 none, 1 relu, 3 relu6
        return v if a == 0 else np.maximum(v, 0.0) if a == 1 else np.clip(v, 0.0, 6.0)

    def windows(v, size, out_w, stride, pad_left):      # [n][out_w][size][c], rows outside the image = NaN (callers mask them)
        n, w, c = v.shape
        idx = np.arange(out_w)[:, None] * stride - pad_left + np.arange(size)[None, :]
        ok = (idx >= 0) & (idx < w)
        g = v[:, np.clip(idx, 0, w - 1), :]
        return g, ok

    for nd in nodes:
        if nd["op"] == stop_before_op:
            break
        const = dequantised_constants(tensors)          # per node: on_output may have re-quantised a bias
        o, p = nd["out"][0], nd["p"]
        dims = tensors[o]["dims"]
        a = act.get(nd["in"][0])
        if nd["op"] == 0:                               # RESHAPE
            n = a.shape[0]
            act[o] = a.reshape(n, -1) if len(dims) == 2 else a.reshape(n, -1, dims[-1])
        elif nd["op"] in (1, 6):                        # CONV_2D / DEPTHWISE_CONV_2D, 1 x K, stride 1
            w = const[nd["in"][1]]
            b = const[nd["in"][2]]
            n, in_w, in_c = a.shape
            taps = w.shape[2]
            out_w = in_w if p[0] == 1 else in_w - taps + 1
            pad_left = max(0, (out_w - 1) + taps - in_w) // 2
            g, ok = windows(a, taps, out_w, 1, pad_left)
            g = np.where(ok[None, :, :, None], g, 0.0)
            if nd["op"] == 1:
                v = np.einsum("nwtc,otc->nwo", g, w[:, 0]) + b
            else:
                mult = p[6]
                v = np.einsum("nwto,to->nwo", np.repeat(g, mult, axis=3), w[0, 0]) + b
            act[o] = fused(v, p[3])
        elif nd["op"] == 2:                             # ADD of a per-channel constant
            act[o] = fused(a + const[nd["in"][1]], p[0])
        elif nd["op"] == 3:                             # MAX_POOL_2D over time
            n, in_w, c = a.shape
            size, stride = p[4], p[2]
            out_w = dims[1]
            pad_left = max(0, (out_w - 1) * stride + size - in_w) // 2 if p[0] == 1 else 0
            g, ok = windows(a, size, out_w, stride, pad_left)
            act[o] = fused(np.where(ok[None, :, :, None], g, -np.inf).max(axis=2), p[5])
        elif nd["op"] == 4:                             # FULLY_CONNECTED
            act[o] = fused(a @ const[nd["in"][1]].T + const[nd["in"][2]], p[0])
        elif nd["op"] == 5:                             # SOFTMAX
            e = np.exp((a - a.max(axis=1, keepdims=True)) * nd["beta"])
            act[o] = e / e.sum(axis=1, keepdims=True)
        else:
            raise ValueError("op %d" % nd["op"])
        if on_output:
            on_output(nd, act[o])
    return act


def calibrate_ranges(tensors, nodes, t_in, x):
    """Post-training quantisation of a synthetic graph's ACTIVATIONS: the drawn activation scales / zero points are arbitrary, so the int8
    graph clamps most of what flows through it and has little to do with its float twin.  Here every CONV_2D / DEPTHWISE_CONV_2D / ADD /
    FULLY_CONNECTED output gets the range its float twin shows on the calibration set x (min .. max, including 0), RESHAPE and
    MAX_POOL_2D outputs keep their input's parameters (as TFLite requires), and every int32 bias is re-quantised at its new scale
    input scale x filter scale[ch] (kernel_util_lite.cc:47-120 checks that product) from its old float value.  Weights and the input
    tensor keep their drawn parameters."""
    def on_output(nd, v):
        o, i0 = nd["out"][0], nd["in"][0]
        if nd["op"] in (0, 3):
            tensors[o]["scale"], tensors[o]["zero"] = list(tensors[i0]["scale"]), list(tensors[i0]["zero"])
            return
        if nd["op"] == 5:
            return
        lo, hi = min(float(v.min()), 0.0), max(float(v.max()), 0.0)
        sc = float(np.float32((hi - lo) / 255.0))
        tensors[o]["scale"], tensors[o]["zero"] = [sc], [int(np.clip(round(-128 - lo / sc), -128, 127))]

    def before(nd):                                     # the bias of a node whose input scale has just been settled
        if nd["op"] not in (1, 4, 6):
            return
        tb, tw, i0 = nd["in"][2], nd["in"][1], nd["in"][0]
        old = np.frombuffer(tensors[tb]["data"], np.int32).astype(np.float64) * np.float64(tensors[tb]["scale"])
        new_scale = [float(np.float32(tensors[i0]["scale"][0]) * np.float32(s_)) for s_ in tensors[tw]["scale"]]
        tensors[tb]["scale"] = new_scale
        tensors[tb]["data"] = np.round(old / np.float64(new_scale)).astype(np.int32).tobytes()

    act = {t_in: np.asarray(x, np.float64)}
    for nd in nodes:
        before(nd)
        sub = float_forward(tensors, [nd], nd["in"][0], act[nd["in"][0]], on_output=on_output) if nd["op"] != 5 else None
        if sub is None:
            break
        act[nd["out"][0]] = sub[nd["out"][0]]


def calibrate_head(tensors, nodes, t_in, tfw, tfb, tfo, logit_std, seed, n_cal=256):
    """Give a synthetic graph a head with a sane logit scale (synth_model_blob's logit_std).  The network's input is cmvnw's output:
    every column standardised over its window -- so the calibration set is n_cal matrices of independent N(0, 1) values (measured: the
    logit statistics on those equal the ones on the bench's synthetic clips to a few percent).  With z = W x the float twin's
    bias-free logits on that set: the FULLY_CONNECTED weight scale is multiplied by logit_std / pooled deviation of (z - class mean), the
    int32 biases become round(-class mean / bias scale) + the drawn ones, and the output tensor's scale spans +-8 logit_std."""
    rng = np.random.default_rng(100000 + seed)
    F = tensors[t_in]["dims"][1]
    x = rng.standard_normal((n_cal, F))
    fc = [nd for nd in nodes if nd["op"] == 4][0]
    calibrate_ranges(tensors, nodes, t_in, x)
    feat = float_forward(tensors, nodes, t_in, x, stop_before_op=4)[fc["in"][0]]
    w = dequantised_constants(tensors)[tfw]
    z = feat @ w.T
    mean = z.mean(axis=0)
    k = logit_std / float(np.sqrt(((z - mean) ** 2).mean()))
    fw_scale = float(np.float32(tensors[tfw]["scale"][0] * k))
    b_scale = float(np.float32(tensors[fc["in"][0]]["scale"][0]) * np.float32(fw_scale))
    drawn = np.frombuffer(tensors[tfb]["data"], np.int32)
    bias = np.round(-mean * k / b_scale).astype(np.int64) + drawn
    assert np.abs(bias).max() < 2 ** 31
    tensors[tfw]["scale"] = [fw_scale]
    tensors[tfb]["scale"] = [b_scale]
    tensors[tfb]["data"] = bias.astype(np.int32).tobytes()
    tensors[tfo]["scale"] = [float(np.float32(logit_std * 8.0 / 128.0))]


def synth_model_blob(seed, ncep=13, win_size=101, low=300, high=4000, blocks=((30, 7, 7), (10, 7, 7)), n_labels=4,
                     conv_bias=False, add_bias=True, num_filters=32, raw_samples=16000, fft_length=256, frame_length=0.02,
                     frame_stride=0.02, pre_cof=0.98, dsp_block="mfcc", logit_std=None, quantize_filterbank=False):
    """blocks: sequence of
         (out_channels, taps, pool)              CONV_2D 1xK (+ optional int32 bias) -> ADD(int8 per-channel)+ReLU -> MAX_POOL
         ("dw", depth_mult, taps, pool, act)     DEPTHWISE_CONV_2D 1xK with int32 bias and fused activation -> MAX_POOL
         ("pw", out_channels, act)               CONV_2D 1x1 with int32 bias and fused activation (pointwise)
       (pool 1 = no pooling node, a negative pool = VALID padding: the ragged tail of the time axis is dropped; act: TfLiteFusedActivation 0 none, 1 relu, 3 relu6).
       logit_std: None = the FULLY_CONNECTED layer as drawn (random weights and biases: one class usually wins every clip by tens
       of logit units, the softmax is saturated and a score says nothing about the logits behind it); a number = the head is
       calibrated (calibrate_head): scale and biases are set so that, over standardised random feature matrices -- which is what
       cmvnw hands the network --, every class's logit has zero mean and the logits' pooled deviation is logit_std (the shipped
       impulse's is 1.5).  No random draw is added or removed: every other tensor is the one the same seed gave before.
       Frame geometry defaults to the shipped one (1 s at 16 kHz, 20 ms frames and stride: 49 frames); raw_samples / frame_length /
       frame_stride / fft_length / pre_cof change the DSP block (the frame count follows speechpy's rule, processing.hpp:260-284)."""
    rng = np.random.default_rng(seed)
    flen_s = int(round(16000 * np.float32(frame_length)))
    n_frames = int(np.floor(np.float32(raw_samples - flen_s) / np.float32(round(16000 * np.float32(frame_stride)))))
    assert n_frames >= 1
    if dsp_block == "mfe":          # extract_mfe_features: the feature matrix is [frames][mel filters]
        ncep = num_filters
    F = n_frames * ncep
    tensors, nodes = [], []

    def T(ttype, dims, const=False, scale=(), zero=(), data=b"", qdim=0):
        nbytes = int(np.prod(dims)) * {1: 4, 2: 4, 9: 1}[ttype]
        if const:
            assert len(data) == nbytes
        tensors.append({"type": ttype, "dims": list(dims), "nbytes": nbytes, "const": const, "scale": list(scale),
                        "zero": list(zero), "qdim": qdim, "data": data})
        return len(tensors) - 1

    def node(op, ins, outs, p=None, beta=0.0):
        pp = [0] * 8
        if p:
            pp[:len(p)] = p
        nodes.append({"op": op, "in": list(ins), "out": list(outs), "p": pp, "beta": beta})

    def shape_const(dims):
        return T(2, [len(dims)], True, data=np.int32(dims).tobytes())

    def rscale(lo, hi):
        return float(np.float32(np.exp(rng.uniform(np.log(lo), np.log(hi)))))

    in_scale, in_zp = rscale(0.03, 0.06), int(rng.integers(-20, 5))
    t_in = T(9, [1, F], scale=[in_scale], zero=[in_zp])
    cur, cur_scale, cur_zp, w, c = t_in, in_scale, in_zp, n_frames, ncep
    t = T(9, [1, 1, w, c], scale=[cur_scale], zero=[cur_zp])
    node(0, [cur, shape_const([1, 1, w, c])], [t])
    cur = t
    def add_pool(pool):
        nonlocal cur, w
        if pool in (0, 1):
            return
        valid = pool < 0                                                # negative: VALID padding (floor, a ragged tail is dropped)
        pool = abs(pool)
        t4 = T(9, [1, w, 1, c_out], scale=[cur_scale], zero=[cur_zp])
        node(0, [cur, shape_const([1, w, 1, c_out])], [t4])
        pw = w // pool if valid else (w + pool - 1) // pool
        assert pw >= 1 and (pw - 1) * pool < w           # SAME: the last window may be ragged (clipped to the tensor)
        tp = T(9, [1, pw, 1, c_out], scale=[cur_scale], zero=[cur_zp])
        node(3, [t4], [tp], [2 if valid else 1, 1, pool, 1, pool, 0])   # VALID / SAME, stride (w1,hP), filter (w1,hP)
        w = pw
        t5 = T(9, [1, 1, w, c_out], scale=[cur_scale], zero=[cur_zp])
        node(0, [tp, shape_const([1, 1, w, c_out])], [t5])
        cur = t5

    for blk in blocks:
        if blk[0] == "dw":
            _, mult, taps, pool, act = blk
            c_out = c * mult
            wq = rng.integers(-127, 128, (1, 1, taps, c_out)).astype(np.int8)
            wscales = [rscale(0.002, 0.02) for _ in range(c_out)]
            tw = T(9, [1, 1, taps, c_out], True, wscales, [0] * c_out, wq.tobytes(), qdim=3)
            bias = rng.integers(-300, 300, c_out).astype(np.int32)
            tb = T(2, [c_out], True, [cur_scale * s_ for s_ in wscales], [0] * c_out, bias.tobytes())
            out_scale, out_zp = rscale(0.03, 0.12), int(rng.integers(-128, -90)) if act else int(rng.integers(-30, 40))
            tc = T(9, [1, 1, w, c_out], scale=[out_scale], zero=[out_zp])
            node(6, [cur, tw, tb], [tc], [1, 1, 1, act, 1, 1, mult])          # SAME, stride 1
            cur, cur_scale, cur_zp = tc, out_scale, out_zp
            add_pool(pool)
            c = c_out
            continue
        if blk[0] == "pw":
            _, c_out, act = blk
            wq = rng.integers(-127, 128, (c_out, 1, 1, c)).astype(np.int8)
            wscales = [rscale(0.001, 0.008) for _ in range(c_out)]
            tw = T(9, [c_out, 1, 1, c], True, wscales, [0] * c_out, wq.tobytes())
            bias = rng.integers(-300, 300, c_out).astype(np.int32)
            tb = T(2, [c_out], True, [cur_scale * s_ for s_ in wscales], [0] * c_out, bias.tobytes())
            out_scale, out_zp = rscale(0.03, 0.12), int(rng.integers(-128, -90)) if act else int(rng.integers(-30, 40))
            tc = T(9, [1, 1, w, c_out], scale=[out_scale], zero=[out_zp])
            node(1, [cur, tw, tb], [tc], [1, 1, 1, act, 1, 1])
            cur, cur_scale, cur_zp = tc, out_scale, out_zp
            c = c_out
            continue
        (oc, taps, pool) = blk
        c_out = oc
        wq = rng.integers(-127, 128, (oc, 1, taps, c)).astype(np.int8)
        wscales = [rscale(0.001, 0.005) for _ in range(oc)]
        tw = T(9, [oc, 1, taps, c], True, wscales, [0] * oc, wq.tobytes())
        bias = rng.integers(-200, 200, oc).astype(np.int32) if conv_bias else np.zeros(oc, np.int32)
        tb = T(2, [oc], True, [cur_scale * s for s in wscales], [0] * oc, bias.tobytes())
        out_scale, out_zp = rscale(0.03, 0.12), int(rng.integers(-30, 40))
        tc = T(9, [1, 1, w, oc], scale=[out_scale], zero=[out_zp])
        node(1, [cur, tw, tb], [tc], [1, 1, 1, 0, 1, 1])                    # SAME, stride 1, no activation
        cur, cur_scale, cur_zp = tc, out_scale, out_zp
        if add_bias:
            t3 = T(9, [1, w, oc], scale=[cur_scale], zero=[cur_zp])
            node(0, [cur, shape_const([1, w, oc])], [t3])
            bq = rng.integers(-127, 1, oc).astype(np.int8)
            tbq = T(9, [oc], True, [rscale(0.001, 0.01)], [0], bq.tobytes())
            a_scale = rscale(0.02, 0.05)
            ta = T(9, [1, w, oc], scale=[a_scale], zero=[-128])
            node(2, [t3, tbq], [ta], [1])                                   # ReLU
            cur, cur_scale, cur_zp = ta, a_scale, -128
        if pool not in (0, 1):
            add_pool(pool)
        elif add_bias:                                                      # back to NHWC for the next convolution
            t6 = T(9, [1, 1, w, oc], scale=[cur_scale], zero=[cur_zp])
            node(0, [cur, shape_const([1, 1, w, oc])], [t6])
            cur = t6
        c = oc
    fc_in = w * c
    tf = T(9, [1, fc_in], scale=[cur_scale], zero=[cur_zp])
    node(0, [cur, shape_const([1, fc_in])], [tf])
    fw = rng.integers(-127, 128, (n_labels, fc_in)).astype(np.int8)
    fw_scale = rscale(0.005, 0.02)
    tfw = T(9, [n_labels, fc_in], True, [fw_scale], [0], fw.tobytes())
    tfb = T(2, [n_labels], True, [cur_scale * fw_scale], [0], rng.integers(-400, 400, n_labels).astype(np.int32).tobytes())
    tfo = T(9, [1, n_labels], scale=[rscale(0.05, 0.2)], zero=[int(rng.integers(-10, 10))])
    node(4, [tf, tfw, tfb], [tfo], [0])
    tso = T(9, [1, n_labels], scale=[0.00390625], zero=[-128])
    node(5, [tfo], [tso], beta=1.0)
    if logit_std is not None:
        calibrate_head(tensors, nodes, t_in, tfw, tfb, tfo, float(logit_std), seed)
    meta = {"labels": ["label%d" % i for i in range(n_labels)],
            "dsp": {"axes": 1, "num_cepstral": ncep, "frame_length": frame_length, "frame_stride": frame_stride, "num_filters": num_filters,
                    "fft_length": fft_length, "win_size": win_size, "low_frequency": low, "high_frequency": high,
                    "pre_cof": pre_cof, "pre_shift": 1, "block": 1 if dsp_block == "mfe" else 0,
                    "quantize_filterbank": 1 if quantize_filterbank else 0},
            "raw_sample_count": raw_samples, "frequency": 16000, "nn_input_frame_size": F}
    return eon_import.serialise(tensors, nodes, t_in, tso, meta)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--ncep", type=int, default=13)
    ap.add_argument("--num-filters", type=int, default=32)
    ap.add_argument("--win-size", type=int, default=101)
    ap.add_argument("--low", type=int, default=300)
    ap.add_argument("--high", type=int, default=4000)
    ap.add_argument("--labels", type=int, default=4)
    ap.add_argument("--blocks", default="30,7,7;10,7,7", help="per block: out_channels,taps,pool | dw,depth_mult,taps,pool,act | pw,out_channels,act")
    ap.add_argument("--logit-std", type=float, default=None, help="calibrate the head: zero-mean class logits of this pooled deviation on standardised features")
    a = ap.parse_args()
    blocks = tuple(tuple(v if v in ("dw", "pw") else int(v) for v in b.split(",")) for b in a.blocks.split(";"))
    blob = synth_model_blob(a.seed, ncep=a.ncep, win_size=a.win_size, low=a.low, high=a.high, blocks=blocks,
                            n_labels=a.labels, num_filters=a.num_filters, logit_std=a.logit_std)
    with open(a.out, "wb") as f:
        f.write(blob)
    print(a.out, len(blob), "bytes")


if __name__ == "__main__":
    main()
