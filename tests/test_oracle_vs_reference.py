"""Pins the C restatement against the UNMODIFIED reference compiled in place by `make -C oracle ref`
(only where /root/reference exists; skipped on the GPU box).  Randomised, broader than the goldens."""
import os

import numpy as np
import pytest

from kws_testlib import L476_CONFIG, ROOT, Reference, bits, special_clips

pytestmark = pytest.mark.skipif(
    not __import__("kws_testlib").have_reference(), reason="oracle/_ref not built")


def test_dsp_stages_random_clips(oracle, reference):
    cfg = L476_CONFIG()
    clips = oracle.synth(11, 100, 24)
    for c in clips:
        for off in (0, 320, 15360):
            assert (bits(oracle.preemphasis(c, 0.98, 1, off, 320)) == bits(reference.preemphasis(c, 0.98, 1, off, 320))).all()
        fr = oracle.preemphasis(c, 0.98, 1, 3200, 320)
        assert (bits(oracle.power_spectrum(fr, 256)) == bits(reference.power_spectrum(fr, 256))).all()
        a, b = oracle.mfe(c, cfg)
        a2, b2 = reference.mfe(c, cfg)
        assert (bits(a) == bits(a2)).all() and (bits(b) == bits(b2)).all()
        assert (bits(oracle.mfcc_nocmvn(c, cfg)) == bits(reference.mfcc_nocmvn(c, cfg))).all()
        assert (bits(oracle.extract_mfcc(c, cfg)) == bits(reference.extract_mfcc(c, cfg))).all()


@pytest.mark.parametrize("kw", [
    dict(),                                                      # shipped L476
    dict(high_frequency=0),                                      # shipped L432 (300..8000 Hz)
    dict(num_filters=40, num_cepstral=40, high_frequency=0),     # BASELINE "40-band" variant (radix-5 DCT FFT)
    dict(num_filters=40, num_cepstral=13, low_frequency=0, high_frequency=0),
    dict(fft_length=512),                                        # zero-padded frames
    dict(fft_length=128, win_size=51),
    dict(frame_stride=0.01, win_size=31),                        # overlapping frames
    dict(pre_cof=0.0),
    dict(fft_length=512, frame_length=0.0200625, frame_stride=0.0100625, win_size=31),   # 321-sample frames every 161 samples (odd both)
])
def test_mfcc_configs(oracle, reference, kw):
    cfg = L476_CONFIG().copy(**kw)
    for c in list(oracle.synth(5, 0, 3)) + [special_clips()["impulses"]]:
        assert oracle.num_frames(c.size, cfg) == reference.num_frames(c.size, cfg)
        assert (bits(oracle.filterbanks(cfg)) == bits(reference.filterbanks(cfg))).all()
        assert (bits(oracle.extract_mfcc(c, cfg)) == bits(reference.extract_mfcc(c, cfg))).all()


def test_short_clips(oracle, reference):
    cfg = L476_CONFIG()
    for n in (4000, 640, 1000, 15999):
        c = oracle.synth(9, 3, 1, n)[0]
        assert oracle.num_frames(n, cfg) == reference.num_frames(n, cfg)
        assert (bits(oracle.extract_mfcc(c, cfg)) == bits(reference.extract_mfcc(c, cfg))).all()


def test_leaf_functions(oracle, reference):
    rng = np.random.default_rng(3)
    xs = np.concatenate([np.exp(rng.uniform(-80, 80, 2000)), [1.1920929e-07, 1.0, 2.5]]).astype(np.float32)
    a = np.float32([oracle.L.kwso_log(float(x)) for x in xs])
    b = np.float32([reference.L.eiref_log(float(x)) for x in xs])
    assert (bits(a) == bits(b)).all()
    for f in rng.uniform(0, 8000, 200):
        assert oracle.L.kwso_frequency_to_mel(f) == reference.L.eiref_frequency_to_mel(f)
    for m in rng.uniform(0, 2900, 200):
        assert oracle.L.kwso_mel_to_frequency(m) == reference.L.eiref_mel_to_frequency(m)
    for n in (256, 32, 40, 512, 64, 16):
        x = rng.standard_normal(n).astype(np.float32)
        assert (bits(oracle.rfft_complex(x)) == bits(reference.rfft_complex(x))).all(), n
    for n in (32, 40, 13):
        if n % 2:
            continue
        x = rng.standard_normal(n).astype(np.float32) * 5
        assert (bits(oracle.dct2_ortho(x)) == bits(reference.dct2_ortho(x))).all(), n
    for rows, cols, win in ((49, 13, 101), (49, 40, 101), (11, 13, 101), (99, 13, 31), (1, 13, 101)):
        m = (rng.standard_normal((rows, cols)) * 3).astype(np.float32)
        assert (bits(oracle.cmvnw(m, win, True)) == bits(reference.cmvnw(m, win, True))).all()
    const = np.full((49, 13), 1.2345, np.float32)     # zero-variance columns (rounding canary)
    assert (bits(oracle.cmvnw(const, 101, True)) == bits(reference.cmvnw(const, 101, True))).all()


def test_nn_every_op_random_int8(oracle, reference, l476):
    rng = np.random.default_rng(0)
    for i in range(300):
        if i % 3 == 0:
            q = rng.integers(-128, 128, 637, dtype=np.int8)
        elif i % 3 == 1:
            q = np.clip(rng.normal(-11, 20, 637), -128, 127).astype(np.int8)
        else:
            q = np.full(637, rng.integers(-128, 128), np.int8)
        out, taps = l476.nn_invoke(q, taps=True)
        rt = reference.nn_taps(q)
        assert len(rt) == 15
        for tid, v in rt.items():
            assert (taps[tid] == v).all(), (i, tid)


def test_quantise_wrap_semantics(reference, l476):
    """ei_run_classifier.h:440 has no clamp: out-of-range features wrap (x86 cvttss2si + low byte)."""
    rng = np.random.default_rng(1)
    for scale in (1, 5, 20, 100, 1e4, 1e9, 1e12):
        for i in range(30):
            f = (rng.standard_normal(637) * scale).astype(np.float32)
            if i % 10 == 0:
                f[rng.integers(0, 637, 5)] = np.float32([np.nan, np.inf, -np.inf, 3e38, -3e38])
            assert (l476.run_inference(f) == reference.run_inference(f)).all()


def test_run_classifier_end_to_end(oracle, reference, l476):
    clips = oracle.synth(21, 0, 48)
    s = l476.run_batch(clips)
    for i, c in enumerate(clips):
        rc, rs, total_length_after, calls = reference.run_classifier(c)
        assert rc == 0 and total_length_after == 16000 and calls == 98   # SURVEY 8(b) ownership row
        assert (bits(s[i]) == bits(rs)).all()
    for name, c in special_clips().items():
        assert (bits(l476.run_batch(c)[0]) == bits(reference.run_classifier(c)[1])).all(), name


def test_float_kernels_pinned_leaf_by_leaf(oracle, reference):
    """No float model ships with the reference, but its SDK carries the float TFLite-Micro kernels.
    The restated float graph (fp32 twin of the shipped model, tools/dequantize_model.py) is pinned op by op: each
    reference kernel is fed the oracle's input tensor of that op and must reproduce the oracle's output tensor."""
    import ctypes as C
    import os
    from kws_testlib import MODELS, OracleModel
    mf = OracleModel(oracle, os.path.join(MODELS, "l476_no_yes_f32.kwsm"))
    assert oracle.L.kwso_model_is_float(mf.h) == 1
    L = reference.L
    fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)
    FLT_MAX = float(np.finfo(np.float32).max)

    def P(a):
        return np.ascontiguousarray(a, np.float32).ctypes.data_as(fp)

    rng = np.random.default_rng(2)
    for trial in range(6):
        feat = (rng.standard_normal(637) * (1 + trial)).astype(np.float32)
        out, t = mf.nn_invoke_f32(feat, taps=True)
        # tensor ids as in SURVEY appendix A: conv1 16->17 (W=7, B=6), add 18+2->19, pool 20->21, conv2 22->23 (W=9, B=8),
        # add 24+3->25, pool 26->27, fc 28->29 (W=5, B=4), softmax 29->30
        y = np.zeros(49 * 30, np.float32)
        L.eiref_f32_conv(P(t[16]), 1, 49, 13, P(t[7]), 30, 1, 7, P(t[6]), 3, 0, C.c_float(-FLT_MAX), C.c_float(FLT_MAX), P(y), 1, 49)
        assert (bits(y) == bits(t[17])).all()
        y = np.zeros(49 * 30, np.float32)
        L.eiref_f32_add_bcast(P(t[18]), (C.c_int * 4)(1, 1, 49, 30), P(t[2]), (C.c_int * 4)(1, 1, 1, 30), (C.c_int * 4)(1, 1, 49, 30),
                              C.c_float(0.0), C.c_float(FLT_MAX), P(y))
        assert (bits(y) == bits(t[19])).all()
        y = np.zeros(7 * 30, np.float32)
        L.eiref_f32_maxpool(P(t[20]), 49, 1, 30, 7, 1, 7, 1, C.c_float(-FLT_MAX), C.c_float(FLT_MAX), P(y), 7, 1)
        assert (bits(y) == bits(t[21])).all()
        y = np.zeros(7 * 10, np.float32)
        L.eiref_f32_conv(P(t[22]), 1, 7, 30, P(t[9]), 10, 1, 7, P(t[8]), 3, 0, C.c_float(-FLT_MAX), C.c_float(FLT_MAX), P(y), 1, 7)
        assert (bits(y) == bits(t[23])).all()
        y = np.zeros(70, np.float32)
        L.eiref_f32_add_bcast(P(t[24]), (C.c_int * 4)(1, 1, 7, 10), P(t[3]), (C.c_int * 4)(1, 1, 1, 10), (C.c_int * 4)(1, 1, 7, 10),
                              C.c_float(0.0), C.c_float(FLT_MAX), P(y))
        assert (bits(y) == bits(t[25])).all()
        y = np.zeros(10, np.float32)
        L.eiref_f32_maxpool(P(t[26]), 7, 1, 10, 7, 1, 7, 1, C.c_float(-FLT_MAX), C.c_float(FLT_MAX), P(y), 1, 1)
        assert (bits(y) == bits(t[27])).all()
        y = np.zeros(4, np.float32)
        L.eiref_f32_fc(P(t[28]), 10, P(t[5]), 4, P(t[4]), C.c_float(-FLT_MAX), C.c_float(FLT_MAX), P(y))
        assert (bits(y) == bits(t[29])).all()
        y = np.zeros(4, np.float32)
        L.eiref_f32_softmax(P(t[29]), 4, C.c_float(1.0), P(y))
        assert (bits(y) == bits(t[30])).all() and (bits(y) == bits(out)).all()


@pytest.mark.parametrize("name", sorted(__import__("kws_testlib").SYNTH_SPECS))
def test_synthetic_graphs_through_reference_op_registrations(name, oracle, reference, tmp_path):
    """Every synthetic graph (int8 and its float32 twin) evaluated by the reference's OWN TFLite-Micro op code
    (init/prepare/invoke of Register_*(), driven by eiref_graph_run) == the restatement, every tensor, bit for bit.
    Covers what no shipped model exercises: DEPTHWISE_CONV_2D (whose int8 path ignores its fused activation,
    depthwise_conv.cc:618-620), pointwise convolutions, fused activations, conv biases, the float kernels."""
    from kws_testlib import SYNTH_SPECS, OracleModel, synth_model_blob    # (puts tools/ on sys.path)
    from dequantize_model import dequantize
    blob = synth_model_blob(**SYNTH_SPECS[name])
    rng = np.random.default_rng(3)
    for kind, b in (("i8", blob), ("f32", dequantize(blob))):
        p = tmp_path / (name + kind + ".kwsm")
        p.write_bytes(b)
        om = OracleModel(oracle, str(p))
        for it in range(12):
            if kind == "i8":
                x = rng.integers(-128, 128, om.n_features).astype(np.int8)
                out, taps = reference.graph_run(b, x)
                oo, ot = om.nn_invoke(x, taps=True)
                assert (out == oo).all()
                for a, t in zip(taps, ot):
                    if a.dtype == np.int8:
                        assert (a == t).all(), (name, it)
            else:
                x = (rng.standard_normal(om.n_features) * np.float32(10.0) ** rng.integers(-2, 2)).astype(np.float32)
                out, taps = reference.graph_run(b, x)
                oo, ot = om.nn_invoke_f32(x, taps=True)
                for a, t in zip(taps[:-1], ot[:-1]):
                    if a.dtype == np.float32:
                        assert (bits(a) == bits(t)).all(), (name, it)           # everything up to the logits
                assert np.abs(out - oo).max() <= 1e-7                          # softmax: libm expf on both sides


def test_l432_model_through_reference_op_registrations(oracle, reference):
    """The second shipped impulse (L432, 3 classes).  Its own SDK copy cannot be built here, but its
    graph and tables can be evaluated by the reference's TFLite-Micro op code all the same: every tensor of
    models/l432_trick_or_treat.kwsm through init/prepare/invoke of Register_*() == the restatement, bit for bit; its DSP
    settings through the reference's extract_mfcc_features are the `high_frequency=0` case of the MFCC configurations above."""
    import os
    from kws_testlib import MODELS, OracleModel
    path = os.path.join(MODELS, "l432_trick_or_treat.kwsm")
    blob = open(path, "rb").read()
    om = OracleModel(oracle, path)
    rng = np.random.default_rng(432)
    for it in range(64):
        x = rng.integers(-128, 128, om.n_features).astype(np.int8)
        out, taps = reference.graph_run(blob, x)
        oo, ot = om.nn_invoke(x, taps=True)
        assert (out == oo).all(), it
        for a, t in zip(taps, ot):
            if a.dtype == np.int8:
                assert (a == t).all(), it


def _same_bits_or_both_nan(a, b):
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    return ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all()


def test_l432_mfe_block_normalisation_pinned(oracle, reference):
    """The MFE block of the newer SDK copy: processing::cmvnw(win, variance_normalization, scale) and numpy::normalize from the
    L432 headers compiled in place (oracle/_ref/libei_ref_l432dsp.so) == the restatement, bit for bit, for every flag
    combination; and extract_mfe_features' composition (feature::mfe on the raw signal -- the same text in both copies, so the
    L476 build's -- then cmvnw(win, false, true)) == kwso_extract_mfe, incl. constant clips whose normalisation is 0 * inf."""
    import os
    from kws_testlib import REF432_SO, ReferenceL432Dsp, L476_CONFIG, special_clips
    if not os.path.exists(REF432_SO):
        pytest.skip("oracle/_ref/libei_ref_l432dsp.so not built (no /root/reference here)")
    r432 = ReferenceL432Dsp()
    rng = np.random.default_rng(5)
    for rows, cols, win in ((49, 32, 101), (49, 40, 101), (12, 13, 5), (7, 3, 51), (49, 40, 13), (1, 4, 3)):
        m = (rng.standard_normal((rows, cols)) * rng.uniform(0.01, 50)).astype(np.float32)
        for vn in (False, True):
            for sc in (False, True):
                assert _same_bits_or_both_nan(r432.cmvnw(m, win, vn, sc), oracle.cmvnw_scale(m, win, vn, sc)), (rows, cols, win, vn, sc)
        a = r432.normalize(m)
        b = m.copy()
        oracle.L.kwso_normalize(b.ctypes.data_as(__import__("ctypes").c_void_p), b.size)
        assert _same_bits_or_both_nan(a, b)
    cfg = L476_CONFIG()
    clips = list(oracle.synth(9, 0, 6)) + list(special_clips().values())
    for kw in (dict(), dict(num_filters=40, num_cepstral=40, high_frequency=0), dict(win_size=51), dict(fft_length=512, num_filters=24)):
        c = cfg.copy(pre_cof=0.0, **kw)
        for x in clips:
            mel, _ = reference.mfe(x, c)
            want = r432.cmvnw(mel, c.win_size, False, True).reshape(-1)
            assert _same_bits_or_both_nan(want, oracle.extract_mfe(x, cfg.copy(**kw))), kw


def test_random_graphs_through_reference_op_registrations(oracle, reference, tmp_path):
    """Fuzz: ~100 random members of the graph family (kws_testlib.random_graph_spec), int8 and float32 twins, every tensor of
    the reference's op registrations == the restatement.  The same draws are replayed on the GPU (test_random_graphs_on_gpu)."""
    from kws_testlib import OracleModel, random_graph_spec, synth_model_blob
    from dequantize_model import dequantize
    rng = np.random.default_rng(9)
    n_ok = 0
    for seed in range(150):
        kw = random_graph_spec(seed)
        if kw is None:
            continue
        blob = synth_model_blob(**kw)
        for kind, b in (("i8", blob), ("f32", dequantize(blob))):
            p = tmp_path / ("fz%d%s.kwsm" % (seed, kind))
            p.write_bytes(b)
            om = OracleModel(oracle, str(p))
            for it in range(3):
                if kind == "i8":
                    x = rng.integers(-128, 128, om.n_features).astype(np.int8)
                    out, taps = reference.graph_run(b, x)
                    oo, ot = om.nn_invoke(x, taps=True)
                    assert (out == oo).all(), (seed, kw)
                    assert all((a == t).all() for a, t in zip(taps, ot) if a.dtype == np.int8), (seed, kw)
                else:
                    x = (rng.standard_normal(om.n_features) * np.float32(4.0)).astype(np.float32)
                    out, taps = reference.graph_run(b, x)
                    oo, ot = om.nn_invoke_f32(x, taps=True)
                    assert all((bits(a) == bits(t)).all() for a, t in zip(taps[:-1], ot[:-1]) if a.dtype == np.float32), (seed, kw)
                    assert np.abs(out - oo).max() <= 1e-7, (seed, kw)
        n_ok += 1
    assert n_ok >= 75, n_ok


def test_random_mfcc_configurations(oracle, reference):
    """Fuzz: 40 random DSP configurations within the GPU kernel's build (kws_testlib.random_dsp_spec): the reference's
    extract_mfcc_features == the restatement, bit for bit; replayed on the GPU by test_random_mfcc_configurations_on_gpu."""
    from kws_testlib import random_dsp_spec
    clips = list(oracle.synth(21, 0, 2)) + [special_clips()["impulses"], special_clips()["ramp"]]
    for seed in range(40):
        cfg_kw, _ = random_dsp_spec(seed)
        cfg = L476_CONFIG().copy(**cfg_kw)
        for c in clips:
            assert (bits(oracle.extract_mfcc(c, cfg)) == bits(reference.extract_mfcc(c, cfg))).all(), (seed, cfg_kw)


def test_quantized_filterbank_option_pinned(oracle):
    """EIDSP_QUANTIZE_FILTERBANK = 1, the SDK's default (SDK/dsp/config.hpp:75-77): a second build of
    the unmodified reference with the option at its default (oracle/_ref/libei_ref_l476_qfb.so) pins (a) the table the restatement
    generates from its rule -- every fraction a / b with b <= 22 and every i / 100 -- entry by entry against numpy.hpp:52's, (b)
    quantize_zero_one incl. its out-of-range quirks on a dense sweep, (c) the filterbank matrices, (d) extract_mfcc_features, bit for bit."""
    from kws_testlib import REF_QFB_SO
    if not os.path.exists(REF_QFB_SO):
        pytest.skip("oracle/_ref/libei_ref_l476_qfb.so not built (make -C oracle ref)")
    ref = Reference(qfb=True)
    tab = np.zeros(256, np.float32)
    n = ref.L.eiref_quantized_table(tab.ctypes.data_as(__import__("ctypes").c_void_p), 256)
    assert n == 231
    for v in tab[:n]:
        assert bits(np.float32(oracle.quantize_zero_one(v))) == bits(np.float32(v))       # every table value is a fixed point ...
    rng = np.random.default_rng(5)
    sweep = np.concatenate([rng.random(20000).astype(np.float32), (tab[:n - 1] + tab[1:n]) / np.float32(2), np.float32([-0.5, -1e-9, 1.0000001, 1.5, 3.0])])
    for v in sweep:
        assert bits(np.float32(oracle.quantize_zero_one(v))) == bits(np.float32(ref.L.eiref_quantize_zero_one(float(v)))), v
    clips = oracle.synth(9, 0, 6)
    changed = 0
    for kw in (dict(), dict(num_filters=40, num_cepstral=40, high_frequency=0), dict(num_filters=40, num_cepstral=13, low_frequency=0, high_frequency=0),
               dict(fft_length=512, num_filters=32, high_frequency=0), dict(fft_length=1024, num_filters=20, low_frequency=0, high_frequency=0),
               dict(low_frequency=50, high_frequency=6000, num_cepstral=20)):
        cfg = L476_CONFIG().copy(quantize_filterbank=1, **kw)
        fo, fr = oracle.filterbanks(cfg), ref.filterbanks(cfg)
        assert (bits(fo) == bits(fr)).all(), kw
        # a triangle weight is (bin - left) / (middle - left): a fraction whose denominator is the filter's half width in bins, so the table
        # (denominators up to 22) reproduces narrow filters exactly -- the shipped fft-256 configurations -- and only moves wide ones
        changed += int((fo != oracle.filterbanks(cfg.copy(quantize_filterbank=0))).any())
        for c in clips:
            assert (bits(oracle.extract_mfcc(c, cfg)) == bits(ref.extract_mfcc(c, cfg))).all(), kw
    assert changed >= 1

def test_random_general_shape_configurations(oracle, reference):
    """The random GENERAL-SHAPE configurations of tests/generic_soak.py (other fft lengths incl. non powers of two, 8 .. 64 filters, other frame
    lengths / strides, 0.25 .. 2 s clips -- what csrc/kws_generic.hip serves): the reference's extract_mfcc_features == the restatement, bit for
    bit.  The soak replays the same seeds on the GPU against the restatement (profiles/r04_generic_soak.txt)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("generic_soak", os.path.join(ROOT, "tests", "generic_soak.py"))
    gs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gs)
    sp = special_clips()
    for seed in range(40, 440):
        cfg_kw, _, n = gs.random_general_spec(seed)
        cfg = L476_CONFIG().copy(**cfg_kw)
        rnd = oracle.synth(1000 + seed, 0, 2).reshape(32000)[:n]
        for c in (rnd, np.concatenate([sp["alternating_fullscale"]] * 2)[:n]):
            a, b = oracle.extract_mfcc(c, cfg), reference.extract_mfcc(c, cfg)
            assert a.shape == b.shape and (bits(a) == bits(b)).all(), (seed, cfg_kw)
