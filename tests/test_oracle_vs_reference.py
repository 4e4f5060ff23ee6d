"""Pins the C restatement against the UNMODIFIED reference compiled in place by `make -C oracle ref`
(only where /root/reference exists; skipped on the GPU box).  Randomised, broader than the goldens."""
import numpy as np
import pytest

from kws_testlib import L476_CONFIG, bits, special_clips

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
