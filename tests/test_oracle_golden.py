"""The C restatement (oracle/kws_oracle.c) against golden vectors generated from the unmodified
reference (tools/make_golden.py).  Runs everywhere (CPU only, no /root/reference needed).
Bar: bit-exact, float32 compared as uint32."""
import os

import numpy as np
import pytest

from kws_testlib import GOLDEN, L476_CONFIG, ROOT, bits, special_clips


def _load(name):
    return np.load(os.path.join(GOLDEN, name))


def test_leaves(oracle):
    g = _load("leaves_l476.npz")
    cfg = L476_CONFIG()
    y = np.float32([oracle.L.kwso_log(float(x)) for x in g["log_x"]])
    assert (bits(y) == bits(g["log_y"])).all()
    # known answers quoted in SURVEY 8(a) rows 8 and 10
    assert oracle.L.kwso_log(1.0) == 0.0
    assert abs(oracle.L.kwso_log(1.1920929e-07) - (-15.9423847)) < 1e-6
    y = np.float32([oracle.L.kwso_frequency_to_mel(float(f)) for f in g["mel_f"]])
    assert (bits(y) == bits(g["mel_y"])).all()
    fb = oracle.filterbanks(cfg)
    assert (bits(fb) == bits(g["filterbank_l476"])).all()
    assert int((fb != 0).sum()) == 39              # 39 non-zeros, bins 2..63 only
    nz = np.nonzero(fb)[0]
    assert nz.min() == 2 and nz.max() <= 63
    assert (bits(oracle.filterbanks(cfg.copy(high_frequency=0))) == bits(g["filterbank_l432"])).all()
    assert (bits(oracle.filterbanks(cfg.copy(num_filters=40, high_frequency=0))) == bits(g["filterbank_40"])).all()
    assert (bits(oracle.dct2_ortho(np.arange(32))) == bits(g["dct_ramp32"])).all()
    assert (bits(oracle.dct2_ortho(np.arange(40))) == bits(g["dct_ramp40"])).all()   # radix-5 path
    # half-output DCT quirk: outputs 17..31 are input*2*sqrt(1/64)
    assert np.allclose(g["dct_ramp32"][17:], np.arange(17, 32) * 0.25)
    assert (bits(oracle.rfft_complex(g["rfft256_x"])) == bits(g["rfft256_y"])).all()
    assert (bits(oracle.rfft_complex(g["rfft32_x"])) == bits(g["rfft32_y"])).all()
    assert (bits(oracle.cmvnw(g["cmvn_x"], 101, True)) == bits(g["cmvn_y"])).all()
    assert (bits(oracle.cmvnw(g["cmvn_x"], 101, False)) == bits(g["cmvn_y_novar"])).all()


def test_end_to_end(oracle, l476):
    g = _load("e2e_l476.npz")
    n = int(g["clips_per_seed"])
    k = 0
    for seed in g["seeds"]:
        clips = oracle.synth(int(seed), 0, n)
        s, f, _ = l476.run_batch(clips, want_features=True)
        assert (bits(f) == bits(g["features"][k:k + n])).all()
        assert (bits(s) == bits(g["scores"][k:k + n])).all()
        k += n
    sp = special_clips()
    for i, name in enumerate(g["special_names"]):
        s, f, _ = l476.run_batch(sp[str(name)], want_features=True)
        assert (bits(f[0]) == bits(g["special_features"][i])).all(), name
        assert (bits(s[0]) == bits(g["special_scores"][i])).all(), name


def test_silence_canary(oracle, l476):
    """all-zero clip -> c0 column is -0.888889 in every frame (a pure summation-order
    artefact), other columns 0, scores {0.25, 0.22266, 0.25, 0.27734}."""
    s, f, _ = l476.run_batch(np.zeros(16000, np.int16), want_features=True)
    f = f.reshape(49, 13)
    assert np.allclose(f[:, 0], -0.888889, atol=1e-6)
    assert (f[:, 1:] == 0).all()
    assert np.allclose(s[0], [0.25, 0.22265625, 0.25, 0.27734375])


def test_every_stage(oracle, l476):
    g = _load("deep_l476.npz")
    cfg = L476_CONFIG()
    sp = special_clips()
    for k in range(int(g["n"])):
        seed, idx = g["ids"][k]
        clip = oracle.synth(int(seed), int(idx), 1)[0] if seed >= 0 else sp[["step", "impulses"][idx]]
        p = f"c{k}_"
        assert (bits(oracle.preemphasis(clip, cfg.pre_cof, 1, 0, 320)) == bits(g[p + "pre_f0"])).all()
        assert (bits(oracle.preemphasis(clip, cfg.pre_cof, 1, 7 * 320, 320)) == bits(g[p + "pre_f7"])).all()
        assert (bits(oracle.power_spectrum(g[p + "pre_f7"], 256)) == bits(g[p + "ps_f7"])).all()
        mel, en = oracle.mfe(clip, cfg)
        assert (bits(mel) == bits(g[p + "mel"])).all()
        assert (bits(en) == bits(g[p + "energy"])).all()
        assert (bits(oracle.mfcc_nocmvn(clip, cfg)) == bits(g[p + "mfcc"])).all()
        f = oracle.extract_mfcc(clip, cfg)
        assert (bits(f) == bits(g[p + "features"])).all()
        q = l476.quantize_input(f)
        assert (q == g[p + "q_in"]).all()
        out, taps = l476.nn_invoke(q, taps=True)
        n_checked = 0
        for tid in range(len(taps)):
            key = p + f"t{tid}"
            if key in g.files:
                assert (taps[tid] == g[key]).all(), (k, tid)
                n_checked += 1
        assert n_checked == 15                      # one output per graph node
        assert (bits(l476.dequantize(out)) == bits(g[p + "scores"])).all()


def test_framing(oracle):
    cfg = L476_CONFIG()
    assert oracle.num_frames(16000, cfg) == 49      # floor((16000-320)/320)
    import ctypes
    assert oracle.L.kwso_frame_length_samples(ctypes.byref(cfg)) == 320
    assert oracle.num_frames(4000, cfg) == 11
    assert oracle.num_frames(640, cfg) == 1


def test_fixed_point_known_answers(oracle):
    L = oracle.L
    assert L.kwso_srdhm(-2**31, -2**31) == 2**31 - 1
    assert L.kwso_srdhm(1 << 30, 1 << 30) == 1 << 29
    assert L.kwso_srdhm(-3, 1 << 30) == -1          # truncating division: (-3*2^30 + 1-2^30)/2^31 -> -1
    assert L.kwso_rdivpot(5, 1) == 3 and L.kwso_rdivpot(-5, 1) == -3 and L.kwso_rdivpot(-6, 2) == -2
    assert oracle.quantize_multiplier(0.5) == (1 << 30, 0)
    assert oracle.quantize_multiplier(1.0) == (1 << 30, 1)
    assert oracle.quantize_multiplier(0.0) == (0, 0)
    assert L.kwso_exp_on_negative_values_q5_26(0) == 2**31 - 1
    # exp(-1) in Q0.31 within gemmlowp's accuracy
    assert abs(L.kwso_exp_on_negative_values_q5_26(-(1 << 26)) / 2**31 - np.exp(-1)) < 1e-6
    assert abs(L.kwso_one_over_one_plus_x(0) / 2**31 - 1.0) < 1e-6
    assert abs(L.kwso_one_over_one_plus_x(1 << 30) / 2**31 - 1 / 1.5) < 1e-6


def test_model_blob(l476, l432):
    assert l476.labels == ["no", "noise", "unknown", "yes"]
    assert l432.labels == ["_noise", "_unknown", "trick_or_treat"]
    assert l476.n_features == 637 and l432.n_features == 637
    assert l476.cfg.high_frequency == 4000 and l432.cfg.high_frequency == 0
    assert len(l476.tensor_bytes) == 31


def test_bad_blob_rejected(oracle):
    assert not oracle.L.kwso_model_load(b"XXXX" + b"\0" * 64, 68)
    blob = open(os.path.join(os.path.dirname(GOLDEN), "..", "models", "l476_no_yes.kwsm"), "rb").read()
    assert not oracle.L.kwso_model_load(blob[:200], 200)   # truncated


def test_continuous_mode_golden(oracle, l476):
    """run_classifier_continuous + run_classifier_init restated: 20 slices, re-init, 6 more (golden from the reference)."""
    from kws_testlib import OracleContinuous
    g = _load("continuous_l476.npz")
    audio = oracle.synth(int(g["audio_seed"]), 0, int(g["n_clips"])).reshape(-1)
    oc = OracleContinuous(l476)
    oc.init()
    k = 0
    for n_slices in (20, 6):
        for i in range(n_slices):
            rc, produced, s = oc.step(audio[i * 4000:(i + 1) * 4000])
            assert rc == 0 and produced == bool(g["produced"][k])
            assert (bits(s) == bits(g["scores"][k])).all(), k
            k += 1
        oc.init()
    assert list(g["produced"][:5]) == [False, False, False, True, True]      # buffer full after 4 slices
    assert g["total_length_after"][0] == 4000 and (g["total_length_after"][1:] == 4320).all()   # first_run quirk


def test_fp32_twin_golden(oracle):
    """fp32 twin of the shipped model (BASELINE config "fp32"): golden = the reference's float kernels chained leaf by
    leaf on the reference's own MFCC features (tools/make_golden.py)."""
    from kws_testlib import MODELS, OracleModel
    g = _load("f32_twin_l476.npz")
    mf = OracleModel(oracle, os.path.join(MODELS, "l476_no_yes_f32.kwsm"))
    clips = oracle.synth(int(g["seed"]), 0, int(g["n"]))
    s = mf.run_batch(clips)
    assert np.abs(s - g["scores"]).max() <= 1e-7        # expf of the host libm is the only non-replayed operation
    out, taps = mf.nn_invoke_f32(oracle.extract_mfcc(clips[3], mf.cfg), taps=True)
    assert (bits(taps[29]) == bits(g["logits"][3])).all()


def test_mfcc40_golden(oracle):
    """BASELINE's 40-band MFCC variants: reference extract_mfcc_features outputs (tools/make_golden.py mfcc40)."""
    from kws_testlib import L476_CONFIG, special_clips
    g = _load("mfcc40_l476.npz")
    sp = special_clips()
    clips = np.concatenate([oracle.synth(int(g["seed"]), 0, int(g["n"])), np.stack([sp[str(k)] for k in g["special_names"]])])
    cfg = L476_CONFIG()
    for name, kw in (("f40c40", dict(num_filters=40, num_cepstral=40, high_frequency=0)),
                     ("f40c13", dict(num_filters=40, num_cepstral=13, low_frequency=0, high_frequency=0)),
                     ("f40c30w51", dict(num_filters=40, num_cepstral=30, win_size=51))):
        c = cfg.copy(**kw)
        for i, x in enumerate(clips):
            assert (bits(oracle.extract_mfcc(x, c)) == bits(g[name][i])).all(), (name, i)


MFE_BLOCK_CASES = (("f32", dict()), ("f40", dict(num_filters=40, num_cepstral=40, high_frequency=0)), ("f32w51", dict(win_size=51)))


def mfe_block_clips(oracle, g):
    from kws_testlib import special_clips
    sp = special_clips()
    return np.concatenate([oracle.synth(int(g["seed"]), 0, int(g["n"])), np.stack([sp[str(k)] for k in g["special_names"]])])


def same_bits_or_both_nan(a, b):
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    return ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all()


def test_mfe_block_golden(oracle):
    """extract_mfe_features of the L432 SDK copy (tools/make_golden.py mfe_block: the reference's feature::mfe, then the L432
    headers' cmvnw(win, false, true) + numpy::normalize); constant clips normalise to 0 * inf = NaN there too."""
    from kws_testlib import L476_CONFIG
    g = _load("mfe_block_l432.npz")
    clips = mfe_block_clips(oracle, g)
    cfg = L476_CONFIG()
    for name, kw in MFE_BLOCK_CASES:
        for i, x in enumerate(clips):
            assert same_bits_or_both_nan(oracle.extract_mfe(x, cfg.copy(**kw)), g[name][i]), (name, i)


def test_synthetic_graph_goldens(oracle, tmp_path):
    """Outputs of the reference's own op registrations for the synthetic graphs (tools/make_golden.py graphs)."""
    from kws_testlib import SYNTH_SPECS, OracleModel, synth_model_blob
    from dequantize_model import dequantize
    g = _load("graphs_l476.npz")
    for name in [str(n) for n in g["names"]]:
        blob = synth_model_blob(**SYNTH_SPECS[name])
        for kind, b in (("i8", blob), ("f32", dequantize(blob))):
            p = tmp_path / (name + kind + ".kwsm")
            p.write_bytes(b)
            om = OracleModel(oracle, str(p))
            rng = np.random.default_rng(int(g["rng_seed"]))
            xi = rng.integers(-128, 128, (int(g["n"]), om.n_features)).astype(np.int8)
            xf = (rng.standard_normal((int(g["n"]), om.n_features)) * 3).astype(np.float32)
            n_t = len(om.tensor_bytes)
            for k in range(int(g["n"])):
                if kind == "i8":
                    o, taps = om.nn_invoke(xi[k], taps=True)
                    assert (o == g[name + "_i8_out"][k]).all() and (taps[n_t - 2] == g[name + "_i8_fc"][k]).all(), (name, k)
                else:
                    o, taps = om.nn_invoke_f32(xf[k], taps=True)
                    assert (bits(taps[n_t - 2]) == bits(g[name + "_f32_logits"][k])).all(), (name, k)
                    assert np.abs(o - g[name + "_f32_scores"][k]).max() <= 1e-7


def test_mfe_model_golden(oracle, tmp_path):
    """A model whose DSP block is MFE (blob version 2), one-shot and in continuous mode, against vectors composed from the
    reference's own leaves (tools/make_golden.py mfe_model: feature::mfe of the L476 build, cmvnw + normalize of the L432
    headers, the graph through the reference's op registrations)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from make_golden import MFE_MODEL_KW
    from kws_testlib import OracleContinuous, OracleModel, synth_model_blob
    g = np.load(os.path.join(GOLDEN, "mfe_model_l432.npz"))
    path = str(tmp_path / "mfe.kwsm")
    open(path, "wb").write(synth_model_blob(**MFE_MODEL_KW))
    om = OracleModel(oracle, path)
    assert oracle.L.kwso_model_dsp_block(om.h) == 1 and om.n_features == 49 * 32
    sp = special_clips()
    clips = np.concatenate([oracle.synth(int(g["seed"]), int(g["first"]), int(g["n"])), np.stack([sp[str(k)] for k in g["special_names"]])])
    s, f, q = om.run_batch(clips, want_features=True)
    assert (bits(f) == bits(g["features"])).all() and (q == g["q"]).all() and (bits(s) == bits(g["scores"])).all()
    audio = oracle.synth(int(g["cont_audio_seed"]), 0, 3).reshape(-1)
    oc = OracleContinuous(om)
    oc.init()
    for k in range(len(g["cont_produced"])):
        rc, produced, sc = oc.step(audio[k * 4000:(k + 1) * 4000])
        assert rc == 0 and produced == bool(g["cont_produced"][k]), k
        if produced:
            assert (bits(sc) == bits(g["cont_scores"][k])).all(), k


def test_eon_import_reads_an_mfe_block_configuration():
    """tools/eon_import.py: a model_metadata.h that instantiates ei_dsp_config_mfe_t (the MFE block of the newer SDK copy, field
    order of L432 model-parameters/model_metadata.h:103-112) yields a version-2 blob whose DSP block is MFE."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import eon_import
    meta_h = '''
#define EI_CLASSIFIER_RAW_SAMPLE_COUNT           16000
#define EI_CLASSIFIER_FREQUENCY                  16000
#define EI_CLASSIFIER_NN_INPUT_FRAME_SIZE        1960
const char* ei_classifier_inferencing_categories[] = { "a", "b" };
ei_dsp_config_mfe_t ei_dsp_config_3 = {
    1,
    0.02f,
    0.02f,
    40,
    256,
    300,
    0,
    101
};
'''
    m = eon_import.parse_metadata(meta_h)
    d = m["dsp"]
    assert d["block"] == 1 and d["num_filters"] == 40 and d["num_cepstral"] == 40 and d["fft_length"] == 256
    assert d["win_size"] == 101 and d["low_frequency"] == 300 and d["high_frequency"] == 0 and d["pre_cof"] == 0.0
    assert abs(d["frame_length"] - 0.02) < 1e-9 and m["labels"] == ["a", "b"]
    # the shipped exports stay MFCC blocks, byte for byte
    for name, sub in (("l476_no_yes.kwsm", "nucleo-l476-keyword-spotting/ei-keyword-spotting"),):
        ref_dir = "/root/reference/embedded-demos/stm32cubeide/" + sub
        if os.path.isdir(ref_dir):
            blob, _ = eon_import.import_export(ref_dir)
            assert blob == open(os.path.join(ROOT, "models", name), "rb").read()


@pytest.mark.parametrize("name", ["cfg2_mfcc40_f32.kwsm", "cfg2_mfcc40_int8.kwsm", "cfg5_dscnn_mfcc40_f32.kwsm", "cfg5_dscnn_mfcc40_int8.kwsm",
                                  "l476_no_yes_f32.kwsm"])
def test_synthetic_models_are_not_saturated(name):
    """with a saturated softmax (max score > 0.999 on 90 % of the clips, p (1 - p) ~ 1e-9) a bar on the SCORES says
    nothing about the logits.  The synthetic heads are calibrated (tools/synth_model.py calibrate_head; tools/make_models.sh): on the bench's
    own clips the winning score's median must sit between 0.4 and 0.95 and the typical p (1 - p) of a clip near its maximum, 1/4."""
    from kws_testlib import MODELS, Oracle, OracleModel
    o = Oracle()
    m = OracleModel(o, os.path.join(MODELS, name))
    s = m.run_batch(o.synth(0, 0, 256))
    mx = s.max(axis=1)
    assert 0.4 <= np.median(mx) <= 0.95, np.median(mx)
    assert (mx > 0.999).mean() < 0.05
    assert np.median((s * (1 - s)).max(axis=1)) > 0.1
    assert (np.bincount(s.argmax(axis=1), minlength=m.n_labels) > 0).sum() >= 3      # not one class for every clip


QFB_CASES = {"l476": dict(), "fft512": dict(fft_length=512, high_frequency=0),
             "fft1024_f20": dict(fft_length=1024, num_filters=20, num_cepstral=12, low_frequency=0, high_frequency=0)}


def test_quantized_filterbank_golden(oracle):
    """EIDSP_QUANTIZE_FILTERBANK = 1 (the SDK's default; the demos build with 0): features and filterbank matrices from the reference built
    with the option at its default (tools/make_golden.py --only-qfb), bit for bit."""
    g = np.load(os.path.join(GOLDEN, "qfb_l476.npz"))
    clips = oracle.synth(int(g["seed"]), 0, int(g["n"]))
    for name, kw in QFB_CASES.items():
        cfg = L476_CONFIG().copy(quantize_filterbank=1, **kw)
        assert (bits(oracle.filterbanks(cfg)) == bits(g[name + "_fb"])).all(), name
        for i, c in enumerate(clips):
            assert (bits(oracle.extract_mfcc(c, cfg)) == bits(g[name][i])).all(), (name, i)
