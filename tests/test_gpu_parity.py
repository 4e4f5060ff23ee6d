"""-m gpu: the HIP path (through the C ABI of libkws_mi355x.so) against the C oracle and the committed golden
vectors.  Bar: BIT-EXACT -- MFCC features compared as uint32, int8 tensors and scores compared exactly."""
import ctypes
import os

import numpy as np
import pytest

from kws_testlib import GOLDEN, MODELS, ROOT, bits, special_clips

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    import sys
    sys.path.insert(0, ROOT)
    import torch  # noqa: F401  (first: libamdhip64 of the torch wheel is the one the process uses)
    from __graft_entry__ import load_package
    return load_package()


@pytest.fixture(scope="module")
def gpu476(pkg):
    return pkg.Model(os.path.join(MODELS, "l476_no_yes.kwsm"), device=0)


@pytest.fixture(scope="module")
def gpu432(pkg):
    return pkg.Model(os.path.join(MODELS, "l432_trick_or_treat.kwsm"), device=0)


def test_native_library_is_what_runs(pkg, gpu476):
    maps = open("/proc/self/maps").read()
    assert "libkws_mi355x.so" in maps
    assert gpu476.labels == ["no", "noise", "unknown", "yes"] and gpu476.n_features == 637 and gpu476.n_frames == 49


def test_synth_generator_matches_host(pkg, oracle):
    import torch
    B = 37
    t = torch.empty((B, 16000), dtype=torch.int16, device="cuda:0")
    pkg.synth_clips_device(5, 1000, B, 16000, t.data_ptr())
    torch.cuda.synchronize()
    assert (t.cpu().numpy() == oracle.synth(5, 1000, B)).all()


def test_golden_end_to_end(gpu476, oracle):
    g = np.load(os.path.join(GOLDEN, "e2e_l476.npz"))
    n = int(g["clips_per_seed"])
    k = 0
    for seed in g["seeds"]:
        clips = oracle.synth(int(seed), 0, n)
        s, f, q = gpu476.run_classifier_batch(clips, want_features=True)
        assert (bits(f) == bits(g["features"][k:k + n])).all()
        assert (bits(s) == bits(g["scores"][k:k + n])).all()
        k += n
    sp = special_clips()
    names = [str(x) for x in g["special_names"]]
    s, f, q = gpu476.run_classifier_batch(np.stack([sp[nm] for nm in names]), want_features=True)
    assert (bits(f) == bits(g["special_features"])).all()
    assert (bits(s) == bits(g["special_scores"])).all()


def test_golden_deep_taps(gpu476):
    g = np.load(os.path.join(GOLDEN, "deep_l476.npz"))
    for k in range(int(g["n"])):
        p = f"c{k}_"
        s, pooled, fc, out = gpu476.nn_batch(g[p + "q_in"])
        assert (pooled[0][:210] == g[p + "t21"]).all()       # MAX_POOL_2D #1  [1,7,1,30]
        assert (pooled[0][210:220] == g[p + "t27"]).all()    # MAX_POOL_2D #2  [1,1,1,10]
        assert (fc[0] == g[p + "t29"]).all()                 # FULLY_CONNECTED
        assert (out[0] == g[p + "t30"]).all()                # SOFTMAX
        assert (bits(s[0]) == bits(g[p + "scores"])).all()


@pytest.mark.parametrize("which", ["l476", "l432"])
def test_oracle_parity_random_clips(which, gpu476, gpu432, l476, l432, oracle):
    gm, om = (gpu476, l476) if which == "l476" else (gpu432, l432)
    for seed, first, n in ((31, 0, 256), (32, 5000, 193), (33, 70000, 1)):
        clips = oracle.synth(seed, first, n)
        s, f, q = gm.run_classifier_batch(clips, want_features=True)
        so, fo, qo = om.run_batch(clips, want_features=True)
        assert (bits(f) == bits(fo)).all(), "%d feature words differ" % int((bits(f) != bits(fo)).sum())
        assert (q == qo).all()
        assert (bits(s) == bits(so)).all()


@pytest.mark.parametrize("which", ["l476", "l432"])
def test_nn_bit_exact_random_int8(which, gpu476, gpu432, l476, l432):
    gm, om = (gpu476, l476) if which == "l476" else (gpu432, l432)
    rng = np.random.default_rng(4)
    qs = np.concatenate([rng.integers(-128, 128, (400, 637)).astype(np.int8),
                         np.clip(rng.normal(-11, 25, (400, 637)), -128, 127).astype(np.int8),
                         np.repeat(np.arange(-128, 128, dtype=np.int8)[:, None], 637, 1)])
    s, pooled, fc, out = gm.nn_batch(qs)
    for i in range(qs.shape[0]):
        o, taps = om.nn_invoke(qs[i], taps=True)
        assert (out[i] == o).all(), i
        assert (fc[i] == taps[29]).all(), i
        assert (pooled[i][:210] == taps[21]).all() and (pooled[i][210:] == taps[27]).all(), i
        assert (bits(s[i]) == bits(om.dequantize(o))).all()


def test_edge_batches(gpu476, l476, oracle):
    # empty batch, batch of one, non-multiple-of-wave batch
    s = gpu476.run_classifier_batch(np.zeros((0, 16000), np.int16))
    assert s.shape == (0, 4)
    for n in (1, 3, 65):
        clips = oracle.synth(77, 0, n)
        assert (bits(gpu476.run_classifier_batch(clips)) == bits(l476.run_batch(clips))).all()


def test_full_size_properties(pkg, gpu476, l476, oracle):
    """BASELINE config-2 size (65 536 clips, 2 GiB of PCM resident in HBM): size-independent properties --
    (a) clip order does not matter (each clip is independent: permuting the batch permutes the scores),
    (b) duplicated clips give identical rows, (c) a strided sample of rows equals the oracle bit for bit,
    (d) scores are a valid quantised softmax (multiples of 1/256 that sum to ~1)."""
    import torch
    B = 65536
    pcm = torch.empty((B, 16000), dtype=torch.int16, device="cuda:0")
    pkg.synth_clips_device(0, 0, B, 16000, pcm.data_ptr())
    pcm[1::2048] = pcm[0::2048]                                   # (b) duplicates
    scores = torch.empty((B, 4), dtype=torch.float32, device="cuda:0")
    gpu476.run_classifier_batch_device(pcm.data_ptr(), B, scores.data_ptr())
    torch.cuda.synchronize()
    s = scores.cpu().numpy()
    assert (s[1::2048] == s[0::2048]).all()
    perm = torch.randperm(B, device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(1))
    pcm2 = pcm[perm].contiguous()
    scores2 = torch.empty_like(scores)
    gpu476.run_classifier_batch_device(pcm2.data_ptr(), B, scores2.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(scores2, scores[perm])                     # (a)
    idx = np.arange(7, B, 997)
    host = pcm[torch.from_numpy(idx).cuda()].cpu().numpy()
    assert (bits(s[idx]) == bits(l476.run_batch(host))).all()     # (c)
    assert (np.abs(s * 256 - np.round(s * 256)) == 0).all()       # (d)
    assert (np.abs(s.sum(1) - 1.0) <= 4 / 256).all()


def test_run_classifier_drop_in(pkg, gpu476, l476, oracle):
    """The SDK call shape of L476/Core/Src/main.cpp:190-199: signal_t + get_data callback -> run_classifier()."""
    gpu476.set_default()
    clip = oracle.synth(55, 9, 1)[0]
    calls = []

    def get_data(offset, length, out):                           # main.cpp:526-531 -> numpy::int16_to_float
        calls.append((offset, length))
        if offset + length > 16000:
            return -1
        seg = clip[offset:offset + length].astype(np.float32) / np.float32(32768)
        ctypes.memmove(out, seg.ctypes.data, 4 * length)
        return 0

    cb = pkg.GET_DATA_FN(get_data)
    sig = pkg.Signal(cb, 16000)
    Result = pkg.result_struct(4)
    res = Result()
    rc = pkg.lib().run_classifier(ctypes.byref(sig), ctypes.byref(res), False)
    assert rc == 0
    assert sig.total_length == 16000                              # caller's struct untouched
    # the callback is asked what the reference asks it, in its order: 98 calls (tests/golden/get_data_trace_l476.npz, recorded from the
    # compiled reference; tests/test_get_data_sequence.py holds the other scenarios on CPU)
    want = np.load(os.path.join(GOLDEN, "get_data_trace_l476.npz"))["oneshot"]
    assert len(calls) == 98 and calls == [(int(o), int(n)) for o, n, _ in want]
    got = np.float32([res.classification[i].value for i in range(4)])
    assert [res.classification[i].label.decode() for i in range(4)] == ["no", "noise", "unknown", "yes"]
    assert (bits(got) == bits(l476.run_batch(clip)[0])).all()
    # the library's error convention (the SDK's return-code semantics; the reference's default build asserts instead, dsp/config.hpp:65-67):
    # a failing get_data -> EI_IMPULSE_DSP_ERROR (-5); a window with more frames than the model's -> -5.  Windows with fewer frames are
    # classified as the reference classifies them: tests/test_other_window_length.py
    bad = pkg.GET_DATA_FN(lambda o, l, p: -7)
    assert pkg.lib().run_classifier(ctypes.byref(pkg.Signal(bad, 16000)), ctypes.byref(res), False) == -5
    assert pkg.lib().run_classifier(ctypes.byref(pkg.Signal(cb, 17000)), ctypes.byref(res), False) == -5
    # run_inference on a feature matrix
    s, f, q = gpu476.run_classifier_batch(clip, want_features=True)
    fm = np.ascontiguousarray(f[0])
    mat = pkg.Matrix(fm.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), 1, 637, False)
    assert pkg.lib().run_inference(ctypes.byref(mat), ctypes.byref(res), False) == 0
    assert (np.float32([res.classification[i].value for i in range(4)]) == s[0]).all()
    mat_bad = pkg.Matrix(fm.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), 1, 600, False)
    assert pkg.lib().run_inference(ctypes.byref(mat_bad), ctypes.byref(res), False) == -1


def test_bad_inputs(pkg):
    with pytest.raises(pkg.KwsError):
        pkg.Model("/nonexistent.kwsm")
    h = ctypes.c_void_p()
    assert pkg.lib().kws_create(b"XXXXXXXXXXXXXXXX", 16, 0, ctypes.byref(h)) == -20


def test_generic_nn_kernel_also_bit_exact(pkg, gpu476, l476):
    """The shipped graph runs on the matrix-core kernel; the generic (dot4) kernel that serves other graph shapes
    must give the same bits."""
    rng = np.random.default_rng(9)
    qs = rng.integers(-128, 128, (300, 637)).astype(np.int8)
    L = pkg.lib()
    L.kws_dev_force_scalar_nn(1)
    try:
        s, pooled, fc, out = gpu476.nn_batch(qs)
    finally:
        L.kws_dev_force_scalar_nn(0)
    s2, pooled2, fc2, out2 = gpu476.nn_batch(qs)
    assert (pooled == pooled2).all() and (fc == fc2).all() and (out == out2).all() and (s == s2).all()
    for i in range(0, 300, 7):
        o, taps = l476.nn_invoke(qs[i], taps=True)
        assert (out[i] == o).all() and (pooled[i][:210] == taps[21]).all() and (pooled[i][210:] == taps[27]).all()


@pytest.mark.parametrize("which", ["l476", "l432"])
def test_stage_api_split_at_cmvn(which, pkg, gpu476, gpu432, l476, l432, oracle):
    """kws_mfcc_batch_device = speechpy::feature::mfcc (cepstra before cmvnw); kws_cmvn_inference_batch_device =
    cmvnw + run_inference.  Both halves bit-exact against the oracle; together they equal the fused path."""
    import torch
    gm, om = (gpu476, l476) if which == "l476" else (gpu432, l432)
    B = 130
    clips = oracle.synth(91, 40, B)
    pcm = torch.from_numpy(clips).cuda()
    mfcc = torch.empty((B, 637), dtype=torch.float32, device="cuda")
    feats = torch.empty((B, 637), dtype=torch.float32, device="cuda")
    q = torch.empty((B, 637), dtype=torch.int8, device="cuda")
    scores = torch.empty((B, gm.n_labels), dtype=torch.float32, device="cuda")
    gm.mfcc_batch_device(pcm.data_ptr(), B, mfcc.data_ptr())
    gm.cmvn_inference_batch_device(mfcc.data_ptr(), B, scores.data_ptr(), feats.data_ptr(), q.data_ptr())
    torch.cuda.synchronize()
    so, fo, qo = om.run_batch(clips, want_features=True)
    m = mfcc.cpu().numpy()
    for i in range(0, B, 13):
        assert (bits(m[i].reshape(49, 13)) == bits(oracle.mfcc_nocmvn(clips[i], om.cfg))).all()
    assert (bits(feats.cpu().numpy()) == bits(fo)).all()
    assert (q.cpu().numpy() == qo).all()
    assert (bits(scores.cpu().numpy()) == bits(so)).all()
    # CMVN-only + generic NN kernel route
    L = pkg.lib()
    L.kws_dev_force_scalar_nn(1)
    try:
        scores2 = torch.empty_like(scores)
        gm.cmvn_inference_batch_device(mfcc.data_ptr(), B, scores2.data_ptr())
        torch.cuda.synchronize()
    finally:
        L.kws_dev_force_scalar_nn(0)
    assert torch.equal(scores2, scores)


def test_run_classifier_continuous(pkg, gpu476, l476, oracle):
    """The demos' real entry point (L476/Core/Src/main.cpp:190-199): 250 ms slices through run_classifier_continuous,
    against the restated reference (oracle) slice by slice, including the caller-visible total_length mutation."""
    from kws_testlib import OracleContinuous
    gpu476.set_default()
    L = pkg.lib()
    L.run_classifier_continuous.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_bool]
    audio = oracle.synth(4, 0, 5).reshape(-1)
    cur = {}

    def get_data(offset, length, out):
        if offset + length > 4000:                  # like the reference harness: refuse reads beyond the slice
            return -1
        seg = cur["s"][offset:offset + length].astype(np.float32) / np.float32(32768)
        ctypes.memmove(out, seg.ctypes.data, 4 * length)
        return 0

    cb = pkg.GET_DATA_FN(get_data)
    Result = pkg.result_struct(4)
    oc = OracleContinuous(l476)
    for n_slices in (20, 6):
        L.run_classifier_init()
        oc.init()
        for i in range(n_slices):
            cur["s"] = audio[i * 4000:(i + 1) * 4000]
            sig = pkg.Signal(cb, 4000)
            res = Result()
            rc = L.run_classifier_continuous(ctypes.byref(sig), ctypes.byref(res), False)
            orc, produced, s = oc.step(cur["s"])
            assert rc == 0 and orc == 0
            assert bool(res.classification[0].label) == produced
            got = np.float32([res.classification[j].value for j in range(4)])
            if produced:
                assert (bits(got) == bits(s)).all(), (n_slices, i)
            assert sig.total_length in (4000, 4320)


def test_many_streams_continuous_mode(pkg, gpu476, l476, oracle):
    """kws_streams_*: S streams in lock step, state in HBM; every stream must follow run_classifier_continuous()
    (restated reference) slice by slice, across a run_classifier_init()."""
    import torch
    from kws_testlib import OracleContinuous
    S, n_steps = 37, 11
    audio = oracle.synth(12, 0, S * 3).reshape(S, 3 * 16000)            # 3 s per stream
    sb = pkg.StreamBatch(gpu476, S)
    ocs = [OracleContinuous(l476) for _ in range(S)]
    for oc in ocs:
        oc.init()
    scores = torch.empty((S, 4), dtype=torch.float32, device="cuda")
    for phase in range(2):
        for k in range(n_steps if phase == 0 else 6):
            sl = np.ascontiguousarray(audio[:, k * 4000:(k + 1) * 4000])
            d = torch.from_numpy(sl).cuda()
            produced = sb.step_device(d.data_ptr(), 4000, scores.data_ptr())
            torch.cuda.synchronize()
            got = scores.cpu().numpy()
            for s in range(S):
                rc, p, want = ocs[s].step(sl[s])
                assert rc == 0 and p == produced, (phase, k, s)
                if p:
                    assert (bits(got[s]) == bits(want)).all(), (phase, k, s)
        sb.init()
        for oc in ocs:
            oc.init()
    sb.close()


from kws_testlib import SYNTH_SPECS  # noqa: E402
SYNTH_MODELS = [SYNTH_SPECS[k] for k in ("seed1", "seed2", "seed4", "seed5", "seed6", "seed7")] + [
    dict(seed=3, blocks=((24, 7, 7),), n_labels=6)]                      # one block: FULLY_CONNECTED over 7 x 24 = 168 inputs


@pytest.mark.parametrize("kw", SYNTH_MODELS, ids=lambda kw: "seed%d" % kw["seed"])
def test_synthetic_models_of_the_same_graph_family(kw, pkg, oracle, tmp_path):
    """Model ingestion + plan builder + both NN kernels on other shapes / quantisations / DSP settings of the Edge
    Impulse 1-D CNN family (random weights): features, int8 tensor, pooled taps and scores vs the oracle."""
    from kws_testlib import OracleModel, synth_model_blob
    blob = synth_model_blob(**kw)
    path = tmp_path / "m.kwsm"
    path.write_bytes(blob)
    om = OracleModel(oracle, str(path))
    gm = pkg.Model(blob=blob)
    assert gm.labels == om.labels and gm.n_features == om.n_features
    clips = oracle.synth(40 + kw["seed"], 0, 70)
    s, f, q = gm.run_classifier_batch(clips, want_features=True)
    so, fo, qo = om.run_batch(clips, want_features=True)
    assert (bits(f) == bits(fo)).all() and (q == qo).all() and (bits(s) == bits(so)).all()
    rng = np.random.default_rng(kw["seed"])
    qs = rng.integers(-128, 128, (200, om.n_features)).astype(np.int8)
    s2, pooled, fc, out = gm.nn_batch(qs)
    pool_ids = [i for i, nb in enumerate(om.tensor_bytes)]       # tensors are compared through the final taps below
    for i in range(0, 200, 9):
        o, taps = om.nn_invoke(qs[i], taps=True)
        assert (out[i] == o).all(), i
        assert (fc[i] == taps[len(taps) - 2]).all(), i
    gm.close()


def test_unsupported_models_fail_loudly(pkg, oracle):
    from kws_testlib import synth_model_blob
    for kw in (dict(seed=3, blocks=((64, 3, 1),), n_labels=6),          # FULLY_CONNECTED input of 49 x 64 = 3136 > kernel limit (1024)
               dict(seed=8, ncep=40),                                     # 40 cepstra > 32 mel filters
               ):
        with pytest.raises(pkg.KwsError) as e:
            pkg.Model(blob=synth_model_blob(**kw))
        assert e.value.code == -18, kw                                    # KWS_ERROR_UNSUPPORTED_MODEL


def test_soak_8192_clips_bit_exact(gpu476, l476, oracle):
    """A longer randomised comparison (5.2 M feature words, 51 M power-spectrum square roots behind them)."""
    n = 8192
    clips = oracle.synth(2024, 123456, n)
    s, f, q = gpu476.run_classifier_batch(clips, want_features=True)
    so, fo, qo = l476.run_batch(clips, want_features=True)
    assert int((bits(f) != bits(fo)).sum()) == 0
    assert (q == qo).all() and (bits(s) == bits(so)).all()


# ---- float32 models (BASELINE.json config "fp32": the de-quantised twin of the shipped model) -------------------------------
# Bar: MFCC features and every tensor up to the logits BIT-EXACT (the float kernels' sequential accumulation order is
# replayed); softmax scores within 1e-6 absolute (expf: device libm vs host libm, both ~1 ulp; tolerance per BASELINE's
# north_star "fp32 scores within 1e-4" with two orders of margin).
F32_SCORE_TOL = 1e-6


@pytest.fixture(scope="module")
def gpu476f(pkg):
    return pkg.Model(os.path.join(MODELS, "l476_no_yes_f32.kwsm"), device=0)


@pytest.fixture(scope="module")
def l476f(oracle):
    from kws_testlib import OracleModel
    return OracleModel(oracle, os.path.join(MODELS, "l476_no_yes_f32.kwsm"))


def _f32_logits(pkg, gm, feats):
    import torch
    f = torch.from_numpy(np.ascontiguousarray(feats, np.float32)).to("cuda:0")
    B = f.shape[0]
    s = torch.empty((B, gm.n_labels), dtype=torch.float32, device="cuda:0")
    lg = torch.empty((B, gm.n_labels), dtype=torch.float32, device="cuda:0")
    gm.nn_f32_batch_device(f.data_ptr(), B, s.data_ptr(), lg.data_ptr())
    torch.cuda.synchronize()
    return s.cpu().numpy(), lg.cpu().numpy()


def test_fp32_twin_golden_and_oracle(pkg, gpu476f, l476f, oracle):
    assert gpu476f.is_float and gpu476f.labels == ["no", "noise", "unknown", "yes"]
    g = np.load(os.path.join(GOLDEN, "f32_twin_l476.npz"))
    clips = oracle.synth(int(g["seed"]), 0, int(g["n"]))
    s, f, q = gpu476f.run_classifier_batch(clips, want_features=True)
    assert q is None
    assert np.abs(s - g["scores"]).max() <= F32_SCORE_TOL
    s2, lg = _f32_logits(pkg, gpu476f, f)
    assert (bits(lg) == bits(g["logits"])).all()                 # logits: the reference's float kernels, bit for bit
    assert (bits(s2) == bits(s)).all()
    # random clips against the oracle's float graph
    clips = oracle.synth(77, 500, 300)
    s, f, _ = gpu476f.run_classifier_batch(clips, want_features=True)
    so, fo, _ = l476f.run_batch(clips, want_features=True)
    assert (bits(f) == bits(fo)).all()
    assert np.abs(s - so).max() <= F32_SCORE_TOL
    _, lg = _f32_logits(pkg, gpu476f, f)
    n_t = len(l476f.tensor_bytes)
    for i in range(0, 300, 13):
        _, taps = l476f.nn_invoke_f32(fo[i], taps=True)
        assert (bits(lg[i]) == bits(taps[n_t - 2])).all(), i
    # adversarial feature vectors: huge / tiny / signed zeros / denormals exercise ReLU clamps and max-pool ordering
    rng = np.random.default_rng(5)
    feats = (rng.standard_normal((64, 637)) * np.float32(10.0) ** rng.integers(-3, 3, (64, 1))).astype(np.float32)
    feats[0] = 0.0; feats[1] = -0.0; feats[2] = 1e-41; feats[3, ::2] = 3e4
    sg, lg = _f32_logits(pkg, gpu476f, feats)
    for i in range(64):
        so_i, taps = l476f.nn_invoke_f32(feats[i], taps=True)
        assert (bits(lg[i]) == bits(taps[n_t - 2])).all(), i
        assert np.abs(sg[i] - so_i).max() <= F32_SCORE_TOL, i


def test_fp32_entry_points(pkg, gpu476f, gpu476, l476f, oracle):
    """Every entry point that serves both kinds of model, plus the loud failures of the int8-only ones."""
    import torch
    B = 130
    clips = oracle.synth(31, 0, B)
    so, fo, _ = l476f.run_batch(clips, want_features=True)
    pcm = torch.from_numpy(clips).to("cuda:0")
    s = torch.empty((B, 4), dtype=torch.float32, device="cuda:0")
    # run_classifier_batch_device without a feature buffer (scratch), then stage-split APIs
    gpu476f.run_classifier_batch_device(pcm.data_ptr(), B, s.data_ptr())
    torch.cuda.synchronize()
    s_ref = s.cpu().numpy().copy()
    assert np.abs(s_ref - so).max() <= F32_SCORE_TOL
    f = torch.empty((B, 637), dtype=torch.float32, device="cuda:0")
    gpu476f.extract_mfcc_batch_device(pcm.data_ptr(), B, f.data_ptr())
    gpu476f.run_inference_batch_device(f.data_ptr(), B, s.data_ptr())
    torch.cuda.synchronize()
    assert (bits(f.cpu().numpy()) == bits(fo)).all() and (bits(s.cpu().numpy()) == bits(s_ref)).all()
    m = torch.empty((B, 637), dtype=torch.float32, device="cuda:0")
    gpu476f.mfcc_batch_device(pcm.data_ptr(), B, m.data_ptr())
    s.zero_()
    gpu476f.cmvn_inference_batch_device(m.data_ptr(), B, s.data_ptr())
    torch.cuda.synchronize()
    assert (bits(s.cpu().numpy()) == bits(s_ref)).all()
    # int8-only entry points refuse a float model, and the float one refuses an int8 model
    q = torch.zeros((B, 637), dtype=torch.int8, device="cuda:0")
    for call in (lambda: gpu476f.nn_batch_device(q.data_ptr(), B, s.data_ptr()),
                 lambda: gpu476f.extract_mfcc_batch_device(pcm.data_ptr(), B, f.data_ptr(), q.data_ptr()),
                 lambda: gpu476f.run_classifier_batch_device(pcm.data_ptr(), B, s.data_ptr(), None, q.data_ptr()),
                 lambda: gpu476.nn_f32_batch_device(f.data_ptr(), B, s.data_ptr())):
        with pytest.raises(pkg.KwsError) as e:
            call()
        assert e.value.code == -18
    # SDK entry point with the float model as the default model
    gpu476f.set_default()
    try:
        buf = clips[5].astype(np.float32) / np.float32(32768)

        @pkg.GET_DATA_FN
        def get_data(offset, length, out):
            ctypes.memmove(out, buf[offset:offset + length].ctypes.data, 4 * length)
            return 0
        sig = pkg.Signal(get_data=get_data, total_length=16000)
        res = pkg.result_struct(4)()
        rc = pkg.lib().run_classifier(ctypes.byref(sig), ctypes.byref(res), False)
        assert rc == 0
        got = np.float32([res.classification[i].value for i in range(4)])
        assert (bits(got) == bits(s_ref[5])).all()
    finally:
        gpu476.set_default()


def test_fp32_twins_of_synthetic_models(pkg, oracle, tmp_path):
    """De-quantised twins of other members of the graph family (different widths / taps / pools / labels)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from dequantize_model import dequantize
    from kws_testlib import OracleModel, synth_model_blob
    for kw in SYNTH_MODELS:
        blob = dequantize(synth_model_blob(**kw))
        p = tmp_path / ("f32_%d.kwsm" % kw["seed"])
        p.write_bytes(blob)
        om = OracleModel(oracle, str(p))
        gm = pkg.Model(blob=blob)
        assert gm.is_float
        clips = oracle.synth(kw["seed"], 0, 40)
        so, fo, _ = om.run_batch(clips, want_features=True)
        s, f, _ = gm.run_classifier_batch(clips, want_features=True)
        assert (bits(f) == bits(fo)).all()
        assert np.abs(s - so).max() <= F32_SCORE_TOL
        _, lg = _f32_logits(pkg, gm, f)
        n_t = len(om.tensor_bytes)
        for i in range(0, 40, 7):
            _, taps = om.nn_invoke_f32(fo[i], taps=True)
            assert (bits(lg[i]) == bits(taps[n_t - 2])).all(), (kw, i)
        gm.close()


# ---- 40 mel filters (BASELINE configs 2 and 5: "40-band MFCC (49x40)") ------------------------------------------------------
MFCC40_MODELS = {          # golden key -> synthetic model around that MFCC block (weights random: only the DSP block is golden)
    "f40c40": dict(seed=11, num_filters=40, ncep=40, low=300, high=0, blocks=((16, 5, 7), (8, 3, 7)), n_labels=3),
    "f40c13": dict(seed=12, num_filters=40, ncep=13, low=0, high=0),
    "f40c30w51": dict(seed=13, num_filters=40, ncep=30, win_size=51, blocks=((24, 3, 7), (10, 7, 7)), n_labels=4),
}


@pytest.mark.parametrize("key", sorted(MFCC40_MODELS))
def test_mfcc40_golden_and_oracle(key, pkg, oracle, tmp_path):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from dequantize_model import dequantize
    from kws_testlib import OracleModel, synth_model_blob
    g = np.load(os.path.join(GOLDEN, "mfcc40_l476.npz"))
    sp = special_clips()
    gclips = np.concatenate([oracle.synth(int(g["seed"]), 0, int(g["n"])), np.stack([sp[str(k)] for k in g["special_names"]])])
    blob = synth_model_blob(**MFCC40_MODELS[key])
    for kind, b in (("int8", blob), ("f32", dequantize(blob))):
        p = tmp_path / ("m40_%s_%s.kwsm" % (key, kind))
        p.write_bytes(b)
        om = OracleModel(oracle, str(p))
        gm = pkg.Model(blob=b)
        s, f, q = gm.run_classifier_batch(gclips, want_features=True)
        assert (bits(f) == bits(g[key])).all(), (key, kind)               # features: the reference itself
        clips = oracle.synth(99, 40, 150)
        so, fo, qo = om.run_batch(clips, want_features=True)
        s, f, q = gm.run_classifier_batch(clips, want_features=True)
        assert (bits(f) == bits(fo)).all(), (key, kind)
        if kind == "int8":
            assert (q == qo).all() and (bits(s) == bits(so)).all(), key
        else:
            assert np.abs(s - so).max() <= F32_SCORE_TOL, key
            _, lg = _f32_logits(pkg, gm, f)
            n_t = len(om.tensor_bytes)
            for i in range(0, 150, 17):
                _, taps = om.nn_invoke_f32(fo[i], taps=True)
                assert (bits(lg[i]) == bits(taps[n_t - 2])).all(), (key, i)
        gm.close()


def test_mfcc40_stage_api_and_streams(pkg, oracle, tmp_path):
    """cepstra before CMVN (stage API) and cmvnw+network in kws_cmvn_nn_kernel for a 49x40 model."""
    import torch
    from kws_testlib import OracleModel, synth_model_blob
    blob = synth_model_blob(**MFCC40_MODELS["f40c40"])
    p = tmp_path / "m40.kwsm"
    p.write_bytes(blob)
    om = OracleModel(oracle, str(p))
    gm = pkg.Model(blob=blob)
    B = 70
    clips = oracle.synth(3, 7, B)
    so, fo, qo = om.run_batch(clips, want_features=True)
    pcm = torch.from_numpy(clips).to("cuda:0")
    m = torch.empty((B, gm.n_features), dtype=torch.float32, device="cuda:0")
    s = torch.empty((B, gm.n_labels), dtype=torch.float32, device="cuda:0")
    f = torch.empty((B, gm.n_features), dtype=torch.float32, device="cuda:0")
    gm.mfcc_batch_device(pcm.data_ptr(), B, m.data_ptr())
    gm.cmvn_inference_batch_device(m.data_ptr(), B, s.data_ptr(), f.data_ptr())
    torch.cuda.synchronize()
    mo = np.stack([oracle.mfcc_nocmvn(c, om.cfg) for c in clips[:8]])
    assert (bits(m.cpu().numpy()[:8]) == bits(mo.reshape(8, -1))).all()
    assert (bits(f.cpu().numpy()) == bits(fo)).all() and (bits(s.cpu().numpy()) == bits(so)).all()
    gm.close()


# ---- every synthetic graph incl. DEPTHWISE_CONV_2D (BASELINE config 5) ----------------------------------
@pytest.mark.parametrize("name", sorted(SYNTH_SPECS))
def test_graph_goldens_from_reference_op_registrations(name, pkg, oracle, tmp_path):
    """Network only: golden outputs computed by the reference's own TFLite-Micro op code (tools/make_golden.py graphs),
    then the whole pipeline against the oracle on audio."""
    import torch
    from dequantize_model import dequantize
    from kws_testlib import OracleModel, synth_model_blob
    g = np.load(os.path.join(GOLDEN, "graphs_l476.npz"))
    blob = synth_model_blob(**SYNTH_SPECS[name])
    n = int(g["n"])
    for kind, b in (("i8", blob), ("f32", dequantize(blob))):
        gm = pkg.Model(blob=b)
        rng = np.random.default_rng(int(g["rng_seed"]))
        xi = rng.integers(-128, 128, (n, gm.n_features)).astype(np.int8)
        xf = (rng.standard_normal((n, gm.n_features)) * 3).astype(np.float32)
        if kind == "i8":
            s, pooled, fc, out = gm.nn_batch(xi)
            assert (out == g[name + "_i8_out"]).all() and (fc == g[name + "_i8_fc"]).all(), name
        else:
            s, lg = _f32_logits(pkg, gm, xf)
            assert (bits(lg) == bits(g[name + "_f32_logits"])).all(), name
            assert np.abs(s - g[name + "_f32_scores"]).max() <= F32_SCORE_TOL, name
        p = tmp_path / (name + kind + ".kwsm")
        p.write_bytes(b)
        om = OracleModel(oracle, str(p))
        clips = oracle.synth(123, 0, 48)
        so, fo, qo = om.run_batch(clips, want_features=True)
        s, f, q = gm.run_classifier_batch(clips, want_features=True)
        assert (bits(f) == bits(fo)).all(), (name, kind)
        if kind == "i8":
            assert (q == qo).all() and (bits(s) == bits(so)).all(), name
        else:
            assert np.abs(s - so).max() <= F32_SCORE_TOL, name
        gm.close()


def test_mfe_front_end(pkg, gpu476, gpu432, l476, l432, oracle):
    """speechpy::feature::mfe: mel filterbank energies + frame energies, against the golden taps of
    the reference (deep_l476.npz: mel / energy of the deep clips) and the oracle on more clips / the other filterbank."""
    import torch
    g = np.load(os.path.join(GOLDEN, "deep_l476.npz"))
    sp = special_clips()
    ids = g["ids"]
    clips = np.stack([oracle.synth(int(s_), int(i), 1)[0] if s_ >= 0 else (sp["step"], sp["impulses"])[int(i)] for s_, i in ids])
    for gm, om, check_golden in ((gpu476, l476, True), (gpu432, l432, False)):
        B = len(clips)
        pcm = torch.from_numpy(clips).to("cuda:0")
        mel = torch.empty((B, gm.n_frames, gm.n_filters), dtype=torch.float32, device="cuda:0")
        en = torch.empty((B, gm.n_frames), dtype=torch.float32, device="cuda:0")
        gm.mfe_batch_device(pcm.data_ptr(), B, mel.data_ptr(), en.data_ptr())
        torch.cuda.synchronize()
        mel, en = mel.cpu().numpy(), en.cpu().numpy()
        for k in range(B):
            mo, eo = oracle.mfe(clips[k], om.cfg)
            assert (bits(mel[k]) == bits(mo)).all() and (bits(en[k]) == bits(eo)).all(), k
            if check_golden:
                assert (bits(mel[k]) == bits(g["c%d_mel" % k])).all() and (bits(en[k]) == bits(g["c%d_energy" % k])).all(), k


def test_full_size_properties_headline_workload(pkg, oracle):
    """bench.py's headline workload (BASELINE configs[1]: 65 536 clips, 49x40 MFCC + 2-Conv fp32) at full size:
    (a) permuting the batch permutes scores AND features bit for bit, (b) duplicated clips give identical rows,
    (c) a strided sample of rows equals the oracle -- features bit for bit, scores within 1e-6, (d) rows are softmaxes."""
    import torch
    from kws_testlib import OracleModel
    path = os.path.join(MODELS, "cfg2_mfcc40_f32.kwsm")
    gm = pkg.Model(path, device=0)
    om = OracleModel(oracle, path)
    B, F, C = 65536, gm.n_features, gm.n_labels
    pcm = torch.empty((B, 16000), dtype=torch.int16, device="cuda:0")
    pkg.synth_clips_device(0, 0, B, 16000, pcm.data_ptr())
    pcm[1::4096] = pcm[0::4096]
    scores = torch.empty((B, C), dtype=torch.float32, device="cuda:0")
    feats = torch.empty((B, F), dtype=torch.float32, device="cuda:0")
    gm.run_classifier_batch_device(pcm.data_ptr(), B, scores.data_ptr(), feats.data_ptr())
    torch.cuda.synchronize()
    s = scores.cpu().numpy()
    assert (s[1::4096] == s[0::4096]).all()                                           # (b)
    perm = torch.randperm(B, device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(2))
    pcm2 = pcm[perm].contiguous()
    scores2, feats2 = torch.empty_like(scores), torch.empty_like(feats)
    gm.run_classifier_batch_device(pcm2.data_ptr(), B, scores2.data_ptr(), feats2.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(scores2, scores[perm]) and torch.equal(feats2.view(torch.int32), feats[perm].view(torch.int32))   # (a)
    idx = np.arange(11, B, 1999)
    host = pcm[torch.from_numpy(idx).cuda()].cpu().numpy()
    so, fo, _ = om.run_batch(host, want_features=True)
    assert (bits(feats[torch.from_numpy(idx).cuda()].cpu().numpy()) == bits(fo)).all()   # (c)
    assert np.abs(s[idx] - so).max() <= F32_SCORE_TOL
    assert (s >= 0).all() and (np.abs(s.sum(1) - 1.0) <= 1e-5).all()                   # (d)
    gm.close()


def test_streams_with_float_and_40band_models(pkg, oracle):
    """Continuous mode for many streams with a float32 model and with the 49x40 headline graph: every stream follows the
    restated run_classifier_continuous() slice by slice (scores within 1e-6: float softmax)."""
    import torch
    from kws_testlib import OracleContinuous, OracleModel
    for name in ("l476_no_yes_f32.kwsm", "cfg2_mfcc40_f32.kwsm", "cfg2_mfcc40_int8.kwsm"):
        path = os.path.join(MODELS, name)
        gm = pkg.Model(path, device=0)
        om = OracleModel(oracle, path)
        S = 9
        audio = oracle.synth(14, 0, S * 2).reshape(S, 2 * 16000)
        sb = pkg.StreamBatch(gm, S)
        ocs = [OracleContinuous(om) for _ in range(S)]
        for oc in ocs:
            oc.init()
        scores = torch.empty((S, gm.n_labels), dtype=torch.float32, device="cuda")
        n_prod = 0
        for k in range(8):
            sl = np.ascontiguousarray(audio[:, k * 4000:(k + 1) * 4000])
            d = torch.from_numpy(sl).cuda()
            produced = sb.step_device(d.data_ptr(), 4000, scores.data_ptr())
            torch.cuda.synchronize()
            got = scores.cpu().numpy()
            for s in range(S):
                rc, p, want = ocs[s].step(sl[s])
                assert rc == 0 and p == produced, (name, k, s)
                if p:
                    n_prod += 1
                    if gm.is_float:
                        assert np.abs(got[s] - want).max() <= F32_SCORE_TOL, (name, k, s)
                    else:
                        assert (bits(got[s]) == bits(want)).all(), (name, k, s)
        assert n_prod > 0
        sb.close()
        gm.close()


def test_soak_headline_model_2048_clips(pkg, oracle):
    """A longer randomised comparison for the 49x40 fp32 graph bench.py times: 4.0 M feature words bit for bit, every
    score within 1e-6 of the restated reference float kernels, logits bit for bit."""
    from kws_testlib import OracleModel
    path = os.path.join(MODELS, "cfg2_mfcc40_f32.kwsm")
    gm = pkg.Model(path, device=0)
    om = OracleModel(oracle, path)
    n = 2048
    clips = oracle.synth(777, 4242, n)
    s, f, _ = gm.run_classifier_batch(clips, want_features=True)
    so, fo, _ = om.run_batch(clips, want_features=True)
    assert int((bits(f) != bits(fo)).sum()) == 0
    assert np.abs(s - so).max() <= F32_SCORE_TOL
    _, lg = _f32_logits(pkg, gm, f)
    n_t = len(om.tensor_bytes)
    for i in range(0, n, 97):
        _, taps = om.nn_invoke_f32(fo[i], taps=True)
        assert (bits(lg[i]) == bits(taps[n_t - 2])).all(), i
    gm.close()


def test_host_buffer_entry_point_is_chunked_and_identical(pkg, oracle):
    """kws_run_classifier_batch (host buffers) pipelines chunks of 8192 clips over two streams: a batch that spans three
    chunks (the last one ragged) returns exactly what the device entry point returns, for an int8 and a float32 model."""
    import torch
    n = 8192 * 2 + 3616
    clips = oracle.synth(31, 900000, n)
    for name in ("l476_no_yes.kwsm", "cfg2_mfcc40_f32.kwsm"):
        gm = pkg.Model(os.path.join(MODELS, name), device=0)
        s, f, q = gm.run_classifier_batch(clips, want_features=True)
        d = torch.from_numpy(clips).to("cuda:0")
        sd = torch.empty((n, gm.n_labels), dtype=torch.float32, device="cuda:0")
        fd = torch.empty((n, gm.n_features), dtype=torch.float32, device="cuda:0")
        qd = None if gm.is_float else torch.empty((n, gm.n_features), dtype=torch.int8, device="cuda:0")
        gm.run_classifier_batch_device(d.data_ptr(), n, sd.data_ptr(), fd.data_ptr(), None if qd is None else qd.data_ptr())
        torch.cuda.synchronize()
        assert (bits(s) == bits(sd.cpu().numpy())).all(), name
        assert (bits(f) == bits(fd.cpu().numpy())).all(), name
        if qd is not None:
            assert (q == qd.cpu().numpy()).all(), name
        s2 = gm.run_classifier_batch(clips[:5])                    # a later, smaller call reuses the buffers
        assert (bits(s2) == bits(s[:5])).all(), name
        gm.close()


def test_mfe_block_golden_and_oracle(pkg, oracle, tmp_path):
    """kws_extract_mfe_batch_device = extract_mfe_features of the L432 SDK copy (feature::mfe on the raw signal, cmvnw(win,
    false, true), numpy::normalize): the reference's own outputs (tests/golden/mfe_block_l432.npz) and the restatement, bit
    for bit -- constant clips normalise to 0 * inf = NaN on both sides (NaN payload/sign not compared)."""
    import torch
    from kws_testlib import synth_model_blob, L476_CONFIG
    from test_oracle_golden import MFE_BLOCK_CASES, mfe_block_clips, same_bits_or_both_nan
    g = np.load(os.path.join(GOLDEN, "mfe_block_l432.npz"))
    gclips = mfe_block_clips(oracle, g)
    cfg0 = L476_CONFIG()
    models = {"f32": dict(seed=1), "f40": dict(seed=11, num_filters=40, ncep=40, low=300, high=0, blocks=((16, 5, 7), (8, 3, 7)), n_labels=3),
              "f32w51": dict(seed=2, ncep=10, win_size=51, high=4000, blocks=((16, 5, 7), (8, 3, 7)), n_labels=3)}
    for name, kw in MFE_BLOCK_CASES:
        gm = pkg.Model(blob=synth_model_blob(**models[name]))
        cfg = cfg0.copy(**kw)
        nf = gm.n_filters
        rows = oracle.num_frames(16000, cfg)

        def run(clips):
            d = torch.from_numpy(np.ascontiguousarray(clips)).to("cuda:0")
            out = torch.empty((len(clips), rows * nf), dtype=torch.float32, device="cuda:0")
            gm.extract_mfe_batch_device(d.data_ptr(), len(clips), out.data_ptr())
            torch.cuda.synchronize()
            return out.cpu().numpy()
        got = run(gclips)
        for i in range(len(gclips)):
            assert same_bits_or_both_nan(got[i], g[name][i]), (name, i)            # the reference itself
        clips = oracle.synth(123, 7, 200)
        got = run(clips)
        for i in range(len(clips)):
            assert same_bits_or_both_nan(got[i], oracle.extract_mfe(clips[i], cfg)), (name, i)
        gm.close()


def test_run_classifier_latency_mode_matches_batch_path(pkg, gpu476, oracle, tmp_path):
    """run_classifier() on one window takes the latency-mode MFCC kernel (7 waves per window, cmvnw dealt out in 5-row tasks);
    its scores must be those of the batch path (one wave per clip) bit for bit -- 32 and 40 filters, 13 / 40 / 30 / 10 cepstra,
    windows 101 and 51, int8 and float32 graphs, random and edge-case clips."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from dequantize_model import dequantize
    from kws_testlib import synth_model_blob
    sp = special_clips()
    clips = np.concatenate([oracle.synth(77, 3, 6), np.stack([sp[k] for k in sorted(sp)])])
    blobs = [open(os.path.join(MODELS, "l476_no_yes.kwsm"), "rb").read(), open(os.path.join(MODELS, "cfg2_mfcc40_f32.kwsm"), "rb").read(),
             synth_model_blob(**MFCC40_MODELS["f40c30w51"]), dequantize(synth_model_blob(**MFCC40_MODELS["f40c13"])),
             synth_model_blob(seed=2, ncep=10, win_size=51, high=0, blocks=((16, 5, 7), (8, 3, 7)), n_labels=3)]
    try:
        for bi, blob in enumerate(blobs):
            gm = pkg.Model(blob=blob)
            gm.set_default()
            want = gm.run_classifier_batch(clips)
            res = pkg.result_struct(gm.n_labels)()
            for ci, clip in enumerate(clips):
                buf = clip.astype(np.float32) / np.float32(32768)

                @pkg.GET_DATA_FN
                def get_data(offset, length, out):
                    ctypes.memmove(out, buf[offset:offset + length].ctypes.data, 4 * length)
                    return 0
                sig = pkg.Signal(get_data=get_data, total_length=16000)
                assert pkg.lib().run_classifier(ctypes.byref(sig), ctypes.byref(res), False) == 0
                got = np.float32([res.classification[i].value for i in range(gm.n_labels)])
                assert (bits(got) == bits(want[ci])).all(), (bi, ci)
            gpu476.set_default()
            gm.close()
    finally:
        gpu476.set_default()


def test_random_graphs_on_gpu(pkg, oracle, tmp_path):
    """Fuzz: the random graphs of test_random_graphs_through_reference_op_registrations (there held to the reference's op code)
    on the HIP path: int8 -- every block's pooled tensor, FC and softmax outputs exactly; float32 twins -- logits bit for bit,
    scores within 1e-6.  Draws the plan refuses (outside documented limits) must fail loudly with KWS_ERROR_UNSUPPORTED_MODEL."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from dequantize_model import dequantize
    from kws_testlib import OracleModel, random_graph_spec, synth_model_blob
    rng = np.random.default_rng(10)
    n_ok = n_refused = 0
    for seed in range(150):
        kw = random_graph_spec(seed)
        if kw is None:
            continue
        blob = synth_model_blob(**kw)
        try:
            gm = pkg.Model(blob=blob)
        except pkg.KwsError as e:
            assert e.code == -18, (seed, kw)
            n_refused += 1
            continue
        p = tmp_path / ("fz%d.kwsm" % seed)
        p.write_bytes(blob)
        om = OracleModel(oracle, str(p))
        qs = rng.integers(-128, 128, (40, om.n_features)).astype(np.int8)
        s, pooled, fc, out = gm.nn_batch(qs)
        for i in range(0, 40, 3):
            o, taps = om.nn_invoke(qs[i], taps=True)
            assert (out[i] == o).all(), (seed, kw)
            assert (fc[i] == taps[len(taps) - 2]).all(), (seed, kw)
        gm.close()
        bf = dequantize(blob)
        try:
            gf = pkg.Model(blob=bf)
        except pkg.KwsError as e:                     # float weights of a wide graph can exceed the CU's LDS: refused, loudly
            assert e.code == -18, (seed, kw)
            n_ok += 1
            continue
        pf = tmp_path / ("fz%df.kwsm" % seed)
        pf.write_bytes(bf)
        of = OracleModel(oracle, str(pf))
        x = (rng.standard_normal((24, of.n_features)) * np.float32(4.0)).astype(np.float32)
        sg, lg = _f32_logits(pkg, gf, x)
        n_t = len(of.tensor_bytes)
        for i in range(0, 24, 2):
            so, taps = of.nn_invoke_f32(x[i], taps=True)
            assert (bits(lg[i]) == bits(taps[n_t - 2])).all(), (seed, kw)
            assert np.abs(sg[i] - so).max() <= F32_SCORE_TOL, (seed, kw)
        gf.close()
        n_ok += 1
    assert n_ok >= 60, (n_ok, n_refused)


def test_random_mfcc_configurations_on_gpu(pkg, oracle):
    """Fuzz: the random DSP configurations of test_random_mfcc_configurations on the HIP path, features bit for bit against the
    restatement (which the CPU test holds to the reference); configurations the kernel is not built for (e.g. a mel filter with
    more taps than it keeps in registers) must be refused with KWS_ERROR_UNSUPPORTED_MODEL."""
    from kws_testlib import L476_CONFIG, random_dsp_spec, synth_model_blob
    sp = special_clips()
    clips = np.concatenate([oracle.synth(21, 0, 10), np.stack([sp["impulses"], sp["ramp"], sp["zeros"], sp["alternating_fullscale"]])])
    n_ok = n_refused = 0
    for seed in range(40):
        cfg_kw, blob_kw = random_dsp_spec(seed)
        try:
            gm = pkg.Model(blob=synth_model_blob(**blob_kw))
        except pkg.KwsError as e:
            assert e.code == -18, (seed, cfg_kw)
            n_refused += 1
            continue
        cfg = L476_CONFIG().copy(**cfg_kw)
        _, f, _ = gm.run_classifier_batch(clips, want_features=True)
        for i, c in enumerate(clips):
            assert (bits(f[i]) == bits(oracle.extract_mfcc(c, cfg))).all(), (seed, cfg_kw, i)
        gm.close()
        n_ok += 1
    assert n_ok >= 25, (n_ok, n_refused)


def test_default_model_from_an_unsupported_file_returns_the_error(pkg, tmp_path):
    """KWS_MODEL names a file that parses but that a plan builder refuses: run_classifier() must come back with the error.  (It used
    to dead-lock: kws_default_model() held its mutex while kws_create()'s failure path asked kws_destroy() -> kws_sdk_forget_default()
    for the same mutex.)  Run in a child process with a time limit so that a regression is a failure, not a hung suite."""
    import subprocess
    import sys
    from kws_testlib import ROOT, synth_model_blob
    bad = tmp_path / "unsupported.kwsm"
    bad.write_bytes(synth_model_blob(seed=3, blocks=((64, 3, 1),), n_labels=6))      # FULLY_CONNECTED input beyond the kernel's limit
    code = (
        "import sys, ctypes\n"
        "sys.path.insert(0, %r)\n"
        "import torch\n"
        "from __graft_entry__ import load_package\n"
        "pkg = load_package(); L = pkg.lib()\n"
        "assert L.kws_default_model() is None\n"
        "res = pkg.result_struct(6)()\n"
        "s = pkg.Signal(pkg.GET_DATA_FN(lambda off, n, out: 0), 16000)\n"
        "rc = L.run_classifier(ctypes.byref(s), ctypes.byref(res), False)\n"
        "assert rc != 0 and len(L.kws_last_error()) > 0, rc\n"
        "print('rc', rc)\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, KWS_MODEL=str(bad)), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=240)
    assert out.returncode == 0, out.stdout + out.stderr[-2000:]
    assert "rc -" in out.stdout
