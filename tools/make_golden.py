#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the UNMODIFIED reference compiled by `make -C oracle ref`.

Runs only where /root/reference exists (the build container).  The vectors are data: inputs are the
deterministic synthetic clips of include/kws/kws_synth.h (regenerated in the tests from seed/index, plus
the hand-made edge-case clips of tests/kws_testlib.special_clips) and the values are what the reference
returned for them at each stage of run_classifier():
  pre-emphasised frames, power spectrum, mel energies, frame energies, MFCC before CMVN,
  features (after CMVN), int8 input tensor, every int8 op output, final scores.

Without arguments: leaves_l476, e2e_l476, deep_l476, continuous_l476, f32_twin_l476.  One fixture at a time:
  --only-mfcc40            mfcc40_l476.npz          the 40-filter MFCC configurations
  --only-mfe-block         mfe_block_l432.npz       the MFE block's leaves (L432 headers compiled in place)
  --only-mfe-model         mfe_model_l432.npz       a model whose DSP block is MFE, composed from the reference's leaves
  --only-graphs            graphs_l476.npz          synthetic graphs through the reference's op registrations
  --only-qfb               qfb_l476.npz             EIDSP_QUANTIZE_FILTERBANK = 1 (the second build of the compiled reference)
  --only-trace             get_data_trace_l476.npz  what the reference asks the application's callback, call by call (FIRST user of continuous mode in its process)
  --only-debug-cancel      debug_cancel_l476.npz    the text debug = true prints and what the cancellation polls do (FIRST user of continuous mode in its process)
  --only-other-length      other_length_l476.npz    run_classifier on windows of another length (1 .. 49 frames)
  --only-mfe-other-length  mfe_other_length_l432.npz  the same for the MFE-block model, composed from the L432 copy's leaves
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from kws_testlib import GOLDEN, L476_CONFIG, Oracle, Reference, special_clips  # noqa: E402

SEEDS = (1, 2, 3)
CLIPS_PER_SEED = 32      # end-to-end vectors
DEEP_PER_SEED = 2        # clips with every intermediate stage stored


def mfcc40(ref, synth, cfg):
    """BASELINE's 40-band MFCC variants through the reference's extract_mfcc_features (radix-5 DCT FFT, and the
    coefficients above N/2 that the transform never writes)."""
    clips = synth.synth(8, 0, 12)
    sp = special_clips()
    clips = np.concatenate([clips, np.stack([sp[k] for k in sorted(sp)])])
    out = {"seed": np.int32(8), "n": np.int32(12), "special_names": np.array(sorted(sp))}
    for name, kw in (("f40c40", dict(num_filters=40, num_cepstral=40, high_frequency=0)),
                     ("f40c13", dict(num_filters=40, num_cepstral=13, low_frequency=0, high_frequency=0)),
                     ("f40c30w51", dict(num_filters=40, num_cepstral=30, win_size=51))):
        c = cfg.copy(**kw)
        out[name] = np.stack([ref.extract_mfcc(x, c) for x in clips])
    np.savez_compressed(os.path.join(GOLDEN, "mfcc40_l476.npz"), **out)
    print("mfcc40_l476.npz", os.path.getsize(os.path.join(GOLDEN, "mfcc40_l476.npz")), "bytes")


def mfe_block(ref, synth, cfg):
    """The MFE block of the L432 SDK copy (extract_mfe_features, ei_run_dsp.h:369-418): feature::mfe on the raw signal -- the
    L476 build's, the function is the same text in both copies -- then cmvnw(win_size, false, true) + numpy::normalize from
    the L432 headers compiled in place (oracle/ref_l432_dsp.cpp)."""
    from kws_testlib import ReferenceL432Dsp
    r432 = ReferenceL432Dsp()
    clips = synth.synth(9, 0, 10)
    sp = special_clips()
    clips = np.concatenate([clips, np.stack([sp[k] for k in sorted(sp) if k != "zeros"])])   # all-zero clip: max == min -> 0 * inf
    out = {"seed": np.int32(9), "n": np.int32(10), "special_names": np.array([k for k in sorted(sp) if k != "zeros"])}
    for name, kw in (("f32", dict()), ("f40", dict(num_filters=40, num_cepstral=40, high_frequency=0)), ("f32w51", dict(win_size=51))):
        c = cfg.copy(pre_cof=0.0, **kw)
        rows = []
        for x in clips:
            mel, _ = ref.mfe(x, c)
            rows.append(r432.cmvnw(mel, c.win_size, False, True).reshape(-1))
        out[name] = np.stack(rows)
    np.savez_compressed(os.path.join(GOLDEN, "mfe_block_l432.npz"), **out)
    print("mfe_block_l432.npz", os.path.getsize(os.path.join(GOLDEN, "mfe_block_l432.npz")), "bytes")


MFE_MODEL_KW = dict(seed=77, blocks=((8, 3, 7), (4, 3, 7)), n_labels=3, dsp_block="mfe")     # 32 filters, 300-4000 Hz, cmvnw window 101


def mfe_model(ref, synth, cfg):
    """A model whose DSP block is MFE (synthetic graph: the reference ships none), composed from the reference's own leaves:
    one-shot = extract_mfe_features (feature::mfe of the L476 build on the raw signal, cmvnw(win, false, true) + normalize of the
    L432 headers) -> input quantisation -> the graph through the reference's op registrations; continuous = run_classifier_continuous
    of the L432 copy (classifier/ei_run_classifier.h:185-296): per slice feature::mfe only (extract_mfe_per_slice_features; every
    slice but the first claims one more frame length), rolling buffer, normalisation of a COPY of the full buffer, network,
    2-tap moving average."""
    from kws_testlib import OracleModel, ReferenceL432Dsp, synth_model_blob
    r432 = ReferenceL432Dsp()
    blob = synth_model_blob(**MFE_MODEL_KW)
    tmp = os.path.join(GOLDEN, "_mfe_model_tmp.kwsm")
    open(tmp, "wb").write(blob)
    om = OracleModel(synth, tmp)                       # only for the input quantisation / output dequantisation leaves
    os.remove(tmp)
    c = cfg.copy(pre_cof=0.0)
    NF, F = c.num_filters, 49 * c.num_filters

    def infer(features):
        q = om.quantize_input(features)
        out, _ = ref.graph_run(blob, q)
        return q, om.dequantize(out)
    sp = special_clips()
    clips = np.concatenate([synth.synth(9, 100, 10), np.stack([sp[k] for k in sorted(sp) if k != "zeros"])])
    feats, qs, scores = [], [], []
    for x in clips:
        mel, _ = ref.mfe(x, c)
        f = r432.cmvnw(mel, c.win_size, False, True).reshape(-1)
        q, s = infer(f)
        feats.append(f); qs.append(q); scores.append(s)
    out = {"seed": np.int32(9), "first": np.int32(100), "n": np.int32(10), "special_names": np.array([k for k in sorted(sp) if k != "zeros"]),
           "features": np.stack(feats), "q": np.stack(qs), "scores": np.stack(scores)}
    # continuous mode: 3 s of audio = 12 slices
    audio = synth.synth(14, 0, 3).reshape(-1)
    buf = np.zeros(F, np.float32)
    slice_offset, full, first_run = 0, False, False
    maf = [dict(idx=0, run=np.float32(0), buf=np.zeros(2, np.float32)) for _ in range(om.n_labels)]
    prod, sc = [], []
    flen = 320
    for k in range(12):
        sl = audio[k * 4000:(k + 1) * 4000]
        claimed = 4000 + (flen if first_run else 0)
        first_run = True
        padded = np.concatenate([sl, np.zeros(claimed - 4000, np.int16)])       # the frame count comes from the claimed length; no frame reads beyond the slice
        mel, _ = ref.mfe(padded, c)
        fsz = mel.size
        assert (mel.shape[0] - 1) * 320 + 320 <= 4000
        buf[slice_offset:slice_offset + fsz] = mel.reshape(-1)
        if not full:
            slice_offset += fsz
            if slice_offset > F - fsz:
                full = True
                slice_offset -= fsz
        s = np.zeros(om.n_labels, np.float32)
        if full:
            _, s = infer(r432.cmvnw(buf.copy().reshape(49, NF), c.win_size, False, True).reshape(-1))
            for i in range(om.n_labels):                                          # run_moving_average_filter
                m = maf[i]
                m["run"] = np.float32(m["run"] - m["buf"][m["idx"]])
                m["run"] = np.float32(m["run"] + s[i])
                m["buf"][m["idx"]] = s[i]
                m["idx"] = (m["idx"] + 1) % 2
                s[i] = np.float32(m["run"] / np.float32(2))
            buf[:F - fsz] = buf[fsz:].copy()
        prod.append(full); sc.append(s)
    out["cont_audio_seed"] = np.int32(14)
    out["cont_produced"] = np.array(prod)
    out["cont_scores"] = np.stack(sc)
    np.savez_compressed(os.path.join(GOLDEN, "mfe_model_l432.npz"), **out)
    print("mfe_model_l432.npz", os.path.getsize(os.path.join(GOLDEN, "mfe_model_l432.npz")), "bytes")


def graphs(ref):
    """Synthetic graphs (kws_testlib.SYNTH_SPECS; int8 and float32 twins) evaluated by the reference's own TFLite-Micro
    op registrations (eiref_graph_run): inputs are regenerated in the tests from the seed, outputs are stored."""
    from kws_testlib import SYNTH_SPECS, synth_model_blob
    from dequantize_model import dequantize
    import eon_import
    out = {"names": np.array(sorted(SYNTH_SPECS)), "n": np.int32(16), "rng_seed": np.int32(17)}
    for name in sorted(SYNTH_SPECS):
        blob = synth_model_blob(**SYNTH_SPECS[name])
        tens, nodes, t_in, t_out, _ = eon_import.parse_blob(blob)
        nfeat = tens[t_in]["nbytes"]
        fc_t = nodes[-1]["in"][0]
        rng = np.random.default_rng(17)
        xi = rng.integers(-128, 128, (16, nfeat)).astype(np.int8)
        xf = (rng.standard_normal((16, nfeat)) * 3).astype(np.float32)
        bf = dequantize(blob)
        qi, fi, lf, sf = [], [], [], []
        for k in range(16):
            o, taps = ref.graph_run(blob, xi[k])
            qi.append(o); fi.append(taps[fc_t].copy())
            o, taps = ref.graph_run(bf, xf[k])
            sf.append(o); lf.append(taps[fc_t].copy())
        out[name + "_i8_out"], out[name + "_i8_fc"] = np.stack(qi), np.stack(fi)
        out[name + "_f32_scores"], out[name + "_f32_logits"] = np.stack(sf), np.stack(lf)
    np.savez_compressed(os.path.join(GOLDEN, "graphs_l476.npz"), **out)
    print("graphs_l476.npz", os.path.getsize(os.path.join(GOLDEN, "graphs_l476.npz")), "bytes")


def qfb(synth, cfg):
    """EIDSP_QUANTIZE_FILTERBANK = 1 (the SDK's default): extract_mfcc_features of the reference built with the option at its default
    (oracle/_ref/libei_ref_l476_qfb.so), for configurations whose filters are wide enough for the table to move a weight (fft 512 / 1024)
    and for the shipped one (where it is the identity)."""
    ref = Reference(qfb=True)
    clips = synth.synth(12, 0, 8)
    out = {"seed": np.int32(12), "n": np.int32(8)}
    for name, kw in (("l476", dict()), ("fft512", dict(fft_length=512, high_frequency=0)),
                     ("fft1024_f20", dict(fft_length=1024, num_filters=20, num_cepstral=12, low_frequency=0, high_frequency=0))):
        c = cfg.copy(quantize_filterbank=1, **kw)
        out[name] = np.stack([ref.extract_mfcc(x, c) for x in clips])
        out[name + "_fb"] = ref.filterbanks(c)
    np.savez_compressed(os.path.join(GOLDEN, "qfb_l476.npz"), **out)
    print("qfb_l476.npz", os.path.getsize(os.path.join(GOLDEN, "qfb_l476.npz")), "bytes")


def get_data_trace(ref):
    """What the reference asks the application's get_data callback, in order: (offset, length, return value) per call, for the scenarios
    tests/sanitize/host_driver.cpp --trace replays through the library (one-shot, one-shot with a window one sample short, the process's first
    continuous call, a later one -- total_length grown by a frame length in the caller's struct, ei_run_dsp.h:318-326 --, a half slice).
    The callback refuses ranges beyond the real buffer (oracle/ref_driver.cpp pcm_get_data).  MUST be the first user of
    run_classifier_continuous in its process (the reference's function-static first_run): run as  make_golden.py --only-trace."""
    n, out = 16000, {}
    clip = Oracle().synth(5, 0, 1)[0]
    (r, _s, tl, _gc), t = ref.traced(ref.run_classifier, clip)
    out["oneshot"], out["oneshot_meta"] = t, np.int64([tl, int(r != 0)])
    (r, _s, tl, _gc), t = ref.traced(ref.run_classifier, clip[:n - 1])
    out["oneshot_short"], out["oneshot_short_meta"] = t, np.int64([tl, int(r != 0)])
    ref.continuous_init()
    for tag, m in (("continuous_first", n // 4), ("continuous_second", n // 4), ("continuous_short", n // 8)):
        (r, _p, _s, tl), t = ref.traced(ref.continuous, clip[:m])
        out[tag], out[tag + "_meta"] = t, np.int64([tl, int(r != 0)])
    np.savez_compressed(os.path.join(GOLDEN, "get_data_trace_l476.npz"), **out)
    for k in ("oneshot", "oneshot_short", "continuous_first", "continuous_second", "continuous_short"):
        print(k, len(out[k]), "calls; total_length after, error:", out[k + "_meta"].tolist(), "first:", out[k][:3].tolist())


# the boundary scenarios, in the order every implementation must run them in a FRESH process (the reference's function-static
# first_run makes the process's first continuous call special): (name, kind, what, debug, cancel_at)
#   what: ("clip", i) = window i of the seed-9 clips; ("features", i) = extract_mfcc_features of window i; ("slice", k) = slice k of the
#   2 s stream made of windows 2 and 3; "init" = run_classifier_init()
BOUNDARY_SCENARIOS = (
    ("oneshot_dbg_a", "oneshot", ("clip", 0), 1, 0), ("oneshot_dbg_b", "oneshot", ("clip", 1), 1, 0),
    ("inference_dbg_a", "inference", ("features", 0), 1, 0),
    ("oneshot_cancel1", "oneshot", ("clip", 0), 0, 1), ("oneshot_cancel2", "oneshot", ("clip", 0), 0, 2), ("oneshot_cancel1_dbg", "oneshot", ("clip", 0), 1, 1),
    ("oneshot_cancel2_dbg", "oneshot", ("clip", 0), 1, 2), ("inference_cancel1", "inference", ("features", 0), 0, 1),
    ("init0", "init", None, 0, 0),
    ("cont_dbg_0", "continuous", ("slice", 0), 1, 0), ("cont_dbg_1", "continuous", ("slice", 1), 1, 0), ("cont_dbg_2", "continuous", ("slice", 2), 1, 0),
    ("cont_dbg_3", "continuous", ("slice", 3), 1, 0), ("cont_dbg_4", "continuous", ("slice", 4), 1, 0),
    ("init1", "init", None, 0, 0),
    ("cont_cancel1_first", "continuous", ("slice", 0), 0, 1),          # cancelled after the DSP block: the slice is not committed
    ("cont_after_0", "continuous", ("slice", 0), 0, 0), ("cont_after_1", "continuous", ("slice", 1), 0, 0), ("cont_after_2", "continuous", ("slice", 2), 0, 0),
    ("cont_after_3", "continuous", ("slice", 3), 0, 0),
    ("cont_cancel2_full", "continuous", ("slice", 4), 0, 2),           # buffer full: cancelled inside run_inference, the filter and the shift still happen
    ("cont_after_5", "continuous", ("slice", 5), 0, 0),
    ("cont_cancel1_full", "continuous", ("slice", 6), 1, 1),           # buffer full, cancelled after the DSP block (debug on: nothing printed yet)
    ("cont_after_7", "continuous", ("slice", 7), 0, 0),
)


def debug_cancel(ref, synth, cfg):
    """What the reference prints with debug = true (ei_run_classifier.h:698-705, 463-479; continuous :242-253) and what its cancellation
    polls do (:689, :489; continuous :221) -- return code, number of polls, the caller's result struct byte for byte (pre-filled with 0xA5:
    what stays 0xA5 the reference did not touch) -- for BOUNDARY_SCENARIOS.  Run as  make_golden.py --only-debug-cancel  (fresh process)."""
    clips = synth.synth(9, 0, 4)
    stream = clips[2:4].reshape(-1)
    out = {"clips": clips, "names": np.array([x[0] for x in BOUNDARY_SCENARIOS])}
    feats = {}
    for name, kind, what, debug, cancel_at in BOUNDARY_SCENARIOS:
        if kind == "init":
            ref.continuous_init()
            continue
        if what[0] == "clip":
            data = clips[what[1]]
        elif what[0] == "features":
            if what[1] not in feats:
                feats[what[1]] = ref.extract_mfcc(clips[what[1]], cfg).reshape(-1)
            data = feats[what[1]]
        else:
            data = stream[what[1] * 4000:(what[1] + 1) * 4000]
        rc, polls, res, lab, text = ref.boundary_call(kind, data, debug, cancel_at)
        out[name + "_meta"] = np.int64([rc, polls])
        out[name + "_result"] = res
        out[name + "_labels"] = lab
        out[name + "_text"] = np.frombuffer(text, np.uint8)
        print("%-22s rc %2d polls %d touched %2d of %d bytes, text %5d bytes: %s" % (name, rc, polls, int((res != 0xA5).sum()), res.size, len(text),
                                                                                  text[:60].decode(errors="replace").replace("\n", "|")))
    out["features_a"] = feats[0]
    out["labels"] = np.array(ref.labels)
    np.savez_compressed(os.path.join(GOLDEN, "debug_cancel_l476.npz"), **out)
    print("debug_cancel_l476.npz", os.path.getsize(os.path.join(GOLDEN, "debug_cancel_l476.npz")), "bytes")


OTHER_LENGTHS = (16319, 16001, 15999, 15681, 15680, 15679, 8000, 1280, 960, 641, 640)
MFE_OTHER_LENGTHS = (16319, 16001, 15999, 15680, 15679, 8000, 1280, 641, 640)


def mfe_other_length(ref, synth, cfg):
    """The MFE-block model (mfe_model above) on one-shot windows of another length, composed from the reference's leaves the way mfe_model is:
    extract_mfe_features sizes its matrix from signal->total_length with the same rule as the MFCC block (L432 ei_run_dsp.h:379-389: only a
    matrix that does not fit is an error), feature::mfe on the frames that fit (the L476 build's leaf), cmvnw(win, false, true) + normalize over
    those rows only (the L432 headers, compiled), the rest of the calloc'd feature matrix (L432 dsp/numpy_types.h:91) at zero, input
    quantisation, the graph through the reference's op registrations.  The L432 copy's run_classifier itself cannot be compiled here: this row
    is pinned by its leaves, not by an end-to-end run."""
    from kws_testlib import OracleModel, ReferenceL432Dsp, synth_model_blob
    r432 = ReferenceL432Dsp()
    blob = synth_model_blob(**MFE_MODEL_KW)
    tmp = os.path.join(GOLDEN, "_mfe_model_tmp.kwsm")
    open(tmp, "wb").write(blob)
    om = OracleModel(synth, tmp)
    os.remove(tmp)
    c = cfg.copy(pre_cof=0.0)
    F = 49 * c.num_filters
    clips = synth.synth(19, 0, 6).reshape(3, 32000)[:, :max(MFE_OTHER_LENGTHS)]
    feats = np.zeros((len(clips), len(MFE_OTHER_LENGTHS), F), np.float32)
    qs = np.zeros(feats.shape, np.int8)
    scores = np.zeros((len(clips), len(MFE_OTHER_LENGTHS), om.n_labels), np.float32)
    for i, x in enumerate(clips):
        for j, L in enumerate(MFE_OTHER_LENGTHS):
            mel, _ = ref.mfe(x[:L], c)
            f = r432.cmvnw(mel, c.win_size, False, True).reshape(-1)
            assert f.size == c.num_filters * ((L - 320) // 320) <= F
            feats[i, j, :f.size] = f
            qs[i, j] = om.quantize_input(feats[i, j])
            out, _ = ref.graph_run(blob, qs[i, j])
            scores[i, j] = om.dequantize(out)
    np.savez_compressed(os.path.join(GOLDEN, "mfe_other_length_l432.npz"), clips=clips, lengths=np.int64(MFE_OTHER_LENGTHS), features=feats, q=qs, scores=scores)
    print("mfe_other_length_l432.npz", os.path.getsize(os.path.join(GOLDEN, "mfe_other_length_l432.npz")), "bytes;", scores[0, :3].round(4).tolist())



def other_length(ref, synth, cfg):
    """run_classifier on windows of another length than the model's 16 000 samples (ei_run_dsp.h:277-284 refuses only a feature matrix that
    would not fit): per clip and length the compiled reference's return value and scores, the feature matrix it classified (its
    extract_mfcc_features on that window, the rest at the calloc'd zeros) and the float32 twin's scores for that matrix through the
    reference's op registrations.  Only lengths with 1 .. 49 frames (640 .. 16 319 samples): there the compiled reference is deterministic and
    memory-safe (checked per length in processes of their own under MALLOC_CHECK_=3 with two MALLOC_PERTURB_ patterns).  A window with more
    frames or none has no defined result in the reference: EIDSP_ERR is printf + assert(false) (dsp/config.hpp:65-67, EIDSP_USE_ASSERTS = 1 --
    this SDK copy does not compile with 0), so it aborts, or under NDEBUG runs on: 16 320 / 16 321 / 16 640 / 17 000 samples overflow its
    feature matrix (glibc: "free(): invalid next size"), 639 crashes after printing EIDSP_INPUT_MATRIX_EMPTY."""
    clips = synth.synth(9, 0, 6).reshape(3, 32000)[:, :max(OTHER_LENGTHS)]
    from kws_testlib import MODELS
    twin = open(os.path.join(MODELS, "l476_no_yes_f32.kwsm"), "rb").read()
    F = ref.n_features
    rc = np.zeros((len(clips), len(OTHER_LENGTHS)), np.int32)
    scores = np.zeros(rc.shape + (ref.n_labels,), np.float32)
    twin_scores = np.zeros_like(scores)
    feats = np.zeros(rc.shape + (F,), np.float32)
    calls = np.zeros(rc.shape, np.int32)
    for i, c in enumerate(clips):
        for j, L in enumerate(OTHER_LENGTHS):
            r, s, tl, gc = ref.run_classifier(c[:L])
            assert tl == L
            rc[i, j], scores[i, j], calls[i, j] = r, s, gc
            if r == 0:
                f = ref.extract_mfcc(c[:L], cfg)
                feats[i, j, :f.size] = f
                twin_scores[i, j] = ref.graph_run(twin, feats[i, j])[0]
    np.savez_compressed(os.path.join(GOLDEN, "other_length_l476.npz"), clips=clips, lengths=np.int64(OTHER_LENGTHS), rc=rc, scores=scores, features=feats,
                        twin_scores=twin_scores, get_data_calls=calls)
    print("other_length_l476.npz", os.path.getsize(os.path.join(GOLDEN, "other_length_l476.npz")), "bytes; rc per length:", dict(zip(OTHER_LENGTHS, rc[0].tolist())),
          "calls:", calls[0].tolist())


def main():
    if "--only-qfb" in sys.argv:
        return qfb(Oracle(), L476_CONFIG())
    ref = Reference()
    if "--only-trace" in sys.argv:
        return get_data_trace(ref)
    if "--only-debug-cancel" in sys.argv:
        return debug_cancel(ref, Oracle(), L476_CONFIG())
    if "--only-other-length" in sys.argv:
        return other_length(ref, Oracle(), L476_CONFIG())
    if "--only-mfe-other-length" in sys.argv:
        return mfe_other_length(ref, Oracle(), L476_CONFIG())
    if "--only-mfcc40" in sys.argv:
        return mfcc40(ref, Oracle(), L476_CONFIG())
    if "--only-mfe-block" in sys.argv:
        return mfe_block(ref, Oracle(), L476_CONFIG())
    if "--only-mfe-model" in sys.argv:
        return mfe_model(ref, Oracle(), L476_CONFIG())
    if "--only-graphs" in sys.argv:
        return graphs(ref)
    synth = Oracle()          # only used for kwso_synth_fill (shared integer generator)
    cfg = L476_CONFIG()
    os.makedirs(GOLDEN, exist_ok=True)

    # ---- leaves / tables --------------------------------------------------------------------
    xs = np.concatenate([np.float32([1.1920929e-07, 1e-30, 1e-10, 0.5, 2 / 3, 1.0, 4 / 3, 2.5, 1e4, 3e38]),
                         np.exp(np.linspace(-40, 40, 200)).astype(np.float32)])
    leaves = {
        "log_x": xs, "log_y": np.float32([ref.L.eiref_log(float(x)) for x in xs]),
        "mel_f": np.float32([0, 300, 1000, 4000, 8000]),
        "mel_y": np.float32([ref.L.eiref_frequency_to_mel(f) for f in (0, 300, 1000, 4000, 8000)]),
        "filterbank_l476": ref.filterbanks(cfg),
        "filterbank_l432": ref.filterbanks(cfg.copy(high_frequency=0)),
        "filterbank_40": ref.filterbanks(cfg.copy(num_filters=40, high_frequency=0)),
        "dct_ramp32": ref.dct2_ortho(np.arange(32, dtype=np.float32)),
        "dct_ramp40": ref.dct2_ortho(np.arange(40, dtype=np.float32)),
    }
    rng = np.random.default_rng(7)
    x256 = rng.standard_normal(256).astype(np.float32)
    leaves["rfft256_x"] = x256
    leaves["rfft256_y"] = ref.rfft_complex(x256)
    x32 = rng.standard_normal(32).astype(np.float32)
    leaves["rfft32_x"] = x32
    leaves["rfft32_y"] = ref.rfft_complex(x32)
    m = (rng.standard_normal((49, 13)) * np.float32([5] + [1] * 12)).astype(np.float32)
    leaves["cmvn_x"] = m
    leaves["cmvn_y"] = ref.cmvnw(m, 101, True)
    leaves["cmvn_y_novar"] = ref.cmvnw(m, 101, False)
    np.savez_compressed(os.path.join(GOLDEN, "leaves_l476.npz"), **leaves)

    # ---- end-to-end -------------------------------------------------------------------------
    e2e = {"seeds": np.int32(SEEDS), "clips_per_seed": np.int32(CLIPS_PER_SEED)}
    feats, scores, qin, outq = [], [], [], []
    t_in, t_out = 0, len(ref.tensor_bytes) - 1
    for seed in SEEDS:
        clips = synth.synth(seed, 0, CLIPS_PER_SEED)
        for c in clips:
            rc, s, total_len_after, n_calls = ref.run_classifier(c)
            assert rc == 0 and total_len_after == 16000 and n_calls == 98
            f = ref.extract_mfcc(c, cfg)
            feats.append(f)
            scores.append(s)
    e2e["features"] = np.stack(feats)
    e2e["scores"] = np.stack(scores)
    sp = special_clips()
    e2e["special_names"] = np.array(sorted(sp))
    e2e["special_features"] = np.stack([ref.extract_mfcc(sp[k], cfg) for k in sorted(sp)])
    e2e["special_scores"] = np.stack([ref.run_classifier(sp[k])[1] for k in sorted(sp)])
    np.savez_compressed(os.path.join(GOLDEN, "e2e_l476.npz"), **e2e)

    # ---- deep (every stage) -----------------------------------------------------------------
    deep = {}
    from kws_testlib import OracleModel, MODELS
    om = OracleModel(synth, os.path.join(MODELS, "l476_no_yes.kwsm"))   # only for the quantise step below
    k = 0
    deep_clips = [synth.synth(seed, i, 1)[0] for seed in SEEDS for i in range(DEEP_PER_SEED)]
    deep_ids = [(seed, i) for seed in SEEDS for i in range(DEEP_PER_SEED)]
    deep_clips += [sp["step"], sp["impulses"]]
    deep_ids += [(-1, 0), (-1, 1)]
    deep["ids"] = np.int32(deep_ids)
    for c in deep_clips:
        p = f"c{k}_"
        deep[p + "pre_f0"] = ref.preemphasis(c, cfg.pre_cof, cfg.pre_shift, 0, 320)
        deep[p + "pre_f7"] = ref.preemphasis(c, cfg.pre_cof, cfg.pre_shift, 7 * 320, 320)
        deep[p + "ps_f7"] = ref.power_spectrum(deep[p + "pre_f7"], cfg.fft_length)
        mel, en = ref.mfe(c, cfg)
        deep[p + "mel"], deep[p + "energy"] = mel, en
        deep[p + "mfcc"] = ref.mfcc_nocmvn(c, cfg)
        f = ref.extract_mfcc(c, cfg)
        deep[p + "features"] = f
        # int8 input tensor: quantised with the restatement (pinned against the reference's
        # run_inference in tests/test_oracle_vs_reference.py), then every op output from the reference
        q = om.quantize_input(f)
        taps = ref.nn_taps(q)
        deep[p + "q_in"] = q
        for tid, v in taps.items():
            deep[p + f"t{tid}"] = v
        deep[p + "scores"] = ref.run_inference(f)
        k += 1
    deep["n"] = np.int32(k)
    np.savez_compressed(os.path.join(GOLDEN, "deep_l476.npz"), **deep)
    # ---- continuous (sliced) mode: 5 s of audio = 20 slices of 4000 samples, then re-init and 6 more ----------
    # (this script is the first and only user of run_classifier_continuous in its process, so the reference's
    #  never-reset `first_run` static is false exactly for slice 0)
    audio = synth.synth(4, 0, 5).reshape(-1)
    cont = {"audio_seed": np.int32(4), "n_clips": np.int32(5)}
    ref.continuous_init()
    prod, sc, tls = [], [], []
    for k in range(20):
        rc, p, s, tl = ref.continuous(audio[k * 4000:(k + 1) * 4000])
        assert rc == 0
        prod.append(p); sc.append(s); tls.append(tl)
    ref.continuous_init()
    for k in range(6):
        rc, p, s, tl = ref.continuous(audio[k * 4000:(k + 1) * 4000])
        assert rc == 0
        prod.append(p); sc.append(s); tls.append(tl)
    cont["produced"] = np.array(prod)
    cont["scores"] = np.stack(sc)
    cont["total_length_after"] = np.int64(tls)
    np.savez_compressed(os.path.join(GOLDEN, "continuous_l476.npz"), **cont)
    # ---- fp32 twin (tools/dequantize_model.py) through the reference's FLOAT kernels, leaf by leaf --------------
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    import eon_import
    from kws_testlib import reference_float_twin
    tens, _, _, _, _ = eon_import.parse_blob(open(os.path.join(MODELS, "l476_no_yes_f32.kwsm"), "rb").read())
    consts = {i: np.frombuffer(t["data"], np.float32).copy() for i, t in enumerate(tens) if t["const"] and t["type"] == 1}
    clips = synth.synth(6, 0, 24)
    f32 = {"seed": np.int32(6), "n": np.int32(24)}
    lg, sc = [], []
    for c in clips:
        a, b = reference_float_twin(ref, consts, ref.extract_mfcc(c, cfg))
        lg.append(a); sc.append(b)
    f32["logits"], f32["scores"] = np.stack(lg), np.stack(sc)
    np.savez_compressed(os.path.join(GOLDEN, "f32_twin_l476.npz"), **f32)
    mfcc40(ref, synth, cfg)
    mfe_model(ref, synth, cfg)
    graphs(ref)
    for fn in ("leaves_l476.npz", "e2e_l476.npz", "deep_l476.npz", "continuous_l476.npz", "f32_twin_l476.npz"):
        print(fn, os.path.getsize(os.path.join(GOLDEN, fn)), "bytes")


if __name__ == "__main__":
    main()
