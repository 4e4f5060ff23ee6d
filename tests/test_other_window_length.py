"""run_classifier on a window of another length than the model's (the drop-in boundary; found while pinning the callback sequence).

The reference's one-shot extractor sizes its frame count from signal->total_length (ei_run_dsp.h:277-286, processing.hpp:194-284): a window with
1 .. 49 frames (640 .. 16 319 samples for the shipped impulse) is classified from the frames that fit -- normalised among themselves, x[-1] = the last
sample of THAT window, the rest of the network's input at its calloc'd zeros.  tests/golden/other_length_l476.npz holds what the compiled reference
returns there (tools/make_golden.py --only-other-length): scores, the feature matrix, the float32 twin's scores.  Outside that range the reference
has no defined result (EIDSP_ERR = printf + assert(false), dsp/config.hpp:65-67: abort, or under NDEBUG a heap overflow / crash -- see the
generator's docstring); the library returns EI_IMPULSE_DSP_ERROR there, which is ITS contract, not the reference's.

CPU: the oracle's functions composed the same way against the fixture (and the compiled reference again, when built).  -m gpu: the library.
"""
import ctypes
import os

import numpy as np
import pytest

from kws_testlib import GOLDEN, L476_CONFIG, MODELS, OracleModel, bits, have_reference


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "other_length_l476.npz"))


def test_oracle_composition_matches_the_reference_on_other_lengths(oracle, l476, gold):
    cfg = L476_CONFIG()
    twin = OracleModel(oracle, os.path.join(MODELS, "l476_no_yes_f32.kwsm"))
    assert (gold["rc"] == 0).all() and sorted(set(gold["get_data_calls"][0].tolist())) == [2, 4, 6, 48, 94, 96, 98]
    for i, clip in enumerate(gold["clips"]):
        for j, L in enumerate(gold["lengths"]):
            f = oracle.extract_mfcc(clip[:L], cfg)
            assert f.size == 13 * ((int(L) - 320) // 320)
            padded = np.zeros(637, np.float32)
            padded[:f.size] = f
            assert (bits(padded) == bits(gold["features"][i, j])).all(), (i, int(L))
            assert (bits(l476.run_inference(padded)) == bits(gold["scores"][i, j])).all(), (i, int(L))
            assert np.abs(twin.run_inference(padded) - gold["twin_scores"][i, j]).max() <= 1e-6, (i, int(L))
    # the wrap sample matters: the same 48 frames, three different windows, three different feature matrices
    j = {int(L): k for k, L in enumerate(gold["lengths"])}
    assert (bits(gold["features"][0, j[15999]]) != bits(gold["features"][0, j[15681]])).any()
    assert (bits(gold["features"][0, j[15681]]) != bits(gold["features"][0, j[15680]])).any()


@pytest.mark.skipif(not have_reference(), reason="oracle/_ref not built (no /root/reference here)")
def test_fixture_is_what_the_compiled_reference_returns(reference, gold):
    for i, clip in enumerate(gold["clips"]):
        for j, L in enumerate(gold["lengths"]):
            rc, s, tl, calls = reference.run_classifier(clip[:L])
            assert rc == 0 and tl == L and calls == gold["get_data_calls"][i, j]
            assert (bits(s) == bits(gold["scores"][i, j])).all(), (i, int(L))


def _sdk_call(pkg, clip, n_claimed, n_labels=4):
    def get_data(offset, length, out):
        if offset + length > len(clip):
            return -1
        seg = clip[offset:offset + length].astype(np.float32) / np.float32(32768)
        ctypes.memmove(out, seg.ctypes.data, 4 * length)
        return 0
    cb = pkg.GET_DATA_FN(get_data)
    sig = pkg.Signal(cb, n_claimed)
    res = pkg.result_struct(n_labels)()
    rc = pkg.lib().run_classifier(ctypes.byref(sig), ctypes.byref(res), False)
    return rc, np.float32([res.classification[i].value for i in range(n_labels)]), sig.total_length


def _oneshot_features(pkg, n):
    """the feature matrix the last run_classifier() call classified (kws_dev_oneshot_features: a test aid outside the public headers)"""
    out = np.zeros(n, np.float32)
    L = pkg.lib()
    L.kws_dev_oneshot_features.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    assert L.kws_dev_oneshot_features(out.ctypes.data_as(ctypes.c_void_p), n) == 0
    return out


@pytest.mark.gpu
def test_run_classifier_on_other_lengths_matches_the_reference(gold):
    from __graft_entry__ import load_package
    pkg = load_package()
    for name, key, tol in (("l476_no_yes.kwsm", "scores", 0.0), ("l476_no_yes_f32.kwsm", "twin_scores", 1e-6)):
        m = pkg.Model(os.path.join(MODELS, name), device=0)
        m.set_default()
        for i, clip in enumerate(gold["clips"]):
            for j, L in enumerate(gold["lengths"]):
                rc, s, tl = _sdk_call(pkg, clip[:L], int(L))
                assert rc == 0 and tl == L, (name, int(L), rc)
                assert (bits(_oneshot_features(pkg, 637)) == bits(gold["features"][i, j])).all(), (name, i, int(L))     # the matrix it classified
                if tol == 0.0:
                    assert (bits(s) == bits(gold[key][i, j])).all(), (name, i, int(L), s, gold[key][i, j])
                else:
                    assert np.abs(s - gold[key][i, j]).max() <= tol, (name, i, int(L))
            # the model's own length in between: the fused path is untouched by the other plans
            rc, s, _ = _sdk_call(pkg, clip[:16000], 16000)
            assert rc == 0
        # the library's own contract where the reference has none: more frames than the model's, or none -> EI_IMPULSE_DSP_ERROR (-5), nothing written
        clip = gold["clips"][0]
        for L in (16320, 16321, 17000, 639, 320, 1, 0):
            rc, s, tl = _sdk_call(pkg, clip[:L], L)
            assert rc == -5 and tl == L and (s == 0).all(), (name, L, rc)
        m.close()


# ---- a model whose DSP block is MFE (L432 copy): the same rule (ei_run_dsp.h:379-389), no pre-emphasis object, cmvnw(win, false, true) + normalize
#      over the rows that fit.  That copy's run_classifier cannot be compiled here: tests/golden/mfe_other_length_l432.npz is composed from its
#      compiled leaves (tools/make_golden.py --only-mfe-other-length), as tests/golden/mfe_model_l432.npz is.
def _mfe_blob():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(GOLDEN), "..", "tools"))
    from kws_testlib import synth_model_blob
    from make_golden import MFE_MODEL_KW
    return synth_model_blob(**MFE_MODEL_KW)


def test_oracle_composition_matches_the_leaves_for_an_mfe_block(oracle, tmp_path):
    g = np.load(os.path.join(GOLDEN, "mfe_other_length_l432.npz"))
    path = str(tmp_path / "mfe.kwsm")
    open(path, "wb").write(_mfe_blob())
    om = OracleModel(oracle, path)
    cfg = L476_CONFIG().copy(pre_cof=0.0)
    F = 49 * cfg.num_filters
    for i, clip in enumerate(g["clips"]):
        for j, L in enumerate(g["lengths"]):
            f = oracle.extract_mfe(clip[:L], cfg)
            padded = np.zeros(F, np.float32)
            padded[:f.size] = f
            assert f.size == cfg.num_filters * ((int(L) - 320) // 320)
            assert (bits(padded) == bits(g["features"][i, j])).all(), (i, int(L))
            assert (om.quantize_input(padded) == g["q"][i, j]).all(), (i, int(L))
            assert (bits(om.run_inference(padded)) == bits(g["scores"][i, j])).all(), (i, int(L))
    assert (g["q"][0, 2] != g["q"][0, 0]).sum() > 50 and (g["q"][0, 5] != g["q"][0, 0]).sum() > 1000      # the fixture tells the lengths apart


@pytest.mark.gpu
def test_run_classifier_on_other_lengths_for_an_mfe_block():
    from __graft_entry__ import load_package
    pkg = load_package()
    g = np.load(os.path.join(GOLDEN, "mfe_other_length_l432.npz"))
    m = pkg.Model(blob=_mfe_blob())
    m.set_default()
    F = m.n_features
    for i, clip in enumerate(g["clips"]):
        for j, L in enumerate(g["lengths"]):
            rc, s, tl = _sdk_call(pkg, clip[:L], int(L), n_labels=m.n_labels)
            assert rc == 0 and tl == L, (int(L), rc)
            assert (bits(_oneshot_features(pkg, F)) == bits(g["features"][i, j])).all(), (i, int(L))
            assert (bits(s) == bits(g["scores"][i, j])).all(), (i, int(L), s, g["scores"][i, j])
    for L in (16320, 17000, 639, 0):
        rc, s, tl = _sdk_call(pkg, g["clips"][0][:L], L, n_labels=m.n_labels)
        assert rc == -5 and (s == 0).all(), (L, rc)
    m.close()
