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
