"""Child process of tests/test_gpu_mfe_model.py (a fresh process: the SDK's continuous mode keeps a never-reset `first_run`
static, like the reference): run_classifier() and run_classifier_continuous() with a model whose DSP block is MFE, against the
golden vectors composed from the reference's leaves.  Exit status 0 = all equal."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402,F401
from __graft_entry__ import load_package  # noqa: E402
from kws_testlib import GOLDEN, Oracle, bits, special_clips, synth_model_blob  # noqa: E402
from make_golden import MFE_MODEL_KW  # noqa: E402

pkg = load_package()
o = Oracle()
g = np.load(os.path.join(GOLDEN, "mfe_model_l432.npz"))
gm = pkg.Model(blob=synth_model_blob(**MFE_MODEL_KW))
gm.set_default()
L = pkg.lib()
L.run_classifier_continuous.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_bool]
Result = pkg.result_struct(gm.n_labels)
sp = special_clips()
clips = np.concatenate([o.synth(int(g["seed"]), int(g["first"]), int(g["n"])), np.stack([sp[str(k)] for k in g["special_names"]])])
cur = {}


def get_data(offset, length, out):
    if offset + length > len(cur["s"]):
        return -1
    seg = cur["s"][offset:offset + length].astype(np.float32) / np.float32(32768)
    ctypes.memmove(out, seg.ctypes.data, 4 * length)
    return 0


cb = pkg.GET_DATA_FN(get_data)
for i, c in enumerate(clips):                                  # run_classifier, one window per call
    cur["s"] = c
    sig, res = pkg.Signal(cb, 16000), Result()
    rc = L.run_classifier(ctypes.byref(sig), ctypes.byref(res), False)
    got = np.float32([res.classification[j].value for j in range(gm.n_labels)])
    assert rc == 0 and (bits(got) == bits(g["scores"][i])).all(), ("run_classifier", i, got, g["scores"][i])
audio = o.synth(int(g["cont_audio_seed"]), 0, 3).reshape(-1)
L.run_classifier_init()
for k in range(len(g["cont_produced"])):                        # run_classifier_continuous, 250 ms slices
    cur["s"] = audio[k * 4000:(k + 1) * 4000]
    sig, res = pkg.Signal(cb, 4000), Result()
    rc = L.run_classifier_continuous(ctypes.byref(sig), ctypes.byref(res), False)
    assert rc == 0, (k, rc)
    produced = bool(res.classification[0].label)
    assert produced == bool(g["cont_produced"][k]), k
    if produced:
        got = np.float32([res.classification[j].value for j in range(gm.n_labels)])
        assert (bits(got) == bits(g["cont_scores"][k])).all(), ("continuous", k, got, g["cont_scores"][k])
print("mfe sdk worker: %d windows, %d slices OK" % (len(clips), len(g["cont_produced"])))
