"""GPU diagnostic: where do HIP features differ from the oracle? (development aid)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from __graft_entry__ import load_package
from kws_testlib import *
pkg = load_package()
gm = pkg.Model(pkg.DEFAULT_MODEL)
o = Oracle(); om = OracleModel(o, os.path.join(MODELS, "l476_no_yes.kwsm"))
clips = o.synth(1, 0, 32)
s, f, q = gm.run_classifier_batch(clips, want_features=True)
so, fo, qo = om.run_batch(clips, want_features=True)
d = bits(f) != bits(fo)
print("mismatching words", d.sum(), "of", d.size, "clips affected", d.any(1).sum())
print("max abs diff", np.abs(f - fo).max(), "q mismatches", (q != qo).sum(), "score mismatches", (s != so).sum())
dd = d.reshape(32, 49, 13)
print("per coef", dd.sum((0, 1)))
print("per frame", dd.sum((0, 2)))
i = np.argwhere(dd)[:10]
for c, fr, k in i:
    print(c, fr, k, f[c, fr * 13 + k], fo[c, fr * 13 + k])
