#!/usr/bin/env python3
"""KWS_MODE_FAST against KWS_MODE_EXACT of the same library over MANY seeds: evidence for the statistical guard (the
tests hold one seed of 65 536 clips + ten families against the oracle).  The exact mode is pinned to the oracle bit for bit on the features and
within 1e-6 on float scores (tests/test_gpu_parity.py), so |fast - exact| <= 1e-4 - 1e-6 here implies the 1e-4 bar against the reference.

For every float model: S seeds x 65 536 clips of the bench's generator + every input family of tests/kws_families.py x F seeds x 2 048 clips.
Prints per model: clips, worst |score difference|, how many clips the fast tiers handed to the exact kernels, the distribution's tail.

    python tools/gpu_fast_sweep.py [S=32] [F=4]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


EDGES = np.array([1e-7, 1e-6, 1e-5, 2.5e-5, 5e-5, 1e-4])


class Stats:
    def __init__(self, pkg, name, B):
        import torch
        from kws_testlib import MODELS
        self.name, self.pkg = name, pkg
        self.gm = pkg.Model(os.path.join(MODELS, name), device=0)
        self.s_e = torch.zeros((B, self.gm.n_labels), dtype=torch.float32, device="cuda:0")
        self.s_f = torch.zeros((B, self.gm.n_labels), dtype=torch.float32, device="cuda:0")
        self.worst, self.n_clips, self.n_back = 0.0, 0, 0
        self.hist = np.zeros(8, dtype=np.int64)          # |d| < 1e-7, < 1e-6, < 1e-5, < 2.5e-5, < 5e-5, < 1e-4, >= 1e-4, nan

    def both(self, p, n, tag):
        import torch
        gm, pkg = self.gm, self.pkg
        gm.set_mode(pkg.MODE_EXACT)
        gm.run_classifier_batch_device(p.data_ptr(), n, self.s_e.data_ptr(), None)
        gm.set_mode(pkg.MODE_FAST)
        gm.run_classifier_batch_device(p.data_ptr(), n, self.s_f.data_ptr(), None)
        torch.cuda.synchronize()
        self.n_back += gm.fast_fallback_count()
        d = (self.s_f[:n] - self.s_e[:n]).abs().max(dim=1).values.cpu().numpy()
        self.hist[7] += int(np.isnan(d).sum())
        d = d[~np.isnan(d)]
        self.hist[:7] += np.bincount(np.searchsorted(EDGES, d, side="right"), minlength=7)
        w = float(d.max()) if len(d) else 0.0
        if w > self.worst:
            self.worst = w
            print("    %-26s %-28s worst so far %.3g" % (self.name, tag, w), flush=True)
        self.n_clips += n


def main():
    import torch
    import kws_families
    from __graft_entry__ import load_package
    pkg = load_package()
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    F = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    B, nf = 65536, 2048
    pcm = torch.empty((B, 16000), dtype=torch.int16, device="cuda:0")
    models = [Stats(pkg, name, B) for name in ("cfg2_mfcc40_f32.kwsm", "l476_no_yes_f32.kwsm", "cfg5_dscnn_mfcc40_f32.kwsm")]
    for seed in range(S):
        pkg.synth_clips_device(9000 + seed, 0, B, 16000, pcm.data_ptr())
        for m in models:
            m.both(pcm, B, "generator seed %d" % (9000 + seed))
    for fam in kws_families.FAMILIES:
        for fs in range(F):
            x = torch.from_numpy(np.ascontiguousarray(kws_families.family(fam, nf, seed=100 + fs))).to("cuda:0")
            for m in models:
                m.both(x, nf, "%s seed %d" % (fam, 100 + fs))
    bad = False
    for m in models:
        print("%s: %d clips (%d generator seeds x %d + %d families x %d seeds x %d): worst |fast - exact| score = %.3g; handed to the exact kernels: %d (%.2f %%); "
              "|d| histogram [<1e-7, <1e-6, <1e-5, <2.5e-5, <5e-5, <1e-4, >=1e-4, nan] = %s"
              % (m.name, m.n_clips, S, B, len(kws_families.FAMILIES), F, nf, m.worst, m.n_back, 100.0 * m.n_back / m.n_clips, m.hist.tolist()), flush=True)
        bad = bad or m.hist[6] != 0 or m.hist[7] != 0
        m.gm.close()
    if bad:
        raise SystemExit("a clip left the 1e-4 bar")


if __name__ == "__main__":
    main()
