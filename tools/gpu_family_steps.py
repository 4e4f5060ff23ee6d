#!/usr/bin/env python3
"""KWS_MODE_FAST steps of the headline graph over ONE input family of tests/kws_families.py (2 048 distinct clips tiled to 65 536, the batch bench.py's
also_inputs times) -- the command rocprofv3 --kernel-trace --stats is wrapped around to see what the tiers behind the fast kernel cost (round 6).

    python tools/gpu_family_steps.py <family> [steps = 50] [model = cfg2_mfcc40_f32.kwsm]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402
import kws_families  # noqa: E402

fam = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
name = sys.argv[3] if len(sys.argv) > 3 else "cfg2_mfcc40_f32.kwsm"
pkg = load_package()
B, NB = 65536, 2048
base = torch.from_numpy(np.ascontiguousarray(kws_families.family(fam, NB, seed=5))).to("cuda:0")
pcm = base.repeat(B // NB, 1).contiguous()
m = pkg.Model(os.path.join(ROOT, "models", name), device=0)
m.set_mode(pkg.MODE_FAST)
s = torch.empty((B, m.n_labels), dtype=torch.float32, device="cuda:0")
for _ in range(5):
    m.run_classifier_batch_device(pcm.data_ptr(), B, s.data_ptr())
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    m.run_classifier_batch_device(pcm.data_ptr(), B, s.data_ptr())
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
print("%s %s: %.4f ms per step (%.1f M clips/s), handed on %d, exact kernels %d of %d" % (name, fam, ms, B / ms / 1e3, m.fast_fallback_count(), m.fast_exact_count(), B))
