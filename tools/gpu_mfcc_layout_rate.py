#!/usr/bin/env python3
"""Development aid: where kws_mfcc8_kernel starts to pay -- extract_mfcc_features (exact mode) for 65 536 windows of 8 .. 49 frames on
both spectral layouts (KWS_DEV_MFCC_OLD_LAYOUT / KWS_DEV_MFCC8_MIN_FRAMES), 32 x 13 and 40 x 40."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402
from kws_testlib import synth_model_blob  # noqa: E402

pkg = load_package()
B = 65536
for shape in (dict(num_filters=32, ncep=13), dict(num_filters=40, ncep=40, high=0)):
    for nfr in (8, 12, 16, 24, 32, 40, 49):
        n = 320 * (nfr + 1)
        gm = pkg.Model(blob=synth_model_blob(seed=1, raw_samples=n, blocks=((8, 3, 1), (4, 3, 1)), n_labels=3, **shape))
        pcm = torch.empty((B, n), dtype=torch.int16, device="cuda:0")
        pkg.synth_clips_device(0, 0, B, n, pcm.data_ptr())
        f = torch.empty((B, gm.n_features), dtype=torch.float32, device="cuda:0")
        ms = {}
        for tag in ("old", "new"):
            os.environ.pop("KWS_DEV_MFCC_OLD_LAYOUT", None)
            os.environ["KWS_DEV_MFCC8_MIN_FRAMES"] = "1"
            if tag == "old":
                os.environ["KWS_DEV_MFCC_OLD_LAYOUT"] = "1"
            for _ in range(3):
                gm.extract_mfcc_batch_device(pcm.data_ptr(), B, f.data_ptr())
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                gm.extract_mfcc_batch_device(pcm.data_ptr(), B, f.data_ptr())
            torch.cuda.synchronize()
            ms[tag] = (time.perf_counter() - t0) / 20 * 1e3
        print("%d filters x %d cepstra, %2d frames: kws_mfcc_kernel %.3f ms, kws_mfcc8_kernel %.3f ms (%+.1f %%)"
              % (shape["num_filters"], shape["ncep"], nfr, ms["old"], ms["new"], 100.0 * (ms["new"] / ms["old"] - 1.0)), flush=True)
        gm.close()
        del pcm, f
