#!/usr/bin/env python3
"""One general-shape DSP case run a few times (for a counter pass or a kernel trace): python tools/gpu_generic_once.py [case substring] [clips] [calls]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    import torch
    from __graft_entry__ import load_package
    from kws_testlib import synth_model_blob
    from gpu_generic_rate import CASES
    want = sys.argv[1] if len(sys.argv) > 1 else "fft512 49"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    calls = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    pkg = load_package()
    for name, kw in CASES.items():
        if want not in name:
            continue
        gm = pkg.Model(blob=synth_model_blob(seed=3, **dict(dict(blocks=((8, 3, 7), (4, 3, 7)), n_labels=3), **kw)))
        pcm = torch.empty((n, gm.clip_samples), dtype=torch.int16, device="cuda:0")
        pkg.synth_clips_device(0, 0, n, gm.clip_samples, pcm.data_ptr())
        mf = torch.zeros((n, gm.n_features), dtype=torch.float32, device="cuda:0")
        for _ in range(calls):
            gm.mfcc_batch_device(pcm.data_ptr(), n, mf.data_ptr())
        torch.cuda.synchronize()
        print(name, gm.mfcc_kernel, "ran", calls, "calls of", n, "clips")
        gm.close()


if __name__ == "__main__":
    main()
