"""Development aid: whole-call rate of kws_run_classifier_batch (host buffers in, scores out) from pageable and pinned memory."""
import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
import torch
from __graft_entry__ import load_package
pkg = load_package()
for name in ("cfg2_mfcc40_f32.kwsm", "l476_no_yes.kwsm"):
    gm = pkg.Model(os.path.join("models", name), device=0)
    B = 32768
    d = torch.empty((B, 16000), dtype=torch.int16, device="cuda:0")
    pkg.synth_clips_device(0, 0, B, 16000, d.data_ptr()); torch.cuda.synchronize()
    pinned = torch.empty((B, 16000), dtype=torch.int16).pin_memory(); pinned.copy_(d.cpu())
    pageable = pinned.numpy().copy()
    for label, arr in (("pageable numpy", pageable), ("pinned host", pinned.numpy())):
        gm.run_classifier_batch(arr[:1024])
        t0 = time.perf_counter(); s = gm.run_classifier_batch(arr); dt = time.perf_counter() - t0
        print("%s kws_run_classifier_batch from %s memory: %d clips in %.1f ms = %.2f M clips/s (%.1f GB/s of PCM)" % (name, label, B, dt * 1e3, B / dt / 1e6, B * 32000 / dt / 1e9), flush=True)
    gm.close()
