"""Development aid: throughput of kws_extract_mfe_batch_device (the MFE DSP block) for 65 536 clips resident in HBM."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_package
pkg = load_package()
B = 65536
pcm = torch.empty((B, 16000), dtype=torch.int16, device="cuda:0")
pkg.synth_clips_device(0, 0, B, 16000, pcm.data_ptr())
for name in ("l476_no_yes.kwsm", "cfg2_mfcc40_f32.kwsm"):
    gm = pkg.Model(os.path.join(ROOT, "models", name), device=0)
    out = torch.empty((B, 49 * gm.n_filters), dtype=torch.float32, device="cuda:0")
    for _ in range(3):
        gm.extract_mfe_batch_device(pcm.data_ptr(), B, out.data_ptr())
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(10):
        gm.extract_mfe_batch_device(pcm.data_ptr(), B, out.data_ptr())
    ev[1].record(); torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 10
    print("%s DSP settings (%d filters): extract_mfe_features for %d clips in %.3f ms = %.1f M clips/s" % (name, gm.n_filters, B, ms, B / ms / 1e3), flush=True)
    gm.close()

# a model whose DSP block IS the MFE block, both arithmetic modes: PCM -> scores
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from make_golden import MFE_MODEL_KW  # noqa: E402
from kws_testlib import synth_model_blob  # noqa: E402
for kw in (MFE_MODEL_KW, dict(MFE_MODEL_KW, num_filters=40, high=0, win_size=51, seed=78)):
    gm = pkg.Model(blob=synth_model_blob(**kw))
    s = torch.empty((B, gm.n_labels), dtype=torch.float32, device="cuda:0")
    for mode, tag in ((pkg.MODE_EXACT, "exact"), (pkg.MODE_FAST, "fast")):
        gm.set_mode(mode)
        for _ in range(3):
            gm.run_classifier_batch_device(pcm.data_ptr(), B, s.data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            gm.run_classifier_batch_device(pcm.data_ptr(), B, s.data_ptr())
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 50 * 1e3
        print("MFE-block model, %d filters, int8 graph, %s mode: %d clips in %.3f ms = %.1f M clips/s" % (gm.n_filters, tag, B, ms, B / ms / 1e3), flush=True)
    gm.close()
