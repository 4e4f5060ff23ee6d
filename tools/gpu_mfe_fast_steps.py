"""Development aid: N KWS_MODE_FAST steps of an MFE-block model (the spectral prefix of the fast kernel, kws_fast_kernel<..., MFE>) for a kernel trace:
    KWS_LIB=<library> rocprofv3 --kernel-trace --stats -- python tools/gpu_mfe_fast_steps.py [filters] [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from __graft_entry__ import load_package
from make_golden import MFE_MODEL_KW
from kws_testlib import synth_model_blob
pkg = load_package()
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 40
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
B = 65536
pcm = torch.empty((B, 16000), dtype=torch.int16, device="cuda:0")
pkg.synth_clips_device(0, 0, B, 16000, pcm.data_ptr())
kw = MFE_MODEL_KW if nf == 32 else dict(MFE_MODEL_KW, num_filters=40, high=0, win_size=51, seed=78)
gm = pkg.Model(blob=synth_model_blob(**kw))
gm.set_mode(pkg.MODE_FAST)
s = torch.empty((B, gm.n_labels), dtype=torch.float32, device="cuda:0")
for _ in range(steps):
    gm.run_classifier_batch_device(pcm.data_ptr(), B, s.data_ptr())
torch.cuda.synchronize()
print("done", gm.n_filters, float(s.sum()))
