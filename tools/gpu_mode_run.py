"""Runs K calls of kws_run_classifier_batch_device in one mode over a resident batch (the command rocprofv3 wraps).
   python tools/gpu_mode_run.py <model.kwsm> <fast|exact> [steps] [batch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_package
pkg = load_package()
path, mode = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
B = int(sys.argv[4]) if len(sys.argv) > 4 else 65536
m = pkg.Model(path)
m.set_mode(pkg.MODE_FAST if mode == "fast" else pkg.MODE_EXACT)
pcm = torch.empty((B, 16000), dtype=torch.int16, device="cuda")
pkg.synth_clips_device(0, 0, B, 16000, pcm.data_ptr())
s = torch.empty((B, m.n_labels), dtype=torch.float32, device="cuda")
m.run_classifier_batch_device(pcm.data_ptr(), B, s.data_ptr())
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    m.run_classifier_batch_device(pcm.data_ptr(), B, s.data_ptr())
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print("%s %s: %.3f ms per call, %.2f M clips/s, checksum %.6f" % (os.path.basename(path), mode, dt * 1e3, B / dt / 1e6, float(s.double().sum())))
