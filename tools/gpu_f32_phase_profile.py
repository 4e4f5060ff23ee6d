"""Development aid: per-phase shader-clock breakdown of kws_nn_f32_kernel (wave 0 of workgroup 0, full-occupancy launch)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_package
pkg = load_package()
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "models", "cfg2_mfcc40_f32.kwsm")
m = pkg.Model(path)
B = 65536
f = torch.randn((B, m.n_features), dtype=torch.float32, device="cuda")
s = torch.empty((B, m.n_labels), dtype=torch.float32, device="cuda")
prof = torch.zeros(16, dtype=torch.int64, device="cuda")
L = pkg.lib()
L.kws_dev_set_f32_prof.argtypes = [ctypes.c_void_p]
L.kws_dev_set_f32_prof(prof.data_ptr())
for _ in range(2):
    m.run_inference_batch_device(f.data_ptr(), B, s.data_ptr())
    torch.cuda.synchronize()
L.kws_dev_set_f32_prof(None)
p = prof.cpu().numpy()
names = ["input"] + ["block %d" % i for i in range(8)] + ["fc+softmax"]
tot = p.sum()
print(os.path.basename(path), "total cycles of wave 0:", tot)
for n, v in zip(names, p):
    if v:
        print("%-12s %12d  %5.1f%%" % (n, v, 100.0 * v / tot))
