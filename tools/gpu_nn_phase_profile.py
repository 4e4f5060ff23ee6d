"""Development aid: per-phase shader-clock breakdown of kws_nn_kernel (generic int8; wave 0 of workgroup 0)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_package
pkg = load_package()
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "models", "cfg5_dscnn_mfcc40_int8.kwsm")
m = pkg.Model(path)
B = 65536
q = torch.randint(-128, 128, (B, m.n_features), dtype=torch.int8, device="cuda")
s = torch.empty((B, m.n_labels), dtype=torch.float32, device="cuda")
prof = torch.zeros(16, dtype=torch.int64, device="cuda")
L = pkg.lib()
L.kws_dev_set_nn_prof.argtypes = [ctypes.c_void_p]
L.kws_dev_force_scalar_nn.argtypes = [ctypes.c_int]
L.kws_dev_force_scalar_nn(1)
L.kws_dev_set_nn_prof(prof.data_ptr())
for _ in range(2):
    m.nn_batch_device(q.data_ptr(), B, s.data_ptr())
    torch.cuda.synchronize()
L.kws_dev_set_nn_prof(None)
L.kws_dev_force_scalar_nn(0)
p = prof.cpu().numpy()
names = ["input"] + ["block %d" % i for i in range(8)] + ["fc+softmax"]
tot = p.sum()
print(os.path.basename(path), m.nn_kernel, "total cycles of wave 0:", tot)
for n, v in zip(names, p):
    if v:
        print("%-12s %12d  %5.1f%%" % (n, v, 100.0 * v / tot))
