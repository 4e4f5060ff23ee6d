"""Development aid: time of the network kernels alone (int8 tensor in HBM -> scores) for 65 536 windows."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_package
pkg = load_package()
for name in (sys.argv[1:] or ["cfg5_dscnn_mfcc40_int8.kwsm", "l476_no_yes.kwsm", "cfg2_mfcc40_int8.kwsm"]):
    m = pkg.Model(os.path.join(ROOT, "models", name))
    B = 65536
    q = torch.randint(-128, 128, (B, m.n_features), dtype=torch.int8, device="cuda")
    s = torch.empty((B, m.n_labels), dtype=torch.float32, device="cuda")
    for _ in range(3):
        m.nn_batch_device(q.data_ptr(), B, s.data_ptr())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        m.nn_batch_device(q.data_ptr(), B, s.data_ptr())
    e1.record()
    torch.cuda.synchronize()
    print("%s (%s): %.3f ms per 65536 windows" % (name, m.nn_kernel, e0.elapsed_time(e1) / 20), flush=True)
    m.close()
