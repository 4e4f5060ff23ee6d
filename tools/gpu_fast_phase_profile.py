"""Development aid: per-phase shader-clock breakdown of kws_fast_kernel (wave 0 of workgroup 0, full-occupancy launch).

    python tools/gpu_fast_phase_profile.py [model.kwsm] [clips = 65536] [input family of tests/kws_families.py; default: the bench's synthetic clips]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_package
pkg = load_package()
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "models", "cfg2_mfcc40_f32.kwsm")
m = pkg.Model(path)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
pcm = torch.empty((B, 16000), dtype=torch.int16, device="cuda")
pkg.synth_clips_device(0, 0, B, 16000, pcm.data_ptr())
if len(sys.argv) > 3:
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import kws_families
    base = torch.from_numpy(np.ascontiguousarray(kws_families.family(sys.argv[3], 2048, seed=5))).to("cuda")
    pcm = base.repeat((B + 2047) // 2048, 1)[:B].contiguous()
    print("input family:", sys.argv[3])
s = torch.empty((B, m.n_labels), dtype=torch.float32, device="cuda")
prof = torch.zeros(24, dtype=torch.int64, device="cuda")
L = pkg.lib()
L.kws_dev_fast_phase_profile.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
for _ in range(2):
    rc = L.kws_dev_fast_phase_profile(m.h, pcm.data_ptr(), B, s.data_ptr(), prof.data_ptr())
    torch.cuda.synchronize()
pa = prof.cpu().numpy()
p = pa[:12]
names = ["load+preemph", "fft", "split+power+energy", "mel+log", "dct", "cmvn", "conv block 0", "conv blocks 1+", "fc+softmax", "(block 0: k loop)", "(block 0: epilogue)", "(block 0: preamble)"]
tot = p[:9].sum()
tolr = m.fast_tolerance()
waves = int(os.environ.get("KWS_DEV_FAST_WAVES", tolr.get("fused_waves") or 8))
nclips = max(1, B // (256 * waves))
# (three waves per SIMD: the clips are dealt out by ticket, wave 0's count is only about B / waves in all -- read the shares, not the clocks per clip)
print(os.path.basename(path), "rc", rc, "waves per workgroup", waves, "(per SIMD: %s)" % tolr.get("fused_waves_per_simd"), "clips by wave 0 ~", nclips, "total cycles", tot, "per clip ~", tot / nclips)
for n, v in zip(names, p):
    if n.startswith("(") and (v <= 0 or v > tot):      # block 0's sub-phase slots: only the fp32-instruction form of the convolution writes them
        continue
    print("%-20s %12d  %5.1f%%  %8.0f cycles/clip" % (n, v, 100.0 * v / tot, v / nclips))
for b in range(1, 8):
    if pa[12 + b]:
        print("  block %d              %12d  %5.1f%%  %8.0f cycles/clip" % (b, pa[12 + b], 100.0 * pa[12 + b] / tot, pa[12 + b] / nclips))
if pa[19]:
    print("  (cmvn: column-0 means) %9d  %5.1f%%  %8.0f cycles/clip" % (pa[19], 100.0 * pa[19] / tot, pa[19] / nclips))
