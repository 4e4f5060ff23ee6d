"""Development aid: throughput of kws_streams_step_device -- S audio streams in continuous mode (250 ms slices), state in HBM."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_package
pkg = load_package()
for name in (sys.argv[1:] or ["l476_no_yes.kwsm", "cfg2_mfcc40_f32.kwsm"]):
  for mode in (pkg.MODE_EXACT, pkg.MODE_FAST):
    gm = pkg.Model(os.path.join(ROOT, "models", name), device=0)
    gm.set_mode(mode)
    for S in (4096, 65536):
        sb = pkg.StreamBatch(gm, S)
        n_sl = 8
        audio = torch.empty((S, n_sl * 4000), dtype=torch.int16, device="cuda:0")
        pkg.synth_clips_device(3, 0, S, n_sl * 4000, audio.data_ptr())
        slices = [audio[:, k * 4000:(k + 1) * 4000].contiguous() for k in range(n_sl)]
        scores = torch.empty((S, gm.n_labels), dtype=torch.float32, device="cuda:0")
        for k in range(5):                                       # fill the rolling feature buffers (4 slices per window)
            sb.step_device(slices[k % n_sl].data_ptr(), 4000, scores.data_ptr())
        torch.cuda.synchronize()
        steps = 20
        t0 = time.perf_counter()
        for k in range(steps):
            sb.step_device(slices[(5 + k) % n_sl].data_ptr(), 4000, scores.data_ptr())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        print("%s [%s]: %d streams, %.3f ms per 250 ms slice step = %.1f M slices/s = %.0f x real time per stream-second (%.0f k concurrent real-time streams)"
              % (name, "fast" if mode else "exact", S, dt * 1e3, S / dt / 1e6, 0.25 / dt, S * 0.25 / dt / 1e3), flush=True)
        sb.close()
    gm.close()
