#!/usr/bin/env python3
"""Development aid: shader clock and socket power while the hot path runs back to back (rocm-smi sampled twice a second from a thread).
    python tools/gpu_clock_watch.py [model] [mode] [seconds]"""
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from __graft_entry__ import load_package
    pkg = load_package()
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg2_mfcc40_f32.kwsm"
    mode = sys.argv[2] if len(sys.argv) > 2 else "fast"
    secs = float(sys.argv[3]) if len(sys.argv) > 3 else 8.0
    B = 65536
    pcm = torch.empty((B, 16000), dtype=torch.int16, device="cuda:0")
    pkg.synth_clips_device(0, 0, B, 16000, pcm.data_ptr())
    m = pkg.Model(os.path.join(ROOT, "models", name), device=0)
    m.set_mode(pkg.MODE_FAST if mode == "fast" else pkg.MODE_EXACT)
    s = torch.empty((B, m.n_labels), dtype=torch.float32, device="cuda:0")
    stop = False
    rows = []

    def watch():
        while not stop:
            out = subprocess.run(["/opt/rocm/bin/rocm-smi", "-d", "0", "--showclocks", "--showpower", "--showtemp"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
            keep = [ln.strip() for ln in out.splitlines() if any(k in ln for k in ("sclk", "mclk", "Power", "junction", "fclk"))]
            rows.append((time.time(), keep))
            time.sleep(0.4)
    print("idle:")
    out = subprocess.run(["/opt/rocm/bin/rocm-smi", "-d", "0", "--showclocks", "--showpower"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
    print("\n".join(ln for ln in out.splitlines() if any(k in ln for k in ("sclk", "Power"))))
    th = threading.Thread(target=watch)
    th.start()
    t0 = time.time()
    n = 0
    while time.time() - t0 < secs:
        for _ in range(200):
            m.run_classifier_batch_device(pcm.data_ptr(), B, s.data_ptr())
        torch.cuda.synchronize()
        n += 200
    dt = time.time() - t0
    stop = True
    th.join()
    print("%s %s: %.4f ms per step over %.1f s" % (name, mode, dt / n * 1e3, dt))
    for t, keep in rows:
        print("%.1fs  %s" % (t - t0, " | ".join(keep)))


if __name__ == "__main__":
    main()
