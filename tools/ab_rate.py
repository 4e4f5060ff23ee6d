#!/usr/bin/env python3
"""Same-box A/B of builds of libkws_mi355x.so (the pool's box-to-box spread is larger than most single optimisations): every variant
ab_tmp/libkws_<name>.so is timed in a process of its own (KWS_LIB), the variants alternating `rounds` times.

    python tools/ab_rate.py old,new [rounds] [model,model,...] [mode]
    (child)  python tools/ab_rate.py --child model,model,... mode
Prints ms per 65 536-clip step per model and variant (median of the rounds)."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(models, mode):
    sys.path.insert(0, ROOT)
    import torch
    from __graft_entry__ import load_package
    pkg = load_package()
    B = 65536
    pcm = torch.empty((B, 16000), dtype=torch.int16, device="cuda:0")
    pkg.synth_clips_device(0, 0, B, 16000, pcm.data_ptr())
    for name in models:
        m = pkg.Model(os.path.join(ROOT, "models", name), device=0)
        m.set_mode(pkg.MODE_FAST if mode == "fast" else pkg.MODE_EXACT)
        s = torch.empty((B, m.n_labels), dtype=torch.float32, device="cuda:0")
        for _ in range(5):
            m.run_classifier_batch_device(pcm.data_ptr(), B, s.data_ptr())
        torch.cuda.synchronize()
        steps = 200
        t0 = time.perf_counter()
        for _ in range(steps):
            m.run_classifier_batch_device(pcm.data_ptr(), B, s.data_ptr())
        torch.cuda.synchronize()
        print("RATE %s %.4f" % (name, (time.perf_counter() - t0) / steps * 1e3), flush=True)
        m.close()


def main():
    if sys.argv[1] == "--child":
        return child(sys.argv[2].split(","), sys.argv[3])
    variants = sys.argv[1].split(",")
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    models = sys.argv[3] if len(sys.argv) > 3 else "cfg2_mfcc40_f32.kwsm,l476_no_yes.kwsm,l476_no_yes_f32.kwsm"
    mode = sys.argv[4] if len(sys.argv) > 4 else "fast"
    res = {}
    for _ in range(rounds):
        for v in variants:
            # a variant "name+VAR" / "name+VAR=value" runs libkws_name.so with that environment variable (development switches of the library)
            name, _, var = v.partition("+")
            env = dict(os.environ, KWS_LIB=os.path.join(ROOT, "ab_tmp", "libkws_%s.so" % name))
            if var:
                key, _, val = var.partition("=")
                env[key] = val or "1"
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", models, mode], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            if out.returncode != 0:
                print(v, "FAILED", out.stderr[-800:])
            for ln in out.stdout.splitlines():
                if ln.startswith("RATE"):
                    _, name, ms = ln.split()
                    res.setdefault((name, v), []).append(float(ms))
    for name in models.split(","):
        print(name, "  ".join("%s %.4f ms (%s)" % (v, sorted(res.get((name, v), [0]))[len(res.get((name, v), [0])) // 2], " ".join("%.3f" % x for x in res.get((name, v), []))) for v in variants))


if __name__ == "__main__":
    main()
