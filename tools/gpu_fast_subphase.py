#!/usr/bin/env python3
"""Development measurement: shader clocks of wave 0 of workgroup 0 inside the split-operand convolution of the headline graph -- image split,
contraction loop, epilogue -- read from a scratch build of the library that carries the clock reads (KWS_LIB; built from a patched copy of
csrc/kws_fast.hip by tools/round5/make_subprof_lib.sh, not part of the product).

    KWS_LIB=ab_tmp/libkws_subprof.so python tools/gpu_fast_subphase.py [steps] [waves per workgroup = 8]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from __graft_entry__ import load_package
    pkg = load_package()
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    B = 65536
    pcm = torch.empty((B, 16000), dtype=torch.int16, device="cuda:0")
    pkg.synth_clips_device(0, 0, B, 16000, pcm.data_ptr())
    m = pkg.Model(os.path.join(ROOT, "models", "cfg2_mfcc40_f32.kwsm"), device=0)
    m.set_mode(pkg.MODE_FAST)
    s = torch.empty((B, m.n_labels), dtype=torch.float32, device="cuda:0")
    for _ in range(5):
        m.run_classifier_batch_device(pcm.data_ptr(), B, s.data_ptr())
    torch.cuda.synchronize()
    L = pkg.lib()
    buf = (ctypes.c_longlong * 8)()
    L.kws_dev_fast_sub(buf)                                        # clear
    for _ in range(steps):
        m.run_classifier_batch_device(pcm.data_ptr(), B, s.data_ptr())
    torch.cuda.synchronize()
    L.kws_dev_fast_sub(buf)
    waves = int(sys.argv[2]) if len(sys.argv) > 2 else 8           # waves per workgroup of the build (11 for the three-waves-per-SIMD build of the 49x40 graph)
    clips = steps * B / (256 * waves)                              # wave 0 of workgroup 0 sees 65 536 / (256 x waves) clips per step
    names = ("split of the image into halves", "contraction loop", "epilogue")
    for b in (0, 1):
        tot = sum(buf[4 * b + i] for i in range(3)) or 1
        print("conv block %d: " % b + ", ".join("%s %.0f clocks per clip (%.0f %%)" % (names[i], buf[4 * b + i] / clips, 100.0 * buf[4 * b + i] / tot) for i in range(3))
              + "; total %.0f" % (tot / clips))


if __name__ == "__main__":
    main()
