#!/usr/bin/env python3
"""General-shape DSP configurations (fft 128 / 512 / 1024, 99 frames, ...): time per clip and per frame of the exact MFCC stage on the
LDS-resident cooperative kernel (kws_spectral_lds_kernel, round 4) against the round-1 kernel with its scratch in HBM
(KWS_DEV_GENERIC_SCRATCH=1, run in a child process) and against the tuned kernel's shape (fft 256, 49 frames).

    python tools/gpu_generic_rate.py [n_clips]
"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CASES = {
    "tuned fft256 49 frames (kws_mfcc8_kernel)": dict(),
    "fft512 49 frames": dict(fft_length=512),
    "fft128 49 frames": dict(fft_length=128, win_size=51),
    "fft256 stride 10 ms, 98 frames": dict(frame_stride=0.01, win_size=31, blocks=((8, 3, 1), (4, 3, 1))),
    "fft1024 50 ms frames, 36 filters": dict(fft_length=1024, num_filters=36, ncep=17, frame_length=0.05, frame_stride=0.025, win_size=21, blocks=((8, 3, 1), (4, 3, 1))),
    "fft512 2 s clips (no pooling)": dict(fft_length=512, raw_samples=32000, blocks=((8, 3, 1), (4, 3, 1))),
}


def child(n):
    import torch
    from __graft_entry__ import load_package
    from kws_testlib import synth_model_blob
    pkg = load_package()
    for name, kw in CASES.items():
        blob = synth_model_blob(seed=3, **dict(dict(blocks=((8, 3, 7), (4, 3, 7)), n_labels=3), **kw))
        try:
            gm = pkg.Model(blob=blob)
        except pkg.KwsError as e:
            print("SKIP|%s|%s" % (name, e), flush=True)
            continue
        ns = gm.clip_samples
        pcm = torch.empty((n, ns), dtype=torch.int16, device="cuda:0")
        pkg.synth_clips_device(0, 0, n, ns, pcm.data_ptr())
        mf = torch.zeros((n, gm.n_features), dtype=torch.float32, device="cuda:0")
        ft = torch.zeros((n, gm.n_features), dtype=torch.float32, device="cuda:0")
        res = []
        for fn in (lambda: gm.mfcc_batch_device(pcm.data_ptr(), n, mf.data_ptr()), lambda: gm.extract_mfcc_batch_device(pcm.data_ptr(), n, ft.data_ptr())):
            for _ in range(8):                                   # (the handle measures its chunk length on its first seven large calls)
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / reps)
        chunk = 0
        if hasattr(pkg.lib(), "kws_dev_generic_chunk"):
            import ctypes
            pkg.lib().kws_dev_generic_chunk.argtypes = [ctypes.c_void_p]
            chunk = pkg.lib().kws_dev_generic_chunk(gm.h)
        print("RATE|%s|%s|%d|%.6f|%.6f|%d" % (name, gm.mfcc_kernel, gm.n_frames, res[0], res[1], chunk), flush=True)
        if gm.mfcc_kernel == "kws_spectral_lds_kernel" and hasattr(pkg.lib(), "kws_dev_generic_prof"):
            import ctypes
            buf = (ctypes.c_longlong * 8)()
            pkg.lib().kws_dev_generic_prof(buf)                  # clear
            gm.mfcc_batch_device(pcm.data_ptr(), n, mf.data_ptr())
            torch.cuda.synchronize()
            pkg.lib().kws_dev_generic_prof(buf)
            tot = float(sum(buf)) or 1.0
            print("PROF|%s|" % name + " ".join("%s %.0f%%" % (nm, 100.0 * v / tot) for nm, v in zip(("load", "samples arrive", "levels", "split+power", "energy", "mel", "dct+out", "-"), buf)) + " | clocks per frame of workgroup 0: %.0f" % (tot / max(1, (n * ((gm.n_frames + 15) // 16) // 4096)) / 16), flush=True)
        gm.close()


VARIANTS = [
    # tag, environment (None: the shipped library, no switches, no phase clocks), library (None: the development build)
    ("product", None, None),
    ("auto", {}, None),
    ("auto, sample wait", {"KWS_DEV_GENERIC_PROF": "1"}, None),
    ("L4 deep", {"KWS_DEV_GENERIC_LCH": "4", "KWS_DEV_GENERIC_WPS": "2"}, None),
    ("L8 deep", {"KWS_DEV_GENERIC_LCH": "8", "KWS_DEV_GENERIC_WPS": "2"}, None),
    ("L4 shallow", {"KWS_DEV_GENERIC_LCH": "4", "KWS_DEV_GENERIC_WPS": "4"}, None),
    ("L8 shallow", {"KWS_DEV_GENERIC_LCH": "8", "KWS_DEV_GENERIC_WPS": "4"}, None),
    ("L4 deep, by sample", {"KWS_DEV_GENERIC_LCH": "4", "KWS_DEV_GENERIC_WPS": "2", "KWS_DEV_GENERIC_NOPAIRS": "1"}, None),
    ("L4 shallow, by sample", {"KWS_DEV_GENERIC_LCH": "4", "KWS_DEV_GENERIC_WPS": "4", "KWS_DEV_GENERIC_NOPAIRS": "1"}, None),
    ("L8 shallow, by sample", {"KWS_DEV_GENERIC_LCH": "8", "KWS_DEV_GENERIC_WPS": "4", "KWS_DEV_GENERIC_NOPAIRS": "1"}, None),
]
# experiment builds (tools/build_generic_variants.sh): libkws_var_<tag>.so = the four-wave build at other batch depths
import glob
for lib in sorted(glob.glob(os.path.join(ROOT, "ei-keyword-spotting_amd", "libkws_var_*.so"))):
    vt = os.path.basename(lib)[len("libkws_var_"):-3]
    for lch in ("4", "8"):
        VARIANTS.append(("L%s %s" % (lch, vt), {"KWS_DEV_GENERIC_LCH": lch, "KWS_DEV_GENERIC_WPS": "4"}, lib))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child(int(sys.argv[2]))
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    only = set(sys.argv[2].split(",")) if len(sys.argv) > 2 else None
    rows = {}
    dev_lib = os.path.join(ROOT, "ei-keyword-spotting_amd", "libkws_mi355x_dev.so")
    tags = []
    for tag, env, lib in VARIANTS:
        if only and tag not in only:
            continue
        tags.append(tag)
        cenv = dict(os.environ) if env is None else dict(os.environ, KWS_LIB=lib or dev_lib, **env)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(n)], env=cenv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        if out.returncode != 0:
            print(tag, "FAILED", out.stderr[-1500:])
        for ln in out.stdout.splitlines():
            if ln.startswith("PROF|") and tag.startswith("auto"):
                print(ln.replace("PROF|", "PROF %s|" % tag, 1))
            if ln.startswith("RATE|"):
                _, name, kern, nfr, t_spec, t_all, chunk = (ln.split("|") + ["0"])[:7]
                rows.setdefault(name, {})[tag] = (kern, int(nfr), float(t_spec), float(t_all), int(chunk))
    print("# %d clips per call; ms per call of speechpy::feature::mfcc (cepstra before cmvnw) | of extract_mfcc_features (with cmvnw)" % n)
    print("# product = the shipped library; auto = the development build without switches (the handle's measured chunk length in brackets); Lx = chunk length x;")
    print("# deep = the build for two waves per SIMD (4 butterflies per lane and round trip, 4 frames requested together), shallow = the build for four (1 and 2);")
    print("# by sample = 2-byte loads requested at the start of the sub-batch (round 4's way) instead of dword pairs requested one sub-batch ahead")
    for name, r in rows.items():
        kern, nfr = next(iter(r.values()))[:2]
        print("%s  (%d frames, %s)" % (name, nfr, kern))
        for tag in tags:
            if tag in r:
                kern, nfr, ts, ta, chunk = r[tag]
                print("    %-22s %8.3f | %8.3f ms   %6.2f ns per frame%s" % (tag, ts * 1e3, ta * 1e3, ts * 1e9 / (n * nfr), (" [chunk %d]" % chunk) if tag in ("product", "auto") else ""))


if __name__ == "__main__":
    main()
