#!/usr/bin/env python3
"""GPU check of kws_mfcc8_kernel (exact mode on the eight-lanes-per-frame spectral layout) against kws_mfcc_kernel, stage by stage and
bit for bit: mel energies and frame energies (kws_mfe_batch_device), cepstra before cmvnw (kws_mfcc_batch_device), the feature matrix
and the int8 tensor (kws_extract_mfcc_batch_device).  The library picks the kernel per launch (development switch
KWS_DEV_MFCC_OLD_LAYOUT), so both run in one process on the same inputs.  A sample of the clips also goes through the C oracle.

    python tools/gpu_mfcc_layout_check.py [clips] [model,model,...]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from kws_families import FAMILIES, family  # noqa: E402
from kws_testlib import MODELS, Oracle, OracleModel  # noqa: E402


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32 if a.dtype == np.float32 else a.dtype)


def describe(tag, a, b, shape):
    """a, b: [n][rows * cols]; prints where they differ"""
    d = bits(a) != bits(b)
    n_bad = int(d.sum())
    print("    %-26s %s  (%d of %d values differ)" % (tag, "identical" if n_bad == 0 else "DIFFERENT", n_bad, d.size))
    if n_bad:
        d3 = d.reshape((len(a),) + shape)
        print("        clips with a difference: %d of %d; by row: %s" % (int(d3.any(axis=(1, 2)).sum()), len(a), d3.sum(axis=(0, 2)).tolist()))
        print("        by column: %s" % d3.sum(axis=(0, 1)).tolist())
        c, r, k = np.argwhere(d3)[0]
        print("        first: clip %d row %d col %d: %r vs %r" % (c, r, k, a.reshape(d3.shape)[c, r, k], b.reshape(d3.shape)[c, r, k]))
    return n_bad


def main():
    import torch
    from __graft_entry__ import load_package
    pkg = load_package()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    models = sys.argv[2].split(",") if len(sys.argv) > 2 else ["cfg2_mfcc40_f32.kwsm", "l476_no_yes.kwsm", "cfg5_dscnn_mfcc40_int8.kwsm"]
    dev = torch.device("cuda:0")
    oracle = Oracle()
    # the bench's clips + every adversarial family + silence / full scale / alternating extremes
    parts = [oracle.synth(3, 0, n // 2)]
    per = max(8, n // 2 // len(FAMILIES))
    parts += [family(f, per, seed=5) for f in FAMILIES if f not in ("word_silence", "word_background")]
    edge = np.zeros((4, 16000), np.int16)
    edge[1] = 32767
    edge[2] = -32768
    edge[3, ::2] = 32767
    edge[3, 1::2] = -32768
    host = np.concatenate(parts + [edge])
    B = len(host)
    pcm = torch.from_numpy(host).to(dev)
    total_bad = 0
    for name in models:
        path = os.path.join(MODELS, name)
        gm = pkg.Model(path, device=0)
        gm.set_mode(pkg.MODE_EXACT)
        nfr, nft = gm.n_frames, gm.n_features
        ncep, nf = nft // nfr, gm.n_filters
        print("== %s: %d clips, %d frames x %d cepstra, %d filters" % (name, B, nfr, ncep, nf))

        def run(old):
            if old:
                os.environ["KWS_DEV_MFCC_OLD_LAYOUT"] = "1"
            else:
                os.environ.pop("KWS_DEV_MFCC_OLD_LAYOUT", None)
            mel = torch.zeros((B, nfr * nf), dtype=torch.float32, device=dev)
            en = torch.zeros((B, nfr), dtype=torch.float32, device=dev)
            cep = torch.zeros((B, nft), dtype=torch.float32, device=dev)
            feat = torch.zeros((B, nft), dtype=torch.float32, device=dev)
            q = torch.zeros((B, nft), dtype=torch.int8, device=dev)
            gm.mfe_batch_device(pcm.data_ptr(), B, mel.data_ptr(), en.data_ptr())
            gm.mfcc_batch_device(pcm.data_ptr(), B, cep.data_ptr())
            gm.extract_mfcc_batch_device(pcm.data_ptr(), B, feat.data_ptr(), None if gm.is_float else q.data_ptr())
            torch.cuda.synchronize()
            os.environ.pop("KWS_DEV_MFCC_OLD_LAYOUT", None)
            return [x.cpu().numpy() for x in (en, mel, cep, feat, q)]
        new, old = run(False), run(True)
        print("  kernel: %s" % gm.mfcc_kernel)
        bad = describe("frame energies", new[0], old[0], (nfr, 1))
        bad += describe("mel energies", new[1], old[1], (nfr, nf))
        bad += describe("cepstra before cmvnw", new[2], old[2], (nfr, ncep))
        bad += describe("features", new[3], old[3], (nfr, ncep))
        if not gm.is_float:
            bad += describe("int8 tensor", new[4], old[4], (nfr, ncep))
        # the oracle on a sample (the old kernel has been held to it since round 1; this pins the new one directly)
        om = OracleModel(oracle, path)
        idx = np.unique(np.concatenate([np.arange(0, B, max(1, B // 192)), np.arange(B - 4, B)]))
        fo = np.stack([oracle.extract_mfcc(host[i], om.cfg) for i in idx]).reshape(len(idx), -1)
        bad += describe("features vs oracle (%d)" % len(idx), new[3][idx], fo.astype(np.float32), (nfr, ncep))
        total_bad += bad
        gm.close()
    print("RESULT: %s" % ("all identical" if total_bad == 0 else "%d differing values" % total_bad))
    return 1 if total_bad else 0


if __name__ == "__main__":
    sys.exit(main())
