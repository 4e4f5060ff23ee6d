#!/usr/bin/env python3
"""Opt-in soak (MI355X), not collected by pytest: random GENERAL-SHAPE DSP configurations (random_general_spec below: fft 64 .. 1024 incl.
lengths that are not powers of two, 8 .. 64 filters, frame lengths and strides, cepstra, cmvnw windows, frequency ranges, 0.25 .. 2 s clips)
through the library's exact mode against the C oracle, features bit for bit.  The report is per kernel name: what matters is how many
configurations kws_spectral_lds_kernel (round 4) served.  (kws_testlib.random_dsp_spec stays inside the tuned shape: a first version of this
script used it and exercised kws_mfcc8_kernel only.)

    python tests/generic_soak.py [first_seed] [n_seeds] [clips_per_configuration]     (defaults 40 400 24; seeds 0..39 are the test suite's)
Prints one line per kernel name and a verdict; exit status 1 on any differing word."""
import collections
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from kws_testlib import L476_CONFIG, Oracle, bits, special_clips, synth_model_blob  # noqa: E402


def random_general_spec(seed):
    """A random MFCC configuration OUTSIDE the tuned kernels' shape most of the time: other fft lengths (powers of two and lengths whose half
    factors into 2 / 3 / 4 / 5), 8 .. 64 mel filters, other frame lengths / strides / window lengths, clips of 0.25 .. 2 s.  What the plan builder
    does not serve comes back as KWS_ERROR_UNSUPPORTED_MODEL and is counted.  Returns (MfccConfig overrides, synth_model_blob arguments, samples)."""
    rng = np.random.default_rng(91000 + seed)
    fft = int(rng.choice([64, 128, 256, 512, 512, 1024, 1024, 240, 320, 400, 480, 600, 640, 800, 1000]))
    nf = int(rng.choice([8, 10, 12, 16, 20, 24, 30, 32, 36, 40, 48, 50, 60, 64]))      # (26 -> a 13-point DCT half: a radix the restatement does not carry)
    ncep = int(rng.integers(2, nf + 1))
    flen = float(rng.choice([0.01, 0.016, 0.02, 0.025, 0.032, 0.04, 0.05]))
    fstr = float(rng.choice([0.01, 0.0125, 0.016, 0.02, 0.025, 0.03]))
    n = int(rng.choice([4000, 8000, 12000, 15999, 16000, 24000, 32000]))
    win = int(rng.choice([11, 21, 31, 51, 101, 151]))
    low = int(rng.choice([0, 100, 300]))
    high = int(rng.choice([0, 3800, 6000]))
    pre = float(rng.choice([0.98, 0.97, 0.0]))
    cfg_kw = dict(num_filters=nf, num_cepstral=ncep, win_size=win, low_frequency=low, high_frequency=high, fft_length=fft, frame_length=flen,
                  frame_stride=fstr, pre_cof=pre)
    blob_kw = dict(seed=500 + seed, num_filters=nf, ncep=ncep, win_size=win, low=low, high=high, fft_length=fft, frame_length=flen, frame_stride=fstr,
                   pre_cof=pre, raw_samples=n, blocks=((8, 3, 1), (4, 3, 1)), n_labels=3)
    return cfg_kw, blob_kw, n


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    n_seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    n_clips = int(sys.argv[3]) if len(sys.argv) > 3 else 24
    from __graft_entry__ import load_package
    pkg = load_package()
    oracle = Oracle()
    sp = special_clips()
    fixed = np.stack([sp["impulses"], sp["ramp"], sp["zeros"], sp["alternating_fullscale"]])
    by_kernel = collections.Counter()
    words = collections.Counter()
    refused = bad = skipped = 0
    for seed in range(first, first + n_seeds):
        cfg_kw, blob_kw, n = random_general_spec(seed)
        try:
            gm = pkg.Model(blob=synth_model_blob(**blob_kw))
        except pkg.KwsError as e:
            if e.code != -18:
                print("seed", seed, "unexpected error", e.code, cfg_kw, flush=True)
                bad += 1
            refused += 1
            continue
        except Exception as e:                                   # the model synthesiser's own limits (e.g. a window with no frame)
            skipped += 1
            continue
        cfg = L476_CONFIG().copy(**cfg_kw)
        rnd = oracle.synth(1000 + seed, 0, 2 * (n_clips - len(fixed))).reshape(n_clips - len(fixed), 32000)[:, :n]
        fx = np.concatenate([fixed, fixed], axis=1)[:, :n]
        clips = np.ascontiguousarray(np.concatenate([rnd, fx]))
        _, f, _ = gm.run_classifier_batch(clips, want_features=True)
        name = gm.mfcc_kernel() if callable(gm.mfcc_kernel) else gm.mfcc_kernel
        for i, c in enumerate(clips):
            want = oracle.extract_mfcc(c, cfg)
            d = int((bits(f[i]) != bits(want)).sum())
            if d:
                bad += 1
                print("MISMATCH seed", seed, "clip", i, d, "of", want.size, "words", name, cfg_kw, flush=True)
            words[name] += want.size
        by_kernel[name] += 1
        gm.close()
    for k in sorted(by_kernel):
        print("%-32s %4d configurations  %12d feature words compared" % (k, by_kernel[k], words[k]))
    print("seeds %d..%d: %d configurations served, %d refused (KWS_ERROR_UNSUPPORTED_MODEL), %d not synthesisable, %d clips each, differing clips: %d"
          % (first, first + n_seeds - 1, sum(by_kernel.values()), refused, skipped, n_clips, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
