#!/usr/bin/env python3
"""The numbers behind the guard's error constants (kws_fast_plan.cpp: kAlphaDct40 / kAlphaDct32 / kAlphaStale / kKappa*), from the per-clip arrays
tools/gpu_guard_study.py wrote on the GPU box with the guard switched off (no GPU needed here).

    python tools/guard_fit_report.py gpurun_out/<study>.npz > profiles/r06_guard_fit.txt

Per float graph: (1) per input family -- clips truly over 1e-4 with the guard off, the largest logit error in units of the product guard's sigma, what the
product guard handed on; (2) per cepstral column, pooled over the clips WITHOUT digitally silent frames -- rms / 95th / 99th percentile of a clip's
|feature - oracle| x window deviation / log-mel level (the quantity alpha_dct[c] x 1.3 is set from); (3) stale columns and column 0 per |window mean|.
"""
import sys

import numpy as np

FAMS = ("synth", "word_background", "word_noise_gain", "word_silence", "amp_sweep", "bursts", "quiet_noise", "clipped", "dc_tone", "pure_tone", "near_constant",
        "detuned_tone")


def main():
    z = np.load(sys.argv[1])
    models = sorted({k.split("/")[0] for k in z.files})
    np.set_printoptions(linewidth=220, precision=2, suppress=True)
    print("# %s  (tools/gpu_guard_study.py: guard off = development build with KWS_DEV_FAST_GUARD_SCALE=0, 2 048 clips per family, seed 5 = bench.py's also_inputs)" % sys.argv[1])
    for m in models:
        nc = z[m + "/gain"].size
        NF = 40 if nc == 40 else 32
        print("\n== %s  (%d columns, %d filters; gain per column %.3f .. %.3f, sigma_net %.3g)" % (m, nc, NF, z[m + "/gain"].min(), z[m + "/gain"].max(), float(z[m + "/sigma_net"])))
        print("  %-16s %9s %9s %10s %9s %11s %11s %12s %12s" % ("family", "handed on", "exact", "truly>1e-4", "precision", "max ds off", "max ds on", "max dz/sigma", "p99 dz/sigma"))
        pooled = []
        for fam in FAMS:
            key = "%s/%s/" % (m, fam)
            if key + "ds" not in z.files:
                continue
            g = lambda k: z[key + k]           # noqa: E731
            on = g("on")
            over = int((g("ds") > 1e-4).sum())
            r = g("dzp") / np.sqrt(g("v_hi"))
            print("  %-16s %9d %9d %10d %9s %11.3g %11.3g %12.3g %12.3g" % (fam, on[0], on[1], over, ("%.3f" % (over / on[0])) if on[0] else "-", np.nanmax(g("ds")),
                                                                          np.nanmax(g("ds_on")), np.nanmax(r), np.nanquantile(r, 0.99)))
            live = g("sil") == 0
            ec, sd, lvl = g("ecep").astype(np.float64), g("sd_mean"), g("lvl").astype(np.float64)
            rr = ec / lvl[:, None]
            rr[sd < 1e-3] = np.nan
            pooled.append((fam, rr[live], (ec / np.maximum(g("m_abs"), 1e-30))[live], g("m_abs")[live], lvl[live]))
        R = np.concatenate([p[1] for p in pooled])
        print("  -- per column, clips without silent frames, all families pooled (%d clips): |error| x deviation / level, x 1e7" % len(R))
        print("     rms   ", np.sqrt(np.nanmean(R ** 2, axis=0)) * 1e7)
        print("     p95   ", np.nanquantile(R, 0.95, axis=0) * 1e7)
        print("     p99   ", np.nanquantile(R, 0.99, axis=0) * 1e7)
        print("     rms x 1.3 of the DCT outputs 1 .. %d (the table in kws_fast_plan.cpp):" % (min(NF // 2, nc - 1)), np.sqrt(np.nanmean(R ** 2, axis=0))[1:NF // 2 + 1] * 1.3e7)
        print("  -- per family: stale columns (c > NF/2) and column 0, a clip's rms |error| x deviation per |window mean|, x 1e7: median / p99")
        for fam, _, em, ma, lvl in pooled:
            if not len(em):
                continue
            st = ("stale %.2f / %.2f  (|mean| median %.2f = %.2f x level)" % (np.median(em[:, NF // 2 + 1:]) * 1e7, np.quantile(em[:, NF // 2 + 1:], 0.99) * 1e7,
                                                                             np.median(ma[:, NF // 2 + 1:]), np.median(ma[:, NF // 2 + 1:]) / np.median(lvl))) if nc > NF // 2 + 1 else ""
            print("     %-16s column 0 %.2f / %.2f  (|mean| median %.1f)   %s" % (fam, np.median(em[:, 0]) * 1e7, np.quantile(em[:, 0], 0.99) * 1e7, np.median(ma[:, 0]), st))


if __name__ == "__main__":
    main()
