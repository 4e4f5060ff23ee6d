#!/usr/bin/env python3
"""Development measurement: which part of kws_fast_kernel the LDS bank conflicts come from.

PMC counters are per kernel, not per phase; but three forms of the kernel that the library launches anyway are nested prefixes of the
headline form, on the same DSP shape (fft 256, 40 mel filters, 49 frames):
    mfe   kws_fast_kernel<..., MFE>        load -> FFT -> power -> mel energies                    (an MFE-block model in fast mode)
    feat  kws_fast_kernel<..., NET=false>  ... + log, DCT, cmvnw -> features                       (scores AND features wanted: the feature-emitting form runs)
    full  kws_fast_kernel<..., NET=true>   ... + the fused float network -> scores                 (the headline launch)
so the differences of their per-clip counters attribute LDS work and bank-conflict cycles to {spectral prefix, DCT + cmvnw, network}.

    (GPU box)  tools/lds_attrib.sh <tag>                     one rocprofv3 --pmc pass per form -> gpurun_out/<tag>/
    (child)    python tools/lds_attrib.py run mfe|feat|full
    (anywhere) python tools/lds_attrib.py report <dir>       table from <dir>/{mfe,feat,full}_counter_collection.csv
"""
import collections
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = 65536
COUNTERS = ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES")


def run(form):
    sys.path.insert(0, ROOT)
    import torch
    from __graft_entry__ import load_package
    pkg = load_package()
    pcm = torch.empty((B, 16000), dtype=torch.int16, device="cuda:0")
    pkg.synth_clips_device(0, 0, B, 16000, pcm.data_ptr())
    if form == "mfe":
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from make_golden import MFE_MODEL_KW
        from kws_testlib import synth_model_blob
        m = pkg.Model(blob=synth_model_blob(**dict(MFE_MODEL_KW, num_filters=40, high=0, win_size=51, seed=78)))
    else:
        m = pkg.Model(os.path.join(ROOT, "models", "cfg2_mfcc40_f32.kwsm"), device=0)
    m.set_mode(pkg.MODE_FAST)
    s = torch.empty((B, m.n_labels), dtype=torch.float32, device="cuda:0")
    f = torch.empty((B, m.n_features), dtype=torch.float32, device="cuda:0") if form == "feat" else None
    for _ in range(3):
        m.run_classifier_batch_device(pcm.data_ptr(), B, s.data_ptr(), f.data_ptr() if f is not None else None)
    torch.cuda.synchronize()
    print("ran", form, m.n_features, "features; clips handed on by the fast tier:", m.fast_fallback_count(), flush=True)
    m.close()


def report(d):
    rows = {}
    for form in ("mfe", "feat", "full"):
        p = os.path.join(d, form + "_counter_collection.csv")
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(p)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        # template arguments <NZ, DG, PROF, FROM_CEP, NET, QCP, MFE>: the form this run is about
        def is_form(k):
            if not k.startswith("kws_fast_kernel<"):
                return False
            a = [x.strip() for x in k[k.index("<") + 1:k.rindex(">")].split(",")]
            net, mfe, from_cep = a[4] == "true", a[6] == "true", a[3] == "true"
            return not from_cep and {"mfe": mfe, "feat": not net and not mfe, "full": net}[form]
        fast = [k for k in acc if is_form(k)]
        print("# %s: kernels seen: %s" % (form, ", ".join("%s x%d" % (k, len(next(iter(acc[k].values())))) for k in sorted(acc) if "kws" in k)))
        if len(fast) != 1:
            print("# %s: expected one kws_fast_kernel form of this kind, found %s -- not attributed" % (form, fast))
            continue
        rows[form] = (fast[0], {c: sum(v) / len(v) for c, v in acc[fast[0]].items()})
    if len(rows) != 3:
        return 1
    print("\nper clip (counter summed over the device / %d clips), one launch of %d clips:" % (B, B))
    print("%-22s %14s %14s %14s   | %14s %14s %14s" % ("counter", "mfe (prefix)", "feat", "full", "spectral", "DCT + cmvnw", "network"))
    for c in COUNTERS:
        a, b, f = (rows[k][1].get(c, float("nan")) / B for k in ("mfe", "feat", "full"))
        print("%-22s %14.1f %14.1f %14.1f   | %14.1f %14.1f %14.1f" % (c, a, b, f, a, b - a, f - b))
    for k in ("mfe", "feat", "full"):
        v = rows[k][1]
        if v.get("SQ_LDS_IDX_ACTIVE"):
            print("%-5s %s: bank-conflict cycles / LDS index-active cycles = %.3f" % (k, rows[k][0][:70], v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"]))
    return 0


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "run":
        run(sys.argv[2])
    elif len(sys.argv) == 3 and sys.argv[1] == "report":
        sys.exit(report(sys.argv[2]))
    else:
        sys.exit(__doc__)
