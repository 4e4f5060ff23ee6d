#!/usr/bin/env python3
"""HBM traffic per kernel launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; --output-format csv).

usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <batch> <out.json> [model file name]

Corrections (MI355X_MICROARCH.md, HBM / rocprofv3 section): the counters report KiB; on gfx950 FETCH_SIZE counts half
of the bytes of 16-byte-per-lane coalesced streaming reads, so it is doubled.  WRITE_SIZE is checked against
kws_synth_kernel, which writes exactly batch * 32000 bytes.
"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        short = name.split("(")[0].split("<")[0].replace("void ", "").strip()
        acc[short].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    fetch, write, batch, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    model = sys.argv[5] if len(sys.argv) > 5 else "cfg2_mfcc40_f32.kwsm"
    f, w = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    res = {"command": "rocprofv3 --pmc <C> --kernel-trace --output-format csv -- python bench.py --steps 3 --warmup 1 "
                      "--no-cpu-baseline --no-also  (one pass per counter: FETCH_SIZE, WRITE_SIZE)",
           "batch": batch, "model": model, "unit": "bytes per launch",
           "correction": "counter values are KiB; FETCH_SIZE doubled (gfx950, 16-byte/lane coalesced streaming reads); "
                         "WRITE_SIZE checked on kws_synth_kernel (batch*32000 B written)",
           "kernels": {}}
    for k in sorted(set(f) | set(w)):
        if not k.startswith("kws"):
            continue
        rd, wr = int(f.get(k, 0.0) * 1024 * 2), int(w.get(k, 0.0) * 1024)
        res["kernels"][k] = {"FETCH_SIZE_KiB_avg": f.get(k, 0.0), "WRITE_SIZE_KiB_avg": w.get(k, 0.0),
                             "hbm_read_bytes": rd, "hbm_write_bytes": wr, "traffic_bytes": rd + wr}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res["kernels"], indent=1))


if __name__ == "__main__":
    main()
