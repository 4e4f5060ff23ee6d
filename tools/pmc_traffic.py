#!/usr/bin/env python3
"""HBM traffic per kernel launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; --output-format csv).

usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <batch> <out.json> [model file name] [mode: fast | exact]

Corrections (MI355X_MICROARCH.md, HBM / rocprofv3 section): the counters report KiB; on gfx950 FETCH_SIZE counts half
of the bytes of 16-byte-per-lane coalesced streaming reads, so it is doubled.  WRITE_SIZE is checked against
kws_synth_kernel, which writes exactly batch * 32000 bytes.
"""
import collections
import csv
import hashlib
import json
import os
import sys


def per_kernel(path, counter):
    """average counter value per launch, per INSTANTIATION (a step launches several instantiations of a kernel template -- the hot one and
    the usually empty re-run forms -- which must not be averaged together); reported under the template's name: the instantiation that
    moves the most"""
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        full = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        acc[full].append(float(r["Counter_Value"]))
    best = {}
    for full, v in acc.items():
        short, mean = full.split("<")[0].strip(), sum(v) / len(v)
        if short not in best or mean > best[short][0]:
            best[short] = (mean, full)
    return {k: m for k, (m, _) in best.items()}, {k: f for k, (_, f) in best.items()}


def main():
    fetch, write, batch, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    model = sys.argv[5] if len(sys.argv) > 5 else "cfg2_mfcc40_f32.kwsm"
    mode = sys.argv[6] if len(sys.argv) > 6 else "fast"
    (f, f_inst), (w, _) = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ei-keyword-spotting_amd", "libkws_mi355x.so")
    res = {"lib_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest(),
           "command": "rocprofv3 --pmc <C> --kernel-trace --output-format csv -- python bench.py --model models/%s --mode %s --steps 3 --warmup 1 "
                      "--no-cpu-baseline --no-also  (one pass per counter: FETCH_SIZE, WRITE_SIZE)" % (model, mode),
           "batch": batch, "model": model, "mode": mode, "unit": "bytes per launch",
           "correction": "counter values are KiB; FETCH_SIZE doubled (gfx950, 16-byte/lane coalesced streaming reads); "
                         "WRITE_SIZE checked on kws_synth_kernel (batch*32000 B written)",
           "kernels": {}}
    for k in sorted(set(f) | set(w)):
        if not k.startswith("kws"):
            continue
        rd, wr = int(f.get(k, 0.0) * 1024 * 2), int(w.get(k, 0.0) * 1024)
        res["kernels"][k] = {"instantiation": f_inst.get(k, k), "FETCH_SIZE_KiB_avg": f.get(k, 0.0), "WRITE_SIZE_KiB_avg": w.get(k, 0.0),
                             "hbm_read_bytes": rd, "hbm_write_bytes": wr, "traffic_bytes": rd + wr}
    # SQ counters of the same library, if tools/pmc_sets.sh wrote a summary next to the output
    sq = os.path.join(os.path.dirname(os.path.abspath(out)), "summary.json")
    if os.path.exists(sq):
        res["sq"] = {}
        for name, cs in sorted(json.load(open(sq)).items(), key=lambda kv: kv[1].get("SQ_INSTS_VALU", 0.0)):      # the busiest instantiation last: it wins
            short = name.split("<")[0].strip()
            if "SQ_INSTS_VALU" in cs and "SQ_WAVE_CYCLES" in cs:
                res["sq"][short] = {"VALU_instructions_per_clip": round(cs["SQ_INSTS_VALU"] / batch, 1),
                                    "MFMA_instructions_per_clip": round(cs.get("SQ_INSTS_MFMA", 0.0) / batch, 1),
                                    "LDS_instructions_per_clip": round(cs.get("SQ_INSTS_LDS", 0.0) / batch, 1),
                                    "SALU_instructions_per_clip": round(cs.get("SQ_INSTS_SALU", 0.0) / batch, 1),
                                    "VALU_active_share_of_wave_cycles": round(cs.get("SQ_ACTIVE_INST_VALU", 0.0) / cs["SQ_WAVE_CYCLES"], 4),
                                    "wait_share_of_wave_cycles": round(cs.get("SQ_WAIT_ANY", 0.0) / cs["SQ_WAVE_CYCLES"], 4),
                                    "issue_stall_share_of_wave_cycles": round(cs.get("SQ_WAIT_INST_ANY", 0.0) / cs["SQ_WAVE_CYCLES"], 4),
                                    "MFMA_busy_cycles_per_clip": round(cs.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / batch, 1),
                                    "LDS_bank_conflict_share": round(cs.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(1.0, cs.get("SQ_LDS_IDX_ACTIVE", 1.0)), 4),
                                    "source": "rocprofv3 --pmc, one pass per counter set (tools/pmc_sets.sh)"}
                # fractions of the pipes' ceilings over the launch (bench.py: roofline.compute).  GRBM_GUI_ACTIVE is summed over the 8 XCDs: / 8 = the
                # launch's shader cycles; 256 CUs x 4 SIMDs; a plain fp32 vector instruction occupies a SIMD's issue for 2.34 cycles (1.05e12 wave-instructions/s
                # at 2.4 GHz, tools/ubench/valu_throughput.hip); SQ_VALU_MFMA_BUSY_CYCLES is in cycles per SIMD; SQ_LDS_IDX_ACTIVE in cycles per CU.
                if cs.get("GRBM_GUI_ACTIVE"):
                    cyc = cs["GRBM_GUI_ACTIVE"] / 8.0
                    valu_plain = cs["SQ_INSTS_VALU"] - cs.get("SQ_INSTS_MFMA", 0.0)
                    res["sq"][short]["compute"] = {
                        "valu_issue_frac": round(valu_plain * 2.34 / (1024.0 * cyc), 4),
                        "valu_active_frac": round(cs.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / (1024.0 * cyc), 4),
                        "mfma_busy_frac": round(cs.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * cyc), 4),
                        "lds_busy_frac": round(cs.get("SQ_LDS_IDX_ACTIVE", 0.0) / (256.0 * cyc), 4),
                        "waves_per_simd": round(cs["SQ_WAVES"] / 1024.0, 2) if cs.get("SQ_WAVES") else None,
                        "shader_cycles_per_launch": round(cyc),
                        "ceilings": "vector ALU: 1.05e12 plain fp32 wave-instructions/s = 2.34 cycles per instruction per SIMD (tools/ubench/valu_throughput.hip, measured on MI355X); "
                                    "valu_active_frac = SQ_ACTIVE_INST_VALU (quad-cycles a wave has a vector instruction executing, summed over waves) x 4 / (1024 SIMDs x launch cycles): the pipe's occupancy whatever the instruction mix; matrix pipe: busy cycles / (1024 SIMDs x launch cycles); LDS: index-active cycles / (256 CUs x launch cycles); launch cycles = GRBM_GUI_ACTIVE / 8 XCDs"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res["kernels"], indent=1))


if __name__ == "__main__":
    main()
