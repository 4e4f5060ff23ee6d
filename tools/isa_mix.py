#!/usr/bin/env python3
"""Static instruction mix of one kernel of an AMDGPU assembly listing (hipcc -S ... or --save-temps), per basic block, priced with
the per-wave issue costs measured by tools/ubench.  Development aid for the issue-bound kernels: it shows where
a wave's instruction slots go without a GPU.

usage: isa_mix.py listing.s kernel-name-substring [min-cycles]
"""
import re, sys
PRICE = [  # (regex, class, cycles of the issuing wave)
    (r"v_mfma_", "mfma", 31.5), (r"ds_write_b64|ds_write2", "lds", 24), (r"ds_read_b(64|96|128)|ds_read2", "lds", 11),
    (r"ds_(read|write|bpermute|swizzle|max|add|permute)", "lds", 15), (r"global_|flat_|buffer_|scratch_", "vmem", 20),
    (r"v_mad_i64_i32|v_mad_u64_u32", "valu", 9), (r"v_.*_dpp|v_mov_b32_dpp", "valu", 7), (r"v_pk_", "valu", 8),
    (r"v_.*f64", "valu", 9), (r"v_", "valu", 5), (r"s_waitcnt", "wait", 0), (r"s_", "salu", 4)]
def main():
    path, name = sys.argv[1], sys.argv[2]
    floor = float(sys.argv[3]) if len(sys.argv) > 3 else 150.0
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and name in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    blocks, cur = [], {"label": "entry", "depth": "", "n": {}, "cyc": 0.0, "line": start}
    for i in range(start + 1, end + 1):
        l = lines[i].strip()
        m = re.match(r"^(\.LBB\S+):\s*(;.*)?$", l)
        if m or l.startswith("; %bb."):
            blocks.append(cur)
            d = re.search(r"Depth=(\d+)", l); h = re.search(r"Header[:=]\s*(\S+)", l)
            cur = {"label": m.group(1) if m else l.split()[1], "depth": (h.group(1) if h else "") + (" d" + d.group(1) if d else ""),
                   "n": {}, "cyc": 0.0, "line": i}
            continue
        if not l or l.startswith(";") or l.startswith("."): continue
        op = l.split()[0]
        for rx, cls, c in PRICE:
            if re.match(rx, op):
                cur["n"][cls] = cur["n"].get(cls, 0) + 1; cur["cyc"] += c; break
    blocks.append(cur)
    tot = {}
    for b in blocks:
        for k, v in b["n"].items(): tot[k] = tot.get(k, 0) + v
    print("static totals:", tot)
    for b in blocks:
        if b["cyc"] >= floor:
            print(f"{b['line']:7d} {b['label']:14s} {b['depth']:18s} {b['cyc']:8.0f} cyc  " + " ".join(f"{k}={v}" for k, v in sorted(b["n"].items())))
main()
