#!/usr/bin/env python3
"""Where a kernel's scratch traffic sits: scratch loads / stores of an AMDGPU listing (hipcc -S) per enclosing loop (LLVM's "in Loop: Header=... Depth=..."
block comments), with the loop's global loads beside them -- a scratch reload shares vmcnt with the loop's prefetches and waits for them (in-order counter).
usage: isa_scratch_by_loop.py listing.s"""
import re, sys, collections
loops = collections.OrderedDict()
cur = ("-", 0)
for ln, l in enumerate(open(sys.argv[1]), 1):
    t = l.strip()
    m = re.match(r"^\.LBB\d+_\d+:\s*;(.*)$", t) or re.match(r"^; %bb\.\d+:\s*;?(.*)$", t)
    if re.match(r"^\.LBB\d+_\d+:", t) or t.startswith("; %bb."):
        h = re.search(r"Header[:=]\s*(\S+)", t); d = re.search(r"Depth=(\d+)", t)
        if "Loop Header" in t:
            cur = (t.split(":")[0], int(d.group(1)) if d else 0)
        elif h:
            cur = (h.group(1), int(d.group(1)) if d else 0)
        elif not t.startswith("; %bb.") or True:
            if not h and "Loop" not in t: cur = ("-", 0)
        continue
    if not t or t[0] in ";.": continue
    op = t.split()[0]
    e = loops.setdefault(cur, collections.Counter(first=ln))
    e["instr"] += 1
    if op.startswith("scratch_load"): e["sc_ld"] += 1
    elif op.startswith("scratch_store"): e["sc_st"] += 1
    elif op.startswith("global_load"): e["gl_ld"] += 1
    elif op.startswith("v_mfma"): e["mfma"] += 1
for (h, d), e in loops.items():
    if e["sc_ld"] or e["sc_st"]:
        print("%-12s depth %d  line %6d  instr %5d  scratch ld %3d st %3d  global ld %3d  mfma %3d" % (h, d, e["first"], e["instr"], e["sc_ld"], e["sc_st"], e["gl_ld"], e["mfma"]))
