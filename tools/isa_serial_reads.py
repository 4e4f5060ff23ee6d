#!/usr/bin/env python3
"""Static check of a kernel's ISA for SERIALISED READS: a memory read (ds_read*, global_load*, buffer_load*) that is followed within a few
instructions by a wait for ALL outstanding reads (s_waitcnt lgkmcnt(0) / vmcnt(0)), several times in a row.  That is what the compiler emits when
every request sits in a conditional block of its own ("if this trip has items: read"): each round trip to LDS or HBM is exposed instead of one per
batch.  Round 5 found two such places by their clocks first and by their ISA second -- the general-shape kernel's guarded sample loads (+ 46 %) and
the split of a convolution block's image in the fast kernel (34 % of the block) -- this script finds them from the ISA alone.

    python tools/isa_serial_reads.py <unit.hip> [kernel-name substring ...]      (compiles the unit for gfx950 with the library's flags, -g1 for lines)
Prints, per kernel, the chains of >= 4 read-then-full-wait pairs and the source line the chain starts at.  No GPU needed."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden", "-DKWS_BUILDING_LIBRARY", "-Wno-unused-function",
         "-S", "--cuda-device-only", "-g1"]
READ = re.compile(r"^(ds_read|global_load|buffer_load|scratch_load)")
FULL = re.compile(r"^s_waitcnt (lgkmcnt\(0\)|vmcnt\(0\))")
INS = re.compile(r"^\s+((?:v_|s_|ds_|global_|buffer_|scratch_)\S*.*)$")


def kernels(lines):
    name, body = None, []
    for ln in lines:
        m = re.match(r"^(_Z\w+):\s", ln)
        if m and name is None and "kernel" in m.group(1):
            name, body = m.group(1), []
            continue
        if name is not None:
            body.append(ln)
            if "s_endpgm" in ln:
                yield name, body
                name = None


def chains(body, gap=6, window=40, least=4):
    """read ... full wait within `gap` instructions = one exposed round trip; consecutive ones at most `window` instructions apart form a chain"""
    ins, loc, cur = [], [], ""
    for ln in body:
        if ".loc" in ln:
            m = re.search(r"; (\S+:\d+)", ln)
            cur = m.group(1) if m else cur
        m = INS.match(ln)
        if m:
            ins.append(m.group(1))
            loc.append(cur)
    exposed = []
    last_read = None
    for n, t in enumerate(ins):
        if READ.match(t):
            last_read = n
        elif FULL.match(t) and last_read is not None and n - last_read <= gap:
            exposed.append(last_read)
            last_read = None
    out, run = [], []
    for n in exposed:
        if run and n - run[-1] > window:
            if len(run) >= least:
                out.append((len(run), loc[run[0]]))
            run = []
        run.append(n)
    if len(run) >= least:
        out.append((len(run), loc[run[0]]))
    return len(ins), sorted(out, reverse=True)


def main():
    unit, want = sys.argv[1], sys.argv[2:]
    extra = ["-fno-slp-vectorize"] if os.path.basename(unit) in ("kws_mfcc.hip", "kws_fast.hip") else []
    with tempfile.TemporaryDirectory() as d:
        s = os.path.join(d, "unit.s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-o", s, unit], check=True, stderr=subprocess.DEVNULL)
        lines = open(s).read().splitlines()
    for name, body in kernels(lines):
        dem = subprocess.run(["c++filt", name], stdout=subprocess.PIPE, text=True).stdout.strip().split("(")[0].replace("void ", "")
        if want and not any(w in dem for w in want):
            continue
        n, ch = chains(body)
        print("%-70s %6d instructions; serialised-read chains (length @ line): %s" % (dem[:70], n, ", ".join("%d @ %s" % c for c in ch[:8]) or "none"))


if __name__ == "__main__":
    main()
