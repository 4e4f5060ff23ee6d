#!/usr/bin/env python3
"""Occupancy facts of every kernel in a built libkws_mi355x.so, read from the code objects themselves (VERDICT round 3, item 6:
rocprofv3's kernel trace reports `vgpr_count` in its own units; DESIGN's occupancy arguments must stand on the code object's numbers).

The library carries one clang offload bundle per translation unit (`__CLANG_OFFLOAD_BUNDLE__`: entry table of offset / size / target
triple); the gfx950 entries are AMDGPU ELF code objects whose NT_AMDGPU_METADATA note lists, per kernel, `.vgpr_count` (architected
VGPRs), `.agpr_count` (accumulation VGPRs; on gfx950 both come out of one 512-entry file per SIMD lane), `.sgpr_count`,
`.group_segment_fixed_size` (static LDS), `.private_segment_fixed_size` (scratch bytes per lane) and the spill counts.  Waves per SIMD
from registers = floor(512 / ceil8(vgpr + agpr)), capped at 8.

usage: codeobj_meta.py [lib.so] [out.md]        (no GPU needed; uses /opt/rocm/lib/llvm/bin/llvm-readelf)
"""
import hashlib
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
CXXFILT = "c++filt"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(blob, arch="gfx950"):
    for m in re.finditer(MAGIC, blob):
        base = m.start()
        (n,) = struct.unpack_from("<Q", blob, base + len(MAGIC))
        pos = base + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, pos)
            triple = blob[pos + 24:pos + 24 + tlen].decode()
            pos += 24 + tlen
            if arch in triple and size:
                yield triple, blob[base + off:base + off + size]


def kernels_of(elf_bytes):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(elf_bytes)
        f.flush()
        notes = subprocess.run([READELF, "--notes", f.name], stdout=subprocess.PIPE, text=True, check=True).stdout
    out = []
    for chunk in re.split(r"\n\s*- \.agpr_count:", "\n" + notes)[1:]:
        chunk = ".agpr_count:" + chunk
        rec = {}
        for key in ("agpr_count", "vgpr_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size", "vgpr_spill_count",
                    "sgpr_spill_count", "max_flat_workgroup_size", "wavefront_size"):
            mm = re.search(r"\.%s:\s+(\d+)" % key, chunk)
            rec[key] = int(mm.group(1)) if mm else 0
        mm = re.search(r"\.name:\s+(\S+)", chunk)
        rec["name"] = mm.group(1) if mm else "?"
        out.append(rec)
    return out


def demangle(names):
    p = subprocess.run([CXXFILT], input="\n".join(names), stdout=subprocess.PIPE, text=True)
    return p.stdout.split("\n")[:len(names)] if p.returncode == 0 else names


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "ei-keyword-spotting_amd", "libkws_mi355x.so")
    out = sys.argv[2] if len(sys.argv) > 2 else None
    blob = open(lib, "rb").read()
    rows = []
    for triple, elf in code_objects(blob):
        rows += kernels_of(elf)
    pretty = demangle([r["name"] for r in rows])
    lines = ["# kernels of %s (sha256 %s), from the gfx950 code objects' NT_AMDGPU_METADATA" % (os.path.basename(lib), hashlib.sha256(blob).hexdigest()[:16]),
             "", "| kernel | VGPR | AGPR | VGPR+AGPR | waves/SIMD (registers) | SGPR | static LDS B | scratch B/lane | VGPR spills | SGPR spills | max WG |",
             "|---|---|---|---|---|---|---|---|---|---|---|"]
    for r, nm in sorted(zip(rows, pretty), key=lambda t: t[1]):
        nm = re.sub(r"^void ", "", nm)
        nm = re.sub(r"\(.*$", "", nm)
        tot = r["vgpr_count"] + r["agpr_count"]
        alloc = (tot + 7) // 8 * 8
        waves = min(8, 512 // alloc) if alloc else 8
        lines.append("| `%s` | %d | %d | %d | %d | %d | %d | %d | %d | %d | %d |" % (
            nm, r["vgpr_count"], r["agpr_count"], tot, waves, r["sgpr_count"], r["group_segment_fixed_size"], r["private_segment_fixed_size"],
            r["vgpr_spill_count"], r["sgpr_spill_count"], r["max_flat_workgroup_size"]))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main()
