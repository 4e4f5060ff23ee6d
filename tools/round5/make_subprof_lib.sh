#!/bin/bash
# Builds ab_tmp/libkws_subprof.so: a SCRATCH copy of the library whose fast_conv_tiles_h carries three clock reads (after the split of the image,
# after the contraction loop, after the epilogue) accumulated for wave 0 of workgroup 0, and an exported reader kws_dev_fast_sub().  Not a product
# build: the sources are patched in a temporary directory.  Read with:  KWS_LIB=ab_tmp/libkws_subprof.so python tools/gpu_fast_subphase.py
set -eu
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
T=${TMPDIR:-/tmp}/kws_subprof_src
rm -rf "$T" && mkdir -p "$T" && cp -r "$ROOT/ei-keyword-spotting_amd" "$T/" && cp -r "$ROOT/include" "$T/"
rm -rf "$T/ei-keyword-spotting_amd/csrc/.obj" "$T/ei-keyword-spotting_amd/csrc/.obj_dev" "$T/ei-keyword-spotting_amd/csrc/.obj_var"
python - "$T/ei-keyword-spotting_amd/csrc/kws_fast.hip" <<'PY'
import sys
p = sys.argv[1]
s = open(p).read()
head = "template <int MT, int NT, bool BG>\n__device__ __forceinline__ float fast_conv_tiles_h("
s = s.replace(head, '''__device__ long long g_fast_sub[8];
extern "C" __attribute__((visibility("default"))) int kws_dev_fast_sub(long long *out8)
{
    long long zero[8] = { 0 };
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_fast_sub), sizeof(zero)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_fast_sub), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#define SUBT(i) do { const long long n_ = clock64(); if (blockIdx.x == 0 && threadIdx.x == 0) g_fast_sub[(k.m_tiles > 2 ? 0 : 4) + (i)] += n_ - ts_; ts_ = n_; } while (0)
''' + head, 1)
a = s.index(head)
first = "    const int lm = lane & 15, lq = lane >> 4;\n    const float inv_s = fast_split_image(in, k.in_w, k.in_c, k.in_cp, k.in_stride, lane, k.inv_ppr20);"
i = s.index(first, a)
s = s[:i] + "    long long ts_ = clock64();\n" + s[i:]
s = s.replace("lane, k.inv_ppr20);\n    v4f acc[MT][NT];", "lane, k.inv_ppr20);\n    SUBT(0);\n    v4f acc[MT][NT];", 1)
fin = "    fast_conv_finish<MT, NT>(k, acc, vout, stage, sstride, shared, lane, sink, scale);"
i = s.index(fin, a)
s = s[:i] + "    SUBT(1);\n" + fin + "\n    SUBT(2);" + s[i + len(fin):]
open(p, "w").write(s)
PY
mkdir -p "$ROOT/ab_tmp"
# EXTRA="-DKWS_FAST_WPS=3" NAME=subprof3: the same for another build of the fast kernel (round 6)
NAME=${NAME:-subprof}
make -s -C "$T/ei-keyword-spotting_amd/csrc" OUT="$ROOT/ab_tmp/libkws_$NAME.so" ${EXTRA:+DEVFLAG="$EXTRA"}
ls -la "$ROOT/ab_tmp/libkws_$NAME.so"
