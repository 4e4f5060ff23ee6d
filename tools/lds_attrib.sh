#!/bin/bash
# Runs ON THE GPU BOX (gpurun): one rocprofv3 --pmc pass (counters only, with --kernel-trace for the kernel names) per form of tools/lds_attrib.py.
# usage: tools/lds_attrib.sh <tag>     -> gpurun_out/<tag>/{mfe,feat,full}_counter_collection.csv + attrib.txt
set -u
TAG=${1:-lds_attrib}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"; export TMPDIR=/tmp
for form in mfe feat full; do
  ( cd "$REPO" && timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv \
      -d "$OUT/p_$form" -o pmc -- python tools/lds_attrib.py run $form > "$OUT/$form.log" 2>&1 ); echo "$form rc=$?" >> "$OUT/$form.log"
  f=$(find "$OUT/p_$form" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/${form}_counter_collection.csv"
  rm -rf "$OUT/p_$form"
  tail -2 "$OUT/$form.log"
done
cd "$REPO" && python tools/lds_attrib.py report "$OUT" > "$OUT/attrib.txt" 2>&1; cat "$OUT/attrib.txt"
