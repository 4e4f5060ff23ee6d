#!/bin/bash
# Runs ON THE GPU BOX: instruction-cache counters (one rocprofv3 --pmc pass, never with other trace domains) over a command.
# usage: tools/pmc_icache.sh <out dir under gpurun_out> <command...>        -> <out dir>/summary.json (tools/pmc_summary.py)
set -u
OUT=$(pwd)/gpurun_out/$1; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$(pwd)
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
           "SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQC_ICACHE_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  ( cd "$REPO" && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/p$i" -o pmc -- "$@" > "$OUT/p$i.log" 2>&1 )
  f=$(find "$OUT/p$i" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/p${i}_counter_collection.csv"
  rm -rf "$OUT/p$i"
done
cd "$REPO"
python tools/pmc_summary.py "$OUT"
