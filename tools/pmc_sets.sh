#!/bin/bash
# Runs ON THE GPU BOX: SQ counter passes (one rocprofv3 --pmc run per set, never with other trace domains) over a command.
# usage: tools/pmc_sets.sh <out dir under gpurun_out> <command...>
set -u
OUT=$(pwd)/gpurun_out/$1; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$(pwd)
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  ( cd "$REPO" && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/p$i" -o pmc -- "$@" > "$OUT/p$i.log" 2>&1 )
  f=$(find "$OUT/p$i" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/p${i}_counter_collection.csv"
  rm -rf "$OUT/p$i"
done
cd "$REPO"
python tools/pmc_summary.py "$OUT"
