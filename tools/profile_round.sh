#!/bin/bash
# Runs ON THE GPU BOX (gpurun): the round's rocprofv3 evidence -> gpurun_out/<tag>/ ; copy the summaries to profiles/.
# usage: tools/profile_round.sh <tag>
set -u
TAG=${1:-r01b}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
run() { ( cd "$REPO" && "$@" ); }
# 1. the official bench line (with the cpu baseline), un-profiled
run python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
for m in l476_no_yes cfg2_mfcc40_int8 cfg5_dscnn_mfcc40_int8 cfg5_dscnn_mfcc40_f32; do
  run python bench.py --no-cpu-baseline --no-also --model models/$m.kwsm >> "$OUT/bench_other_models.jsonl" 2>/dev/null
done
# 2. kernel trace + stats of the same command
for m in cfg2_f32:models/cfg2_mfcc40_f32.kwsm int8:models/l476_no_yes.kwsm f32_twin:models/l476_no_yes_f32.kwsm; do
  name=${m%%:*}; path=${m#*:}
  ( cd "$REPO" && timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace_$name" -o t -- python bench.py --no-cpu-baseline --no-also --model $path > "$OUT/trace_$name.log" 2>&1 )
  db=$(find "$OUT/trace_$name" -name "*.db" | head -1)
  [ -n "$db" ] && run python tools/rocprof_summary.py "$db" "$OUT/${name}_kernel_stats.md" "$TAG: python bench.py --model $path (65536 clips, 10 steps + 3 warm-up)"
  find "$OUT/trace_$name" -name "*.db" -delete
done
# 3. HBM traffic: one PMC pass per counter (never together with other trace domains)
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd "$REPO" && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc_$c" -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also > "$OUT/pmc_$c.log" 2>&1 )
done
f=$(find "$OUT/pmc_FETCH_SIZE" -name "*counter_collection.csv" | head -1)
w=$(find "$OUT/pmc_WRITE_SIZE" -name "*counter_collection.csv" | head -1)
run python tools/pmc_traffic.py "$f" "$w" 65536 "$OUT/traffic.json" cfg2_mfcc40_f32.kwsm > "$OUT/traffic.log" 2>&1
cat "$OUT/bench.json"
ls -la "$OUT"
