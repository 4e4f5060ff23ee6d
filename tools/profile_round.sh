#!/bin/bash
# Runs ON THE GPU BOX (gpurun): the round's rocprofv3 evidence -> gpurun_out/<tag>/ ; copy the summaries to profiles/.
# usage: tools/profile_round.sh <tag>
set -u
TAG=${1:-r02}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
run() { ( cd "$REPO" && "$@" ); }
# 1. the official bench line (with the cpu baseline), un-profiled
run python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
# 2. kernel trace + stats of the same command, per mode
# (the headline mode with bench.py's default step count: a 50-step run ends before the clocks have settled and reads ~1.5 % slower)
for mode in fast exact; do
  steps=600; warm=10; [ $mode = exact ] && { steps=100; warm=5; }
  ( cd "$REPO" && timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace_$mode" -o t -- python bench.py --mode $mode --steps $steps --warmup $warm --no-cpu-baseline --no-also > "$OUT/trace_$mode.log" 2>&1 )
  db=$(find "$OUT/trace_$mode" -name "*.db" | head -1)
  [ -n "$db" ] && run python tools/rocprof_summary.py "$db" "$OUT/cfg2_f32_${mode}_kernel_stats.md" "$TAG: python bench.py --mode $mode --steps $steps --warmup $warm --no-cpu-baseline --no-also (65536 clips per launch)"
  find "$OUT/trace_$mode" -name "*.db" -delete
done
# 3. SQ counters + HBM traffic of the headline (fast) command: one PMC pass per counter set (never with other trace domains)
run tools/pmc_sets.sh $TAG/pmc python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also > "$OUT/pmc_sets.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd "$REPO" && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc_$c" -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also > "$OUT/pmc_$c.log" 2>&1 )
done
f=$(find "$OUT/pmc_FETCH_SIZE" -name "*counter_collection.csv" | head -1)
w=$(find "$OUT/pmc_WRITE_SIZE" -name "*counter_collection.csv" | head -1)
cp "$f" "$OUT/pmc/fetch_counter_collection.csv"; cp "$w" "$OUT/pmc/write_counter_collection.csv"
run python tools/pmc_traffic.py "$f" "$w" 65536 "$OUT/pmc/traffic.json" cfg2_mfcc40_f32.kwsm > "$OUT/traffic.log" 2>&1
rm -rf "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE"
cat "$OUT/bench.json"
ls -la "$OUT" "$OUT/pmc"
