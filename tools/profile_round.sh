#!/bin/bash
# Runs ON THE GPU BOX (gpurun): rocprofv3 evidence for one (model, mode) of bench.py -> gpurun_out/<tag>/ ; copy the summaries to profiles/.
# usage: tools/profile_round.sh <tag> [model file under models/ = cfg2_mfcc40_f32.kwsm] [modes = "fast exact"] [bench = 1: also the plain bench line]
#   r06            -> the headline graph, both modes, bench line, PMC of the fast mode      (gpurun_out/r06/{cfg2_f32_<mode>_kernel_stats.md, pmc/traffic.json})
#   r06_int8 l476_no_yes.kwsm exact 0  -> BASELINE configs[3]: kernel stats + PMC of the exact mode
set -u
TAG=${1:-r06}
MODEL=${2:-cfg2_mfcc40_f32.kwsm}
MODES=${3:-"fast exact"}
BENCH=${4:-1}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
run() { ( cd "$REPO" && "$@" ); }
short=$(basename "$MODEL" .kwsm); short=${short/mfcc40_/}      # cfg2_mfcc40_f32 -> cfg2_f32 (the names profiles/ has used since round 2)
# 1. the official bench line (with the cpu baseline), un-profiled
if [ "$BENCH" = 1 ]; then
  run python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
  cp "$REPO/bench_detail.json" "$OUT/bench_detail.json" 2>/dev/null
fi
# 2. kernel trace + stats of the same command, per mode
# (the headline mode with bench.py's default step count: a 50-step run ends before the clocks have settled and reads ~1.5 % slower)
for mode in $MODES; do
  steps=600; warm=10; [ $mode = exact ] && { steps=100; warm=5; }
  ( cd "$REPO" && timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace_$mode" -o t -- python bench.py --model models/$MODEL --mode $mode --steps $steps --warmup $warm --no-cpu-baseline --no-also > "$OUT/trace_$mode.log" 2>&1 )
  db=$(find "$OUT/trace_$mode" -name "*.db" | head -1)
  [ -n "$db" ] && run python tools/rocprof_summary.py "$db" "$OUT/${short}_${mode}_kernel_stats.md" "$TAG: python bench.py --model models/$MODEL --mode $mode --steps $steps --warmup $warm --no-cpu-baseline --no-also (65536 clips per launch)"
  find "$OUT/trace_$mode" -name "*.db" -delete
done
# 3. SQ counters + HBM traffic of the FIRST mode listed: one PMC pass per counter set (never with other trace domains)
pm=${MODES%% *}
run tools/pmc_sets.sh $TAG/pmc python bench.py --model models/$MODEL --mode $pm --steps 3 --warmup 1 --no-cpu-baseline --no-also > "$OUT/pmc_sets.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd "$REPO" && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc_$c" -o pmc -- python bench.py --model models/$MODEL --mode $pm --steps 3 --warmup 1 --no-cpu-baseline --no-also > "$OUT/pmc_$c.log" 2>&1 )
done
f=$(find "$OUT/pmc_FETCH_SIZE" -name "*counter_collection.csv" | head -1)
w=$(find "$OUT/pmc_WRITE_SIZE" -name "*counter_collection.csv" | head -1)
cp "$f" "$OUT/pmc/fetch_counter_collection.csv"; cp "$w" "$OUT/pmc/write_counter_collection.csv"
run python tools/pmc_traffic.py "$f" "$w" 65536 "$OUT/pmc/traffic.json" $MODEL $pm > "$OUT/traffic.log" 2>&1
rm -rf "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE"
[ "$BENCH" = 1 ] && cat "$OUT/bench.json"
ls -la "$OUT" "$OUT/pmc"
