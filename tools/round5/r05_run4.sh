set -x
mkdir -p gpurun_out/r05d
timeout 900 python tools/ab_rate.py dev,dev+KWS_DEV_FAST_WAVES=4,dev+KWS_DEV_FAST_WAVES=6,dev+KWS_DEV_FAST_SPLIT,dev+KWS_DEV_FAST_B_GLOBAL 2 cfg2_mfcc40_f32.kwsm,l476_no_yes_f32.kwsm > gpurun_out/r05d/ab_occupancy.txt 2>&1
cat gpurun_out/r05d/ab_occupancy.txt
cd /tmp && export TMPDIR=/tmp
KWS_LIB=$GRAFT_REPO_ROOT/ab_tmp/libkws_dev.so KWS_DEV_FAST_SPLIT=1 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r05d/trace_split -o t -- python $GRAFT_REPO_ROOT/tools/ab_rate.py --child cfg2_mfcc40_f32.kwsm fast > $GRAFT_REPO_ROOT/gpurun_out/r05d/trace_split.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(find gpurun_out/r05d/trace_split -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py "$db" gpurun_out/r05d/split_kernel_stats.md "r05 split prototype: KWS_DEV_FAST_SPLIT=1 tools/ab_rate.py --child cfg2_mfcc40_f32.kwsm fast (65536 clips per launch)"
find gpurun_out/r05d/trace_split -name "*.db" -delete
cat gpurun_out/r05d/split_kernel_stats.md | head -20
timeout 900 python -m pytest tests/test_gpu_fast_mode.py -m gpu -x -q -k "guard_follows or depthwise_separable_graph_is" > gpurun_out/r05d/pytest_sel.txt 2>&1
tail -5 gpurun_out/r05d/pytest_sel.txt
