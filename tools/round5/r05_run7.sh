set -x
mkdir -p gpurun_out/r05g
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r05g/trace_cfg5 -o t -- python $GRAFT_REPO_ROOT/tools/ab_rate.py --child cfg5_dscnn_mfcc40_f32.kwsm fast > $GRAFT_REPO_ROOT/gpurun_out/r05g/trace_cfg5.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(find gpurun_out/r05g/trace_cfg5 -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py "$db" gpurun_out/r05g/cfg5_fast_kernel_stats.md "r05: tools/ab_rate.py --child cfg5_dscnn_mfcc40_f32.kwsm fast (65536 clips per launch)"
find gpurun_out/r05g/trace_cfg5 -name "*.db" -delete
head -12 gpurun_out/r05g/cfg5_fast_kernel_stats.md
timeout 600 python tools/gpu_streams_rate.py > gpurun_out/r05g/streams.txt 2>&1
cat gpurun_out/r05g/streams.txt | tail -12
timeout 600 python -m pytest tests/test_gpu_fast_mode.py -m gpu -x -q -k "split_operand" -s 2>&1 | grep -E "first convolution|passed|failed" 
