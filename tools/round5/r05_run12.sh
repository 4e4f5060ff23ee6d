set -x
mkdir -p gpurun_out/r05j
cd /tmp && export TMPDIR=/tmp
for v in dev wps3; do
  KWS_LIB=$GRAFT_REPO_ROOT/ab_tmp/libkws_$v.so timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r05j/trace_$v -o t -- python $GRAFT_REPO_ROOT/tools/gpu_mfe_fast_steps.py 40 40 > $GRAFT_REPO_ROOT/gpurun_out/r05j/trace_$v.log 2>&1
  db=$(find $GRAFT_REPO_ROOT/gpurun_out/r05j/trace_$v -name "*.db" | head -1)
  [ -n "$db" ] && python $GRAFT_REPO_ROOT/tools/rocprof_summary.py "$db" $GRAFT_REPO_ROOT/gpurun_out/r05j/mfe_${v}_kernel_stats.md "r05 MFE-form of the fast kernel, library $v"
  find $GRAFT_REPO_ROOT/gpurun_out/r05j/trace_$v -name "*.db" -delete
  grep "kws_fast_kernel" $GRAFT_REPO_ROOT/gpurun_out/r05j/mfe_${v}_kernel_stats.md | head -3
done
