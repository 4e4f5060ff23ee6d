# evidence on the final library: fast against exact mode over 128 generator seeds + 11 families x 8 seeds per float model; rocprofv3 kernel trace of the general-shape path
set -x
mkdir -p gpurun_out/r05w
sha256sum ei-keyword-spotting_amd/libkws_mi355x.so > gpurun_out/r05w/lib_sha256.txt
(time timeout 1200 python tools/gpu_fast_sweep.py 128 8) > gpurun_out/r05w/fast_sweep.txt 2>&1
tail -5 gpurun_out/r05w/fast_sweep.txt
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
for c in "fft512 49" "fft512 2 s"; do
  tag=$(echo "$c" | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/r05w/trace_$tag -o t -- python $REPO/tools/gpu_generic_once.py "$c" 8192 40 > $REPO/gpurun_out/r05w/trace_$tag.log 2>&1
  db=$(find $REPO/gpurun_out/r05w/trace_$tag -name "*.db" | head -1)
  [ -n "$db" ] && python $REPO/tools/rocprof_summary.py "$db" $REPO/gpurun_out/r05w/generic_${tag}_kernel_stats.md "r05: python tools/gpu_generic_once.py \"$c\" 8192 40 (mfcc_batch_device: cepstra before cmvnw, 8 192 clips per launch)"
  find $REPO/gpurun_out/r05w/trace_$tag -name "*.db" -delete
  grep "kws_spectral" $REPO/gpurun_out/r05w/generic_${tag}_kernel_stats.md | head -3
done
