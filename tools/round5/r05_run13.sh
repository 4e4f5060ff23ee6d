# generic DSP kernel with multi-wave workgroups (tables shared): bit-exactness tests, then the (chunk, frames-together, waves) grid
set -x
mkdir -p gpurun_out/r05g
(time timeout 900 python -m pytest tests/test_gpu_generic_dsp.py tests/test_audio_ingest.py -m gpu -q -x) > gpurun_out/r05g/pytest_generic.txt 2>&1
tail -4 gpurun_out/r05g/pytest_generic.txt
(time timeout 900 python tools/gpu_generic_rate.py 8192) > gpurun_out/r05g/generic_rate.txt 2>&1
cat gpurun_out/r05g/generic_rate.txt
