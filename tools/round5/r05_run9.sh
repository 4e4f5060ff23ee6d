set -x
mkdir -p gpurun_out/r05h
timeout 900 python -m pytest tests/test_gpu_generic_dsp.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r05h/pytest_generic.txt 2>&1
tail -5 gpurun_out/r05h/pytest_generic.txt
timeout 900 python tools/gpu_generic_rate.py 8192 > gpurun_out/r05h/generic_rate.txt 2>&1
cat gpurun_out/r05h/generic_rate.txt
