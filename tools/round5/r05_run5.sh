set -x
mkdir -p gpurun_out/r05e
(time timeout 1800 python -m pytest tests -m gpu -q) > gpurun_out/r05e/pytest_gpu.txt 2>&1
tail -15 gpurun_out/r05e/pytest_gpu.txt
