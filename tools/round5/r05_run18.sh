# general-shape DSP kernel: integer divisions out of the per-chunk phases; tests, rates, SQ counters of the fft 512 case
set -x
mkdir -p gpurun_out/r05l
(time timeout 900 python -m pytest tests/test_gpu_generic_dsp.py -m gpu -q -x) > gpurun_out/r05l/pytest_generic.txt 2>&1
tail -4 gpurun_out/r05l/pytest_generic.txt
(time timeout 1200 python tools/gpu_generic_rate.py 8192) > gpurun_out/r05l/generic_rate.txt 2>&1
cat gpurun_out/r05l/generic_rate.txt
tools/pmc_sets.sh r05l/pmc_generic python tools/gpu_generic_once.py "fft512 49" 8192 12 > gpurun_out/r05l/pmc_generic.txt 2>&1
grep -A22 "== kws_spectral_lds" gpurun_out/r05l/pmc_generic.txt
