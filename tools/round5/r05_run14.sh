# instruction-cache counters of the headline (190 KB of code), the exact-mode kernels and the general-shape kernel (68 KB); fast-mode sweep over seeds
set -x
mkdir -p gpurun_out/r05i
tools/pmc_icache.sh r05i/icache_fast python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also > gpurun_out/r05i/icache_fast.txt 2>&1
tools/pmc_icache.sh r05i/icache_exact python bench.py --mode exact --steps 3 --warmup 1 --no-cpu-baseline --no-also > gpurun_out/r05i/icache_exact.txt 2>&1
tools/pmc_icache.sh r05i/icache_generic python tools/gpu_generic_once.py "fft512 49" 8192 12 > gpurun_out/r05i/icache_generic.txt 2>&1
grep -A14 "== kws_fast_kernel<4, 5, false, false, true" gpurun_out/r05i/icache_fast.txt
grep -A14 "== kws_mfcc8_kernel<true, 8, 40\|== kws_nn_f32" gpurun_out/r05i/icache_exact.txt
grep -A14 "== kws_spectral_lds" gpurun_out/r05i/icache_generic.txt
(time timeout 900 python tools/gpu_fast_sweep.py 32 4) > gpurun_out/r05i/fast_sweep.txt 2>&1
tail -12 gpurun_out/r05i/fast_sweep.txt
