// tools/mag_bench.hip -- development measurement, not part of the library ("implement the float-float magnitude").
// Times software_rfft's magnitude + power_spectrum's scaling per bin (numpy.hpp:1410, processing.hpp:306-309),
//     mag = (float)sqrt(pow(re,2) + pow(im,2)) [double];  P = (1.0/fft) * (mag*mag),
// in two formulations, and counts the bins on which they differ:
//   f64 : kws_device.h's bin_power (what the exact kernels run)
//   ff  : fp32 only -- exact squares and sum as float pairs, v_rsq_f32 seed, one residual correction, and a test that sends a bin to the
//         f64 form when the result sits within 2^-16 ulp of a rounding boundary (where the reference's double rounding could decide)
//   base: the loop and the input generator alone (subtracted)
// Build (here):   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I ei-keyword-spotting_amd/csrc -o ab_tmp/mag_bench tools/mag_bench.hip
// Run (GPU box):  ab_tmp/mag_bench
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdint>
#include <cstdio>
#include "kws_device.h"

__device__ __forceinline__ float ff_power(cf f, float inv_fft, unsigned &n_hard)
{
    const float a = f.r, b = f.i;
    const float p = a * a, q = b * b;
    const float pe = __fmaf_rn(a, a, -p), qe = __fmaf_rn(b, b, -q);          // a^2 = p + pe, b^2 = q + qe exactly
    const float hi = fmaxf(p, q), lo = fminf(p, q);
    const float s = hi + lo;
    const float se = lo - (s - hi);                                            // hi + lo = s + se exactly (hi >= lo >= 0)
    const float t = se + (pe + qe);
    const float y = __builtin_amdgcn_rsqf(s);
    const float m = s * y;                                                     // sqrt(s) within two ulp
    const float e = __fmaf_rn(-m, m, s);
    const float R = e + t;                                                     // a^2 + b^2 - m^2, to ~2^-23 of itself
    const float c = R * (0.5f * y);
    const float M = m + c;
    const float rho = (m - M) + c;                                             // what the last addition rounded away
    const unsigned Mb = __float_as_uint(M);
    const float ulp = __uint_as_float((Mb & 0x7f800000u) - (23u << 23));
    const bool hard = !(s > 0x1p-60f && s < 0x1p100f) || (Mb & 0x7fffffu) == 0u || fabsf(rho) > 0.49999f * ulp;
    if (hard) { ++n_hard; return bin_power(f, inv_fft); }
    return (M * M) * inv_fft;
}

template <int V>
__global__ __launch_bounds__(64, 2) void mag_kernel(int iters, unsigned seed, float *out, unsigned long long *counts)
{
    unsigned x = seed ^ (blockIdx.x * 64u + threadIdx.x) * 2654435761u;
    float acc = 0.0f;
    unsigned n_hard = 0, n_diff = 0;
    for (int i = 0; i < iters; ++i) {
        x = x * 1664525u + 1013904223u;
        const unsigned x2 = x * 22695477u + 1u;
        // spectrum-like values: 24 random mantissa bits, exponents over 2^-12 .. 2^4, both signs
        cf f;
        f.r = __uint_as_float((x & 0x807fffffu) | ((115u + ((x >> 23) & 15u)) << 23));
        f.i = __uint_as_float((x2 & 0x807fffffu) | ((115u + ((x2 >> 23) & 15u)) << 23));
        if (V == 0) acc += f.r + f.i;
        if (V == 1) acc += bin_power(f, 0.00390625f);
        if (V == 2) acc += ff_power(f, 0.00390625f, n_hard);
        if (V == 3) {
            const float u = bin_power(f, 0.00390625f), v = ff_power(f, 0.00390625f, n_hard);
            n_diff += __float_as_uint(u) != __float_as_uint(v);
            acc += v;
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
    if (V >= 2) { atomicAdd(&counts[0], (unsigned long long)n_hard); atomicAdd(&counts[1], (unsigned long long)n_diff); }
}

template <int V>
static double run(const char *name, int blocks, int iters, float *out, unsigned long long *counts, double base_ms)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipMemset(counts, 0, 16);
    mag_kernel<V><<<blocks, 64>>>(iters / 8, 1u, out, counts);
    hipDeviceSynchronize();
    hipMemset(counts, 0, 16);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        mag_kernel<V><<<blocks, 64>>>(iters, 12345u + rep, out, counts);
        hipEventRecord(e1);
        if (hipEventSynchronize(e1) != hipSuccess) { printf("%s: launch failed\n", name); return 0; }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    unsigned long long h[2];
    hipMemcpy(h, counts, 16, hipMemcpyDeviceToHost);
    const double bins = 5.0 * blocks * 64.0 * iters, per_launch = (double)blocks * 64.0 * iters;
    // 256 CUs x 4 SIMDs, 2 waves per SIMD resident: clocks per wave-bin on one SIMD at 2.4 GHz = time x 2.4e9 x 1024 SIMDs / (wave-bins)
    const double clk = (best - base_ms) * 1e-3 * 2.4e9 * 1024.0 / (per_launch / 64.0);
    printf("%-5s best %.3f ms per launch (%d blocks x 64 lanes x %d bins)  minus base: %.1f SIMD clocks per 64-bin wave step @2.4 GHz", name, best, blocks, iters, clk);
    if (V >= 2) printf("  to-f64 share %.3g  differing bins %llu of %.3g", (double)h[0] / (bins * (V == 3 ? 1 : 1)), h[1], bins);
    printf("\n");
    return best;
}

int main()
{
    const int blocks = 256 * 8 * 4, iters = 4096;      // 2 waves per SIMD resident, four rounds of them
    float *out; unsigned long long *counts;
    hipMalloc(&out, (size_t)blocks * 64 * 4); hipMalloc(&counts, 16);
    const double b = run<0>("base", blocks, iters, out, counts, 0.0);
    run<1>("f64", blocks, iters, out, counts, b);
    run<2>("ff", blocks, iters, out, counts, b);
    run<3>("both", blocks, iters, out, counts, b);
    return 0;
}
