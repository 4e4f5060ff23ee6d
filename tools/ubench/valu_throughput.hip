// micro-benchmark: aggregate VALU throughput (wave-instructions per second, whole GPU) against waves per SIMD, measured
// with HIP events around the launch -- the number the kernels' instruction budgets are priced against.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
enum { ADD32, MULADD32, PKMULADD, FMA64, CMVN_TERM };
template <int OP> __global__ __launch_bounds__(64) void k(float *out, int iters, float seed)
{
    float a[16]; v2f p[8]; double d[8];
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = seed + j + threadIdx.x;
#pragma unroll
    for (int j = 0; j < 8; ++j) { p[j].x = seed + j; p[j].y = seed - j; d[j] = seed * j; }
    const v2f m = { 1.0000001f, 0.9999999f }, c = { 0.5f, 0.25f };
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (OP == ADD32) a[j] = a[j] + 1.0000001f;                                        // 16 instr
            if (OP == MULADD32) { float t = a[j] * 1.0000001f; a[j] = t + 0.5f; }              // 32 instr
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (OP == PKMULADD) { v2f t = p[j] * m; p[j] = t + c; }                             // 16 instr (pk_mul + pk_add)
            if (OP == FMA64) d[j] = __fma_rn(d[j], 1.0000001, 0.5);                             // 8 instr
            if (OP == CMVN_TERM) { float x = a[j] - a[j + 8]; double dd = (double)x; a[j] = (float)__fma_rn(dd, dd, (double)a[j]); }   // sub, cvt, cvt, fma, cvt = 5
        }
    }
    float s = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += a[j];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += p[j].x + p[j].y + (float)d[j];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int OP> static void run(const char *name, int instr_per_iter)
{
    float *out; hipMalloc(&out, 16384 * 64 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps : { 1, 2, 3, 4, 8 }) {
        const int blocks = 256 * 4 * wps, iters = 20000;
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, out, 100, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double n = (double)blocks * iters * instr_per_iter;
        printf("%-34s waves/SIMD=%d  %8.3f ms  %7.1f G wave-instr/s\n", name, wps, ms, n / (ms * 1e-3) * 1e-9);
    }
    hipFree(out);
}
int main()
{
    run<ADD32>("v_add_f32", 16);
    run<MULADD32>("v_mul_f32 + v_add_f32", 32);
    run<PKMULADD>("v_pk_mul_f32 + v_pk_add_f32", 16);
    run<FMA64>("v_fma_f64", 8);
    run<CMVN_TERM>("cmvn term (sub,cvt,cvt,fma64,cvt)", 40);
    return 0;
}
