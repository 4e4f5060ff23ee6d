// micro-benchmark: issue cycles per fp32 MFMA (independent accumulators) for one wave and for several waves per SIMD
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
template <int OP, int NACC> __global__ __launch_bounds__(64) void k(float *out, int iters, long long *cyc)
{
    v4f acc[NACC];
    v16f big[2];
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = v4f{ 0.f, 0.f, 0.f, 0.f };
    big[0] = big[1] = v16f{ 0 };
    float a = threadIdx.x * 0.001f, b = 1.0f - a;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) {
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
        } else if (OP == 1) {
#pragma unroll
            for (int j = 0; j < 2; ++j) big[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, big[j], 0, 0, 0);
        } else if (OP == 2) {
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[j], 0, 0, 0);
        }
        asm volatile("" : "+v"(a), "+v"(b));
    }
    long long t1 = clock64();
    float s = 0;
#pragma unroll
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    s += big[0][0] + big[1][5];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int OP, int NACC> void run(const char *name, int per_iter, int wps)
{
    float *out; long long *cyc, h;
    const int blocks = 256 * 4 * wps, iters = 4000;
    hipMalloc(&out, 4 * 64 * blocks); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP, NACC><<<blocks, 64>>>(out, 10, cyc); hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP, NACC><<<blocks, 64>>>(out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s acc=%2d waves/SIMD=%d: %.1f ticks per MFMA per wave; %.2f ms -> %.1f G MFMA/s whole GPU\n", name, NACC, wps,
           (double)h / ((double)iters * per_iter), ms, (double)blocks * iters * per_iter / (ms * 1e-3) * 1e-9);
    hipFree(out); hipFree(cyc);
}
int main()
{
    for (int w : { 1, 2 }) {
        run<0, 1>("v_mfma_f32_16x16x4_f32", 1, w);
        run<0, 2>("v_mfma_f32_16x16x4_f32", 2, w);
        run<0, 4>("v_mfma_f32_16x16x4_f32", 4, w);
        run<0, 8>("v_mfma_f32_16x16x4_f32", 8, w);
        run<1, 2>("v_mfma_f32_32x32x2_f32", 2, w);
        run<2, 8>("v_mfma_f32_4x4x1_16B_f32", 8, w);
    }
    return 0;
}
