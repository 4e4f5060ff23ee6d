// micro-benchmark: the k-loop of kws_fast.hip's fast_conv_tiles<4, 2> in isolation, feature by feature
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int LEVEL> __global__ __launch_bounds__(64) void k(float *out, int iters, long long *cyc, int in_w, int stride)
{
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = i * 1e-3f;
    __syncthreads();
    const int lane = threadIdx.x, lm = lane & 15, lq = lane >> 4;
    v4f acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m) { acc[m][0] = v4f{ 0, 0, 0, 0 }; acc[m][1] = v4f{ 0, 0, 0, 0 }; }
    const float *abase = lds + lm * stride + 2 * lq;
    const float *wbase = lds + 4096 + 2 * lq * 30;
    float2 a[4], b[2];
#pragma unroll
    for (int m = 0; m < 4; ++m) a[m] = *(const float2 *)(abase + m * 16 * stride);
    b[0] = *(const float2 *)(wbase + 2 * lm); b[1] = *(const float2 *)(wbase + 2 * min(16 + lm, 29));
    long long t0 = clock64();
    for (int rep = 0; rep < iters; ++rep) {
        int aoff = 0, boff = 0, tap = 0;
        for (int it = 0; it < 15; ++it) {
            float2 an[4], bn[2];
            if (LEVEL >= 1) {
                aoff = (aoff + 8) & 1023; boff = (boff + 240) & 2047;
                tap = (it + 1) / 5;
#pragma unroll
                for (int m = 0; m < 4; ++m) an[m] = *(const float2 *)(abase + aoff + m * 16 * stride);
                bn[0] = *(const float2 *)(wbase + boff + 2 * lm); bn[1] = *(const float2 *)(wbase + boff + 2 * min(16 + lm, 29));
            }
            if (LEVEL >= 3) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].x, b[n].x, acc[m][n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].y, b[n].y, acc[m][n], 0, 0, 0);
            if (LEVEL >= 3) __builtin_amdgcn_sched_barrier(0);
            if (LEVEL >= 1) {
                if (LEVEL >= 2) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const bool in_img = (unsigned)(lm - 1 + tap + 16 * m) < (unsigned)in_w;
                        an[m].x = in_img ? an[m].x : 0.0f; an[m].y = in_img ? an[m].y : 0.0f;
                    }
                }
#pragma unroll
                for (int m = 0; m < 4; ++m) a[m] = an[m];
                b[0] = bn[0]; b[1] = bn[1];
            } else {
                asm volatile("" : "+v"(a[0].x), "+v"(b[0].x));
            }
        }
    }
    long long t1 = clock64();
    float s = 0;
#pragma unroll
    for (int m = 0; m < 4; ++m) s += acc[m][0][0] + acc[m][1][1] + acc[m][0][2] + acc[m][1][3];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int LEVEL> void run(const char *name, int wps)
{
    float *out; long long *cyc, h;
    const int blocks = 256 * 4 * wps, iters = 200;
    hipMalloc(&out, 4 * 64 * blocks); hipMalloc(&cyc, 8);
    k<LEVEL><<<blocks, 64>>>(out, 2, cyc, 49, 40); hipDeviceSynchronize();
    k<LEVEL><<<blocks, 64>>>(out, iters, cyc, 49, 40); hipDeviceSynchronize();
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-40s waves/SIMD=%d: %.0f ticks per k-step of 16 MFMAs (floor 512)\n", name, wps, (double)h / (iters * 15.0));
    hipFree(out); hipFree(cyc);
}
int main()
{
    for (int w : { 1, 2 }) {
        run<0>("MFMAs only", w);
        run<1>("+ operand prefetch from LDS", w);
        run<2>("+ SAME-padding selects", w);
        run<3>("+ sched_barriers", w);
    }
    return 0;
}
