// micro-benchmark: what does one wave pay, in issue time, for LDS / scalar / MFMA instructions placed between its vector
// instructions?  Body = 64 v_add_f32 with N other instructions spread through the first half; cycles per body for a lone wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
#define V(i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(f[(i) & 7]) : "v"(g))
template <int KIND, int N> __global__ void k(float *out, int iters, long long *cyc)
{
    __shared__ float lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = i;
    __syncthreads();
    float f[8], g = threadIdx.x * 0.5f;
    float d1[8] = { 0 };
    v2f d2[8];
    v4f d4[8], acc[4];
    unsigned s0 = 1, s1 = 2;
#pragma unroll
    for (int i = 0; i < 8; ++i) { f[i] = i; d2[i] = v2f{ 0, 0 }; d4[i] = v4f{ 0, 0, 0, 0 }; }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = v4f{ 0, 0, 0, 0 };
    const unsigned addr = threadIdx.x * 16;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            if (i < 2 * N && (i & 1) == 0) {
                const int j = i >> 1;
                if (KIND == 1) asm volatile("ds_read_b32 %0, %1" : "=v"(d1[j & 7]) : "v"(addr));
                if (KIND == 2) asm volatile("ds_read_b64 %0, %1" : "=v"(d2[j & 7]) : "v"(addr));
                if (KIND == 3) asm volatile("ds_read_b128 %0, %1" : "=v"(d4[j & 7]) : "v"(addr));
                if (KIND == 4) asm volatile("ds_write_b32 %0, %1" : : "v"(addr), "v"(g) : "memory");
                if (KIND == 5) asm volatile("ds_write_b64 %0, %1" : : "v"(addr), "v"(d2[j & 7]) : "memory");
                if (KIND == 6) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(d1[j & 7]) : "v"(addr), "v"(g));
                if (KIND == 7) { asm volatile("s_add_u32 %0, %0, 3" : "+s"(s0) : : "scc"); asm volatile("s_xor_b32 %0, %0, 5" : "+s"(s1) : : "scc"); }
                if (KIND == 8) acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(g, g, acc[j & 3], 0, 0, 0);
                if (KIND == 9) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(d1[j & 7]) : "v"(g));
            }
            V(i);
        }
        asm volatile("s_waitcnt lgkmcnt(0)");
    }
    long long t1 = clock64();
    float r = s0 + s1;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += f[i] + d1[i] + d2[i][0] + d4[i][1];
    out[blockIdx.x * 64 + threadIdx.x] = r + acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int KIND, int N> double run()
{
    float *out; long long *cyc, h;
    const int blocks = 256 * 4, iters = 4000;
    hipMalloc(&out, 4 * 64 * blocks); hipMalloc(&cyc, 8);
    k<KIND, N><<<blocks, 64>>>(out, 10, cyc); hipDeviceSynchronize();
    k<KIND, N><<<blocks, 64>>>(out, iters, cyc); hipDeviceSynchronize();
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    hipFree(out); hipFree(cyc);
    return (double)h / iters;
}
template <int KIND> void row(const char *name)
{
    const double b = run<0, 0>(), a = run<KIND, 8>(), c = run<KIND, 16>();
    printf("%-22s 64 VALU: %6.1f | + 8: %6.1f | + 16: %6.1f  ->  %.1f ticks per extra instruction\n", name, b, a, c, (c - a) / 8.0);
}
int main()
{
    row<1>("ds_read_b32"); fflush(stdout); row<2>("ds_read_b64"); fflush(stdout); row<3>("ds_read_b128"); fflush(stdout);
    row<7>("2 x SALU"); fflush(stdout); row<8>("v_mfma_f32_16x16x4"); fflush(stdout); row<6>("ds_bpermute_b32"); fflush(stdout);
    row<4>("ds_write_b32"); fflush(stdout); row<5>("ds_write_b64"); fflush(stdout); row<9>("v_mov_b32_dpp"); fflush(stdout);
    return 0;
}
