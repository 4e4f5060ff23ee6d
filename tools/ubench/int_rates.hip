// micro-benchmark: issue cycles per wave64 instruction of the integer ops the int8 network kernels lean on
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP 64
template <int OP> __global__ void k(int *out, int iters, long long *cyc)
{
    int a0 = threadIdx.x + 1, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19;
    long long d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
            if (OP == 0) { asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(a0) : "v"(a1)); asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(a2) : "v"(a3)); asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(a4) : "v"(a5)); asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(a6) : "v"(a7)); asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(a1) : "v"(a0)); asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(a3) : "v"(a2)); asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(a5) : "v"(a4)); asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(a7) : "v"(a6)); }
            if (OP == 1) { asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(d0) : "v"(a0), "v"(a1) : "vcc"); asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(d1) : "v"(a2), "v"(a3) : "vcc"); asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(d2) : "v"(a4), "v"(a5) : "vcc"); asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(d3) : "v"(a6), "v"(a7) : "vcc"); asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(d0) : "v"(a1), "v"(a2) : "vcc"); asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(d1) : "v"(a3), "v"(a4) : "vcc"); asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(d2) : "v"(a5), "v"(a6) : "vcc"); asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(d3) : "v"(a7), "v"(a0) : "vcc"); }
            if (OP == 2) { asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(a0) : "v"(a1), "v"(a2)); asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(a3) : "v"(a4), "v"(a5)); asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(a6) : "v"(a7), "v"(a1)); asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(a2) : "v"(a4), "v"(a5)); asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(a0) : "v"(a1), "v"(a2)); asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(a3) : "v"(a4), "v"(a5)); asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(a6) : "v"(a7), "v"(a1)); asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(a2) : "v"(a4), "v"(a5)); }
            if (OP == 3) { asm volatile("v_bfe_i32 %0, %1, 8, 8" : "=v"(a0) : "v"(a1)); asm volatile("v_bfe_i32 %0, %1, 8, 8" : "=v"(a2) : "v"(a3)); asm volatile("v_bfe_i32 %0, %1, 8, 8" : "=v"(a4) : "v"(a5)); asm volatile("v_bfe_i32 %0, %1, 8, 8" : "=v"(a6) : "v"(a7)); asm volatile("v_bfe_i32 %0, %1, 16, 8" : "=v"(a1) : "v"(a0)); asm volatile("v_bfe_i32 %0, %1, 16, 8" : "=v"(a3) : "v"(a2)); asm volatile("v_bfe_i32 %0, %1, 16, 8" : "=v"(a5) : "v"(a4)); asm volatile("v_bfe_i32 %0, %1, 16, 8" : "=v"(a7) : "v"(a6)); }
            if (OP == 4) { asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(a0) : "v"(a1), "v"(a2)); asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(a3) : "v"(a4), "v"(a5)); asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(a6) : "v"(a7), "v"(a1)); asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(a2) : "v"(a4), "v"(a5)); asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(a0) : "v"(a1), "v"(a2)); asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(a3) : "v"(a4), "v"(a5)); asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(a6) : "v"(a7), "v"(a1)); asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(a2) : "v"(a4), "v"(a5)); }
            if (OP == 5) { asm volatile("v_mul_hi_i32 %0, %1, %0" : "+v"(a0) : "v"(a1)); asm volatile("v_mul_hi_i32 %0, %1, %0" : "+v"(a2) : "v"(a3)); asm volatile("v_mul_hi_i32 %0, %1, %0" : "+v"(a4) : "v"(a5)); asm volatile("v_mul_hi_i32 %0, %1, %0" : "+v"(a6) : "v"(a7)); asm volatile("v_mul_hi_i32 %0, %1, %0" : "+v"(a1) : "v"(a0)); asm volatile("v_mul_hi_i32 %0, %1, %0" : "+v"(a3) : "v"(a2)); asm volatile("v_mul_hi_i32 %0, %1, %0" : "+v"(a5) : "v"(a4)); asm volatile("v_mul_hi_i32 %0, %1, %0" : "+v"(a7) : "v"(a6)); }
            if (OP == 6) { asm volatile("v_add_u32 %0, %1, %0" : "+v"(a0) : "v"(a1)); asm volatile("v_add_u32 %0, %1, %0" : "+v"(a2) : "v"(a3)); asm volatile("v_add_u32 %0, %1, %0" : "+v"(a4) : "v"(a5)); asm volatile("v_add_u32 %0, %1, %0" : "+v"(a6) : "v"(a7)); asm volatile("v_add_u32 %0, %1, %0" : "+v"(a1) : "v"(a0)); asm volatile("v_add_u32 %0, %1, %0" : "+v"(a3) : "v"(a2)); asm volatile("v_add_u32 %0, %1, %0" : "+v"(a5) : "v"(a4)); asm volatile("v_add_u32 %0, %1, %0" : "+v"(a7) : "v"(a6)); }
            if (OP == 7) { asm volatile("v_alignbit_b32 %0, %1, %2, 31" : "=v"(a0) : "v"(a1), "v"(a2)); asm volatile("v_alignbit_b32 %0, %1, %2, 31" : "=v"(a3) : "v"(a4), "v"(a5)); asm volatile("v_alignbit_b32 %0, %1, %2, 31" : "=v"(a6) : "v"(a7), "v"(a1)); asm volatile("v_alignbit_b32 %0, %1, %2, 31" : "=v"(a2) : "v"(a4), "v"(a5)); asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(a0) : "v"(a1), "v"(a2), "v"(a3)); asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(a4) : "v"(a5), "v"(a6), "v"(a7)); asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(a1) : "v"(a0), "v"(a2), "v"(a3)); asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(a5) : "v"(a4), "v"(a6), "v"(a7)); }
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (int)(d0 + d1 + d2 + d3);
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int OP> void run(const char *name, int wps)
{
    int *out; long long *cyc, h;
    int blocks = 256 * 4 * wps;
    hipMalloc(&out, sizeof(int) * blocks * 64); hipMalloc(&cyc, 8);
    const int iters = 2000;
    k<OP><<<blocks, 64>>>(out, iters, cyc); hipDeviceSynchronize();
    k<OP><<<blocks, 64>>>(out, iters, cyc); hipDeviceSynchronize();
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s waves/SIMD=%d : %.2f ticks per wave-instruction per wave, %.2f per SIMD\n", name, wps, (double)h / (iters * (double)REP), (double)h / (iters * (double)REP) / wps);
    hipFree(out); hipFree(cyc);
}
int main()
{
    for (int w : { 1, 2, 4 }) {
        run<6>("v_add_u32", w); run<0>("v_mul_lo_u32", w); run<5>("v_mul_hi_i32", w); run<1>("v_mad_i64_i32", w); run<2>("v_mad_i32_i24", w);
        run<3>("v_bfe_i32", w); run<4>("v_dot4_i32_i8", w); run<7>("v_alignbit / v_perm", w);
    }
    return 0;
}
