// micro-benchmark: issue cycles per wave64 instruction for the fp64/convert ops the CMVN and magnitude stages use
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP 64
template <int OP> __global__ void k(double *out, int iters, long long *cyc) {
    double a0 = threadIdx.x * 1e-3 + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float f0 = (float)a0, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
            if (OP == 0) { asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a0) : "v"(f0)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a1) : "v"(f1)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a2) : "v"(f2)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a3) : "v"(f3)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a4) : "v"(f4)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a5) : "v"(f5)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a6) : "v"(f6)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a7) : "v"(f7)); }
            if (OP == 1) { asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f0) : "v"(a0)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f1) : "v"(a1)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f2) : "v"(a2)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f3) : "v"(a3)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f4) : "v"(a4)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f5) : "v"(a5)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f6) : "v"(a6)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f7) : "v"(a7)); }
            if (OP == 2) { asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(a0) : "v"(a1)); asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(a2) : "v"(a3)); asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(a4) : "v"(a5)); asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(a6) : "v"(a7)); asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(a1) : "v"(a0)); asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(a3) : "v"(a2)); asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(a5) : "v"(a4)); asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(a7) : "v"(a6)); }
            if (OP == 3) { asm volatile("v_add_f32 %0, %1, %0" : "+v"(f0) : "v"(f1)); asm volatile("v_add_f32 %0, %1, %0" : "+v"(f2) : "v"(f3)); asm volatile("v_add_f32 %0, %1, %0" : "+v"(f4) : "v"(f5)); asm volatile("v_add_f32 %0, %1, %0" : "+v"(f6) : "v"(f7)); asm volatile("v_add_f32 %0, %1, %0" : "+v"(f1) : "v"(f0)); asm volatile("v_add_f32 %0, %1, %0" : "+v"(f3) : "v"(f2)); asm volatile("v_add_f32 %0, %1, %0" : "+v"(f5) : "v"(f4)); asm volatile("v_add_f32 %0, %1, %0" : "+v"(f7) : "v"(f6)); }
            if (OP == 4) { asm volatile("v_sqrt_f64 %0, %1" : "=v"(a0) : "v"(a1)); asm volatile("v_sqrt_f64 %0, %1" : "=v"(a2) : "v"(a3)); asm volatile("v_sqrt_f64 %0, %1" : "=v"(a4) : "v"(a5)); asm volatile("v_sqrt_f64 %0, %1" : "=v"(a6) : "v"(a7)); asm volatile("v_sqrt_f64 %0, %1" : "=v"(a1) : "v"(a0)); asm volatile("v_sqrt_f64 %0, %1" : "=v"(a3) : "v"(a2)); asm volatile("v_sqrt_f64 %0, %1" : "=v"(a5) : "v"(a4)); asm volatile("v_sqrt_f64 %0, %1" : "=v"(a7) : "v"(a6)); }
            if (OP == 5) { asm volatile("v_sqrt_f32 %0, %1" : "=v"(f0) : "v"(f1)); asm volatile("v_sqrt_f32 %0, %1" : "=v"(f2) : "v"(f3)); asm volatile("v_sqrt_f32 %0, %1" : "=v"(f4) : "v"(f5)); asm volatile("v_sqrt_f32 %0, %1" : "=v"(f6) : "v"(f7)); asm volatile("v_sqrt_f32 %0, %1" : "=v"(f1) : "v"(f0)); asm volatile("v_sqrt_f32 %0, %1" : "=v"(f3) : "v"(f2)); asm volatile("v_sqrt_f32 %0, %1" : "=v"(f5) : "v"(f4)); asm volatile("v_sqrt_f32 %0, %1" : "=v"(f7) : "v"(f6)); }
            if (OP == 6) { asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a0) : "v"(a1)); asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a2) : "v"(a3)); asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a4) : "v"(a5)); asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a6) : "v"(a7)); asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a1) : "v"(a0)); asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a3) : "v"(a2)); asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a5) : "v"(a4)); asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a7) : "v"(a6)); }
            if (OP == 7) { asm volatile("v_mul_f64 %0, %1, %0" : "+v"(a0) : "v"(a1)); asm volatile("v_mul_f64 %0, %1, %0" : "+v"(a2) : "v"(a3)); asm volatile("v_mul_f64 %0, %1, %0" : "+v"(a4) : "v"(a5)); asm volatile("v_mul_f64 %0, %1, %0" : "+v"(a6) : "v"(a7)); asm volatile("v_add_f64 %0, %1, %0" : "+v"(a1) : "v"(a0)); asm volatile("v_add_f64 %0, %1, %0" : "+v"(a3) : "v"(a2)); asm volatile("v_add_f64 %0, %1, %0" : "+v"(a5) : "v"(a4)); asm volatile("v_add_f64 %0, %1, %0" : "+v"(a7) : "v"(a6)); }
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int OP> void run(const char *name, int waves_per_simd) {
    double *out; long long *cyc, h;
    int blocks = 256 * 4 * waves_per_simd;   // 64-thread blocks
    hipMalloc(&out, sizeof(double) * blocks * 64); hipMalloc(&cyc, 8);
    const int iters = 2000;
    k<OP><<<blocks, 64>>>(out, iters, cyc); hipDeviceSynchronize();
    k<OP><<<blocks, 64>>>(out, iters, cyc); hipDeviceSynchronize();
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-16s waves/SIMD=%d : %.2f clock64 ticks per wave-instruction (per wave), %.2f per SIMD\n", name, waves_per_simd, (double)h / (iters * (double)REP), (double)h / (iters * (double)REP) / waves_per_simd);
    hipFree(out); hipFree(cyc);
}
int main() {
    // VALU issue rate against occupancy: a lone wave issues one VALU instruction per ~5.3 cycles; how far does the SIMD scale?
    for (int w : {1, 2, 3, 4, 6, 8}) { run<3>("v_add_f32", w); run<6>("v_pk_add_f32", w); run<2>("v_fma_f64", w); run<0>("v_cvt_f64_f32", w); }
    for (int w : {1, 2}) {
        run<0>("v_cvt_f64_f32", w); run<1>("v_cvt_f32_f64", w); run<2>("v_fma_f64", w); run<3>("v_add_f32", w);
        run<4>("v_sqrt_f64", w); run<5>("v_sqrt_f32", w); run<6>("v_pk_add_f32", w); run<7>("v_mul/add_f64", w);
    }
    return 0;
}
