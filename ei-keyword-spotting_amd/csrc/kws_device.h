// kws_device.h -- what the gfx950 kernels share (internal to libkws_mi355x.so): wave-local synchronisation, the complex
// arithmetic of the KissFFT replay, the reference's fast log and double-precision magnitude, shape constants, the int8
// input quantisation and cmvnw.
//
// BIT-EXACTNESS CONTRACT.  Every floating-point operation in the kernels is performed in the same order, at the same
// precision and with the same (separate) roundings as the reference's x86-64 build: the FFT replays KissFFT's
// radix-4,4,4,2 decimation (kissfft/kiss_fft.cpp:15-84, kiss_fftr.cpp:66-120) butterfly by butterfly, no
// multiply-add is ever contracted (every translation unit is compiled with -ffp-contract=off, the only fused operations are
// the ones the reference itself writes as fmaf()), the magnitude and the CMVN variance go through fp64 exactly
// as the reference's pow()/sqrt() calls do, and sequential fp32 sums keep their order.  Integer work is exact.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>
#include <type_traits>

#include "kws_plan.h"
#include "../../include/kws/kws_synth.h"

#pragma clang fp contract(off)

#define KWS_WAVE 64

// Optional clip selection: KWS_MODE_FAST hands clips whose cmvnw is ill-conditioned back to the exact kernels as a list in HBM
// (sel[0] = how many, sel[1 + i] = clip index); the kernels then walk the list instead of 0..n_clips-1.
__device__ __forceinline__ int sel_count(const int *sel, int n_clips) { return sel ? min(sel[0], n_clips) : n_clips; }
__device__ __forceinline__ int sel_clip(const int *sel, int i) { return sel ? sel[1 + i] : i; }

// Wave-local LDS hand-off: lanes of ONE wave exchange data through LDS.  LDS operations of a wave execute in
// issue order, so only the compiler has to be kept from reordering (same idiom as rocPRIM's wave barrier).
#define WAVE_SYNC()                                                   \
    do {                                                              \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");        \
        __builtin_amdgcn_wave_barrier();                              \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");        \
    } while (0)

// rolling-buffer rows (KwsDspPlan::ring_*)
__device__ __forceinline__ int ring_out_row(const KwsDspPlan &P, int r) { return P.ring_rows ? (P.ring_row0 + r) % P.ring_rows : r; }
__device__ __forceinline__ int ring_in_row(const KwsDspPlan &P, int i) { return (P.ring_rows && i < P.ring_rows) ? (i + P.ring_head) % P.ring_rows : i; }

struct cf { float r, i; };

__device__ __forceinline__ cf cmul(cf a, cf b)   // C_MUL, _kiss_fft_guts.h: four products, one sub, one add
{
    cf m;
    float rr = a.r * b.r, ii = a.i * b.i, ri = a.r * b.i, ir = a.i * b.r;
    m.r = rr - ii;
    m.i = ri + ir;
    return m;
}
__device__ __forceinline__ cf cadd(cf a, cf b) { cf c; c.r = a.r + b.r; c.i = a.i + b.i; return c; }
__device__ __forceinline__ cf csub(cf a, cf b) { cf c; c.r = a.r - b.r; c.i = a.i - b.i; return c; }

// kf_bfly4, forward transform (kiss_fft.cpp:38-84)
__device__ __forceinline__ void bfly4(cf &f0, cf &f1, cf &f2, cf &f3, cf t1, cf t2, cf t3)
{
    cf s0 = cmul(f1, t1), s1 = cmul(f2, t2), s2 = cmul(f3, t3);
    cf s5 = csub(f0, s1);
    f0 = cadd(f0, s1);
    cf s3 = cadd(s0, s2), s4 = csub(s0, s2);
    f2 = csub(f0, s3);
    f0 = cadd(f0, s3);
    f1.r = s5.r + s4.i;
    f1.i = s5.i - s4.r;
    f3.r = s5.r - s4.i;
    f3.i = s5.i + s4.r;
}

// kf_bfly4 with unit twiddles (k = 0): the three products by (1, 0) are exact up to the sign of a zero
__device__ __forceinline__ void bfly4_unit(cf &f0, cf &f1, cf &f2, cf &f3)
{
    const cf s0 = f1, s1 = f2, s2 = f3;
    cf s5 = csub(f0, s1);
    f0 = cadd(f0, s1);
    cf s3 = cadd(s0, s2), s4 = csub(s0, s2);
    f2 = csub(f0, s3);
    f0 = cadd(f0, s3);
    f1.r = s5.r + s4.i;
    f1.i = s5.i - s4.r;
    f3.r = s5.r - s4.i;
    f3.i = s5.i + s4.r;
}

// Four consecutive samples x[e - 2 .. e + 1] (e even) as two dwords: complex FFT input n of a frame is (y[2n], y[2n + 1]) with
// y[s] = x[s] - cof x[s - 1] (processing.hpp:104-106), so a point needs three of them.  4-byte aligned, not 8.
typedef int fast_i2 __attribute__((ext_vector_type(2), aligned(4)));

// the point of fast_i2 v in the reference's own arithmetic: numpy::int16_to_float (numpy.hpp:1289), then processing::preemphasis
// (processing.hpp:104-106): y[s] = x[s] - (cof * x[s - 1]), product and difference rounded separately
__device__ __forceinline__ cf exact_point(fast_i2 v, float pre_cof)
{
    const float prev = (float)(v.x >> 16) * (1.0f / 32768.0f);
    const float lo = (float)(short)(v.y & 0xffff) * (1.0f / 32768.0f);
    const float hi = (float)(v.y >> 16) * (1.0f / 32768.0f);
    cf z;
    const float pl = pre_cof * prev;
    z.r = lo - pl;
    const float ph_ = pre_cof * lo;
    z.i = hi - ph_;
    return z;
}

__device__ __forceinline__ cf ld_cf(const float *b, int n) { n += 8 * (n >> 5); float2 v = *(const float2 *)(b + 2 * n); cf c; c.r = v.x; c.i = v.y; return c; }
__device__ __forceinline__ void st_cf(float *b, int n, cf c) { n += 8 * (n >> 5); *(float2 *)(b + 2 * n) = make_float2(c.r, c.i); }
__device__ __forceinline__ cf to_cf(float2 v) { cf c; c.r = v.x; c.i = v.y; return c; }

// numpy::log (SDK/dsp/numpy.hpp:1350-1371): the fmaf() calls are the reference's own
__device__ __forceinline__ float fast_log(float a)
{
    int g = __float_as_int(a);
    int e = (int)(((unsigned)g - 0x3f2aaaabu) & 0xff800000u);
    g = (int)((unsigned)g - (unsigned)e);
    float m = __int_as_float(g);
    float i = (float)e * 1.19209290e-7f;
    float f = m - 1.0f;
    float s = f * f;
    float r = __fmaf_rn(0.230836749f, f, -0.279208571f);
    float t = __fmaf_rn(0.331826031f, f, -0.498910338f);
    r = __fmaf_rn(r, s, t);
    r = __fmaf_rn(r, s, f);
    r = __fmaf_rn(i, 0.693147182f, r);
    return r;
}

// software_rfft's magnitude + power_spectrum's scaling (numpy.hpp:1410, processing.hpp:306-309):
//   mag = (float)sqrt(pow(re,2) + pow(im,2))  [double];  P = (1.0/fft) * (mag*mag)
// Correctly rounded fp64 sqrt for x == 0 or x >= 2^-298 (a sum of two squared floats): clang's own expansion of
// sqrt(double) -- v_rsq_f64 + Goldschmidt -- minus the rescaling it needs only for inputs below 2^-767.
__device__ __forceinline__ double dsqrt_sumsq(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = __fma_rn(-h, g, 0.5);
    g = __fma_rn(g, r, g);
    h = __fma_rn(h, r, h);
    double d = __fma_rn(-g, g, x);
    g = __fma_rn(d, h, g);
    d = __fma_rn(-g, g, x);
    g = __fma_rn(d, h, g);
    return x == 0.0 ? x : g;
}

__device__ __forceinline__ float bin_power(cf f, float inv_fft)
{
    double re = (double)f.r, im = (double)f.i;
    double s = __fma_rn(re, re, im * im);     // both squares are exact in fp64: one rounding, as re*re + im*im
    float mag = (float)dsqrt_sumsq(s);
    float sq = mag * mag;
    return sq * inv_fft;                       // power-of-two fft length: exact scaling
}

// ---------------------------------------------------------------------------------------------------------
//  Kernel 1: MFCC + CMVN + input quantisation.  FFT 256, 32 mel filters (shipped configs); 64 threads = 1 clip.
// ---------------------------------------------------------------------------------------------------------
constexpr int KWS_FFT = 256;      // real FFT length the kernel is specialised for (host checks the model)
constexpr int KWS_NC = 128;        // complex FFT size
constexpr int KWS_NBINS = 129;
// mel filter counts the kernel is instantiated for: 32 (both shipped impulses) and 40 (BASELINE's 49x40 configs)
constexpr int KWS_NF_MAX = 40;
constexpr int KWS_MAXF = 52;       // frames per clip supported by the lane=frame stages (4 CMVN row groups x 13 rows)
constexpr int KWS_MAXNZ = 12;      // longest mel filter kept in registers
constexpr int KWS_MAXPROW = 192;   // rows of the symmetric-padded CMVN matrix (n_frames + win_size - 1)
// rows of the log-mel / cepstra buffer: LDS per wave must stay <= 20 KB (8 waves per CU)
__host__ __device__ constexpr int kws_mel_rows(int nf) { return nf <= 32 ? 52 : 50; }


// complex FFT slot of element c: 8 slots of padding after every 32 make every butterfly stage bank-conflict free
__device__ __forceinline__ int zi(int c) { return c + 8 * (c >> 5); }
constexpr int KWS_ZF = 2 * (KWS_NC + 8 * (KWS_NC / 32));   // floats per frame buffer

// ---------------------------------------------------------------------------------------------------------
// static_cast<int8_t>(round(f / scale) + zero_point) of run_inference (ei_run_classifier.h:436-444): no clamp, x86 wrap semantics
__device__ __forceinline__ int8_t quantize_feature(float o, float in_scale, int in_zp)
{
    const float qv = roundf(o / in_scale) + (float)in_zp;
    const int iv = (qv >= -2147483648.0f && qv < 2147483648.0f) ? (int)qv : (int)0x80000000;
    return (int8_t)(iv & 0xff);
}

//  cmvnw (processing.hpp:326-389) over the cepstra in LDS (row stride MELS) + optional outputs.
//  A lane owns one column and CR consecutive rows r0..r0+CR-1 (CG lanes = CG columns per row group, 64/CG row groups;
//  <13,16> for up to 16 cepstra, <17,20> for up to 40 in two passes).  Row r's window is padded rows r..r+win-1, so the
//  CR windows overlap: ONE walk over padded rows r0..r0+win+CR-2 feeds all CR running sums, each of which still receives
//  its win terms in the reference's ascending order (fp32 sum; fp64 square-accumulate rounded to fp32 after every term,
//  numpy.hpp:818-824).  CR independent chains per lane hide the fp64 latency.
//  Row offsets of the walk, per row group: off[g][p] = map[min(g*CR + p, prow-1)] * MELS, laid out in walk order so that a
//  lane fetches four of them with one 16-byte read a batch ahead: a term costs ONE dependent LDS read (prefetched too).
// ---------------------------------------------------------------------------------------------------------
//  emit(row, c, o): called once per normalised element o = (x - mean) / (std + eps).
//  VARIANCE = false: cmvnw's variance_normalization = false branch (processing.hpp:379-385), o = x - mean, no second walk.
//  g0 / cb_begin / cb_end: this wave's share when several waves split one matrix (latency mode): its first row group and
//  the column range it walks (multiples of CG); one wave doing everything passes 0, 0, ncep.
template <int CR, int CG, typename Emit, bool VARIANCE = true>
__device__ __forceinline__ void cmvn_columns(const float *__restrict__ mel, const int MELS, const int *__restrict__ map, int *__restrict__ offt,
                                             int lane, int nfr, int ncep, int prow, int win, Emit emit,
                                             int g0 = 0, int cb_begin = 0, int cb_end = 0x7fffffff)
{
    constexpr int NG = KWS_WAVE / CG;                              // row groups
    static_assert(((CR - 1) & 3) == 0, "16-byte aligned offset batches");
    const float fwin = (float)win;
    const int cgrp = min(lane / CG, NG - 1), cl = lane - (lane / CG) * CG;
    const bool lane_on = lane < NG * CG;
    const int r0 = (g0 + cgrp) * CR;
    const int offn = ((win + CR - 1 + 3) & ~3) + 8;
    for (int i = lane; i < NG * offn; i += KWS_WAVE) {
        const int g = i / offn, pp = i - g * offn;
        offt[i] = map[min((g0 + g) * CR + pp, prow - 1)] * MELS;
    }
    WAVE_SYNC();
    const int *myoff = offt + cgrp * offn;
    for (int cb = cb_begin; cb < min(ncep, cb_end); cb += CG) {
        const int c = cb + cl;
        const bool act = lane_on && (c < ncep) && (r0 < nfr);
        const int cc = min(c, ncep - 1);
        const float *col = mel + cc;
        auto val = [&](int p) { return col[myoff[p]]; };
        // body(x) for the padded rows p = CR-1 .. win-1 (every row's window is open), in order
        auto main_walk = [&](auto &&body) {
            int p = CR - 1;
            int4 a = *(const int4 *)(myoff + p);
            float xq[4] = { col[a.x], col[a.y], col[a.z], col[a.w] };
            a = *(const int4 *)(myoff + p + 4);
            for (; p + 4 <= win; p += 4) {
                const float x0 = xq[0], x1 = xq[1], x2 = xq[2], x3 = xq[3];
                xq[0] = col[a.x]; xq[1] = col[a.y]; xq[2] = col[a.z]; xq[3] = col[a.w];    // next batch in flight
                a = *(const int4 *)(myoff + p + 8);
                body(x0); body(x1); body(x2); body(x3);
            }
#pragma unroll
            for (int u = 0; u < 3; ++u)
                if (p + u < win) body(xq[u]);
        };
        float sum[CR], mean[CR], sd[CR];
#pragma unroll
        for (int r = 0; r < CR; ++r) { sum[r] = 0.0f; sd[r] = 0.0f; }
#pragma unroll
        for (int p = 0; p < CR - 1; ++p) {               // ramp-up: rows 0..p have started
            const float x = val(p);
#pragma unroll
            for (int r = 0; r <= p; ++r) sum[r] += x;
        }
        main_walk([&](float x) {
#pragma unroll
            for (int r = 0; r < CR; ++r) sum[r] += x;
        });
#pragma unroll
        for (int q = 0; q < CR - 1; ++q) {               // ramp-down: rows 0..q have finished
            const float x = val(win + q);
#pragma unroll
            for (int r = q + 1; r < CR; ++r) sum[r] += x;
        }
#pragma unroll
        for (int r = 0; r < CR; ++r) mean[r] = sum[r] / fwin;
        if constexpr (VARIANCE) {
            auto sq_acc = [&](float x, int r) {
                const float d = x - mean[r];
                const double dd = (double)d;
                sd[r] = (float)__fma_rn(dd, dd, (double)sd[r]);   // std += pow(d, 2)
            };
#pragma unroll
            for (int p = 0; p < CR - 1; ++p) {
                const float x = val(p);
#pragma unroll
                for (int r = 0; r <= p; ++r) sq_acc(x, r);
            }
            main_walk([&](float x) {
#pragma unroll
                for (int r = 0; r < CR; ++r) sq_acc(x, r);
            });
#pragma unroll
            for (int q = 0; q < CR - 1; ++q) {
                const float x = val(win + q);
#pragma unroll
                for (int r = q + 1; r < CR; ++r) sq_acc(x, r);
            }
        }
#pragma unroll
        for (int r = 0; r < CR; ++r) {
            const int row = r0 + r;
            if (act && row < nfr) {
                const float xv = mel[row * MELS + c];
                if constexpr (VARIANCE) {
                    const float dev = sqrtf(sd[r] / fwin);    // correctly rounded (clang expands v_sqrt_f32 + fix-up)
                    emit(row, c, (xv - mean[r]) / (dev + FLT_EPSILON));
                } else {
                    emit(row, c, xv - mean[r]);
                }
            }
        }
    }
}

