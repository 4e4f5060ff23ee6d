// kws_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X / CDNA4), wave64.
//
// One wavefront owns one 1 s / 16 kHz clip.  Kernel 1 (kws_mfcc_kernel) replaces the reference's
// extract_mfcc_features() (SDK/classifier/ei_run_dsp.h:256-308): coalesced 16-byte int16 loads, pre-emphasis in
// registers, the 256-point real FFT staged in LDS, power spectrum, sparse mel gather, fast log, DCT, windowed
// CMVN and the int8 quantisation of ei_run_classifier.h:436-444.  Kernel 2 (kws_nn_kernel) replaces the
// EON-compiled TFLite-Micro graph (MODEL/tflite-model/trained_model_compiled.cpp:312-328).
//
// BIT-EXACTNESS CONTRACT.  Every floating-point operation below is performed in the same order, at the same
// precision and with the same (separate) roundings as the reference's x86-64 build: the FFT replays KissFFT's
// radix-4,4,4,2 decimation (kissfft/kiss_fft.cpp:15-84, kiss_fftr.cpp:66-120) butterfly by butterfly, no
// multiply-add is ever contracted (this file is compiled with -ffp-contract=off, the only fused operations are
// the ones the reference itself writes as fmaf()), the magnitude and the CMVN variance go through fp64 exactly
// as the reference's pow()/sqrt() calls do, and sequential fp32 sums keep their order.  Integer work is exact.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>

#include "kws_plan.h"
#include "../../include/kws/kws_synth.h"

#pragma clang fp contract(off)

#define KWS_WAVE 64

// Wave-local LDS hand-off: lanes of ONE wave exchange data through LDS.  LDS operations of a wave execute in
// issue order, so only the compiler has to be kept from reordering (same idiom as rocPRIM's wave barrier).
#define WAVE_SYNC()                                                   \
    do {                                                              \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");        \
        __builtin_amdgcn_wave_barrier();                              \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");        \
    } while (0)

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

// one frame pair's worth of samples for this lane: 8 samples + the sample before them
template <bool F32IN> struct RawSamples;
template <> struct RawSamples<false> { int4 v; short prev; };
template <> struct RawSamples<true> { float4 v0, v1; float prev; };

template <bool F32IN>
__device__ __forceinline__ RawSamples<F32IN> fetch_samples(const void *clip_base, int s0, int n_samples)
{
    // x[n-1] for the first of the 8 samples; at n = 0 the reference uses the LAST sample of the window
    // (processing.hpp:68, 104-106); the caller may override that value (continuous mode), see wrap below
    const int ip = (s0 == 0) ? (n_samples - 1) : (s0 - 1);
    RawSamples<F32IN> r;
    if constexpr (F32IN) {
        const float *xf = (const float *)clip_base;
        r.v0 = *(const float4 *)(xf + s0);
        r.v1 = *(const float4 *)(xf + s0 + 4);
        r.prev = xf[ip];
    } else {
        const int16_t *x = (const int16_t *)clip_base;
        r.v = *(const int4 *)(x + s0);
        r.prev = x[ip];
    }
    return r;
}

// kf_bfly5 with m = 1 (kiss_fft.cpp:131-192): every product and sum in the reference's order
__device__ __forceinline__ void bfly5(cf &F0, cf &F1, cf &F2, cf &F3, cf &F4, cf t1, cf t2, cf t3, cf t4, cf ya, cf yb)
{
    const cf s0 = F0;
    const cf s1 = cmul(F1, t1), s2 = cmul(F2, t2), s3 = cmul(F3, t3), s4 = cmul(F4, t4);
    const cf s7 = cadd(s1, s4), s10 = csub(s1, s4), s8 = cadd(s2, s3), s9 = csub(s2, s3);
    float tt, a, b;
    tt = s7.r + s8.r; F0.r = F0.r + tt;
    tt = s7.i + s8.i; F0.i = F0.i + tt;
    cf s5, s6, s11, s12;
    a = s7.r * ya.r; b = s8.r * yb.r; s5.r = (s0.r + a) + b;
    a = s7.i * ya.r; b = s8.i * yb.r; s5.i = (s0.i + a) + b;
    a = s10.i * ya.i; b = s9.i * yb.i; s6.r = a + b;
    a = s10.r * ya.i; b = s9.r * yb.i; s6.i = (-a) - b;
    F1 = csub(s5, s6);
    F4 = cadd(s5, s6);
    a = s7.r * yb.r; b = s8.r * ya.r; s11.r = (s0.r + a) + b;
    a = s7.i * yb.r; b = s8.i * ya.r; s11.i = (s0.i + a) + b;
    a = s10.i * yb.i; b = s9.i * ya.i; s12.r = (-a) + b;
    a = s10.r * yb.i; b = s9.r * ya.i; s12.i = a - b;
    F2 = cadd(s11, s12);
    F3 = csub(s11, s12);
}

// numpy::dct2 of one frame (numpy.hpp:378-401 -> dct::transform, fast-dct-fft.cpp:37-80 -> kiss_fftr(NF)): v holds the NF
// log-mel energies; R receives the NF/2+1 spectrum points the transform reads.  The complex FFT of NF/2 points is
// kf_work's recursion unrolled: NF = 32 -> 16 = 4 x 4 (kf_bfly4, kf_bfly4); NF = 40 -> 20 = 4 x 5 (kf_bfly5 leaves of
// stride 4, then kf_bfly4 with m = 5).
template <int NF, typename Emit>
__device__ __forceinline__ void dct_spectrum(const float (&v)[NF], const KwsDspPlan &P, Emit emit)   // emit(i, R[i]), i = 0..NF/2
{
    constexpr int NC = NF / 2;
    // even/odd reorder (in[i] = v[2i], in[NF-1-i] = v[2i+1]) read as NC complex points
    auto rin = [&](int i) { return (i < NC) ? v[2 * i] : v[2 * (NF - 1 - i) + 1]; };
    cf F[NC];
    if constexpr (NF == 32) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = q + 4 * j;                           // complex input index of leaf q
                F[4 * q + j].r = rin(2 * n);
                F[4 * q + j].i = rin(2 * n + 1);
            }
        const cf d0 = to_cf(P.dct_tw[0]);
#pragma unroll
        for (int q = 0; q < 4; ++q) bfly4(F[4 * q], F[4 * q + 1], F[4 * q + 2], F[4 * q + 3], d0, d0, d0);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            bfly4(F[k], F[k + 4], F[k + 8], F[k + 12], to_cf(P.dct_tw[k]), to_cf(P.dct_tw[2 * k]), to_cf(P.dct_tw[3 * k]));
    } else {
        static_assert(NF == 40, "DCT sizes: 32, 40");
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int n = q + 4 * j;
                F[5 * q + j].r = rin(2 * n);
                F[5 * q + j].i = rin(2 * n + 1);
            }
        const cf d0 = to_cf(P.dct_tw[0]), ya = to_cf(P.dct_tw[4]), yb = to_cf(P.dct_tw[8]);   // tw[fstride*m], tw[2*fstride*m]
#pragma unroll
        for (int q = 0; q < 4; ++q) bfly5(F[5 * q], F[5 * q + 1], F[5 * q + 2], F[5 * q + 3], F[5 * q + 4], d0, d0, d0, d0, ya, yb);
#pragma unroll
        for (int k = 0; k < 5; ++k)
            bfly4(F[k], F[k + 5], F[k + 10], F[k + 15], to_cf(P.dct_tw[k]), to_cf(P.dct_tw[2 * k]), to_cf(P.dct_tw[3 * k]));
    }
    // kiss_fftr split (kiss_fftr.cpp:84-119); every spectrum point is handed on as soon as it exists
    cf r0, rn;
    r0.r = F[0].r + F[0].i; r0.i = 0.0f;
    rn.r = F[0].r - F[0].i; rn.i = 0.0f;
    emit(0, r0);
    emit(NC, rn);
#pragma unroll
    for (int k = 1; k <= NC / 2; ++k) {
        cf fpk = F[k], fpnk;
        fpnk.r = F[NC - k].r; fpnk.i = -F[NC - k].i;
        cf f1k = cadd(fpk, fpnk), f2k = csub(fpk, fpnk);
        cf twv = cmul(f2k, to_cf(P.dct_stw[k - 1]));
        cf lo, hi;
        lo.r = (f1k.r + twv.r) * 0.5f;
        lo.i = (f1k.i + twv.i) * 0.5f;
        hi.r = (f1k.r - twv.r) * 0.5f;
        hi.i = (twv.i - f1k.i) * 0.5f;
        if (k != NC - k) emit(k, lo);                              // k == ncfft/2: overwritten by the "ncfft-k" store
        emit(NC - k, hi);
    }
}

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
template <int CR, int CG, typename Emit>
__device__ __forceinline__ void cmvn_columns(const float *__restrict__ mel, const int MELS, const int *__restrict__ map, int *__restrict__ offt,
                                             int lane, int nfr, int ncep, int prow, int win, Emit emit)
{
    constexpr int NG = KWS_WAVE / CG;                              // row groups
    static_assert(((CR - 1) & 3) == 0, "16-byte aligned offset batches");
    const float fwin = (float)win;
    const int cgrp = min(lane / CG, NG - 1), cl = lane - (lane / CG) * CG;
    const bool lane_on = lane < NG * CG;
    const int r0 = cgrp * CR;
    const int offn = ((win + CR - 1 + 3) & ~3) + 8;
    for (int i = lane; i < NG * offn; i += KWS_WAVE) {
        const int g = i / offn, pp = i - g * offn;
        offt[i] = map[min(g * CR + pp, prow - 1)] * MELS;
    }
    WAVE_SYNC();
    const int *myoff = offt + cgrp * offn;
    for (int cb = 0; cb < ncep; cb += CG) {
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
#pragma unroll
        for (int r = 0; r < CR; ++r) {
            const int row = r0 + r;
            if (act && row < nfr) {
                const float dev = sqrtf(sd[r] / fwin);        // correctly rounded (clang expands v_sqrt_f32 + fix-up)
                const float xv = mel[row * MELS + c];
                emit(row, c, (xv - mean[r]) / (dev + FLT_EPSILON));
            }
        }
    }
}

// PROF: development aid -- per-phase shader-clock totals of block 0 are written to prof_out (tools/gpu_phase_profile.py)
#define KWS_NPHASE 10
#define PH(i) do { if (PROF) { long long now_ = clock64(); ph[i] += now_ - tlast; tlast = now_; } } while (0)

// ---------------------------------------------------------------------------------------------------------
//  Kernel 1: MFCC.  WITH_CMVN = true is the batch hot path: extract_mfcc_features = mfcc + cmvnw + input
//  quantisation in ONE launch, the cepstra never leave LDS.  WITH_CMVN = false stops after speechpy::feature::mfcc
//  (feature.hpp:370-439) and writes the cepstra BEFORE cmvnw to HBM (stage API, continuous mode: there cmvnw runs
//  over a rolling window, in kws_cmvn_nn_kernel).
//  F32IN: samples arrive as float (the SDK's signal_t callback), else int16 PCM.  NZ: mel taps kept in registers.
//  wrap (optional, one float per window): the value the reference's pre-emphasis uses as x[-1]; NULL = x[N-1].
// ---------------------------------------------------------------------------------------------------------
// complex FFT slot of element c: 8 slots of padding after every 32 make every butterfly stage bank-conflict free
__device__ __forceinline__ int zi(int c) { return c + 8 * (c >> 5); }
constexpr int KWS_ZF = 2 * (KWS_NC + 8 * (KWS_NC / 32));   // floats per frame buffer
template <int CHP, int NF>   // frame PAIRS per chunk, mel filters
struct alignas(16) MfccSmem {
    static constexpr int CHF = 2 * CHP;
    static constexpr int MELS = NF + 1;  // padded (odd) row stride of the log-mel / cepstra buffer
    float z[2][KWS_ZF];                  // per half-wave: pre-emphasised frame, then the in-place complex FFT
    // power spectrum [bin][frame in chunk]; after the last chunk of a clip the same storage holds the
    // pad_1d_symmetric row map for cmvnw
    union {
        float p[KWS_NBINS * CHF];
        int map[KWS_MAXPROW];
    } u;
    // log-mel energies [frame][filter]; the DCT overwrites each row in place with that frame's cepstra
    float mel[kws_mel_rows(NF) * MELS];
    float energy[kws_mel_rows(NF)];
    float dcny[2 * CHF];                 // tmp[0] of each frame of the chunk (DC / Nyquist source)
};
static_assert(sizeof(MfccSmem<9, 32>) <= 20 * 1024 && sizeof(MfccSmem<9, 40>) <= 20 * 1024, "8 waves per CU need <= 20 KB LDS each");

template <int CHP, bool F32IN, bool WITH_CMVN, int NZ, int NF = 32, bool PROF = false, bool WIDE = false>
__global__ __launch_bounds__(KWS_WAVE, 2) void kws_mfcc_kernel(KwsDspPlan P, const void *__restrict__ pcm_v, int n_clips,
                                                            float *__restrict__ features, int8_t *__restrict__ q_out,
                                                            float in_scale, int in_zp, const float *__restrict__ wrap, int out_stride,
                                                            long long *prof_out = nullptr)
{
    constexpr int CHF = 2 * CHP;
    constexpr int MELS = NF + 1, NCEPT = NF / 2 + 1;     // DCT only produces outputs 0..NF/2 (fast-dct-fft.cpp:71)
    __shared__ MfccSmem<CHP, NF> sm;
    const int lane = threadIdx.x;
    const int half = lane >> 5, t = lane & 31;

    // ---- per-lane constants, fixed for the whole launch --------------------------------------------------
    const int k01 = t & 1, g01 = t >> 1;
    const int n0 = (g01 >> 2) + 4 * (g01 & 3);            // digit-reversed input base of this lane's radix-4 group
    const cf a1 = to_cf(P.tw[16 * k01]), a2 = to_cf(P.tw[32 * k01]), a3 = to_cf(P.tw[48 * k01]);
    const int K2 = t & 7, G2 = t >> 3;
    const cf b1 = to_cf(P.tw[4 * K2]), b2 = to_cf(P.tw[8 * K2]), b3 = to_cf(P.tw[12 * K2]);
    const cf c1 = to_cf(P.tw[t]), c2 = to_cf(P.tw[2 * t]), c3 = to_cf(P.tw[3 * t]);
    const cf st1 = to_cf(P.stw[t]), st2 = to_cf(P.stw[t + 32]);
    const int nfr = P.n_frames, ncep = P.n_cepstral;
    const int n_pairs = (nfr + 1) >> 1;
    const int prow = nfr + 2 * P.pad;
    float *zb = sm.z[half];
    // NF == 32: this lane's mel filter (filter index = lane & 31 in every pass of the mel stage) keeps its ascending-bin
    // taps in registers; other filter counts walk the CSR table
    int fbin[NZ];
    float fwt[NZ];
    {
        // NF == 32: filter lane & 31 (two frames per pass of the mel stage); other counts: filter `lane` (lanes >= NF idle)
        const int fj = NF == 32 ? t : min(lane, NF - 1);
        const int b0 = P.filt_start[fj], b1e = P.filt_start[fj + 1];
#pragma unroll
        for (int n = 0; n < NZ; ++n) {
            const bool on = b0 + n < b1e;
            fbin[n] = on ? P.filt_bin[b0 + n] * CHF : 0;
            fwt[n] = on ? P.filt_w[b0 + n] : 0.0f;
        }
    }
    int mapreg[KWS_MAXPROW / KWS_WAVE];    // numpy::pad_1d_symmetric row map (numpy.hpp:479-541), KWS_WAVE entries apart
#pragma unroll
    for (int i = 0; i < KWS_MAXPROW / KWS_WAVE; ++i) mapreg[i] = (lane + i * KWS_WAVE < prow) ? P.pad_map[lane + i * KWS_WAVE] : 0;
    long long ph[KWS_NPHASE] = { 0 }, tlast = PROF ? clock64() : 0;

    for (int clip = blockIdx.x; clip < n_clips; clip += gridDim.x) {
        const void *xbase = F32IN ? (const void *)((const float *)pcm_v + (size_t)clip * P.n_samples)
                                  : (const void *)((const int16_t *)pcm_v + (size_t)clip * P.n_samples);
        // software prefetch, two frame pairs deep: the samples of pair p+2 are requested before pair p is transformed
        RawSamples<F32IN> nxt = fetch_samples<F32IN>(xbase, min(half, nfr - 1) * P.frame_stride + 8 * t, P.n_samples);
        RawSamples<F32IN> nxt2 = fetch_samples<F32IN>(xbase, min(2 + half, nfr - 1) * P.frame_stride + 8 * t, P.n_samples);
        const bool has_wrap = wrap != nullptr;
        const float wrapv = has_wrap ? wrap[clip] : 0.0f;

        for (int pair0 = 0; pair0 < n_pairs; pair0 += CHP) {
            const int pair1 = min(pair0 + CHP, n_pairs);
            for (int pr = pair0; pr < pair1; ++pr) {
                // ---- 8 samples/lane (16 B, coalesced: 32 lanes = the 256 samples of a frame that rfft keeps) -----
                const int f = 2 * pr + half;
                const RawSamples<F32IN> cur = nxt;
                nxt = nxt2;
                if (pr + 2 < n_pairs)
                    nxt2 = fetch_samples<F32IN>(xbase, min(f + 4, nfr - 1) * P.frame_stride + 8 * t, P.n_samples);
                const bool first_sample = has_wrap && (min(f, nfr - 1) * P.frame_stride + 8 * t == 0);
                float y[8];
                if constexpr (F32IN) {
                    const float v[8] = { cur.v0.x, cur.v0.y, cur.v0.z, cur.v0.w, cur.v1.x, cur.v1.y, cur.v1.z, cur.v1.w };
                    float prev = first_sample ? wrapv : cur.prev;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float pl = P.pre_cof * prev;                                   // cof * prev, then subtract
                        y[j] = v[j] - pl;
                        prev = v[j];
                    }
                } else {
                    float prev = first_sample ? wrapv : (float)cur.prev * (1.0f / 32768.0f);
                    const int w[4] = { cur.v.x, cur.v.y, cur.v.z, cur.v.w };
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float lo = (float)(short)(w[j] & 0xffff) * (1.0f / 32768.0f);   // numpy::int16_to_float
                        float hi = (float)(short)(w[j] >> 16) * (1.0f / 32768.0f);
                        float pl = P.pre_cof * prev;
                        y[2 * j] = lo - pl;
                        float ph_ = P.pre_cof * lo;
                        y[2 * j + 1] = hi - ph_;
                        prev = hi;
                    }
                }
                *(float4 *)(zb + 2 * zi(4 * t)) = make_float4(y[0], y[1], y[2], y[3]);        // complex slots 4t..4t+3
                *(float4 *)(zb + 2 * zi(4 * t) + 4) = make_float4(y[4], y[5], y[6], y[7]);
                WAVE_SYNC();
                PH(0);

                // ---- kf_bfly2 (m=1) fused with kf_bfly4 (m=2): kiss_fft.cpp:232-296 levels 4 and 3 ---------
                cf u[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    cf a = ld_cf(zb, n0 + 16 * i), b = ld_cf(zb, n0 + 16 * i + 64);
                    u[i] = k01 ? csub(a, b) : cadd(a, b);       // b * tw[0], tw[0] = (1, -0)
                }
                bfly4(u[0], u[1], u[2], u[3], a1, a2, a3);
#pragma unroll
                for (int i = 0; i < 4; ++i) st_cf(zb, 8 * g01 + k01 + 2 * i, u[i]);
                WAVE_SYNC();
                // ---- kf_bfly4 m=8, fstride=4 ----------------------------------------------------------------
#pragma unroll
                for (int i = 0; i < 4; ++i) u[i] = ld_cf(zb, 32 * G2 + K2 + 8 * i);
                bfly4(u[0], u[1], u[2], u[3], b1, b2, b3);
#pragma unroll
                for (int i = 0; i < 4; ++i) st_cf(zb, 32 * G2 + K2 + 8 * i, u[i]);
                WAVE_SYNC();
                // ---- kf_bfly4 m=32, fstride=1 ---------------------------------------------------------------
#pragma unroll
                for (int i = 0; i < 4; ++i) u[i] = ld_cf(zb, t + 32 * i);
                bfly4(u[0], u[1], u[2], u[3], c1, c2, c3);
#pragma unroll
                for (int i = 0; i < 4; ++i) st_cf(zb, t + 32 * i, u[i]);
                WAVE_SYNC();

                PH(1);
                // ---- kiss_fftr split (kiss_fftr.cpp:84-119) + power spectrum -------------------------------
                const int fr = f - 2 * pair0;                 // frame slot in the chunk
                float *pcol = sm.u.p + fr;
                const bool live = f < nfr;
#pragma unroll
                for (int rep = 0; rep < 2; ++rep) {
                    const int k = t + 1 + 32 * rep;            // 1..64
                    const cf stw = rep ? st2 : st1;
                    cf fpk = ld_cf(zb, k), fq = ld_cf(zb, KWS_NC - k);
                    cf fpnk; fpnk.r = fq.r; fpnk.i = -fq.i;
                    cf f1k = cadd(fpk, fpnk), f2k = csub(fpk, fpnk);
                    cf twv = cmul(f2k, stw);
                    cf lo, hi;
                    lo.r = (f1k.r + twv.r) * 0.5f;             // HALF_OF
                    lo.i = (f1k.i + twv.i) * 0.5f;
                    hi.r = (f1k.r - twv.r) * 0.5f;
                    hi.i = (twv.i - f1k.i) * 0.5f;
                    if (live) {
                        if (k != KWS_NC / 2) pcol[k * CHF] = bin_power(lo, P.inv_fft);   // k == 64: overwritten by the
                        pcol[(KWS_NC - k) * CHF] = bin_power(hi, P.inv_fft);            // "ncfft-k" store
                    }
                }
                // DC / Nyquist bins (kiss_fftr.cpp:84-96) need tmp[0] only: parked per frame, evaluated once per chunk
                // with one frame per lane instead of one lane per wave here
                if (t == 0 && live) *(float2 *)(sm.dcny + 2 * fr) = *(const float2 *)zb;
                WAVE_SYNC();
                PH(2);
            }

            // ---- per chunk: frame energy (sequential fp32 sum, numpy.hpp:88-94) -------------------------------
            const int f_base = 2 * pair0;
            const int nfc = min(2 * pair1, nfr) - f_base;
            if (lane < nfc) {
                {
                    const float2 d = *(const float2 *)(sm.dcny + 2 * lane);
                    cf dc, ny;
                    dc.r = d.x + d.y; dc.i = 0.0f;
                    ny.r = d.x - d.y; ny.i = 0.0f;
                    sm.u.p[lane] = bin_power(dc, P.inv_fft);
                    sm.u.p[KWS_NC * CHF + lane] = bin_power(ny, P.inv_fft);
                }
                // 129 ordered adds; the operands arrive 16 at a time, one batch ahead of the adds (the chain would otherwise
                // wait for an LDS round trip per batch with only two waves per SIMD to cover it)
                float e = 0.0f;
                const float *pl = sm.u.p + lane;
                static_assert((KWS_NBINS - 1) % 16 == 0, "batches of 16 bins");
                float cur[16], nxt[16];
                const float last = pl[(KWS_NBINS - 1) * CHF];
#pragma unroll
                for (int u = 0; u < 16; ++u) cur[u] = pl[u * CHF];
                for (int k0 = 0; k0 < KWS_NBINS - 1; k0 += 16) {
                    const int kn = min(k0 + 16, KWS_NBINS - 1 - 16);
#pragma unroll
                    for (int u = 0; u < 16; ++u) nxt[u] = pl[(kn + u) * CHF];
#pragma unroll
                    for (int u = 0; u < 16; ++u) e += cur[u];
#pragma unroll
                    for (int u = 0; u < 16; ++u) cur[u] = nxt[u];
                }
                e += last;
                if (e == 0.0f) e = FLT_EPSILON;                                       // feature.hpp:296-298
                sm.energy[f_base + lane] = e;
                if constexpr (!WITH_CMVN)
                    if (P.mfe_energy) P.mfe_energy[(size_t)clip * nfr + f_base + lane] = e;
            }
            PH(3);
            // ---- mel: sparse ascending-bin gather == dot_by_row (numpy.hpp:183-211), zero_handling, log ------
            if constexpr (NF == 32) {
                for (int idx = lane; idx < nfc * NF; idx += KWS_WAVE) {
                    const int fr = idx >> 5;                                          // filter j == lane & 31 == t
                    float acc = 0.0f;
#pragma unroll
                    for (int n = 0; n < NZ; ++n) {         // taps beyond a filter's end have weight 0: power >= 0 is
                        float prod = sm.u.p[fbin[n] + fr] * fwt[n];   // finite, so they add an exact +0
                        acc += prod;
                    }
                    if (acc == 0.0f) acc = FLT_EPSILON;                                // functions.hpp:63-69
                    if constexpr (!WITH_CMVN)
                        if (P.mfe_mel) P.mfe_mel[((size_t)clip * nfr + f_base + fr) * NF + t] = acc;
                    sm.mel[(f_base + fr) * MELS + t] = fast_log(acc);
                }
            } else {
                // one frame per pass, lane = filter (taps in registers as above; walking the CSR table from memory
                // instead cost a third of the kernel)
                if (lane < NF) {
                    for (int fr = 0; fr < nfc; ++fr) {
                        float acc = 0.0f;
#pragma unroll
                        for (int n = 0; n < NZ; ++n) {
                            float prod = sm.u.p[fbin[n] + fr] * fwt[n];
                            acc += prod;
                        }
                        if (acc == 0.0f) acc = FLT_EPSILON;
                        if constexpr (!WITH_CMVN)
                            if (P.mfe_mel) P.mfe_mel[((size_t)clip * nfr + f_base + fr) * NF + lane] = acc;
                        sm.mel[(f_base + fr) * MELS + lane] = fast_log(acc);
                    }
                }
            }
            WAVE_SYNC();
            PH(4);
        }

        if constexpr (!WITH_CMVN)
            if (P.mfe_mel) { WAVE_SYNC(); continue; }                                   // MFE block: no log / DCT output
        // ---- DCT-II via NF-point kiss_fftr, one frame per lane (numpy.hpp:378-401, fast-dct-fft.cpp:37-80) ------
        // the cepstra of a frame replace its log-mel row in place (row stride MELS)
#pragma unroll
        for (int i = 0; i < KWS_MAXPROW / KWS_WAVE; ++i) sm.u.map[lane + i * KWS_WAVE] = mapreg[i];
        if (lane < nfr) {
            float v[NF];
            float *mrow = sm.mel + lane * MELS;
#pragma unroll
            for (int i = 0; i < NF; ++i) v[i] = mrow[i];
            float *orow = WITH_CMVN ? mrow : features + (size_t)clip * out_stride + lane * ncep;
            auto put = [&](int i, cf R) {
                if (i < ncep) {
                    float a = R.r * P.dct_cos[i];
                    float b = R.i * P.dct_sin[i];
                    float d = (a + b) * 2.0f;
                    d = d * (i == 0 ? P.dct_s0 : P.dct_s1);
                    orow[i] = d;
                }
            };
            // coefficients above N/2 are never written by the transform: they keep the log-mel input (x2, scaled)
            if constexpr (NF == 32) {            // 17 spectrum points fit the register budget: scale + store after the split
                cf R[NCEPT];
                dct_spectrum<NF>(v, P, [&](int i, cf r) { R[i] = r; });
#pragma unroll
                for (int i = 0; i < NCEPT; ++i) put(i, R[i]);
#pragma unroll
                for (int i = NCEPT; i < NF; ++i)
                    if (i < ncep) orow[i] = (v[i] * 2.0f) * P.dct_s1;
            } else {                              // 40 filters: hand every point on as soon as it exists (no spills);
                // in place: element i >= NCEPT is read, then written, by this lane only
                for (int i = NCEPT; i < ncep; ++i) orow[i] = (mrow[i] * 2.0f) * P.dct_s1;
                dct_spectrum<NF>(v, P, put);
            }
            orow[0] = fast_log(sm.energy[lane]);                                       // feature.hpp:425-429
        }
        WAVE_SYNC();
        PH(5);
        if constexpr (!WITH_CMVN) continue;

        // ---- cmvnw (processing.hpp:326-389) + input quantisation ---------------------------------------------
        {
            float *fout = features ? features + (size_t)clip * (nfr * ncep) : nullptr;
            int8_t *qclip = q_out ? q_out + (size_t)clip * (nfr * ncep) : nullptr;
            int *offt = (int *)&sm.z[0][0];                       // the FFT buffers are dead by now
            // WIDE (more than 16 cepstra, chosen at launch): 20 columns x 3 row groups of 17 rows per pass instead of
            // 16 x 4 x 13 -- 40 cepstra take 2 passes instead of 3.  One layout per instantiation keeps the registers.
            auto emit = [&](int row, int c, float o) {
                const int idx = row * ncep + c;
                if (fout) fout[idx] = o;                      // optional output (extract_mfcc_features' matrix)
                if (qclip) qclip[idx] = quantize_feature(o, in_scale, in_zp);
            };
            if constexpr (WIDE) cmvn_columns<17, 20>(sm.mel, MELS, sm.u.map, offt, lane, nfr, ncep, prow, P.win_size, emit);
            else cmvn_columns<13, 16>(sm.mel, MELS, sm.u.map, offt, lane, nfr, ncep, prow, P.win_size, emit);
        }
        WAVE_SYNC();
        PH(7);
    }
    if (PROF && blockIdx.x == 0 && threadIdx.x == 0 && prof_out)
        for (int i = 0; i < KWS_NPHASE; ++i) prof_out[i] = ph[i];
}

// ---------------------------------------------------------------------------------------------------------
//  gemmlowp / TFLite fixed-point helpers (fixedpoint.h:329-368, TFL/kernels/internal/common.h:138-162)
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int srdhm(int a, int b)
{
    const bool overflow = (a == b) && (a == (int)0x80000000);
    const long long ab = (long long)a * (long long)b;
    const int nudge = ab >= 0 ? (1 << 30) : (1 - (1 << 30));
    const int hi = (int)((ab + nudge) / (1ll << 31));          // truncating division
    return overflow ? 0x7fffffff : hi;
}
__device__ __forceinline__ int rdivpot(int x, int e)
{
    const int mask = (int)((1ll << e) - 1);
    const int rem = x & mask;
    const int thr = (mask >> 1) + (x < 0 ? 1 : 0);
    return (x >> e) + (rem > thr ? 1 : 0);
}
__device__ __forceinline__ int mbqm(int x, int mult, int shift)
{
    const int ls = shift > 0 ? shift : 0, rs = shift > 0 ? 0 : -shift;
    return rdivpot(srdhm((int)((unsigned)x << ls), mult), rs);
}
__device__ __forceinline__ int sat_shl(int x, int e)
{
    const int thr = (int)((1u << (31 - e)) - 1);
    if (x > thr) return 0x7fffffff;
    if (x < -thr) return (int)0x80000000;
    return x << e;
}
__device__ __forceinline__ int one_over_one_plus_x(int a)     // fixedpoint.h:842-862
{
    const long long sum = (long long)a + 0x7fffffffll;
    const int half_den = (int)((sum + (sum >= 0 ? 1 : -1)) / 2);
    int x = (int)(1515870810u + (unsigned)srdhm(half_den, -1010580540));
    for (int i = 0; i < 3; ++i) {
        const int hdx = srdhm(half_den, x);
        const int one_minus = (int)((1u << 29) - (unsigned)hdx);
        x = (int)((unsigned)x + (unsigned)sat_shl(srdhm(x, one_minus), 2));
    }
    return sat_shl(x, 1);
}

// ---------------------------------------------------------------------------------------------------------
//  Kernel 2: the int8 CNN.  4 waves per workgroup share the weights and the ADD look-up tables in LDS; each wave
//  owns one clip.  conv accumulators are exact int32 (v_dot4_i32_i8); because requantisation, the folded
//  ADD+ReLU table and the clamps are all monotonically non-decreasing, max-pooling is applied to the raw
//  accumulators first (max commutes with a non-decreasing map), then ONE requantisation per pooled output.
// ---------------------------------------------------------------------------------------------------------
constexpr int KWS_NN_WAVES = 4;
constexpr int KWS_POOL_MAX = 8;
// rows of a block's padded int8 input image in the generic kernel: un-pooled blocks are walked KWS_POOL_MAX time steps at
// a time, so reads reach up to ceil(out_w / 8) * 8 + taps - 1
__host__ __device__ inline int nn_rows(const KwsConvBlock &k) { return max(k.in_w, (k.out_w + KWS_POOL_MAX - 1) & ~(KWS_POOL_MAX - 1)) + k.taps; }

struct NnTaps {            // optional debug outputs for the parity tests (all int8, per clip)
    int8_t *pooled;        // concatenation of every block's pooled output [pool_w][out_c]
    int pooled_stride;
    int8_t *fc;            // [fc_out]
    int8_t *out_q;         // [n_labels]
};

// FULLY_CONNECTED (integer_ops/fully_connected.h:23-63) + SOFTMAX int8->int8 (reference/softmax.h:66-144) for one clip.
// vec: per-wave LDS scratch; bytes [0,64) hold the last pooled vector (int8), ints [16, 16+fc_out) receive the logits.
__device__ __forceinline__ void nn_head(const KwsNnPlan &N, int *vec, int lane, int clip, float *__restrict__ scores,
                                        const NnTaps &taps)
{
    // ---- FULLY_CONNECTED (integer_ops/fully_connected.h:23-63): input = last pooled vector ------------------
    const int8_t *xin = (const int8_t *)vec;
    int logit = 0;
    if (lane < N.fc_out) {
        int acc = 0;
        for (int d = 0; d < N.fc_in; ++d)
            acc += ((int)N.fc_w[lane * N.fc_in + d] + N.fc_w_off) * ((int)xin[d] + N.fc_in_off);
        acc += N.fc_bias[lane];
        acc = mbqm(acc, N.fc_mult, N.fc_shift) + N.fc_out_zp;
        logit = min(max(acc, N.fc_act_min), N.fc_act_max);
    }
    WAVE_SYNC();
    int *lg = vec + 16;     // logits as int32, after the (<=64 byte) pooled vector
    if (lane < N.fc_out) {
        lg[lane] = logit;
        if (taps.fc) taps.fc[(size_t)clip * N.fc_out + lane] = (int8_t)logit;
    }
    WAVE_SYNC();
    // ---- SOFTMAX int8 -> int8 (reference/softmax.h:66-144), every lane < n_labels redundantly ---------------
    if (lane < N.fc_out) {
        int mx = -128;
        for (int c = 0; c < N.fc_out; ++c) mx = max(mx, lg[c]);
        int sum = 0;
        for (int c = 0; c < N.fc_out; ++c) {
            const int d = mx - lg[c];
            if (N.sm_valid[d]) sum = (int)((unsigned)sum + (unsigned)rdivpot(N.sm_exp[d], 12));
        }
        const int hp1 = sum ? __clz(sum) : 32;                                  // GetReciprocal, common.h:530-546
        const int nbits = 12 - hp1;
        const int ssm1 = (int)(((unsigned)sum << hp1) - (1u << 31));
        const int scale = one_over_one_plus_x(ssm1);
        const int d = mx - logit;
        int o = -128;
        if (N.sm_valid[d]) {
            const int unsat = rdivpot(srdhm(scale, N.sm_exp[d]), nbits + 31 - 8);
            o = min(max(unsat - 128, -128), 127);
        }
        if (taps.out_q) taps.out_q[(size_t)clip * N.fc_out + lane] = (int8_t)o;
        scores[(size_t)clip * N.fc_out + lane] = (float)(o - N.out_zp) * N.out_scale;   // ei_run_classifier.h:470
    }
    WAVE_SYNC();
}

__global__ __launch_bounds__(KWS_WAVE * KWS_NN_WAVES) void kws_nn_kernel(KwsNnPlan N, const int8_t *__restrict__ q_in, int n_clips,
                                                                         float *__restrict__ scores, NnTaps taps)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    // ---- shared: weights + ADD tables of every block ---------------------------------------------------------
    unsigned char *sp = smem_raw;
    const int8_t *s_w[KWS_MAX_BLOCKS];
    const int8_t *s_lut[KWS_MAX_BLOCKS];
    int act_bytes = 0;
    for (int b = 0; b < N.n_blocks; ++b) {
        const KwsConvBlock &k = N.blk[b];
        const int wbytes = k.w_bytes;
        for (int i = threadIdx.x * 4; i < wbytes; i += blockDim.x * 4) *(int *)(sp + i) = *(const int *)(k.w + i);
        s_w[b] = (const int8_t *)sp;
        sp += (wbytes + 15) & ~15;
        const int lbytes = k.has_lut ? k.out_c * 256 : 0;
        for (int i = threadIdx.x * 4; i < lbytes; i += blockDim.x * 4) *(int *)(sp + i) = *(const int *)(k.add_lut + i);
        s_lut[b] = (const int8_t *)sp;
        sp += lbytes;
        const int ab = nn_rows(k) * k.in_cpad;
        act_bytes = max(act_bytes, ab);
    }
    act_bytes = (act_bytes + 15) & ~15;
    // per wave: two activation buffers (ping-pong) + a small vector for FC/softmax
    int8_t *actA = (int8_t *)(sp + wave * (2 * act_bytes + 64 * 4));
    int8_t *actB = actA + act_bytes;
    int *vec = (int *)(actB + act_bytes);
    __syncthreads();

    const int F = N.n_features;
    for (int clip = blockIdx.x * KWS_NN_WAVES + wave; clip < n_clips; clip += gridDim.x * KWS_NN_WAVES) {
        // ---- stage the int8 input as [pad_left + t][in_cpad], padding = zero point ((x + offset) == 0) -------
        {
            const KwsConvBlock &k = N.blk[0];
            const int rows = nn_rows(k);
            const int zp4 = (int)((unsigned)(k.in_zp & 0xff) * 0x01010101u);
            for (int i = lane * 4; i < rows * k.in_cpad; i += 64 * 4) *(int *)(actA + i) = zp4;
            WAVE_SYNC();
            const int8_t *src = q_in + (size_t)clip * F;
            for (int i = lane; i < k.in_w * k.in_c; i += 64) {
                const int tt = i / k.in_c, c = i - tt * k.in_c;
                actA[(tt + k.pad_left) * k.in_cpad + c] = src[i];
            }
            WAVE_SYNC();
        }
        int8_t *cur = actA, *nxt = actB;
        int pooled_off = 0;
        for (int b = 0; b < N.n_blocks; ++b) {
            const KwsConvBlock &k = N.blk[b];
            const bool last = (b + 1 == N.n_blocks);
            const int nrows = last ? 0 : nn_rows(N.blk[b + 1]);
            const int ncp = last ? k.out_c : N.blk[b + 1].in_cpad;
            const int npl = last ? 0 : N.blk[b + 1].pad_left;
            if (!last) {
                const int zp4 = (int)((unsigned)(N.blk[b + 1].in_zp & 0xff) * 0x01010101u);
                for (int i = lane * 4; i < nrows * ncp; i += 64 * 4) *(int *)(nxt + i) = zp4;
                WAVE_SYNC();
            }
            const int n_out = k.pool_w * k.out_c;
            // requantise + folded ADD/ReLU of one pooled accumulator, store (integer_ops/conv.h:111-116, add.h)
            auto finish = [&](int m, int pw, int oc) {
                m += k.bias_eff[oc];
                int r = mbqm(m, k.mult[oc], k.shift[oc]) + k.out_zp;
                r = min(max(r, k.act_min), k.act_max);
                const int8_t o = k.has_lut ? s_lut[b][oc * 256 + (r + 128)] : (int8_t)r;
                const int idx = pw * k.out_c + oc;
                if (last) ((int8_t *)vec)[idx] = o;
                else nxt[(npl + pw) * ncp + oc] = o;
                if (taps.pooled) taps.pooled[(size_t)clip * taps.pooled_stride + pooled_off + idx] = o;
            };
            if (k.depthwise) {                           // integer_ops/depthwise_conv.h:64-103: one input channel per output
                const int tp4 = (k.taps + 3) & ~3;
                for (int idx = lane; idx < n_out; idx += 64) {
                    const int pw = idx / k.out_c, oc = idx - pw * k.out_c;
                    const int t0 = pw * k.pool_stride;
                    int acc[KWS_POOL_MAX];
#pragma unroll
                    for (int i = 0; i < KWS_POOL_MAX; ++i) acc[i] = 0;
                    const int8_t *wrow = s_w[b] + oc * tp4;
                    const int8_t *xcol = cur + t0 * k.in_cpad + oc / k.depth_mult;
                    for (int tap = 0; tap < k.taps; ++tap) {
                        const int wv = wrow[tap];
#pragma unroll
                        for (int i = 0; i < KWS_POOL_MAX; ++i)
                            if (i < k.pool) acc[i] += wv * (int)xcol[(i + tap) * k.in_cpad];
                    }
                    int m = (int)0x80000000;
#pragma unroll
                    for (int i = 0; i < KWS_POOL_MAX; ++i)
                        if (i < k.pool && t0 + i < k.out_w) m = max(m, acc[i]);
                    finish(m, pw, oc);
                }
            } else {
                // a lane owns one pooling window of OB = 2 (or 1) output channels: every 16-byte activation read feeds
                // 4 * OB dot products, every 16-byte weight read `pool` of them
                // (an un-pooled block is walked in groups of KWS_POOL_MAX time steps, each stored on its own)
                const bool pooled = k.pool > 1;
                const int tb = pooled ? k.pool : KWS_POOL_MAX, tstride = pooled ? k.pool_stride : KWS_POOL_MAX;
                const int n_tb = pooled ? k.pool_w : (k.out_w + KWS_POOL_MAX - 1) / KWS_POOL_MAX;
                const int ob = (n_tb * k.out_c > 64 && (k.out_c & 1) == 0) ? 2 : 1;
                const int n_ocb = k.out_c / ob;
                const int c16n = k.in_cpad >> 4;
                for (int item = lane; item < n_tb * n_ocb; item += 64) {
                    const int pw = item / n_ocb, oc0 = (item - pw * n_ocb) * ob;
                    const int t0 = pw * tstride;
                    int acc[KWS_POOL_MAX][2];
#pragma unroll
                    for (int i = 0; i < KWS_POOL_MAX; ++i) acc[i][0] = acc[i][1] = 0;
                    const int8_t *w0 = s_w[b] + (size_t)oc0 * k.taps * k.in_cpad;
                    const int8_t *w1 = w0 + (ob == 2 ? k.taps * k.in_cpad : 0);
                    for (int tap = 0; tap < k.taps; ++tap) {
                        const int8_t *xrow = cur + (t0 + tap) * k.in_cpad;
                        for (int c16 = 0; c16 < c16n; ++c16) {
                            const int4 wa = *(const int4 *)(w0 + tap * k.in_cpad + 16 * c16);
                            const int4 wb = *(const int4 *)(w1 + tap * k.in_cpad + 16 * c16);
#pragma unroll
                            for (int i = 0; i < KWS_POOL_MAX; ++i) {
                                if (i < tb) {
                                    const int4 xv = *(const int4 *)(xrow + i * k.in_cpad + 16 * c16);
                                    int a = acc[i][0], c = acc[i][1];
                                    a = __builtin_amdgcn_sdot4(wa.x, xv.x, a, false);
                                    a = __builtin_amdgcn_sdot4(wa.y, xv.y, a, false);
                                    a = __builtin_amdgcn_sdot4(wa.z, xv.z, a, false);
                                    a = __builtin_amdgcn_sdot4(wa.w, xv.w, a, false);
                                    if (ob == 2) {
                                        c = __builtin_amdgcn_sdot4(wb.x, xv.x, c, false);
                                        c = __builtin_amdgcn_sdot4(wb.y, xv.y, c, false);
                                        c = __builtin_amdgcn_sdot4(wb.z, xv.z, c, false);
                                        c = __builtin_amdgcn_sdot4(wb.w, xv.w, c, false);
                                    }
                                    acc[i][0] = a; acc[i][1] = c;
                                }
                            }
                        }
                    }
                    for (int o = 0; o < ob; ++o) {
                        if (pooled) {
                            int m = (int)0x80000000;
#pragma unroll
                            for (int i = 0; i < KWS_POOL_MAX; ++i)
                                if (i < tb && t0 + i < k.out_w) m = max(m, acc[i][o]);
                            finish(m, pw, oc0 + o);
                        } else {
#pragma unroll
                            for (int i = 0; i < KWS_POOL_MAX; ++i)
                                if (t0 + i < k.out_w) finish(acc[i][o], t0 + i, oc0 + o);
                        }
                    }
                }
            }
            pooled_off += n_out;
            WAVE_SYNC();
            int8_t *tmp = cur; cur = nxt; nxt = tmp;
        }
        nn_head(N, vec, lane, clip, scores, taps);
    }
}

// ---------------------------------------------------------------------------------------------------------
//  Kernel 2, matrix-core path for the shipped graph shape (two conv blocks: time<=64 x 16ch x <=8 taps -> <=32 ch,
//  pool 7/7; then time<=16 x 32ch x <=8 taps -> <=16 ch, global pool).  The 1xK convolutions are genuine
//  contractions: per clip  [time x (taps*16)] x [(taps*16) x out_c]  on v_mfma_i32_32x32x32_i8 (8 per clip) and
//  [time x (taps*32)] x [(taps*32) x out_c] on v_mfma_i32_16x16x64_i8 (4 per clip).  int32 accumulation is exact, so
//  the result is bit-identical to the reference's scalar loops whatever the summation order.  One wave per clip;
//  weight fragments stay in registers for the whole launch; activations are read from LDS as aligned 16-byte rows.
// ---------------------------------------------------------------------------------------------------------
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int KWS_A1_ROWS = 72;    // >= 63 + 8 + 1 rows of 16 B: activations of block 1, row = time + tap
constexpr int KWS_A2_ROWS = 24;    // >= 15 + 8 + 1 rows of 32 B
constexpr int KWS_MFMA_POOL = 7;

// per-wave constants of the matrix-core path, fixed for the whole launch
template <int CP>               // bytes (= padded channels) per activation row of block 1: 16, or 64 for up to 64 input channels
struct NnMfmaCtx {
    static constexpr int KS1 = CP == 16 ? 4 : 16;      // k-steps of 32 for block 1: 2 taps per step, or 2 steps per tap
    v4i wb1[KS1], wb2[4];        // weight fragments
    int b1, m1, sh1, b2, m2, sh2;
    bool oc1_ok, oc2_ok;
};

// Weight fragments.  Block 1, CP = 16: k-slot (h, j) of k-step s is tap 2s+h, channel j; CP = 64: k-step s is tap s>>1,
// channels 32*(s&1) + 16*h + j (channels beyond the model's padded count are zero weights).  Block 2: the 16-byte group
// G = 4s+g of k-step s is tap G>>1, channel half G&1.  A and B use the same slot->k map, so the instruction's internal
// ordering of k is irrelevant.
template <int CP>
__device__ __forceinline__ void nn_mfma_init(NnMfmaCtx<CP> &c, const KwsNnPlan &N, int lane)
{
    const KwsConvBlock &k1 = N.blk[0], &k2 = N.blk[1];
    const int oc = lane & 31, h = lane >> 5;
#pragma unroll
    for (int s = 0; s < NnMfmaCtx<CP>::KS1; ++s) {
        const int tap = CP == 16 ? 2 * s + h : s >> 1;
        const int ch = CP == 16 ? 0 : 32 * (s & 1) + 16 * h;
        v4i w = { 0, 0, 0, 0 };
        if (oc < k1.out_c && tap < k1.taps && ch < k1.in_cpad) w = *(const v4i *)(k1.w + ((size_t)oc * k1.taps + tap) * k1.in_cpad + ch);
        c.wb1[s] = w;
    }
    const int oc2 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int G = 4 * s + g, tap = G >> 1, ch = G & 1;
        v4i w = { 0, 0, 0, 0 };
        if (oc2 < k2.out_c && tap < k2.taps) w = *(const v4i *)(k2.w + ((size_t)oc2 * k2.taps + tap) * 32 + ch * 16);
        c.wb2[s] = w;
    }
    c.oc1_ok = oc < k1.out_c;
    c.b1 = c.oc1_ok ? k1.bias_eff[oc] : 0; c.m1 = c.oc1_ok ? k1.mult[oc] : 0; c.sh1 = c.oc1_ok ? k1.shift[oc] : 0;
    c.oc2_ok = oc2 < k2.out_c;
    c.b2 = c.oc2_ok ? k2.bias_eff[oc2] : 0; c.m2 = c.oc2_ok ? k2.mult[oc2] : 0; c.sh2 = c.oc2_ok ? k2.shift[oc2] : 0;
}

// padding rows/columns of the activation buffers hold the input zero point ((x + input_offset) == 0) for the whole launch
template <int CP>
__device__ __forceinline__ void nn_mfma_fill_padding(const KwsNnPlan &N, int8_t *act1, int8_t *act2, int lane)
{
    const int z1 = (int)((unsigned)(N.blk[0].in_zp & 0xff) * 0x01010101u), z2 = (int)((unsigned)(N.blk[1].in_zp & 0xff) * 0x01010101u);
    for (int i = lane; i < KWS_A1_ROWS * (CP / 4); i += 64) ((int *)act1)[i] = z1;
    for (int i = lane; i < KWS_A2_ROWS * 8; i += 64) ((int *)act2)[i] = z2;
}

// One clip through both conv blocks, FC and softmax; act1 already holds the int8 input rows.
template <int CP>
__device__ __forceinline__ void nn_mfma_clip(const NnMfmaCtx<CP> &c, const KwsNnPlan &N, const int8_t *act1, int8_t *act2, int *vec,
                                             const int8_t *s_lut1, const int8_t *s_lut2, int lane, int clip,
                                             float *__restrict__ scores, const NnTaps &taps)
{
    const KwsConvBlock &k1 = N.blk[0], &k2 = N.blk[1];
    const int oc1 = lane & 31, hh = lane >> 5, oc2 = lane & 15, g4 = lane >> 4;
    // ---- conv 1: two 32-row tiles x KS1 k-steps ---------------------------------------------------------------
    v16i acc0 = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 }, acc1 = acc0;
#pragma unroll
    for (int s = 0; s < NnMfmaCtx<CP>::KS1; ++s) {
        const int tap = CP == 16 ? 2 * s + hh : s >> 1;
        const int ch = CP == 16 ? 0 : 32 * (s & 1) + 16 * hh;
        const v4i a0 = *(const v4i *)(act1 + (oc1 + tap) * CP + ch);              // row = time (lane&31) + tap
        const v4i a1 = *(const v4i *)(act1 + (32 + oc1 + tap) * CP + ch);
        acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, c.wb1[s], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, c.wb1[s], acc1, 0, 0, 0);
    }
    // ---- max-pool 7/7 on the raw accumulators (monotone requantisation, see the scalar kernel) --------------
    // accumulator register r of a 32x32 tile holds row (r&3) + 8*(r>>2) + 4*(lane>>5), column lane&31
    int pm[KWS_MFMA_POOL];
#pragma unroll
    for (int i = 0; i < KWS_MFMA_POOL; ++i) pm[i] = (int)0x80000000;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int v = mt ? acc1[r] : acc0[r];
            const int t0 = 32 * mt + (r & 3) + 8 * (r >> 2), t1 = t0 + 4;       // time if lane>>5 is 0 / 1
            if (t0 / KWS_MFMA_POOL < KWS_MFMA_POOL) pm[t0 / KWS_MFMA_POOL] = max(pm[t0 / KWS_MFMA_POOL], hh == 0 ? v : (int)0x80000000);
            if (t1 / KWS_MFMA_POOL < KWS_MFMA_POOL) pm[t1 / KWS_MFMA_POOL] = max(pm[t1 / KWS_MFMA_POOL], hh == 1 ? v : (int)0x80000000);
        }
    }
#pragma unroll
    for (int i = 0; i < KWS_MFMA_POOL; ++i) pm[i] = max(pm[i], __shfl_xor(pm[i], 32));
    // requantise + ADD/ReLU table: half-wave 0 takes pooled rows 0..3, half-wave 1 rows 4..6
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pw = i + 4 * hh;
        const int m = (hh == 0) ? pm[i] : pm[(i + 4 < KWS_MFMA_POOL) ? i + 4 : 0];
        if (c.oc1_ok && pw < k1.pool_w) {
            int rq = mbqm(m + c.b1, c.m1, c.sh1) + k1.out_zp;
            rq = min(max(rq, k1.act_min), k1.act_max);
            const int8_t o = s_lut1[oc1 * 256 + (rq + 128)];
            act2[(pw + k2.pad_left) * 32 + oc1] = o;
            if (taps.pooled) taps.pooled[(size_t)clip * taps.pooled_stride + pw * k1.out_c + oc1] = o;
        }
    }
    WAVE_SYNC();
    // ---- conv 2: one 16-row tile x four k-steps of 64 ----------------------------------------------------------
    v4i c2 = { 0, 0, 0, 0 };
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int G = 4 * s + g4;
        const v4i a = *(const v4i *)(act2 + (oc2 + (G >> 1)) * 32 + (G & 1) * 16);  // row = time (lane&15) + tap
        c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, c.wb2[s], c2, 0, 0, 0);
    }
    // accumulator register r of a 16x16 tile holds row 4*(lane>>4) + r, column lane&15; global max-pool over time
    int pm2 = (int)0x80000000;
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (4 * g4 + r < k2.out_w) pm2 = max(pm2, c2[r]);
    pm2 = max(pm2, __shfl_xor(pm2, 16));
    pm2 = max(pm2, __shfl_xor(pm2, 32));
    if (lane < 16 && c.oc2_ok) {
        int rq = mbqm(pm2 + c.b2, c.m2, c.sh2) + k2.out_zp;
        rq = min(max(rq, k2.act_min), k2.act_max);
        const int8_t o = s_lut2[oc2 * 256 + (rq + 128)];
        ((int8_t *)vec)[oc2] = o;
        if (taps.pooled) taps.pooled[(size_t)clip * taps.pooled_stride + k1.pool_w * k1.out_c + oc2] = o;
    }
    WAVE_SYNC();
    nn_head(N, vec, lane, clip, scores, taps);
}

template <int CP>
__global__ __launch_bounds__(KWS_WAVE * KWS_NN_WAVES) void kws_nn_mfma_kernel(KwsNnPlan N, const int8_t *__restrict__ q_in, int n_clips,
                                                                              float *__restrict__ scores, NnTaps taps)
{
    __shared__ __attribute__((aligned(16))) int8_t s_lut1[32 * 256];
    __shared__ __attribute__((aligned(16))) int8_t s_lut2[16 * 256];
    __shared__ __attribute__((aligned(16))) int8_t s_act1[KWS_NN_WAVES][KWS_A1_ROWS * CP];
    __shared__ __attribute__((aligned(16))) int8_t s_act2[KWS_NN_WAVES][KWS_A2_ROWS * 32];
    __shared__ int s_vec[KWS_NN_WAVES][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const KwsConvBlock &k1 = N.blk[0], &k2 = N.blk[1];
    for (int i = threadIdx.x * 4; i < k1.out_c * 256; i += blockDim.x * 4) *(int *)(s_lut1 + i) = *(const int *)(k1.add_lut + i);
    for (int i = threadIdx.x * 4; i < k2.out_c * 256; i += blockDim.x * 4) *(int *)(s_lut2 + i) = *(const int *)(k2.add_lut + i);
    int8_t *act1 = s_act1[wave], *act2 = s_act2[wave];
    nn_mfma_fill_padding<CP>(N, act1, act2, lane);
    NnMfmaCtx<CP> ctx;
    nn_mfma_init<CP>(ctx, N, lane);
    __syncthreads();
    const int F = N.n_features;
    for (int clip = blockIdx.x * KWS_NN_WAVES + wave; clip < n_clips; clip += gridDim.x * KWS_NN_WAVES) {
        // ---- int8 input tensor [time][in_c] -> LDS rows of CP bytes at row (time + pad_left) --------------------
        const int8_t *src = q_in + (size_t)clip * F;
        for (int i = lane; i < F; i += 64) {
            const int tt = i / k1.in_c, c = i - tt * k1.in_c;
            act1[(tt + k1.pad_left) * CP + c] = src[i];
        }
        WAVE_SYNC();
        nn_mfma_clip<CP>(ctx, N, act1, act2, s_vec[wave], s_lut1, s_lut2, lane, clip, scores, taps);
    }
}

// ---------------------------------------------------------------------------------------------------------
//  Kernel 2': cmvnw (processing.hpp:326-389) + input quantisation (ei_run_classifier.h:436-444) [+ the network when
//  FUSE and the graph fits the matrix-core path].  One wave per window, 4 waves per workgroup.
//  CMVN: cmvn_columns<13, 16> (shared with kws_mfcc_kernel).
// ---------------------------------------------------------------------------------------------------------
template <bool FUSE>
__global__ __launch_bounds__(KWS_WAVE * KWS_NN_WAVES, 2) void kws_cmvn_nn_kernel(KwsDspPlan P, KwsNnPlan N, const float *__restrict__ mfcc,
                                                                              int n_clips, float *__restrict__ features,
                                                                              int8_t *__restrict__ q_out, float *__restrict__ scores,
                                                                              NnTaps taps)
{
    __shared__ int s_map[KWS_MAXPROW];                                    // numpy::pad_1d_symmetric row map (numpy.hpp:479-541)
    __shared__ float s_mfcc[KWS_NN_WAVES][KWS_MAXF * KWS_NF_MAX];       // cepstra before CMVN, [frame][coef] (coef <= filters)
    __shared__ __attribute__((aligned(16))) int s_off[KWS_NN_WAVES][2 * KWS_ZF];   // cmvn_columns' row-offset table (same bound as in kws_mfcc_kernel)
    __shared__ __attribute__((aligned(16))) int8_t s_lut1[FUSE ? 32 * 256 : 16];
    __shared__ __attribute__((aligned(16))) int8_t s_lut2[FUSE ? 16 * 256 : 16];
    __shared__ __attribute__((aligned(16))) int8_t s_act1[KWS_NN_WAVES][FUSE ? KWS_A1_ROWS * 16 : 16];
    __shared__ __attribute__((aligned(16))) int8_t s_act2[KWS_NN_WAVES][FUSE ? KWS_A2_ROWS * 32 : 16];
    __shared__ int s_vec[KWS_NN_WAVES][FUSE ? 64 : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nfr = P.n_frames, ncep = P.n_cepstral, nfeat = nfr * ncep;
    const int prow = nfr + 2 * P.pad;
    for (int i = threadIdx.x; i < prow; i += blockDim.x) s_map[i] = P.pad_map[i];
    NnMfmaCtx<16> ctx;
    int8_t *act1 = s_act1[wave], *act2 = s_act2[wave];
    if constexpr (FUSE) {
        const KwsConvBlock &k1 = N.blk[0], &k2 = N.blk[1];
        for (int i = threadIdx.x * 4; i < k1.out_c * 256; i += blockDim.x * 4) *(int *)(s_lut1 + i) = *(const int *)(k1.add_lut + i);
        for (int i = threadIdx.x * 4; i < k2.out_c * 256; i += blockDim.x * 4) *(int *)(s_lut2 + i) = *(const int *)(k2.add_lut + i);
        nn_mfma_fill_padding<16>(N, act1, act2, lane);
        nn_mfma_init<16>(ctx, N, lane);
    }
    __syncthreads();
    float *mf = s_mfcc[wave];
    const int win = P.win_size;
    const float in_scale = N.in_scale;
    const int in_zp = N.in_zp;

    for (int clip = blockIdx.x * KWS_NN_WAVES + wave; clip < n_clips; clip += gridDim.x * KWS_NN_WAVES) {
        const float *src = mfcc + (size_t)clip * nfeat;
        for (int i = lane; i < nfeat; i += 64) mf[i] = src[i];
        WAVE_SYNC();
        cmvn_columns<13, 16>(mf, ncep, s_map, s_off[wave], lane, nfr, ncep, prow, win, [&](int row, int c, float o) {
            const int idx = row * ncep + c;
            if (features) features[(size_t)clip * nfeat + idx] = o;
            const int8_t qb = quantize_feature(o, in_scale, in_zp);
            if (q_out) q_out[(size_t)clip * nfeat + idx] = qb;
            if constexpr (FUSE) act1[(row + N.blk[0].pad_left) * 16 + c] = qb;
        });
        WAVE_SYNC();
        if constexpr (FUSE) nn_mfma_clip<16>(ctx, N, act1, act2, s_vec[wave], s_lut1, s_lut2, lane, clip, scores, taps);
    }
}

// the matrix-core kernel covers this graph shape; anything else runs on kws_nn_kernel
static bool nn_fits_mfma(const KwsNnPlan &N)
{
    if (N.n_blocks != 2) return false;
    const KwsConvBlock &a = N.blk[0], &b = N.blk[1];
    if (a.depthwise || b.depthwise) return false;
    return (a.in_cpad == 16 || a.in_cpad <= 64) && a.taps <= 8 && a.out_c <= 32 && a.in_w <= 64 && a.pool == KWS_MFMA_POOL && a.pool_stride == KWS_MFMA_POOL &&
           a.pool_w <= KWS_MFMA_POOL && b.in_cpad == 32 && b.taps <= 8 && b.out_c <= 16 && b.in_w <= 16 && b.pool_w == 1 &&
           b.pool >= b.out_w && N.fc_in == b.out_c;
}

// ---------------------------------------------------------------------------------------------------------
//  Kernel 2f: float32 models.  One wave per clip.  Every accumulation replays the reference's sequential
//  `total += input * filter` order (tap outer, channel inner; product and sum rounded separately -- the build has
//  -ffp-contract=off), so everything up to the logits is bit-identical to the float TFLite-Micro kernels; only softmax's
//  expf is the device's.  Zero-padded activation rows stand in for the reference's skipped out-of-image taps: they add
//  an exact 0 (x*0 = +-0, and total + (+-0) == total for every total the chain can hold, +0 included).
//
//  The 148 k multiply-adds per clip are order-constrained only WITHIN one output's chain, so a lane runs TB x OB chains
//  (TB consecutive time steps x OB consecutive output channels) side by side: per chain step it reads TB activations and
//  one OB-wide weight vector from LDS for TB*OB independent mul+add pairs (packed v_pk_mul_f32 / v_pk_add_f32).  The plan
//  picks (TB, OB) per conv block so that the work items fill the 64 lanes (conv1 49x30: 7x4 -> 56 lanes, 91 steps;
//  conv2 7x10: 1x2 -> 35 lanes, 210 steps).
//  LDS: weights transposed to [tap*in_c + c][out_c padded to 4] once per workgroup; per wave X (block input, rows
//  zero-padded), Y (conv+bias+ADD output before pooling) and a small vector for the FULLY_CONNECTED input / logits.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float act_clamp(float x, float lo, float hi)   // ActivationFunctionWithMinMax
{
    const float a = x < lo ? lo : x;
    return hi < a ? hi : a;
}

// how a block's outputs leave the conv: pooled in registers (fused_pool), written straight to the next image (no pooling
// node), or staged in Y for a separate pooling pass
__host__ __device__ __forceinline__ bool nnf_direct(const KwsConvBlockF32 &k) { return k.pool == 1; }
__host__ __device__ __forceinline__ bool nnf_staged(const KwsConvBlockF32 &k) { return !k.fused_pool && !nnf_direct(k); }
__host__ __device__ __forceinline__ int nnf_ntb(const KwsConvBlockF32 &k) { return k.fused_pool ? k.pool_w : (k.out_w + k.tb - 1) / k.tb; }
__host__ __device__ __forceinline__ int nnf_rows(const KwsConvBlockF32 &k)          // rows of the zero-padded input image
{
    const int a = k.pad_left + k.in_w, b = nnf_ntb(k) * k.tb + k.taps - 1;    // rows the blocked walk touches (k.tb is final)
    return a > b ? a : b;
}
__host__ __device__ __forceinline__ int nnf_ocp(const KwsConvBlockF32 &k) { return (k.out_c + 3) & ~3; }

// where a block's (pooled) output goes: the next block's zero-padded input image, or the FULLY_CONNECTED input vector
struct NnfDst { float *p; int row0, stride; };

template <int TB, int OB, bool VEC4>
__device__ __forceinline__ void nnf_conv(const KwsConvBlockF32 &k, const float *__restrict__ x, const float *__restrict__ wt,
                                         float *__restrict__ y, const NnfDst &dst, int lane)
{
    const int J = k.taps * k.in_c, ocp = nnf_ocp(k);
    const int n_ob = (k.out_c + OB - 1) / OB, n_tb = nnf_ntb(k);
    const bool direct = nnf_direct(k);
    for (int item = lane; item < n_tb * n_ob; item += 64) {
        const int tb = item / n_ob, ob = item - tb * n_ob;
        const int t0 = tb * TB, oc0 = ob * OB;
        float acc[TB][OB];
#pragma unroll
        for (int i = 0; i < TB; ++i)
#pragma unroll
            for (int o = 0; o < OB; ++o) acc[i][o] = 0.0f;
        const float *xp = x + t0 * k.in_c;          // rows are contiguous: x[(t+tap)*in_c + c] == x[t*in_c + (tap*in_c + c)]
        const float *wp = wt + oc0;
        // software pipeline: the TB activations and the OB-wide weight vector of step j+1 are in flight while step j's
        // TB*OB multiply-adds issue (LDS latency would otherwise be exposed once per step at 2 waves per SIMD)
        float wn[OB], xn[TB];
        auto load_w = [&](int j, float (&w)[OB]) {
            if constexpr (OB == 4) { const float4 v = *(const float4 *)(wp + j * ocp); w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; }
            else if constexpr (OB == 2) { const float2 v = *(const float2 *)(wp + j * ocp); w[0] = v.x; w[1] = v.y; }
            else w[0] = wp[j * ocp];
        };
        if constexpr (TB == 1) {
            // one time step per lane (small blocks): a chain step is a single multiply-add, so the loop is bound by the LDS
            // round trip unless many steps' operands are requested at once -- 8 steps per batch
            constexpr int U = 8;
            float wn_[U][OB], xn_[U];
            auto load_batch1 = [&](int j0) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int jj = min(j0 + u, J - 1);
                    load_w(jj, wn_[u]);
                    xn_[u] = xp[jj];
                }
            };
            if constexpr (VEC4) load_batch1(0);                     // 256-register build: batches double-buffered
            for (int j0 = 0; j0 < J; j0 += U) {
                float wu[U][OB], xu[U];
                if constexpr (!VEC4) load_batch1(j0);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    xu[u] = xn_[u];
#pragma unroll
                    for (int o = 0; o < OB; ++o) wu[u][o] = wn_[u][o];
                }
                if constexpr (VEC4) load_batch1(min(j0 + U, J - 1));   // the next batch is in flight during this one's chain
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (j0 + u < J) {
#pragma unroll
                        for (int o = 0; o < OB; ++o) {
                            const float prod = xu[u] * wu[u][o];
                            acc[0][o] += prod;
                        }
                    }
                }
            }
        } else if (VEC4 && (k.in_c & 3) == 0) {
            // channel counts that are multiples of 4 (rows 16-byte aligned): FOUR chain steps per batch -- one 16-byte read
            // per time row brings the activations of 4 consecutive steps, the batch after next is in flight meanwhile
            // (TB + 4 LDS instructions per 4 steps instead of 4 * (TB + 1)); needs the 256-register build of the kernel
            float4 xq[TB];
            float wq[4][OB];
            auto load_batch = [&](int j) {
#pragma unroll
                for (int u = 0; u < 4; ++u) load_w(j + u, wq[u]);
#pragma unroll
                for (int i = 0; i < TB; ++i) xq[i] = *(const float4 *)(xp + i * k.in_c + j);
            };
            load_batch(0);
            for (int j = 0; j < J; j += 4) {
                float4 xv[TB];
                float w[4][OB];
#pragma unroll
                for (int i = 0; i < TB; ++i) xv[i] = xq[i];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int o = 0; o < OB; ++o) w[u][o] = wq[u][o];
                load_batch(min(j + 4, J - 4));
#pragma unroll
                for (int u = 0; u < 4; ++u) {                      // chain order: steps j, j+1, j+2, j+3
#pragma unroll
                    for (int i = 0; i < TB; ++i) {
                        const float xs_ = u == 0 ? xv[i].x : u == 1 ? xv[i].y : u == 2 ? xv[i].z : xv[i].w;
#pragma unroll
                        for (int o = 0; o < OB; ++o) {
                            const float prod = xs_ * w[u][o];
                            acc[i][o] += prod;
                        }
                    }
                }
            }
        } else {
            load_w(0, wn);
#pragma unroll
            for (int i = 0; i < TB; ++i) xn[i] = xp[i * k.in_c];
            for (int j = 0; j < J; ++j) {
                float w[OB], xv[TB];
#pragma unroll
                for (int o = 0; o < OB; ++o) w[o] = wn[o];
#pragma unroll
                for (int i = 0; i < TB; ++i) xv[i] = xn[i];
                const int jn = min(j + 1, J - 1);
                load_w(jn, wn);
#pragma unroll
                for (int i = 0; i < TB; ++i) xn[i] = xp[i * k.in_c + jn];
#pragma unroll
                for (int i = 0; i < TB; ++i) {
#pragma unroll
                    for (int o = 0; o < OB; ++o) {
                        const float prod = xv[i] * w[o];
                        acc[i][o] += prod;
                    }
                }
            }
        }
#pragma unroll
        for (int o = 0; o < OB; ++o) {
            const int oc = oc0 + o;
            if (oc >= k.out_c) continue;
            const float bv = k.bias[oc], av = k.addc[oc];
            float mx = -FLT_MAX;
#pragma unroll
            for (int i = 0; i < TB; ++i) {
                if (t0 + i < k.out_w) {
                    float v = act_clamp(acc[i][o] + bv, k.conv_min, k.conv_max);
                    if (k.has_add) v = act_clamp(v + av, k.add_min, k.add_max);
                    if (k.fused_pool) mx = mx < v ? v : mx;             // MAX_POOL_2D: std::max(max, v), window order
                    else if (direct) dst.p[(dst.row0 + t0 + i) * dst.stride + oc] = v;
                    else y[(t0 + i) * k.out_c + oc] = v;
                }
            }
            if (k.fused_pool) dst.p[(dst.row0 + tb) * dst.stride + oc] = act_clamp(mx, k.pool_min, k.pool_max);
        }
    }
}

// DEPTHWISE_CONV_2D float (reference/depthwiseconv_float.h:25-97): the chain of output (t, oc) runs over the taps of input
// channel oc / depth_mult only.  A lane owns TB consecutive time steps of one output channel.
template <int TB>
__device__ __forceinline__ void nnf_dwconv(const KwsConvBlockF32 &k, const float *__restrict__ x, const float *__restrict__ wt,
                                           float *__restrict__ y, const NnfDst &dst, int lane)
{
    const int ocp = nnf_ocp(k), n_tb = nnf_ntb(k);
    const bool direct = nnf_direct(k);
    for (int item = lane; item < n_tb * k.out_c; item += 64) {
        const int tb = item / k.out_c, oc = item - tb * k.out_c;
        const int t0 = tb * TB;
        const float *xp = x + t0 * k.in_c + oc / k.depth_mult;
        float acc[TB];
#pragma unroll
        for (int i = 0; i < TB; ++i) acc[i] = 0.0f;
        for (int tap = 0; tap < k.taps; ++tap) {
            const float w = wt[tap * ocp + oc];
#pragma unroll
            for (int i = 0; i < TB; ++i) {
                const float prod = xp[(i + tap) * k.in_c] * w;
                acc[i] += prod;
            }
        }
        const float bv = k.bias[oc], av = k.addc[oc];
        float mx = -FLT_MAX;
#pragma unroll
        for (int i = 0; i < TB; ++i) {
            if (t0 + i < k.out_w) {
                float v = act_clamp(acc[i] + bv, k.conv_min, k.conv_max);
                if (k.has_add) v = act_clamp(v + av, k.add_min, k.add_max);
                if (k.fused_pool) mx = mx < v ? v : mx;
                else if (direct) dst.p[(dst.row0 + t0 + i) * dst.stride + oc] = v;
                else y[(t0 + i) * k.out_c + oc] = v;
            }
        }
        if (k.fused_pool) dst.p[(dst.row0 + tb) * dst.stride + oc] = act_clamp(mx, k.pool_min, k.pool_max);
    }
}

template <int TB, bool VEC4>
__device__ __forceinline__ void nnf_conv_ob(const KwsConvBlockF32 &k, const float *x, const float *wt, float *y, const NnfDst &dst, int lane)
{
    if (k.depthwise) nnf_dwconv<TB>(k, x, wt, y, dst, lane);
    else if (k.ob == 4) nnf_conv<TB, 4, VEC4>(k, x, wt, y, dst, lane);
    else if (k.ob == 2) nnf_conv<TB, 2, VEC4>(k, x, wt, y, dst, lane);
    else nnf_conv<TB, 1, VEC4>(k, x, wt, y, dst, lane);
}

// LDS layout shared by host and device: weights of every block, then per wave the ping-pong input images A (even blocks) and
// B (odd blocks), the un-pooled conv output Y (only when some block cannot pool in registers) and 128 floats for FC/softmax
struct NnfLayout { int w_floats, a_floats, b_floats, y_floats; };
__host__ __device__ __forceinline__ NnfLayout nnf_layout(const KwsNnPlanF32 &N)
{
    NnfLayout L = { 0, 0, 0, 0 };
    for (int b = 0; b < N.n_blocks; ++b) {
        const KwsConvBlockF32 &k = N.blk[b];
        L.w_floats += (k.depthwise ? k.taps : k.taps * k.in_c) * nnf_ocp(k);
        const int img = nnf_rows(k) * k.in_c;
        if (b & 1) L.b_floats = img > L.b_floats ? img : L.b_floats;
        else L.a_floats = img > L.a_floats ? img : L.a_floats;
        if (nnf_staged(k)) { const int yf = k.out_w * k.out_c; L.y_floats = yf > L.y_floats ? yf : L.y_floats; }
    }
    L.a_floats = (L.a_floats + 3) & ~3; L.b_floats = (L.b_floats + 3) & ~3; L.y_floats = (L.y_floats + 3) & ~3;
    return L;
}

template <int MAXT>     // threads per workgroup the build allows: 1024 (<= 128 VGPRs) or 512 (<= 256 VGPRs, vectorised conv steps)
__global__ __launch_bounds__(MAXT) void kws_nn_f32_kernel(const KwsNnPlanF32 *__restrict__ Np, const float *__restrict__ features,
                                                          int n_clips, float *__restrict__ scores,
                                                          float *__restrict__ tap_logits, long long *__restrict__ prof)
{
    // the plan is read from memory (scalar loads, any block index); by value in the kernel arguments the compiler copies it to
    // scratch as soon as a block is indexed dynamically
    const KwsNnPlanF32 &N = *Np;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    // development aid: shader-clock totals per phase of wave 0 of workgroup 0 (input, each block, head)
    const bool profiling = prof != nullptr && blockIdx.x == 0 && wave == 0;
    long long ph[KWS_MAX_BLOCKS + 2] = { 0 }, tlast = profiling ? clock64() : 0;
    auto mark = [&](int i) { if (profiling) { const long long now = clock64(); ph[i] += now - tlast; tlast = now; } };
    float *sp = (float *)smem_raw;
    const float *s_w[KWS_MAX_BLOCKS];
    for (int b = 0; b < N.n_blocks; ++b) {
        const KwsConvBlockF32 &k = N.blk[b];
        const int J = k.depthwise ? k.taps : k.taps * k.in_c, ocp = nnf_ocp(k);
        for (int i = threadIdx.x; i < J * ocp; i += blockDim.x) {      // [oc][j] -> [j][oc], zero in the padding channels
            const int j = i / ocp, oc = i - j * ocp;
            sp[i] = oc < k.out_c ? (k.depthwise ? k.w[j * k.out_c + oc] : k.w[oc * J + j]) : 0.0f;
        }
        s_w[b] = sp;
        sp += J * ocp;
    }
    const NnfLayout L = nnf_layout(N);
    float *A = sp + wave * (L.a_floats + L.b_floats + L.y_floats + 128);
    float *B = A + L.a_floats;
    float *Y = B + L.b_floats;
    float *vec = Y + L.y_floats;
    __syncthreads();

    for (int clip = blockIdx.x * n_waves + wave; clip < n_clips; clip += gridDim.x * n_waves) {
        {
            const KwsConvBlockF32 &k = N.blk[0];
            const int lo = k.pad_left * k.in_c, hi = lo + k.in_w * k.in_c, tot = nnf_rows(k) * k.in_c;
            const float *src = features + (size_t)clip * N.n_features;
            if (((lo | hi | N.n_features) & 3) == 0) {
                // 16-byte copies (the feature vector of a clip and its place in the image are both 16-byte aligned)
                const float4 *src4 = (const float4 *)src;
                float4 *A4 = (float4 *)A;
                const int lo4 = lo >> 2, hi4 = hi >> 2, tot4 = (tot + 3) >> 2;
                for (int i0 = lane; i0 < tot4; i0 += 64 * 4) {
                    float4 v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int i = i0 + 64 * u;
                        v[u] = (i >= lo4 && i < hi4) ? src4[i - lo4] : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int i = i0 + 64 * u;
                        if (i < tot4) A4[i] = v[u];
                    }
                }
            } else {
                // 8 loads per lane in flight (one at a time this stage is a chain of global-memory round trips)
                for (int i0 = lane; i0 < tot; i0 += 64 * 8) {
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int i = i0 + 64 * u;
                        v[u] = (i >= lo && i < hi) ? src[i - lo] : 0.0f;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int i = i0 + 64 * u;
                        if (i < tot) A[i] = v[u];
                    }
                }
            }
            WAVE_SYNC();
        }
        mark(0);
        for (int b = 0; b < N.n_blocks; ++b) {
            const KwsConvBlockF32 &k = N.blk[b];
            const bool last = (b + 1 == N.n_blocks);
            const float *cur = (b & 1) ? B : A;
            NnfDst dst;
            const int n_out = k.pool_w * k.out_c;
            if (last) { dst.p = vec; dst.row0 = 0; dst.stride = k.out_c; }
            else {
                const KwsConvBlockF32 &nk = N.blk[b + 1];
                dst.p = (b & 1) ? A : B; dst.row0 = nk.pad_left; dst.stride = nk.in_c;
                // the zero padding rows of the next block's input image (its real rows are written below)
                const int lo = nk.pad_left * nk.in_c, hi = lo + n_out, tot = nnf_rows(nk) * nk.in_c;
                for (int i = lane; i < tot; i += 64)
                    if (i < lo || i >= hi) dst.p[i] = 0.0f;
            }
            switch (k.tb) {
            case 8: nnf_conv_ob<8, (MAXT <= 512)>(k, cur, s_w[b], Y, dst, lane); break;
            case 7: nnf_conv_ob<7, (MAXT <= 512)>(k, cur, s_w[b], Y, dst, lane); break;
            case 4: nnf_conv_ob<4, (MAXT <= 512)>(k, cur, s_w[b], Y, dst, lane); break;
            case 2: nnf_conv_ob<2, (MAXT <= 512)>(k, cur, s_w[b], Y, dst, lane); break;
            default: nnf_conv_ob<1, (MAXT <= 512)>(k, cur, s_w[b], Y, dst, lane); break;
            }
            WAVE_SYNC();
            if (nnf_staged(k)) {
                // MAX_POOL_2D over time (pooling.h:189-237) from the staged conv output
                for (int idx = lane; idx < n_out; idx += 64) {
                    const int pw = idx / k.out_c, oc = idx - pw * k.out_c;
                    float mx = -FLT_MAX;
                    for (int q = 0; q < k.pool; ++q) {
                        const float v = Y[(pw * k.pool_stride + q) * k.out_c + oc];
                        mx = mx < v ? v : mx;                          // std::max(max, v)
                    }
                    dst.p[(dst.row0 + pw) * dst.stride + oc] = act_clamp(mx, k.pool_min, k.pool_max);
                }
                WAVE_SYNC();
            }
            mark(1 + b);
        }
        // FULLY_CONNECTED (fully_connected.h:26-60) + SOFTMAX (softmax.h:31-63)
        float *lg = vec + 64;
        if (lane < N.fc_out) {
            float total = 0.0f;
            for (int d = 0; d < N.fc_in; ++d) {
                const float prod = vec[d] * N.fc_w[lane * N.fc_in + d];
                total += prod;
            }
            const float lgt = act_clamp(total + N.fc_bias[lane], N.fc_min, N.fc_max);
            lg[lane] = lgt;
            if (tap_logits) tap_logits[(size_t)clip * N.fc_out + lane] = lgt;
        }
        WAVE_SYNC();
        if (lane < N.fc_out) {
            float mx = -FLT_MAX;
            for (int c = 0; c < N.fc_out; ++c) mx = mx < lg[c] ? lg[c] : mx;
            float sum = 0.0f;
            for (int c = 0; c < N.fc_out; ++c) sum += expf((lg[c] - mx) * N.beta);
            scores[(size_t)clip * N.fc_out + lane] = expf((lg[lane] - mx) * N.beta) / sum;
        }
        WAVE_SYNC();
        mark(1 + KWS_MAX_BLOCKS);
    }
    if (profiling && lane == 0)
        for (int i = 0; i < KWS_MAX_BLOCKS + 2; ++i) prof[i] = ph[i];
}

size_t kws_nn_f32_smem_bytes(const KwsNnPlanF32 &N, int n_waves)
{
    const NnfLayout L = nnf_layout(N);
    return ((size_t)L.w_floats + (size_t)n_waves * (L.a_floats + L.b_floats + L.y_floats + 128)) * sizeof(float);
}

// (TB, OB) of a conv block: fewest lane passes x chain work, LDS reads as the tie breaker
void kws_nn_f32_pick_blocking(KwsConvBlockF32 *k)
{
    static const int tbs[] = { 1, 2, 4, 7, 8 }, obs[] = { 1, 2, 4 };
    // a lane that owns exactly one pooling window can max-pool in registers (no staging buffer, fewer LDS bytes per wave);
    // taken when that blocking costs no more than the best free one
    const bool can_fuse = k->pool > 1 && k->pool == k->pool_stride &&
                          (k->pool == 2 || k->pool == 4 || k->pool == 7 || k->pool == 8);
    float best[2] = { 1e30f, 1e30f };
    int btb[2] = { 1, 1 }, bob[2] = { 1, 1 };
    for (int fused = 0; fused < 2; ++fused)
        for (int tb : tbs)
            for (int ob : obs) {
                if (k->depthwise && ob != 1) continue;
                if (fused && (!can_fuse || tb != k->pool)) continue;
                const int n_tb = fused ? k->pool_w : (k->out_w + tb - 1) / tb;
                const int items = n_tb * ((k->out_c + ob - 1) / ob);
                const int passes = (items + 63) / 64;
                const float cost = (float)passes * (2.0f * tb * ob + 1.0f * (tb + 1));
                if (cost < best[fused]) { best[fused] = cost; btb[fused] = tb; bob[fused] = ob; }
            }
    k->fused_pool = (can_fuse && best[1] <= best[0]) ? 1 : 0;
    k->tb = btb[k->fused_pool];
    k->ob = bob[k->fused_pool];
}

long long *kws_dev_f32_prof = nullptr;      // development aid: device buffer of KWS_MAX_BLOCKS + 2 phase counters, or NULL

// waves per workgroup: as many as fit the CU's 160 KB of LDS (they share one copy of the weights), at most 16
int kws_nn_f32_waves(const KwsNnPlanF32 &N)
{
    // a conv block with a channel count that is a multiple of 4 wants the 256-register build (16-byte activation reads),
    // which serves at most 8 waves per workgroup
    bool vec4 = false;
    for (int b = 0; b < N.n_blocks; ++b) vec4 |= !N.blk[b].depthwise && N.blk[b].tb > 1 && (N.blk[b].in_c & 3) == 0;
    for (int w = vec4 ? 8 : 16; w > 4; --w)
        if (kws_nn_f32_smem_bytes(N, w) <= 158 * 1024) return w;
    return 4;
}

int kws_launch_nn_f32(const KwsNnPlanF32 &N, const KwsNnPlanF32 *d_plan, const float *features, int n_clips, float *scores,
                      float *tap_logits, int n_cu, hipStream_t stream)
{
    (void)hipGetLastError();      // the status returned below is this launch's, not a stale error of an earlier call
    if (n_clips <= 0) return 0;
    const int n_waves = kws_nn_f32_waves(N), grid_mult = 2;
    const size_t smem = kws_nn_f32_smem_bytes(N, n_waves);
    const int per_cu = (int)std::max<size_t>(1, (160 * 1024) / smem);
    int grid = (n_clips + n_waves - 1) / n_waves;
    if (grid > n_cu * per_cu * grid_mult) grid = n_cu * per_cu * grid_mult;
    const void *fn = n_waves <= 8 ? (const void *)kws_nn_f32_kernel<512> : (const void *)kws_nn_f32_kernel<1024>;
    if (smem > 64 * 1024) {                    // opt in to more than the default 64 KB of dynamic LDS
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
    }
    if (n_waves <= 8)
        hipLaunchKernelGGL(kws_nn_f32_kernel<512>, dim3(grid), dim3(KWS_WAVE * n_waves), smem, stream, d_plan, features, n_clips,
                           scores, tap_logits, kws_dev_f32_prof);
    else
        hipLaunchKernelGGL(kws_nn_f32_kernel<1024>, dim3(grid), dim3(KWS_WAVE * n_waves), smem, stream, d_plan, features, n_clips,
                           scores, tap_logits, kws_dev_f32_prof);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
//  float features -> int8 input tensor (the quantise loop of run_inference, ei_run_classifier.h:436-444)
// ---------------------------------------------------------------------------------------------------------
__global__ void kws_quantize_kernel(const float *__restrict__ f, int8_t *__restrict__ q, size_t n, float scale, int zp)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float qv = roundf(f[i] / scale) + (float)zp;
        int iv = (qv >= -2147483648.0f && qv < 2147483648.0f) ? (int)qv : (int)0x80000000;
        q[i] = (int8_t)(iv & 0xff);
    }
}

// ---------------------------------------------------------------------------------------------------------
//  synthetic clips, generated in HBM (include/kws/kws_synth.h; bit-identical to the host generator)
// ---------------------------------------------------------------------------------------------------------
__global__ void kws_synth_kernel(uint32_t seed, uint32_t first_clip, uint32_t n_clips, uint32_t clip_len, int16_t *out)
{
    for (uint32_t c = blockIdx.y; c < n_clips; c += gridDim.y) {
        const kws_synth_params p = kws_synth_clip_params(seed, first_clip + c);
        for (uint32_t n = blockIdx.x * blockDim.x + threadIdx.x; n < clip_len; n += gridDim.x * blockDim.x)
            out[(size_t)c * clip_len + n] = kws_synth_sample(&p, n);
    }
}

// ---------------------------------------------------------------------------------------------------------
//  launchers (called from kws_api.cpp)
// ---------------------------------------------------------------------------------------------------------
int kws_mfcc_max_prow(void) { return KWS_MAXPROW; }
// the CMVN offset table (row groups x walk length) lives in the FFT buffers: 4 x (win + 12 + 3 + 8) or 3 x (win + 16 + 3 + 8) ints
int kws_mfcc_max_win(int n_cepstral) { (void)n_cepstral; return 2 * KWS_ZF / 4 - 23; }                 // the narrower of the two layouts
int kws_mfcc_max_frames_for(int n_filters, int n_cepstral) { const int rows = kws_mel_rows(n_filters); return (n_filters == 40 && n_cepstral > 16) ? (rows < 51 ? rows : 51) : (rows < 52 ? rows : 52); }
int kws_mfcc_max_nz(void) { return KWS_MAXNZ; }
int kws_mfcc_cmvn_rows(void) { return 13; }
int kws_mfcc_max_frames(int n_filters) { return kws_mel_rows(n_filters); }
int kws_mfcc_fft_length(void) { return KWS_FFT; }

constexpr int KWS_CHP = 9;

template <bool F32IN, bool WITH_CMVN, bool PROF>
static int launch_mfcc_t(const KwsDspPlan &P, const void *pcm, int n_clips, float *out, int8_t *q_out, float in_scale, int in_zp,
                         const float *wrap, int out_stride, int grid_cap, long long *prof, hipStream_t stream)
{
    (void)hipGetLastError();      // the status returned below is this launch's, not a stale error of an earlier call
    if (n_clips <= 0) return 0;
    const int grid = n_clips < grid_cap ? n_clips : grid_cap;
    if (P.n_filters == 40 && P.max_nz <= 8 && WITH_CMVN && P.n_cepstral > 16)
        hipLaunchKernelGGL((kws_mfcc_kernel<KWS_CHP, F32IN, WITH_CMVN, 8, 40, PROF, true>), dim3(grid), dim3(KWS_WAVE), 0, stream, P,
                           pcm, n_clips, out, q_out, in_scale, in_zp, wrap, out_stride, prof);
    else if (P.n_filters == 40 && P.max_nz <= 8)
        hipLaunchKernelGGL((kws_mfcc_kernel<KWS_CHP, F32IN, WITH_CMVN, 8, 40, PROF>), dim3(grid), dim3(KWS_WAVE), 0, stream, P,
                           pcm, n_clips, out, q_out, in_scale, in_zp, wrap, out_stride, prof);
    else if (P.n_filters == 40 && P.max_nz <= KWS_MAXNZ)
        hipLaunchKernelGGL((kws_mfcc_kernel<KWS_CHP, F32IN, WITH_CMVN, KWS_MAXNZ, 40, PROF>), dim3(grid), dim3(KWS_WAVE), 0, stream, P,
                           pcm, n_clips, out, q_out, in_scale, in_zp, wrap, out_stride, prof);
    else if (P.n_filters != 32)
        return (int)hipErrorInvalidValue;
    else if (P.max_nz <= 4)
        hipLaunchKernelGGL((kws_mfcc_kernel<KWS_CHP, F32IN, WITH_CMVN, 4, 32, PROF>), dim3(grid), dim3(KWS_WAVE), 0, stream, P, pcm,
                           n_clips, out, q_out, in_scale, in_zp, wrap, out_stride, prof);
    else
        hipLaunchKernelGGL((kws_mfcc_kernel<KWS_CHP, F32IN, WITH_CMVN, KWS_MAXNZ, 32, PROF>), dim3(grid), dim3(KWS_WAVE), 0, stream, P,
                           pcm, n_clips, out, q_out, in_scale, in_zp, wrap, out_stride, prof);
    return (int)hipGetLastError();
}

// extract_mfcc_features (+ quantisation) for n_clips windows in one launch
int kws_launch_mfcc_fused(const KwsDspPlan &P, const void *pcm, int pcm_is_float, int n_clips, float *features, int8_t *q_out,
                          float in_scale, int in_zp, int grid_cap, hipStream_t stream)
{
    (void)hipGetLastError();      // the status returned below is this launch's, not a stale error of an earlier call
    return pcm_is_float ? launch_mfcc_t<true, true, false>(P, pcm, n_clips, features, q_out, in_scale, in_zp, nullptr, 0, grid_cap, nullptr, stream)
                        : launch_mfcc_t<false, true, false>(P, pcm, n_clips, features, q_out, in_scale, in_zp, nullptr, 0, grid_cap, nullptr, stream);
}

int kws_launch_mfcc_fused_prof(const KwsDspPlan &P, const void *pcm, int n_clips, float *features, int8_t *q_out, float in_scale,
                               int in_zp, int grid_cap, long long *prof_out, hipStream_t stream)
{
    (void)hipGetLastError();      // the status returned below is this launch's, not a stale error of an earlier call
    return launch_mfcc_t<false, true, true>(P, pcm, n_clips, features, q_out, in_scale, in_zp, nullptr, 0, grid_cap, prof_out, stream);
}

// speechpy::feature::mfcc for n_clips windows -> mfcc_out[n_clips][n_frames*n_cepstral] (cepstra before cmvnw)
// out_stride: floats between consecutive windows' outputs (0 = packed, n_frames*n_cepstral)
int kws_launch_spectral(const KwsDspPlan &P, const void *pcm, int pcm_is_float, int n_clips, float *mfcc_out, const float *wrap,
                        int out_stride, int grid_cap, hipStream_t stream)
{
    (void)hipGetLastError();      // the status returned below is this launch's, not a stale error of an earlier call
    if (out_stride == 0) out_stride = P.n_frames * P.n_cepstral;
    return pcm_is_float ? launch_mfcc_t<true, false, false>(P, pcm, n_clips, mfcc_out, nullptr, 0.f, 0, wrap, out_stride, grid_cap, nullptr, stream)
                        : launch_mfcc_t<false, false, false>(P, pcm, n_clips, mfcc_out, nullptr, 0.f, 0, wrap, out_stride, grid_cap, nullptr, stream);
}

// speechpy::feature::mfe (feature.hpp:193-318) for n_clips windows: mel energies + frame energies
int kws_launch_mfe(const KwsDspPlan &P0, const void *pcm, int n_clips, float *mel_out, float *energy_out, int grid_cap,
                   hipStream_t stream)
{
    (void)hipGetLastError();      // the status returned below is this launch's, not a stale error of an earlier call
    KwsDspPlan P = P0;
    P.mfe_mel = mel_out;
    P.mfe_energy = energy_out;
    return launch_mfcc_t<false, false, false>(P, pcm, n_clips, nullptr, nullptr, 0.f, 0, nullptr, P.n_frames * P.n_cepstral, grid_cap,
                                              nullptr, stream);
}

// ---- continuous mode, many streams: per-class 2-tap moving average (ei_run_classifier.h:134-145) and the feature-buffer
//      shift (ei_run_classifier.h:277-279) for S streams advancing in lock step
__global__ void kws_maf_kernel(float *__restrict__ scores, float *__restrict__ running_sum, float *__restrict__ maf_buf, int n,
                               int buf_idx, int taps)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float rs = running_sum[i];
        const float v = scores[i];
        rs -= maf_buf[(size_t)i * taps + buf_idx];
        rs += v;
        maf_buf[(size_t)i * taps + buf_idx] = v;
        running_sum[i] = rs;
        scores[i] = rs / (float)taps;
    }
}

__global__ void kws_shift_kernel(const float *__restrict__ src, float *__restrict__ dst, int n_streams, int F, int shift)
{
    const size_t total = (size_t)n_streams * F;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % F);
        dst[i] = (k < F - shift) ? src[i + shift] : src[i];     // the tail keeps its old values, as in the reference
    }
}

int kws_launch_maf(float *scores, float *running_sum, float *maf_buf, int n, int buf_idx, int taps, hipStream_t stream)
{
    (void)hipGetLastError();      // the status returned below is this launch's, not a stale error of an earlier call
    if (n <= 0) return 0;
    hipLaunchKernelGGL(kws_maf_kernel, dim3((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024), dim3(256), 0, stream, scores, running_sum,
                       maf_buf, n, buf_idx, taps);
    return (int)hipGetLastError();
}

int kws_launch_shift(const float *src, float *dst, int n_streams, int F, int shift, hipStream_t stream)
{
    (void)hipGetLastError();      // the status returned below is this launch's, not a stale error of an earlier call
    if (n_streams <= 0) return 0;
    size_t blocks = ((size_t)n_streams * F + 255) / 256;
    hipLaunchKernelGGL(kws_shift_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, stream, src, dst, n_streams, F, shift);
    return (int)hipGetLastError();
}

static bool nn_fits_mfma(const KwsNnPlan &N);
int kws_force_scalar_nn = 0;   // tests: run the generic (dot4) kernel even when the matrix-core kernel applies
int kws_nn_uses_mfma(const KwsNnPlan &N) { return nn_fits_mfma(N) && !kws_force_scalar_nn; }

// cmvnw + quantise (+ the network when it fits the matrix-core path and scores != NULL).  Returns 1 in *ran_nn if the
// network ran inside this launch.
int kws_launch_cmvn_nn(const KwsDspPlan &P, const KwsNnPlan &N, const float *mfcc, int n_clips, float *features, int8_t *q_out,
                       float *scores, int8_t *tap_pooled, int pooled_stride, int8_t *tap_fc, int8_t *tap_out_q, int grid_cap,
                       int *ran_nn, hipStream_t stream)
{
    (void)hipGetLastError();      // the status returned below is this launch's, not a stale error of an earlier call
    *ran_nn = 0;
    if (n_clips <= 0) return 0;
    int grid = (n_clips + KWS_NN_WAVES - 1) / KWS_NN_WAVES;
    if (grid > grid_cap) grid = grid_cap;
    NnTaps taps = { tap_pooled, pooled_stride, tap_fc, tap_out_q };
    if (scores && nn_fits_mfma(N) && N.blk[0].in_cpad == 16 && !kws_force_scalar_nn) {      // (64-byte rows: separate network launch)
        hipLaunchKernelGGL((kws_cmvn_nn_kernel<true>), dim3(grid), dim3(KWS_WAVE * KWS_NN_WAVES), 0, stream, P, N, mfcc, n_clips,
                           features, q_out, scores, taps);
        *ran_nn = 1;
    } else {
        hipLaunchKernelGGL((kws_cmvn_nn_kernel<false>), dim3(grid), dim3(KWS_WAVE * KWS_NN_WAVES), 0, stream, P, N, mfcc, n_clips,
                           features, q_out, scores, taps);
    }
    return (int)hipGetLastError();
}
size_t kws_nn_smem_bytes(const KwsNnPlan &N)
{
    size_t s = 0;
    int act = 0;
    for (int b = 0; b < N.n_blocks; ++b) {
        const KwsConvBlock &k = N.blk[b];
        s += ((size_t)k.w_bytes + 15) & ~(size_t)15;
        s += k.has_lut ? (size_t)k.out_c * 256 : 0;
        const int ab = nn_rows(k) * k.in_cpad;
        act = ab > act ? ab : act;
    }
    act = (act + 15) & ~15;
    return s + (size_t)KWS_NN_WAVES * (2 * act + 64 * 4);
}

int kws_launch_nn(const KwsNnPlan &N, const int8_t *q_in, int n_clips, float *scores, int8_t *tap_pooled,
                  int pooled_stride, int8_t *tap_fc, int8_t *tap_out_q, int grid_cap, hipStream_t stream)
{
    (void)hipGetLastError();      // the status returned below is this launch's, not a stale error of an earlier call
    if (n_clips <= 0) return 0;
    int grid = (n_clips + KWS_NN_WAVES - 1) / KWS_NN_WAVES;
    if (grid > grid_cap) grid = grid_cap;
    NnTaps taps = { tap_pooled, pooled_stride, tap_fc, tap_out_q };
    if (nn_fits_mfma(N) && !kws_force_scalar_nn) {
        if (N.blk[0].in_cpad == 16)
            hipLaunchKernelGGL(kws_nn_mfma_kernel<16>, dim3(grid), dim3(KWS_WAVE * KWS_NN_WAVES), 0, stream, N, q_in, n_clips, scores, taps);
        else
            hipLaunchKernelGGL(kws_nn_mfma_kernel<64>, dim3(grid), dim3(KWS_WAVE * KWS_NN_WAVES), 0, stream, N, q_in, n_clips, scores, taps);
        return (int)hipGetLastError();
    }
    const size_t smem = kws_nn_smem_bytes(N);
    if (smem > 64 * 1024) {                    // wide models: opt in to more than the default 64 KB of dynamic LDS
        hipError_t e = hipFuncSetAttribute((const void *)kws_nn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kws_nn_kernel, dim3(grid), dim3(KWS_WAVE * KWS_NN_WAVES), smem, stream, N, q_in, n_clips, scores, taps);
    return (int)hipGetLastError();
}

int kws_launch_quantize(const float *f, int8_t *q, size_t n, float scale, int zp, hipStream_t stream)
{
    (void)hipGetLastError();      // the status returned below is this launch's, not a stale error of an earlier call
    if (n == 0) return 0;
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(kws_quantize_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, f, q, n, scale, zp);
    return (int)hipGetLastError();
}

int kws_launch_synth(uint32_t seed, uint32_t first_clip, uint32_t n_clips, uint32_t clip_len, int16_t *out, hipStream_t stream)
{
    (void)hipGetLastError();      // the status returned below is this launch's, not a stale error of an earlier call
    if (n_clips == 0) return 0;
    dim3 grid((clip_len + 255) / 256, n_clips < 65535u ? n_clips : 65535u);
    hipLaunchKernelGGL(kws_synth_kernel, grid, dim3(256), 0, stream, seed, first_clip, n_clips, clip_len, out);
    return (int)hipGetLastError();
}
