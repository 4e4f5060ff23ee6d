// kws_generic.hip -- the exact MFCC block for every configuration the tuned kernel (kws_mfcc.hip) is not instantiated for: any even
// fft_length whose half factors into 2, 3, 4, 5 (numpy::rfft zero-pads or truncates the frame to it, SDK/dsp/numpy.hpp:1091-1156;
// kf_factor kiss_fft.cpp:303-324), any such mel filter count up to 64, any frame count / stride / clip length (no alignment
// rules), mel filters of any width.  Same bit-exactness contract as kws_device.h: every floating-point operation in the
// reference's order and precision.
//
// Shape: lane = frame (64 frames per workgroup pass), each lane runs the scalar algorithm on its own frame with its working
// arrays in a lane-interleaved global scratch (element e of lane l at ws[e * 64 + l]: the lanes execute identical, data-
// independent index sequences, so every access is coalesced and L2-resident).  It is the slow, general path: the shapes the
// reference ships and BASELINE.json names all take the tuned kernels.
//   kws_spectral_generic_kernel : pre-emphasis, frame, KissFFT of any factorisation, power spectrum, energy, mel, log, DCT
//                                 -> cepstra before cmvnw [clip][frame][ncep]   (speechpy::feature::mfcc, feature.hpp:370-439)
//   kws_cmvn_generic_kernel     : processing::cmvnw (processing.hpp:326-389) + the int8 input quantisation, one thread per element
#include <atomic>
#include <cstdlib>

#include "kws_device.h"

bool kws_generic_uses_lds(const KwsDspPlan &P);

#define GL 64                                     // lanes interleaved in the scratch

// S = lanes interleaved in the array (GL: the global scratch of kws_spectral_generic_kernel; LCH: the per-lane DCT arrays in LDS of
// kws_spectral_lds_kernel)
template <int S> __device__ __forceinline__ cf g_ld(const float *c, int idx) { cf v; v.r = c[(size_t)(2 * idx) * S]; v.i = c[(size_t)(2 * idx + 1) * S]; return v; }
template <int S> __device__ __forceinline__ void g_st(float *c, int idx, cf v) { c[(size_t)(2 * idx) * S] = v.r; c[(size_t)(2 * idx + 1) * S] = v.i; }

// kf_bfly2 / kf_bfly3 / kf_bfly4 / kf_bfly5 (kiss_fft.cpp:15-192) on the sub-array starting at complex index `base`
template <int S>
__device__ void g_bfly(float *F, int base, int fstride, int m, int p, const float2 *__restrict__ tw)
{
    if (p == 2) {
        for (int k = 0; k < m; k++) {
            const cf t = cmul(g_ld<S>(F, base + k + m), to_cf(tw[k * fstride]));
            const cf a = g_ld<S>(F, base + k);
            g_st<S>(F, base + k + m, csub(a, t));
            g_st<S>(F, base + k, cadd(a, t));
        }
    } else if (p == 4) {
        for (int k = 0; k < m; k++) {
            cf f0 = g_ld<S>(F, base + k), f1 = g_ld<S>(F, base + k + m), f2 = g_ld<S>(F, base + k + 2 * m), f3 = g_ld<S>(F, base + k + 3 * m);
            bfly4(f0, f1, f2, f3, to_cf(tw[k * fstride]), to_cf(tw[k * fstride * 2]), to_cf(tw[k * fstride * 3]));
            g_st<S>(F, base + k, f0); g_st<S>(F, base + k + m, f1); g_st<S>(F, base + k + 2 * m, f2); g_st<S>(F, base + k + 3 * m, f3);
        }
    } else if (p == 3) {
        const cf epi3 = to_cf(tw[fstride * m]);
        for (int k = 0; k < m; k++) {
            cf f0 = g_ld<S>(F, base + k);
            const cf s1 = cmul(g_ld<S>(F, base + k + m), to_cf(tw[k * fstride]));
            const cf s2 = cmul(g_ld<S>(F, base + k + 2 * m), to_cf(tw[k * fstride * 2]));
            const cf s3 = cadd(s1, s2);
            cf s0 = csub(s1, s2);
            cf f1, f2;
            f1.r = f0.r - s3.r * 0.5f;
            f1.i = f0.i - s3.i * 0.5f;
            s0.r *= epi3.i;
            s0.i *= epi3.i;
            f0 = cadd(f0, s3);
            f2.r = f1.r + s0.i;
            f2.i = f1.i - s0.r;
            f1.r -= s0.i;
            f1.i += s0.r;
            g_st<S>(F, base + k, f0); g_st<S>(F, base + k + m, f1); g_st<S>(F, base + k + 2 * m, f2);
        }
    } else {    // 5
        const cf ya = to_cf(tw[fstride * m]), yb = to_cf(tw[fstride * 2 * m]);
        for (int u = 0; u < m; u++) {
            cf F0 = g_ld<S>(F, base + u), F1 = g_ld<S>(F, base + u + m), F2 = g_ld<S>(F, base + u + 2 * m), F3 = g_ld<S>(F, base + u + 3 * m), F4 = g_ld<S>(F, base + u + 4 * m);
            const cf s0 = F0;
            const cf s1 = cmul(F1, to_cf(tw[u * fstride])), s2 = cmul(F2, to_cf(tw[2 * u * fstride]));
            const cf s3 = cmul(F3, to_cf(tw[3 * u * fstride])), s4 = cmul(F4, to_cf(tw[4 * u * fstride]));
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
            g_st<S>(F, base + u, F0); g_st<S>(F, base + u + m, F1); g_st<S>(F, base + u + 2 * m, F2); g_st<S>(F, base + u + 3 * m, F3); g_st<S>(F, base + u + 4 * m, F4);
        }
    }
}

// kiss_fftr (kiss_fftr.cpp:66-120) of the real array `in` [nfft] -> spectrum `spec` [nfft/2 + 1] complex; `tmp` [nfft/2] complex.
// kf_work's recursion (kiss_fft.cpp:232-296) is replayed level by level: the leaves' strided copies first (a mixed-radix digit
// reversal), then the butterflies of every level from the innermost out -- sub-transforms of one level are independent, so the
// order between them does not matter, the order inside one butterfly is the reference's.
template <int S>
__device__ void g_rfft(const float *in, float *tmp, float *spec, int nfft, const int *__restrict__ fac, int n_levels,
                       const float2 *__restrict__ tw, const float2 *__restrict__ stw)
{
    const int ncfft = nfft >> 1;
    for (int o = 0; o < ncfft; o++) {
        int rem = o, i = 0, stride = 1;
        for (int l = 0; l < n_levels; l++) {
            const int p = fac[2 * l], m = fac[2 * l + 1];
            const int k = rem / m;
            rem -= k * m;
            i += k * stride;
            stride *= p;
        }
        cf v; v.r = in[(size_t)(2 * i) * S]; v.i = in[(size_t)(2 * i + 1) * S];
        g_st<S>(tmp, o, v);
    }
    for (int l = n_levels - 1; l >= 0; l--) {
        const int p = fac[2 * l], m = fac[2 * l + 1];
        int fstride = 1;
        for (int q = 0; q < l; q++) fstride *= fac[2 * q];
        for (int base = 0; base < ncfft; base += p * m) g_bfly<S>(tmp, base, fstride, m, p, tw);
    }
    const cf t0 = g_ld<S>(tmp, 0);
    cf dc, ny;
    dc.r = t0.r + t0.i; dc.i = 0.0f;
    ny.r = t0.r - t0.i; ny.i = 0.0f;
    g_st<S>(spec, 0, dc);
    g_st<S>(spec, ncfft, ny);
    for (int k = 1; k <= ncfft / 2; k++) {
        const cf fpk = g_ld<S>(tmp, k);
        cf fpnk = g_ld<S>(tmp, ncfft - k);
        fpnk.i = -fpnk.i;
        const cf f1k = cadd(fpk, fpnk), f2k = csub(fpk, fpnk);
        const cf twv = cmul(f2k, to_cf(stw[k - 1]));
        cf lo, hi;
        lo.r = (f1k.r + twv.r) * 0.5f;
        lo.i = (f1k.i + twv.i) * 0.5f;
        hi.r = (f1k.r - twv.r) * 0.5f;
        hi.i = (twv.i - f1k.i) * 0.5f;
        g_st<S>(spec, k, lo);
        g_st<S>(spec, ncfft - k, hi);
    }
}

__device__ __forceinline__ double g_dsqrt(double x) { return x == 0.0 ? 0.0 : dsqrt_sumsq(x); }

// ws floats per lane: in[max(fft, NF)] | tmp[2 * max(fft, NF) / 2] | spec[2 * (max(fft, NF) / 2 + 1)] | ps[fft/2 + 1] | mel[NF]
__host__ __device__ inline int g_ws_floats(int fft_len, int nf)
{
    const int n = fft_len > nf ? fft_len : nf;
    return n + n + (n + 2) + (fft_len / 2 + 1) + nf;
}

template <bool F32IN>
__global__ __launch_bounds__(GL) void kws_spectral_generic_kernel(KwsDspPlan P, const void *__restrict__ pcm_v, int n_clips, float *__restrict__ mfcc_out,
                                                                  const float *__restrict__ wrap, int out_stride, float *__restrict__ ws_all)
{
    const int lane = threadIdx.x;
    const int nfr = P.n_frames, NF = P.n_filters, ncep = P.n_cepstral, fft = P.fft_len, nbins = P.n_bins;
    const int chunks = (nfr + GL - 1) / GL;
    const int n_max = fft > NF ? fft : NF;
    float *ws = ws_all + (size_t)blockIdx.x * g_ws_floats(fft, NF) * GL + lane;
    float *w_in = ws, *w_tmp = w_in + (size_t)n_max * GL, *w_spec = w_tmp + (size_t)n_max * GL, *w_ps = w_spec + (size_t)(n_max + 2) * GL,
          *w_mel = w_ps + (size_t)nbins * GL;
    const int used = P.frame_len < fft ? P.frame_len : fft;      // numpy::rfft: truncate to fft_length or zero-pad (numpy.hpp:1097-1111)
    const double inv_fft = 1.0 / (double)(float)fft;             // processing.hpp:306-309
    for (int item = blockIdx.x; item < n_clips * chunks; item += gridDim.x) {
        const int clip = item / chunks, f = (item - clip * chunks) * GL + lane;
        if (f >= nfr) continue;
        const size_t cbase = (size_t)clip * P.n_samples;
        auto sample = [&](int n) -> float {
            return F32IN ? ((const float *)pcm_v)[cbase + n] : (float)((const int16_t *)pcm_v)[cbase + n] * (1.0f / 32768.0f);     // numpy::int16_to_float
        };
        // ---- pre-emphasis (processing.hpp:52-138; x[-1] = the window's last sample, or the caller's override) + framing ----
        const int off = f * P.frame_stride;
        for (int n = 0; n < fft; n++) {
            float y = 0.0f;
            if (n < used) {
                const int s = off + n;
                const float prev = (s == 0) ? (wrap ? wrap[clip] : sample(P.n_samples - 1)) : sample(s - 1);
                const float pl = P.pre_cof * prev;
                y = sample(s) - pl;
            }
            w_in[(size_t)n * GL] = y;
        }
        // ---- power spectrum: kiss_fftr, sqrt(re^2 + im^2) in double, (1/fft) * mag^2 (numpy.hpp:1410, processing.hpp:306-309)
        g_rfft<GL>(w_in, w_tmp, w_spec, fft, P.fft_fac, P.fft_levels, P.tw, P.stw);
        float energy = 0.0f;
        for (int k = 0; k < nbins; k++) {
            const cf v = g_ld<GL>(w_spec, k);
            const double re = (double)v.r, im = (double)v.i;
            const float mag = (float)g_dsqrt(__fma_rn(re, re, im * im));
            const float sq = mag * mag;
            const float pw = (float)(inv_fft * (double)sq);
            w_ps[(size_t)k * GL] = pw;
            energy += pw;                                          // numpy::sum, ascending (numpy.hpp:88-94)
        }
        if (energy == 0.0f) energy = FLT_EPSILON;
        if (P.mfe_energy) P.mfe_energy[(size_t)clip * nfr + f] = energy;
        // ---- mel filterbank: dot_by_row over the non-zero weights in ascending bin order, zero handling, log ----------------
        for (int j = 0; j < NF; j++) {
            float acc = 0.0f;
            for (int n = P.filt_start[j]; n < P.filt_start[j + 1]; n++) {
                const float prod = w_ps[(size_t)P.filt_bin[n] * GL] * P.filt_w[n];
                acc += prod;
            }
            if (acc == 0.0f) acc = FLT_EPSILON;
            if (P.mfe_mel) P.mfe_mel[(size_t)clip * out_stride + (size_t)ring_out_row(P, f) * NF + j] = acc;
            w_mel[(size_t)j * GL] = fast_log(acc);
        }
        if (P.mfe_mel) continue;                                   // MFE block: no log / DCT output
        // ---- numpy::dct2 (numpy.hpp:378-401) -> dct::transform (fast-dct-fft.cpp:37-80): even/odd reorder, kiss_fftr(NF),
        //      v[i] = re cos + im sin for i <= NF/2 only, x2, ortho scale; c0 <- log(energy) --------------------------------
        const int half = NF / 2;
        for (int i = 0; i < half; i++) {
            w_in[(size_t)i * GL] = w_mel[(size_t)(2 * i) * GL];
            w_in[(size_t)(NF - 1 - i) * GL] = w_mel[(size_t)(2 * i + 1) * GL];
        }
        g_rfft<GL>(w_in, w_tmp, w_spec, NF, P.dct_fac, P.dct_levels, P.dct_tw, P.dct_stw);
        float *orow = mfcc_out + (size_t)clip * out_stride + (size_t)ring_out_row(P, f) * ncep;
        for (int i = 0; i < ncep; i++) {
            float d;
            if (i <= half) {
                const cf r = g_ld<GL>(w_spec, i);
                const float a = r.r * P.dct_cos[i];
                const float b = r.i * P.dct_sin[i];
                d = a + b;
            } else {
                d = w_mel[(size_t)i * GL];                         // never written by the transform: the input stays
            }
            d = d * 2.0f;
            d = d * (i == 0 ? P.dct_s0 : P.dct_s1);
            orow[i] = d;
        }
        orow[0] = fast_log(energy);                                // feature.hpp:425-429
    }
}

// ---------------------------------------------------------------------------------------------------------
//  kws_spectral_lds_kernel (round 4; round 5: how it occupies the chip): the same function as
//  kws_spectral_generic_kernel -- any factorisation, any frame / filter count, every operation in the reference's order -- without the
//  lane-interleaved scratch in HBM.  A WAVE owns a chunk of LCH consecutive frames of a clip; the waves of a workgroup (up to eight) share
//  one copy of the model's tables and nothing else:
//    * per sub-batch of up to fb frames, the wave transforms them COOPERATIVELY: the samples are fetched (int16 on even sample offsets: as
//      dword pairs, requested one sub-batch ahead), pre-emphasised, and dealt into kf_work's leaf order (a mixed-radix digit reversal,
//      tabulated once per workgroup); then level by level, innermost first, the butterflies of a level -- ncfft / p per frame, all
//      independent -- are spread over the 64 lanes as (frame, butterfly) items; each is kf_bfly2 / 3 / 4 / 5 exactly as g_bfly computes it.
//      kiss_fftr's split and the power spectrum (double-precision magnitude) run per bin pair, straight into the chunk's power rows in LDS.
//    * per chunk: frame energies (a lane per frame: numpy::sum is a sequential sum), the mel filterbank ((frame, filter) items over the
//      lanes, ascending-bin dot products), log, and the DCT of all the chunk's frames at once ((frame, butterfly) items over the lanes),
//      c0 <- log(energy), cepstra to HBM.
//  LDS (lds_layout): per workgroup the tables (leaf orders, twiddles, filterbank: 8 - 13 KB); per wave fb complex work buffers of
//  2 (ncfft + ncfft/16 + 1) floats -- the mel rows and the DCT's arrays lie over them once they are dead -- and LCH power rows of n_bins | 1
//  floats: 8.5 KB for fft 512 at LCH = 4, fb = 2, i.e. sixteen waves per CU in the build for four waves per SIMD (see launch_spectral_lds).
// ---------------------------------------------------------------------------------------------------------
// frames per chunk: a template parameter, 8 or 4 (16: 2.29 ms for 8 192 clips of fft 512 x 49 frames, 8: 1.29 / 1.30 ms in two calls, 4: 1.30 ms; 4 is
// 22 % / 15 % faster on 98-frame windows and on fft 1024, 8 % slower on fft 128: profiles/r04_generic_rate.txt).  The LDS per wave bounds how many
// waves a CU holds, the per-chunk phases favour longer chunks: no rule here predicts the winner, so kws_api.cpp measures it per handle on the
// handle's own first large calls (generic_chunk_begin) and passes the choice to kws_launch_spectral_generic.
constexpr int KWS_LCH_DEFAULT = 8;
// a 24 x 24-bit multiply (full rate; v_mul_lo_u32 is a quarter-rate instruction): LDS offsets, item indices and their reciprocals all fit
#define M24(a, b) __mul24((int)(a), (int)(b))
// development aid: shader-clock totals per phase of workgroup 0 (kws_dev_generic_prof; tools/gpu_generic_rate.py --prof)
__device__ long long g_gen_prof[8];
#ifdef KWS_DEV_SWITCHES
#define GPH(i) do { const long long now_ = clock64(); if (blockIdx.x == 0 && threadIdx.x == 0) g_gen_prof[i] += now_ - tlast_; tlast_ = now_; } while (0)
#else
#define GPH(i) do { } while (0)          // (a clock read drains the wave's outstanding LDS and scalar requests: development build only)
#endif
__device__ __forceinline__ int zpad(int p) { return p + (p >> 4); }             // complex index -> padded complex index
__device__ __forceinline__ cf z_ld(const float *z, int p) { const float2 v = *(const float2 *)(z + 2 * zpad(p)); cf c; c.r = v.x; c.i = v.y; return c; }
__device__ __forceinline__ void z_st(float *z, int p, cf v) { *(float2 *)(z + 2 * zpad(p)) = make_float2(v.r, v.i); }

// one butterfly of a level: kf_bfly2 / 3 / 4 / 5 (kiss_fft.cpp:15-192) at index k of the sub-transform starting at `base` -- the body of
// g_bfly's loops, operation for operation
__device__ __forceinline__ void z_bfly_one(float *Z, int base, int k, int fstride, int m, int p, const float2 *__restrict__ tw)
{
    const int kf = M24(k, fstride), fm = M24(fstride, m);        // (24-bit multiplies: table indices)
    if (p == 2) {
        const cf t = cmul(z_ld(Z, base + k + m), to_cf(tw[kf]));
        const cf a = z_ld(Z, base + k);
        z_st(Z, base + k + m, csub(a, t));
        z_st(Z, base + k, cadd(a, t));
    } else if (p == 4) {
        cf f0 = z_ld(Z, base + k), f1 = z_ld(Z, base + k + m), f2 = z_ld(Z, base + k + 2 * m), f3 = z_ld(Z, base + k + 3 * m);
        bfly4(f0, f1, f2, f3, to_cf(tw[kf]), to_cf(tw[2 * kf]), to_cf(tw[3 * kf]));
        z_st(Z, base + k, f0); z_st(Z, base + k + m, f1); z_st(Z, base + k + 2 * m, f2); z_st(Z, base + k + 3 * m, f3);
    } else if (p == 3) {
        const cf epi3 = to_cf(tw[fm]);
        cf f0 = z_ld(Z, base + k);
        const cf s1 = cmul(z_ld(Z, base + k + m), to_cf(tw[kf]));
        const cf s2 = cmul(z_ld(Z, base + k + 2 * m), to_cf(tw[2 * kf]));
        const cf s3 = cadd(s1, s2);
        cf s0 = csub(s1, s2);
        cf f1, f2;
        f1.r = f0.r - s3.r * 0.5f;
        f1.i = f0.i - s3.i * 0.5f;
        s0.r *= epi3.i;
        s0.i *= epi3.i;
        f0 = cadd(f0, s3);
        f2.r = f1.r + s0.i;
        f2.i = f1.i - s0.r;
        f1.r -= s0.i;
        f1.i += s0.r;
        z_st(Z, base + k, f0); z_st(Z, base + k + m, f1); z_st(Z, base + k + 2 * m, f2);
    } else {    // 5
        const cf ya = to_cf(tw[fm]), yb = to_cf(tw[2 * fm]);
        const int u = k;
        cf F0 = z_ld(Z, base + u), F1 = z_ld(Z, base + u + m), F2 = z_ld(Z, base + u + 2 * m), F3 = z_ld(Z, base + u + 3 * m), F4 = z_ld(Z, base + u + 4 * m);
        const cf s0 = F0;
        const cf s1 = cmul(F1, to_cf(tw[kf])), s2 = cmul(F2, to_cf(tw[2 * kf]));
        const cf s3 = cmul(F3, to_cf(tw[3 * kf])), s4 = cmul(F4, to_cf(tw[4 * kf]));
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
        z_st(Z, base + u, F0); z_st(Z, base + u + m, F1); z_st(Z, base + u + 2 * m, F2); z_st(Z, base + u + 3 * m, F3); z_st(Z, base + u + 4 * m, F4);
    }
}

// + the model's tables, staged once per workgroup (every read of them sits on a lane's serial path: from L2 a tap of a mel filter or a
// twiddle costs a round trip of ~1 us; measured: the first version, tables in L2, ran at 9.9 ns per frame against the scratch kernel's 12.4)
// shared: the tables, one copy per workgroup (offsets from the start of the dynamic LDS); wave: each wave's own buffers (offsets from its block at
// shared + wave index * wave)
struct LdsLayout { int z, zs, ps, ps_stride, mel, mel_stride, dct, wave, perm, tw, stw, dtw, dstw, dcs, fstart, fbin, fw, dperm, shared, total; };
// fb: frames of a chunk transformed TOGETHER (round 5): fb work buffers, the butterflies of a level dealt over (frame, butterfly) items.
// total = the tables + ONE wave's buffers
__host__ __device__ inline LdsLayout lds_layout(int fft, int nf, int nbins, int nnz, int LCH, int fb = 1)
{
    LdsLayout L;
    const int ncfft = fft / 2;
    L.perm = 0;                                           // (the sample buffer of the first version is gone: samples go straight to their leaves)
    L.tw = (L.perm + ncfft + 1) & ~1;                     // float2 tables: 8-byte aligned
    L.stw = L.tw + 2 * ncfft;
    L.dtw = L.stw + 2 * (ncfft / 2 + 1);
    L.dstw = L.dtw + 2 * (nf / 2 + 1);
    L.dcs = L.dstw + 2 * (nf / 4 + 1);
    L.fstart = L.dcs + 2 * (nf / 2 + 1);
    L.fbin = L.fstart + nf + 1;
    L.fw = L.fbin + nnz + 4;                              // (+4: the mel loop reads whole batches of four taps)
    L.dperm = L.fw + nnz + 4;                             // the DCT's leaf order (nf / 2 ints)
    L.shared = (L.dperm + nf / 2 + 3) & ~3;
    L.z = 0;
    L.zs = 2 * (ncfft + (ncfft >> 4) + 1);
    // the mel rows and the DCT's buffers are written after the chunk's last transform: they lie over the (then dead) work buffers
    L.mel = 0;
    L.mel_stride = nf | 1;
    L.dct = (L.mel + LCH * L.mel_stride + 1) & ~1;        // (the DCT's buffers per frame hold float2 points)
    const int tail = L.dct + (2 * (nf / 2 + (nf >> 5) + 1) + 2 * (nf / 2 + 1)) * LCH;
    L.ps = ((fb * L.zs > tail ? fb * L.zs : tail) + 1) & ~1;
    L.ps_stride = nbins | 1;
    L.wave = (L.ps + LCH * L.ps_stride + 3) & ~3;
    L.total = L.shared + L.wave;
    return L;
}

// WPS: waves per SIMD the registers are budgeted for -- 2: up to 256 VGPRs, deep batches; 4: 128 VGPRs, shallow batches (experiment knobs:
// -DKWS_GEN_UB / _FB / _EB = radix-4 butterflies per lane and LDS round trip, frames whose samples are requested together, ordered additions
// per read batch of the four-wave build)
#ifndef KWS_GEN_UB
#define KWS_GEN_UB 1
#endif
#ifndef KWS_GEN_FB
#define KWS_GEN_FB 2
#endif
#ifndef KWS_GEN_EB
#define KWS_GEN_EB 8
#endif
template <bool F32IN, int LCH, int WPS, bool PK>
__global__ __launch_bounds__(WPS >= 4 ? 512 : 256, WPS) void kws_spectral_lds_kernel(KwsDspPlan P, const void *__restrict__ pcm_v, int n_clips, float *__restrict__ mfcc_out,
                                                                    const float *__restrict__ wrap, int out_stride, int fb_arg)
{
    extern __shared__ __attribute__((aligned(16))) float glds[];
    const int fb = fb_arg & 255;
#ifdef KWS_DEV_SWITCHES
    const bool fine_prof = (fb_arg & 256) != 0;          // KWS_DEV_GENERIC_PROF: a forced wait for the samples, so that the phase clocks tell arrival from dealing
#endif
    const int lane = threadIdx.x & (KWS_WAVE - 1), wave = __builtin_amdgcn_readfirstlane(threadIdx.x / KWS_WAVE), n_waves = blockDim.x / KWS_WAVE, tid = threadIdx.x, nthr = blockDim.x;
    const int nfr = P.n_frames, NF = P.n_filters, ncep = P.n_cepstral, fft = P.fft_len, nbins = P.n_bins, ncfft = fft >> 1;
    const int nnz = P.filt_nnz;
    // batch depths: radix-4 butterflies per lane and LDS round trip, frames whose samples are requested together, ordered additions per read batch
    constexpr int UB = WPS >= 3 ? KWS_GEN_UB : 4, FBMAX = WPS >= 3 ? KWS_GEN_FB : 4, EB = WPS >= 3 ? KWS_GEN_EB : 32;
    const LdsLayout L = lds_layout(fft, NF, nbins, nnz, LCH, fb);
    float *wbase = glds + L.shared + wave * L.wave;
    float *Z = wbase + L.z, *PS = wbase + L.ps, *MEL = wbase + L.mel, *DCT = wbase + L.dct;
    int *perm = (int *)(glds + L.perm);
    float2 *l_tw = (float2 *)(glds + L.tw), *l_stw = (float2 *)(glds + L.stw), *l_dtw = (float2 *)(glds + L.dtw), *l_dstw = (float2 *)(glds + L.dstw);
    float *l_dcos = glds + L.dcs, *l_dsin = l_dcos + NF / 2 + 1, *l_fw = glds + L.fw;
    int *l_fstart = (int *)(glds + L.fstart), *l_fbin = (int *)(glds + L.fbin);
    for (int i = tid; i < ncfft; i += nthr) l_tw[i] = P.tw[i];
    for (int i = tid; i < ncfft / 2; i += nthr) l_stw[i] = P.stw[i];
    for (int i = tid; i < NF / 2; i += nthr) l_dtw[i] = P.dct_tw[i];
    for (int i = tid; i < NF / 4; i += nthr) l_dstw[i] = P.dct_stw[i];
    for (int i = tid; i <= NF / 2; i += nthr) { l_dcos[i] = P.dct_cos[i]; l_dsin[i] = P.dct_sin[i]; }
    for (int i = tid; i <= NF; i += nthr) l_fstart[i] = P.filt_start[i];
    for (int i = tid; i < nnz + 4; i += nthr) { l_fbin[i] = i < nnz ? P.filt_bin[i] : 0; l_fw[i] = i < nnz ? P.filt_w[i] : 0.0f; }
    const int chunks = (nfr + LCH - 1) / LCH;
    const int used = P.frame_len < fft ? P.frame_len : fft;      // numpy::rfft: truncate to fft_length or zero-pad (numpy.hpp:1097-1111)
    const double inv_fft = 1.0 / (double)(float)fft;             // processing.hpp:306-309
    // kf_work's leaf order (kiss_fft.cpp:232-296): output position o of the strided copies reads input point i; perm[i] = o
    for (int o = tid; o < ncfft; o += nthr) {
        int rem = o, i = 0, stride = 1;
        for (int l = 0; l < P.fft_levels; l++) {
            const int p = P.fft_fac[2 * l], m = P.fft_fac[2 * l + 1];
            const int k = rem / m;
            rem -= k * m;
            i += k * stride;
            stride *= p;
        }
        perm[i] = o;                                               // inverse: input point i is leaf o
    }
    // the DCT's leaf order likewise (kf_work on NF / 2 points): leaf o reads input point dperm[o]
    int *dperm = (int *)(glds + L.dperm);
    for (int o = tid; o < NF / 2; o += nthr) {
        int rem = o, i = 0, stride = 1;
        for (int l = 0; l < P.dct_levels; l++) {
            const int p = P.dct_fac[2 * l], m = P.dct_fac[2 * l + 1];
            const int k = rem / m;
            rem -= k * m;
            i += k * stride;
            stride *= p;
        }
        dperm[o] = i;
    }
    __syncthreads();                                               // the tables are the workgroup's; from here on every wave is on its own
    // x / d for the small item counts of the per-chunk phases (x d < 2^20): (x * (2^20 / d + 1)) >> 20 -- an integer division costs ~35 instructions
    auto recip20 = [](int d) -> unsigned { return (1u << 20) / (unsigned)d + 1u; };
    auto div20 = [](int x, unsigned r) -> int { return (int)(__umul24((unsigned)x, r) >> 20); };
    const unsigned r_nf = recip20(NF), r_nc = recip20(max(NF >> 1, 1)), r_half = recip20((NF >> 2) + 1), r_ncep = recip20(ncep);
    // the levels' parameters, once: (p, m, twiddle stride, butterflies, 2^20 / m + 1 for b / m with b < 4096)
    constexpr int MAXLEV = 8;
    int lv_p[MAXLEV], lv_m[MAXLEV], lv_fs[MAXLEV], lv_inv[MAXLEV];
#pragma unroll
    for (int l = 0; l < MAXLEV; l++) {
        const int ll = min(l, P.fft_levels - 1);
        lv_p[l] = P.fft_fac[2 * ll]; lv_m[l] = P.fft_fac[2 * ll + 1];
        int fsv = 1;
        for (int q = 0; q < ll; q++) fsv *= P.fft_fac[2 * q];
        lv_fs[l] = fsv;
        lv_inv[l] = (int)((1u << 20) / (unsigned)lv_m[l] + 1u);
    }
#ifdef KWS_DEV_SWITCHES
    long long tlast_ = clock64();
#endif
    // ---- the samples of a sub-batch.  int16 input whose frames start on even samples (PK, decided by the launch): PAIRS -- lane l of slot j
    //      loads the dword that holds samples 2 i, 2 i + 1 of kiss_fftr's input point i = 64 j + l, so a lane owns a whole complex point: one
    //      8-byte LDS store per point, the even sample's predecessor one lane down (a DPP wave shift), the odd sample's in the lane itself.
    //      The requests run one sub-batch AHEAD: as soon as a sub-batch's samples have been dealt into their leaves, the same registers are
    //      asked for the next sub-batch's (of this chunk, or the first of the wave's next chunk) and the transform hides the round trip to HBM
    //      -- measured before: 40 % of the wave's clocks were spent waiting for samples (profiles/r05_generic_rate.txt).
    constexpr int SLOTS = WPS >= 4 ? 4 : 8;                          // dwords per lane and frame: fft <= 128 SLOTS
    const int total_items = n_clips * chunks, item_step = gridDim.x * n_waves;
    unsigned pk[PK ? FBMAX : 1][PK ? SLOTS : 1];
    float pk_prev[PK ? FBMAX : 1];
    auto issue = [&](int clip_n, int frame_n) {
      if constexpr (PK) {
        const int16_t *src = (const int16_t *)pcm_v + (size_t)clip_n * P.n_samples;
#pragma unroll
        for (int fj = 0; fj < FBMAX; fj++) {
            const int off = min(frame_n + fj, nfr - 1) * P.frame_stride;     // (frames past the clip's last: re-read, never used)
#pragma unroll
            for (int j = 0; j < SLOTS; j++) pk[fj][j] = *(const unsigned *)(src + off + 2 * min(64 * j + lane, (used >> 1) - 1));
            pk_prev[fj] = (float)src[off == 0 ? P.n_samples - 1 : off - 1] * (1.0f / 32768.0f);
        }
      }
    };
    int item = blockIdx.x * n_waves + wave;
    // (clip, chunk) of the wave's item, advanced by the stride's own (clips, chunks): three integer divisions per wave instead of three per chunk
    int clip = item / chunks, ck = item - clip * chunks;
    const int step_clips = item_step / chunks, step_ck = item_step - step_clips * chunks;
    if constexpr (PK) {
        if (item < total_items) issue(clip, ck * LCH);
    }
    for (; item < total_items; item += item_step, clip += step_clips + (ck + step_ck >= chunks ? 1 : 0), ck = ck + step_ck >= chunks ? ck + step_ck - chunks : ck + step_ck) {
        const int f0 = ck * LCH;
        const int nfc = min(LCH, nfr - f0);
        const size_t cbase = (size_t)clip * P.n_samples;
        auto sample = [&](int n) -> float {
            return F32IN ? ((const float *)pcm_v)[cbase + n] : (float)((const int16_t *)pcm_v)[cbase + n] * (1.0f / 32768.0f);     // numpy::int16_to_float
        };
        for (int fb0 = 0; fb0 < nfc; fb0 += fb) {
          const int nbf = min(fb, nfc - fb0);                        // frames of this sub-batch: transformed together
          if constexpr (PK) {
            {
              // where point i = 64 j + lane goes: its leaf (frame-independent; worked out per sub-batch from the table rather than held in registers)
              int dpt[SLOTS];
#pragma unroll
              for (int j = 0; j < SLOTS; j++) dpt[j] = 2 * zpad(perm[min(64 * j + lane, ncfft - 1)]);
#ifdef KWS_DEV_SWITCHES
              if (fine_prof) { __builtin_amdgcn_s_waitcnt(0x0F70); GPH(1); }      // phase clocks: "the samples have arrived" apart from "dealt into their leaves"
#endif
#pragma unroll
              for (int fj = 0; fj < FBMAX; fj++) if (fj < nbf) {
                  float *Zf = Z + fj * L.zs;
                  const int off = M24(f0 + fb0 + fj, P.frame_stride);
                  // pre-emphasis (processing.hpp:52-138): x[-1] = the window's last sample, or the caller's override
                  float carry = (wrap && off == 0) ? wrap[clip] : pk_prev[fj];        // the sample before point 64 j: lane 63's odd sample of the slot before
#pragma unroll
                  for (int j = 0; j < SLOTS; j++) if (64 * j < ncfft) {
                      const unsigned w = pk[fj][j];
                      const float x0 = (float)(int)(short)(w & 0xffffu) * (1.0f / 32768.0f), x1 = (float)((int)w >> 16) * (1.0f / 32768.0f);   // numpy::int16_to_float
                      const float p0 = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(carry), __float_as_int(x1), 0x138, 0xf, 0xf, false));   // wave_shr:1, lane 0 keeps carry
                      carry = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x1), KWS_WAVE - 1));
                      const float pl0 = P.pre_cof * p0, pl1 = P.pre_cof * x0;
                      const int i = 64 * j + lane;
                      const bool in = 2 * i < used;                            // (used is even here: both samples of a point or neither)
                      const float y0 = in ? x0 - pl0 : 0.0f, y1 = in ? x1 - pl1 : 0.0f;
                      if (i < ncfft) *(float2 *)(Zf + dpt[j]) = make_float2(y0, y1);
                  }
              }
              // the next sub-batch's samples: this chunk's next frames, or the first frames of the wave's next chunk
              if (fb0 + fb < nfc) issue(clip, f0 + fb0 + fb);
              else if (item + item_step < total_items) {
                  const bool carry_ck = ck + step_ck >= chunks;
                  issue(clip + step_clips + (carry_ck ? 1 : 0), (carry_ck ? ck + step_ck - chunks : ck + step_ck) * LCH);
              }
            }
          }
          if constexpr (!PK) {
          // the first trip (a whole frame up to fft 512) of EVERY frame of the sub-batch is requested before any of it is used (round 5: one
          // exposed round trip to HBM per sub-batch instead of one per frame).  A sample's destination in Z (-1: beyond the fft length) is
          // frame-independent: worked out per sub-batch from the leaf table rather than held in registers for the life of the wave
          int dst0[8];
#pragma unroll
          for (int j = 0; j < 8; j++) {
              const int n = 64 * j + lane;
              dst0[j] = n < fft ? 2 * zpad(perm[min(n, fft - 1) >> 1]) + (n & 1) : -1;
          }
          float xq[FBMAX][8], fpq[FBMAX];
#pragma unroll
          for (int fj = 0; fj < FBMAX; fj++) if (fj < nbf) {
              const int off = (f0 + fb0 + fj) * P.frame_stride;
#pragma unroll
              for (int j = 0; j < 8; j++) xq[fj][j] = sample(off + min(64 * j + lane, used - 1));      // (unconditional: a guard per request serialises them)
              fpq[fj] = sample(off == 0 ? P.n_samples - 1 : off - 1);
          }
#ifdef KWS_DEV_SWITCHES
          if (fine_prof) { __builtin_amdgcn_s_waitcnt(0x0F70); GPH(1); }
#endif
#pragma unroll
          for (int fj = 0; fj < FBMAX; fj++) if (fj < nbf) {
            const int fi = fb0 + fj;
            float *Zf = Z + fj * L.zs;
            // ---- pre-emphasis (processing.hpp:52-138; x[-1] = the window's last sample, or the caller's override) + framing: the predecessor of
            //      sample n is what the neighbouring lane has just loaded (a DPP wave shift) -- lane 0's is lane 63's of the slot before, or
            //      one more load for the frame's first sample.
            const int off = (f0 + fi) * P.frame_stride;
            {
                float first_prev = fpq[fj];
                if (wrap && off == 0) first_prev = wrap[clip];
                float carry = first_prev;                                // sample 64 j - 1: lane 63 of the slot before
#pragma unroll
                for (int j = 0; j < 8; j++) if (64 * j < fft) {
                    const float xvj = xq[fj][j];
                    const float prev = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(carry), __float_as_int(xvj), 0x138, 0xf, 0xf, false));   // wave_shr:1, lane 0 keeps carry
                    carry = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xvj), KWS_WAVE - 1));
                    const float pl = P.pre_cof * prev;
                    // sample n is the real (n even) or imaginary (n odd) part of kiss_fftr's input point n / 2, which sits at leaf perm[n / 2]
                    if (dst0[j] >= 0) Zf[dst0[j]] = 64 * j + lane < used ? xvj - pl : 0.0f;
                }
            }
            for (int n0 = 8 * 64; n0 < fft; n0 += 8 * 64) {
                float xv[8], pv[8];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int n = n0 + 64 * j + lane;
                    const int sidx = off + min(n, used - 1);          // clamped: lanes past the used samples re-read one (their y is 0)
                    xv[j] = sample(sidx);
                    pv[j] = sample(sidx == 0 ? P.n_samples - 1 : sidx - 1);
                }
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int n = n0 + 64 * j + lane;
                    const float prev = (wrap && off + n == 0) ? wrap[clip] : pv[j];
                    const float pl = P.pre_cof * prev;
                    if (n < fft) Zf[2 * zpad(perm[n >> 1]) + (n & 1)] = n < used ? xv[j] - pl : 0.0f;
                }
            }
          }
          }
            WAVE_SYNC();
            GPH(0);
            // ---- kf_work's levels, innermost first.  The butterflies of a level are independent of each other, and so are the frames of the
            //      sub-batch: (frame, butterfly) items over the lanes.  Radix-4 levels (every level of a power-of-two length but one) take four
            //      items per lane at a time -- sixteen points and twelve twiddles requested together, then the four butterflies, then the
            //      stores: one LDS round trip per four butterflies instead of one per butterfly (round 5).
#pragma unroll
            for (int l = MAXLEV - 1; l >= 0; l--) {
                if (l >= P.fft_levels) continue;
                const int p = lv_p[l], m = lv_m[l], fstride = lv_fs[l];
                const int nb = ncfft / p, items = nbf * nb;
                const unsigned inv_nb = (1u << 20) / (unsigned)nb + 1u;          // it / nb for it < 4096
                if (p == 4) {
                    for (int it0 = lane; it0 - lane < items; it0 += UB * 64) {
                        cf f[UB][4], t1[UB], t2[UB], t3[UB];
                        int zo[UB], kk[UB];
                        const int slot0 = it0 - lane;                     // wave-uniform: slots past the last item are skipped whole
#pragma unroll
                        for (int u = 0; u < UB; u++) if (slot0 + 64 * u < items) {
                            const int it = min(it0 + 64 * u, items - 1);
                            const int fj = (int)(__umul24((unsigned)it, inv_nb) >> 20), b = it - M24(fj, nb);
                            const int g = (int)(__umul24((unsigned)b, (unsigned)lv_inv[l]) >> 20), k = b - M24(g, m);
                            zo[u] = M24(fj, L.zs); kk[u] = M24(g, 4 * m) + k;
                            const float *Zf = Z + zo[u];
                            f[u][0] = z_ld(Zf, kk[u]); f[u][1] = z_ld(Zf, kk[u] + m); f[u][2] = z_ld(Zf, kk[u] + 2 * m); f[u][3] = z_ld(Zf, kk[u] + 3 * m);
                            const int kf = M24(k, fstride);
                            t1[u] = to_cf(l_tw[kf]); t2[u] = to_cf(l_tw[2 * kf]); t3[u] = to_cf(l_tw[3 * kf]);
                        }
#pragma unroll
                        for (int u = 0; u < UB; u++) if (slot0 + 64 * u < items) bfly4(f[u][0], f[u][1], f[u][2], f[u][3], t1[u], t2[u], t3[u]);
#pragma unroll
                        for (int u = 0; u < UB; u++) {
                            if (it0 + 64 * u < items) {                  // (a lane past the last item re-did the last butterfly: not stored)
                                float *Zf = Z + zo[u];
                                z_st(Zf, kk[u], f[u][0]); z_st(Zf, kk[u] + m, f[u][1]); z_st(Zf, kk[u] + 2 * m, f[u][2]); z_st(Zf, kk[u] + 3 * m, f[u][3]);
                            }
                        }
                    }
                } else {
                    for (int it = lane; it < items; it += 64) {
                        const int fj = (int)(__umul24((unsigned)it, inv_nb) >> 20), b = it - M24(fj, nb);
                        const int g = (int)(__umul24((unsigned)b, (unsigned)lv_inv[l]) >> 20), k = b - M24(g, m);
                        z_bfly_one(Z + M24(fj, L.zs), M24(g, p * m), k, fstride, m, p, l_tw);
                    }
                }
                WAVE_SYNC();
            }
            GPH(2);
            // ---- kiss_fftr's split (kiss_fftr.cpp:84-119) + power spectrum: sqrt(re^2 + im^2) in double, (1/fft) * mag^2; (frame, bin pair)
            //      items over the lanes ---------
            auto power = [&](cf v) -> float {
                const double re = (double)v.r, im = (double)v.i;
                const float mag = (float)g_dsqrt(__fma_rn(re, re, im * im));
                const float sq = mag * mag;
                return (float)(inv_fft * (double)sq);
            };
            if (lane < nbf) {
                const cf t0 = z_ld(Z + M24(lane, L.zs), 0);
                float *ps = PS + M24(fb0 + lane, L.ps_stride);
                cf dc, ny;
                dc.r = t0.r + t0.i; dc.i = 0.0f;
                ny.r = t0.r - t0.i; ny.i = 0.0f;
                ps[0] = power(dc);
                ps[ncfft] = power(ny);
            }
            {
                const int hb = ncfft / 2, items = nbf * hb;
                const unsigned inv_hb = (1u << 20) / (unsigned)hb + 1u;
                for (int it = lane; it < items; it += 64) {
                    const int fj = (int)(__umul24((unsigned)it, inv_hb) >> 20), k = 1 + (it - M24(fj, hb));
                    const float *Zf = Z + M24(fj, L.zs);
                    float *ps = PS + M24(fb0 + fj, L.ps_stride);
                    const cf fpk = z_ld(Zf, k);
                    cf fpnk = z_ld(Zf, ncfft - k);
                    fpnk.i = -fpnk.i;
                    const cf f1k = cadd(fpk, fpnk), f2k = csub(fpk, fpnk);
                    const cf twv = cmul(f2k, to_cf(l_stw[k - 1]));
                    cf lo, hi;
                    lo.r = (f1k.r + twv.r) * 0.5f;
                    lo.i = (f1k.i + twv.i) * 0.5f;
                    hi.r = (f1k.r - twv.r) * 0.5f;
                    hi.i = (twv.i - f1k.i) * 0.5f;
                    // bin ncfft / 2 is written twice by the reference (k == ncfft - k): the second store wins -- the same value pair here
                    const float plo = power(lo), phi = power(hi);
                    ps[k] = plo;
                    ps[ncfft - k] = phi;
                }
            }
            WAVE_SYNC();
            GPH(3);
        }
        // ---- frame energies: numpy::sum, ascending (numpy.hpp:88-94): a lane per frame --------------------------------------
        float energy = 0.0f;
        if (lane < nfc) {
            const float *ps = PS + M24(lane, L.ps_stride);
            int k = 0;
            for (; k + EB <= nbins; k += EB) {                     // thirty-two (sixteen) reads in flight, then the ordered additions
                float v[EB];
#pragma unroll
                for (int e = 0; e < EB; e++) v[e] = ps[k + e];
#pragma unroll
                for (int e = 0; e < EB; e++) energy += v[e];
            }
            for (; k + 8 <= nbins; k += 8) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; e++) v[e] = ps[k + e];
#pragma unroll
                for (int e = 0; e < 8; e++) energy += v[e];
            }
            for (; k < nbins; k++) energy += ps[k];
            if (energy == 0.0f) energy = FLT_EPSILON;
            if (P.mfe_energy) P.mfe_energy[(size_t)clip * nfr + f0 + lane] = energy;
        }
        GPH(4);
        // ---- mel filterbank: dot_by_row over the non-zero weights in ascending bin order, zero handling, log ----------------
        for (int it = lane; it < nfc * NF; it += 64) {
            const int fi = div20(it, r_nf), j = it - M24(fi, NF);
            const float *ps = PS + M24(fi, L.ps_stride);
            float acc = 0.0f;
            const int n1 = l_fstart[j + 1];
            for (int n = l_fstart[j]; n < n1; n += 4) {            // four taps per trip (the tables are padded by four): bins, then their power values, then
                int bn[4];                                         // the ordered sum; a tap past the filter's end adds an exact +0 (acc >= +0: no bit moves)
                float wv[4], pv[4];
#pragma unroll
                for (int e = 0; e < 4; e++) { bn[e] = l_fbin[n + e]; wv[e] = l_fw[n + e]; }
#pragma unroll
                for (int e = 0; e < 4; e++) pv[e] = ps[bn[e]];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float prod = n + e < n1 ? pv[e] * wv[e] : 0.0f;
                    acc += prod;
                }
            }
            if (acc == 0.0f) acc = FLT_EPSILON;
            if (P.mfe_mel) P.mfe_mel[(size_t)clip * out_stride + (size_t)ring_out_row(P, f0 + fi) * NF + j] = acc;
            MEL[M24(fi, L.mel_stride) + j] = fast_log(acc);
        }
        WAVE_SYNC();
        GPH(5);
        if (!P.mfe_mel) {
            // ---- numpy::dct2 (numpy.hpp:378-401) -> dct::transform (fast-dct-fft.cpp:37-80) for the chunk's frames AT ONCE: even/odd reorder
            //      into kf_work's leaf order, then the NF/2-point transform's levels with (frame, butterfly) items over the lanes (a lane per
            //      frame walking g_rfft on generic pointers was 30 % of the first version), kiss_fftr's split, v[i] = re cos + im sin for
            //      i <= NF/2 only, x2, ortho scale; c0 <- log(energy).  Per frame: a padded complex buffer of NF/2 points + the split's output.
            const int nc = NF >> 1, zs = 2 * (nc + (nc >> 4) + 1), ss = 2 * (nc + 1);      // floats per frame: transform buffer, spectrum
            float *DZ = DCT, *DS = DCT + LCH * zs;
            for (int it = lane; it < nfc * nc; it += 64) {
                const int fi = div20(it, r_nc), o = it - M24(fi, nc);
                const int i = dperm[o];
                // dct::transform's reorder: in[j] = v[2 j], in[NF - 1 - j] = v[2 j + 1] for j < NF / 2; kiss_fftr reads (in[2 i], in[2 i + 1])
                const float *mel = MEL + M24(fi, L.mel_stride);
                auto reord = [&](int q) { return q < nc ? mel[2 * q] : mel[2 * (NF - 1 - q) + 1]; };
                cf v; v.r = reord(2 * i); v.i = reord(2 * i + 1);
                z_st(DZ + M24(fi, zs), o, v);
            }
            WAVE_SYNC();
            for (int l = P.dct_levels - 1; l >= 0; l--) {
                const int p = P.dct_fac[2 * l], m = P.dct_fac[2 * l + 1];
                int fstride = 1;
                for (int q = 0; q < l; q++) fstride *= P.dct_fac[2 * q];
                const int nb = nc / p;
                const unsigned r_nb = recip20(nb), r_m = recip20(m);
                for (int it = lane; it < nfc * nb; it += 64) {
                    const int fi = div20(it, r_nb), b = it - M24(fi, nb);
                    const int g = div20(b, r_m), k = b - M24(g, m);
                    z_bfly_one(DZ + M24(fi, zs), M24(g, p * m), k, fstride, m, p, l_dtw);
                }
                WAVE_SYNC();
            }
            for (int it = lane; it < nfc * (nc / 2 + 1); it += 64) {
                const int fi = div20(it, r_half), k = it - M24(fi, nc / 2 + 1);
                const float *Zf = DZ + M24(fi, zs);
                float *sp = DS + M24(fi, ss);
                if (k == 0) {
                    const cf t0 = z_ld(Zf, 0);
                    sp[0] = t0.r + t0.i; sp[1] = 0.0f;
                    sp[2 * nc] = t0.r - t0.i; sp[2 * nc + 1] = 0.0f;
                } else {
                    const cf fpk = z_ld(Zf, k);
                    cf fpnk = z_ld(Zf, nc - k);
                    fpnk.i = -fpnk.i;
                    const cf f1k = cadd(fpk, fpnk), f2k = csub(fpk, fpnk);
                    const cf twv = cmul(f2k, to_cf(l_dstw[k - 1]));
                    sp[2 * k] = (f1k.r + twv.r) * 0.5f;
                    sp[2 * k + 1] = (f1k.i + twv.i) * 0.5f;
                    sp[2 * (nc - k)] = (f1k.r - twv.r) * 0.5f;             // (k == nc - k: written second, as in the reference)
                    sp[2 * (nc - k) + 1] = (twv.i - f1k.i) * 0.5f;
                }
            }
            WAVE_SYNC();
            // the energies sit in the lanes of the first frames: hand them to the items through LDS (the transform buffer is dead)
            if (lane < nfc) DZ[lane] = energy;
            WAVE_SYNC();
            for (int it = lane; it < nfc * ncep; it += 64) {
                const int fi = div20(it, r_ncep), i = it - M24(fi, ncep);
                const float *mel = MEL + M24(fi, L.mel_stride), *sp = DS + M24(fi, ss);
                float d;
                if (i <= nc) {
                    const float a = sp[2 * i] * l_dcos[i];
                    const float b = sp[2 * i + 1] * l_dsin[i];
                    d = a + b;
                } else {
                    d = mel[i];                                        // never written by the transform: the input stays
                }
                d = d * 2.0f;
                d = d * (i == 0 ? P.dct_s0 : P.dct_s1);
                if (i == 0) d = fast_log(DZ[fi]);                      // c0 <- log(energy), feature.hpp:425-429
                mfcc_out[(size_t)clip * out_stride + (size_t)ring_out_row(P, f0 + fi) * ncep + i] = d;
            }
        }
        GPH(6);
        WAVE_SYNC();
    }
}

// processing::cmvnw over cepstra in HBM: one thread per (clip, row, column); the window's terms in the reference's order (fp32
// running mean, double square accumulated into a float after every term: numpy.hpp:746-836)
__global__ void kws_cmvn_generic_kernel(KwsDspPlan P, const float *__restrict__ mfcc, size_t n_elems, float *__restrict__ features,
                                        int8_t *__restrict__ q_out, float in_scale, int in_zp)
{
    const int nfr = P.n_frames, ncep = P.n_cepstral, win = P.win_size;
    const size_t per_clip = (size_t)nfr * ncep;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_elems; e += (size_t)gridDim.x * blockDim.x) {
        const size_t clip = e / per_clip;
        const int rc = (int)(e - clip * per_clip);
        const int r = rc / ncep, c = rc - r * ncep;
        const float *m = mfcc + clip * per_clip + c;
        float sum = 0.0f;
        for (int j = 0; j < win; j++) sum += m[(size_t)P.pad_map[r + j] * ncep];
        const float fwin = (float)win;
        const float mean = sum / fwin;
        float sd = 0.0f;
        for (int j = 0; j < win; j++) {
            const float d = m[(size_t)P.pad_map[r + j] * ncep] - mean;
            const double dd = (double)d;
            sd = (float)__fma_rn(dd, dd, (double)sd);
        }
        const float dev = sqrtf(sd / fwin);
        const float o = (m[(size_t)r * ncep] - mean) / (dev + FLT_EPSILON);
        if (features) features[e] = o;
        if (q_out) q_out[e] = quantize_feature(o, in_scale, in_zp);
    }
}

// processing::cmvnw (processing.hpp:326-389) with the clip's PADDED cepstra matrix in LDS (round 4): the kernel above asks global memory for the
// pad map's entry and then for the cepstrum it names, twice per window term -- two dependent round trips where the arithmetic is three
// instructions.  Here one wave stages rows pad_map[0 .. nfr + 2 pad) of its clip once (numpy::pad_1d_symmetric, numpy.hpp:479-541, as an index map),
// and the window of output row r is then padded rows r .. r + win - 1: one LDS read per term, lanes on consecutive elements (consecutive
// addresses: no bank conflict), the sums in the reference's order term by term as before -- the same expressions on the same values.
__global__ __launch_bounds__(64) void kws_cmvn_lds_kernel(KwsDspPlan P, const float *__restrict__ mfcc, int n_clips, float *__restrict__ features,
                                                          int8_t *__restrict__ q_out, float in_scale, int in_zp)
{
    extern __shared__ __attribute__((aligned(16))) float cpm[];
    const int nfr = P.n_frames, ncep = P.n_cepstral, win = P.win_size, lane = threadIdx.x;
    const int per_clip = nfr * ncep, padded = (nfr + 2 * P.pad) * ncep;
    const float fwin = (float)win;
    for (int clip = blockIdx.x; clip < n_clips; clip += gridDim.x) {
        const float *src = mfcc + (size_t)clip * per_clip;
        for (int i = lane; i < padded; i += 64) {
            const int row = i / ncep, c = i - row * ncep;
            cpm[i] = src[P.pad_map[row] * ncep + c];
        }
        WAVE_SYNC();
        for (int e = lane; e < per_clip; e += 64) {
            const float *w0 = cpm + e;                                  // padded row r, column c of element e = r ncep + c
            float sum = 0.0f;
#pragma unroll 8
            for (int j = 0; j < win; j++) sum += w0[j * ncep];
            const float mean = sum / fwin;
            float sd = 0.0f;
#pragma unroll 8
            for (int j = 0; j < win; j++) {
                const float d = w0[j * ncep] - mean;
                const double dd = (double)d;
                sd = (float)__fma_rn(dd, dd, (double)sd);
            }
            const float dev = sqrtf(sd / fwin);
            const float o = (w0[P.pad * ncep] - mean) / (dev + FLT_EPSILON);      // the row itself: padded row r + pad
            const size_t out = (size_t)clip * per_clip + e;
            if (features) features[out] = o;
            if (q_out) q_out[out] = quantize_feature(o, in_scale, in_zp);
        }
        WAVE_SYNC();                                                    // the next clip overwrites the matrix
    }
}

// ---------------------------------------------------------------------------------------------------------
// the LDS-resident kernel serves every general configuration whose arrays fit a CU's LDS twice over (fft up to 2048 with 64 filters);
// KWS_DEV_GENERIC_SCRATCH=1 keeps the round-1 kernel with its scratch in HBM (same-box comparisons)
bool kws_generic_uses_lds(const KwsDspPlan &P)
{
    static const bool forced_scratch = KWS_DEV_ENV("KWS_DEV_GENERIC_SCRATCH") != nullptr;
    return !forced_scratch && (size_t)lds_layout(P.fft_len, P.n_filters, P.n_bins, P.filt_nnz, KWS_LCH_DEFAULT).total * sizeof(float) <= 72 * 1024;
}

// development aid (not in the public headers): read and clear the phase clocks of kws_spectral_lds_kernel's workgroup 0
extern "C" __attribute__((visibility("default"))) int kws_dev_generic_prof(long long *out8)
{
    long long zero[8] = { 0 };
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_gen_prof), sizeof(zero)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_gen_prof), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}

size_t kws_generic_ws_bytes(const KwsDspPlan &P, int grid) { if (kws_generic_uses_lds(P)) return 64; return (size_t)grid * g_ws_floats(P.fft_len, P.n_filters) * GL * sizeof(float); }

// How a launch of kws_spectral_lds_kernel is shaped: fb = frames of a chunk transformed together (4, 2, 1; at most the chunk), waves = waves per
// workgroup (they share one copy of the tables), per_cu = workgroups a CU holds.  The registers allow two waves per SIMD (eight per CU); the LDS
// decides how many of them a CU gets: the tables once per workgroup + each wave's buffers.
struct GenericLaunch { int fb, waves, per_cu; size_t smem; };
static GenericLaunch pick_launch(const KwsDspPlan &P, int LCH, int force_fb, int force_waves, int wps = 2)
{
    const size_t cu_lds = 160 * 1024;
    GenericLaunch best = { 1, 1, 1, 0 };
    double best_score = -1.0;
    for (int fb = 4; fb >= 1; fb >>= 1) {
        if (fb > LCH || (force_fb && fb != force_fb) || (wps >= 3 && fb > KWS_GEN_FB)) continue;          // (the four-wave build requests fewer frames' samples together)
        const LdsLayout L = lds_layout(P.fft_len, P.n_filters, P.n_bins, P.filt_nnz, LCH, fb);
        for (int w = wps >= 4 ? 8 : 4; w >= 1; w >>= 1) {
            if (force_waves && w != force_waves) continue;
            const size_t smem = ((size_t)(L.shared + w * L.wave) * sizeof(float) + 1023) & ~(size_t)1023;      // (allocation granularity: assumed 1 KB at most)
            if (smem > cu_lds) continue;
            const int per_cu = (int)std::min<size_t>(4 * wps / w, cu_lds / smem);
            // measured (profiles/r05_generic_rate.txt): frames transformed together are worth more than the same factor in resident waves up to
            // four, and resident waves are worth their number
            const double score = (double)(per_cu * w) * (fb == 4 ? 1.3 : fb == 2 ? 1.15 : 1.0);
            if (score > best_score) { best_score = score; best = GenericLaunch{ fb, w, per_cu, smem }; }
        }
    }
    if (best_score < 0.0) {                                   // nothing fits (kws_generic_uses_lds admits only layouts that do): one wave, one frame
        const LdsLayout L = lds_layout(P.fft_len, P.n_filters, P.n_bins, P.filt_nnz, LCH, 1);
        best = GenericLaunch{ 1, 1, 1, (size_t)L.total * sizeof(float) };
    }
    return best;
}

// one instantiation's launch: its dynamic-LDS ceiling is raised once per device
template <bool F32IN, int LCH, int WPS, bool PK>
static int launch_spectral_lds_at(const KwsDspPlan &P, const void *pcm, int n_clips, float *mfcc_out, const float *wrap, int out_stride,
                                  dim3 gd, dim3 bd, size_t smem, int fb_arg, hipStream_t stream)
{
    static std::atomic<unsigned long long> attr_done{ 0 };
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return (int)hipGetLastError();
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_done.load(std::memory_order_acquire) & bit)) {
        if (hipFuncSetAttribute((const void *)kws_spectral_lds_kernel<F32IN, LCH, WPS, PK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return (int)hipGetLastError();
        attr_done.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL((kws_spectral_lds_kernel<F32IN, LCH, WPS, PK>), gd, bd, smem, stream, P, pcm, n_clips, mfcc_out, wrap, out_stride, fb_arg);
    return (int)hipGetLastError();
}

template <int LCH>
static int launch_spectral_lds(const KwsDspPlan &P, const void *pcm, int pcm_is_float, int n_clips, float *mfcc_out, const float *wrap, int out_stride,
                               int grid, hipStream_t stream)
{
    // development aids: KWS_DEV_GENERIC_FB = frames together (1 = round 4's schedule), KWS_DEV_GENERIC_WAVES = waves per workgroup (1 = tables per wave)
    static const char *fb_env = KWS_DEV_ENV("KWS_DEV_GENERIC_FB"), *w_env = KWS_DEV_ENV("KWS_DEV_GENERIC_WAVES");
    const int force_fb = fb_env ? std::max(1, std::min(std::min(LCH, 4), atoi(fb_env))) : 0;
    const int force_w = w_env ? std::max(1, std::min(8, atoi(w_env))) : 0;
    const int fw = (force_w == 3 || (force_w > 4 && force_w < 8)) ? 4 : force_w, ffb = force_fb == 3 ? 2 : force_fb;
    // Two builds of the kernel: registers for FOUR waves per SIMD with shallow batches (one radix-4 butterfly per lane and round trip, two
    // frames' samples requested together), and for TWO with deep ones (four and four).  Measured on five shapes (profiles/r05_generic_rate.txt):
    // the kernel is bound by latency, not by any pipe -- sixteen resident waves with shallow batches beat eight with deep ones by 9 - 20 %
    // wherever the LDS lets a CU hold more than eight (fft <= 512); where it does not (fft 1024: 12.6 KB per wave), the deep build is 8 % ahead.
    int wps = 4;
    {
        const GenericLaunch G4 = pick_launch(P, LCH, ffb > KWS_GEN_FB ? 0 : ffb, fw, 4);
        if (G4.per_cu * G4.waves <= 8 || P.fft_len > 512) wps = 2;          // (the four-wave build holds four sample dwords per lane and frame: fft <= 512)
    }
    const char *wps_env = KWS_DEV_ENV("KWS_DEV_GENERIC_WPS");              // development aid: 2 or 4 pins the build (read per call: a test toggles it)
    if (wps_env && (atoi(wps_env) == 2 || atoi(wps_env) == 4)) wps = atoi(wps_env);
    const GenericLaunch G = pick_launch(P, LCH, (wps == 4 && ffb > KWS_GEN_FB) ? 0 : ffb, fw, wps);
    const long litems = (long)n_clips * ((P.n_frames + LCH - 1) / LCH);
    // persistent-ish grid: twice the workgroups the chip holds at once (grid = 8 x CUs from the caller), at most one wave per item
    long lgrid = (long)(grid / 8) * G.per_cu * 2;
    if (lgrid * G.waves > litems) lgrid = (litems + G.waves - 1) / G.waves;
    if (lgrid < 1) lgrid = 1;
    const dim3 gd((unsigned)lgrid), bd(KWS_WAVE * G.waves);
    static const char *prof_env = KWS_DEV_ENV("KWS_DEV_GENERIC_PROF");
    const int fb_arg = G.fb | (prof_env ? 256 : 0);
    // pair loads (see the kernel's item loop): int16 input whose frames start on even samples and use an even number of them
    const int used = std::min(P.frame_len, P.fft_len);
    const char *nopairs_env = KWS_DEV_ENV("KWS_DEV_GENERIC_NOPAIRS");             // development aid: the sample-by-sample loads (A/B runs, tests; read per call)
    const bool pairs = !pcm_is_float && !nopairs_env && P.n_samples % 2 == 0 && P.frame_stride % 2 == 0 && used % 2 == 0 && used >= 2 && ((uintptr_t)pcm & 3) == 0 &&
                       P.fft_len <= (wps == 4 ? 512 : 1024);
#define KWS_GEN_GO(F, W, K) launch_spectral_lds_at<F, LCH, W, K>(P, pcm, n_clips, mfcc_out, wrap, out_stride, gd, bd, G.smem, fb_arg, stream)
    if (pcm_is_float) return wps == 4 ? KWS_GEN_GO(true, 4, false) : KWS_GEN_GO(true, 2, false);
    if (pairs) return wps == 4 ? KWS_GEN_GO(false, 4, true) : KWS_GEN_GO(false, 2, true);
    return wps == 4 ? KWS_GEN_GO(false, 4, false) : KWS_GEN_GO(false, 2, false);
#undef KWS_GEN_GO
}

// lch: frames per chunk of the LDS kernel, 4 or 8 (anything else: 8); ignored by the scratch kernel
int kws_launch_spectral_generic(const KwsDspPlan &P, const void *pcm, int pcm_is_float, int n_clips, float *mfcc_out, const float *wrap,
                                int out_stride, float *ws, int grid, int lch, hipStream_t stream)
{
    (void)hipGetLastError();
    if (n_clips <= 0) return 0;
    if (out_stride == 0) out_stride = P.n_frames * (P.mfe_mel ? P.n_filters : P.n_cepstral);
    if (kws_generic_uses_lds(P))
        return lch == 4 ? launch_spectral_lds<4>(P, pcm, pcm_is_float, n_clips, mfcc_out, wrap, out_stride, grid, stream)
                        : launch_spectral_lds<8>(P, pcm, pcm_is_float, n_clips, mfcc_out, wrap, out_stride, grid, stream);
    const long items = (long)n_clips * ((P.n_frames + GL - 1) / GL);
    if (items < grid) grid = (int)items;
    if (pcm_is_float)
        hipLaunchKernelGGL(kws_spectral_generic_kernel<true>, dim3(grid), dim3(GL), 0, stream, P, pcm, n_clips, mfcc_out, wrap, out_stride, ws);
    else
        hipLaunchKernelGGL(kws_spectral_generic_kernel<false>, dim3(grid), dim3(GL), 0, stream, P, pcm, n_clips, mfcc_out, wrap, out_stride, ws);
    return (int)hipGetLastError();
}

int kws_launch_cmvn_generic(const KwsDspPlan &P, const float *mfcc, int n_clips, float *features, int8_t *q_out, float in_scale, int in_zp,
                            hipStream_t stream)
{
    (void)hipGetLastError();
    if (n_clips <= 0) return 0;
    // the padded matrix of one clip in LDS when it fits 64 KB (199 x 13 floats = 10 KB for a 2 s window; 342 x 64 would be 87 KB); KWS_DEV_CMVN_GLOBAL=1
    // keeps the one-thread-per-element kernel (same-box comparisons)
    const size_t smem = (size_t)(P.n_frames + 2 * P.pad) * P.n_cepstral * sizeof(float);
    static const bool forced_global = KWS_DEV_ENV("KWS_DEV_CMVN_GLOBAL") != nullptr;
    if (smem <= 64 * 1024 && !forced_global) {
        // one-wave workgroups, as many as the LDS lets a CU hold (up to 16), over 256 CUs' worth
        const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(16, (160 * 1024) / std::max<size_t>(smem, 1)));
        const int grid = std::min(n_clips, 256 * per_cu);
        hipLaunchKernelGGL(kws_cmvn_lds_kernel, dim3(grid), dim3(64), smem, stream, P, mfcc, n_clips, features, q_out, in_scale, in_zp);
        return (int)hipGetLastError();
    }
    const size_t n = (size_t)n_clips * P.n_frames * P.n_cepstral;
    const int grid = (int)std::min<size_t>((n + 255) / 256, 65536);
    hipLaunchKernelGGL(kws_cmvn_generic_kernel, dim3(grid), dim3(256), 0, stream, P, mfcc, n, features, q_out, in_scale, in_zp);
    return (int)hipGetLastError();
}
