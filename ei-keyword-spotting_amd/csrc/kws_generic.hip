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
#include "kws_device.h"

#define GL 64                                     // lanes interleaved in the scratch

__device__ __forceinline__ cf g_ld(const float *c, int idx) { cf v; v.r = c[(size_t)(2 * idx) * GL]; v.i = c[(size_t)(2 * idx + 1) * GL]; return v; }
__device__ __forceinline__ void g_st(float *c, int idx, cf v) { c[(size_t)(2 * idx) * GL] = v.r; c[(size_t)(2 * idx + 1) * GL] = v.i; }

// kf_bfly2 / kf_bfly3 / kf_bfly4 / kf_bfly5 (kiss_fft.cpp:15-192) on the sub-array starting at complex index `base`
__device__ void g_bfly(float *F, int base, int fstride, int m, int p, const float2 *__restrict__ tw)
{
    if (p == 2) {
        for (int k = 0; k < m; k++) {
            const cf t = cmul(g_ld(F, base + k + m), to_cf(tw[k * fstride]));
            const cf a = g_ld(F, base + k);
            g_st(F, base + k + m, csub(a, t));
            g_st(F, base + k, cadd(a, t));
        }
    } else if (p == 4) {
        for (int k = 0; k < m; k++) {
            cf f0 = g_ld(F, base + k), f1 = g_ld(F, base + k + m), f2 = g_ld(F, base + k + 2 * m), f3 = g_ld(F, base + k + 3 * m);
            bfly4(f0, f1, f2, f3, to_cf(tw[k * fstride]), to_cf(tw[k * fstride * 2]), to_cf(tw[k * fstride * 3]));
            g_st(F, base + k, f0); g_st(F, base + k + m, f1); g_st(F, base + k + 2 * m, f2); g_st(F, base + k + 3 * m, f3);
        }
    } else if (p == 3) {
        const cf epi3 = to_cf(tw[fstride * m]);
        for (int k = 0; k < m; k++) {
            cf f0 = g_ld(F, base + k);
            const cf s1 = cmul(g_ld(F, base + k + m), to_cf(tw[k * fstride]));
            const cf s2 = cmul(g_ld(F, base + k + 2 * m), to_cf(tw[k * fstride * 2]));
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
            g_st(F, base + k, f0); g_st(F, base + k + m, f1); g_st(F, base + k + 2 * m, f2);
        }
    } else {    // 5
        const cf ya = to_cf(tw[fstride * m]), yb = to_cf(tw[fstride * 2 * m]);
        for (int u = 0; u < m; u++) {
            cf F0 = g_ld(F, base + u), F1 = g_ld(F, base + u + m), F2 = g_ld(F, base + u + 2 * m), F3 = g_ld(F, base + u + 3 * m), F4 = g_ld(F, base + u + 4 * m);
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
            g_st(F, base + u, F0); g_st(F, base + u + m, F1); g_st(F, base + u + 2 * m, F2); g_st(F, base + u + 3 * m, F3); g_st(F, base + u + 4 * m, F4);
        }
    }
}

// kiss_fftr (kiss_fftr.cpp:66-120) of the real array `in` [nfft] -> spectrum `spec` [nfft/2 + 1] complex; `tmp` [nfft/2] complex.
// kf_work's recursion (kiss_fft.cpp:232-296) is replayed level by level: the leaves' strided copies first (a mixed-radix digit
// reversal), then the butterflies of every level from the innermost out -- sub-transforms of one level are independent, so the
// order between them does not matter, the order inside one butterfly is the reference's.
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
        cf v; v.r = in[(size_t)(2 * i) * GL]; v.i = in[(size_t)(2 * i + 1) * GL];
        g_st(tmp, o, v);
    }
    for (int l = n_levels - 1; l >= 0; l--) {
        const int p = fac[2 * l], m = fac[2 * l + 1];
        int fstride = 1;
        for (int q = 0; q < l; q++) fstride *= fac[2 * q];
        for (int base = 0; base < ncfft; base += p * m) g_bfly(tmp, base, fstride, m, p, tw);
    }
    const cf t0 = g_ld(tmp, 0);
    cf dc, ny;
    dc.r = t0.r + t0.i; dc.i = 0.0f;
    ny.r = t0.r - t0.i; ny.i = 0.0f;
    g_st(spec, 0, dc);
    g_st(spec, ncfft, ny);
    for (int k = 1; k <= ncfft / 2; k++) {
        const cf fpk = g_ld(tmp, k);
        cf fpnk = g_ld(tmp, ncfft - k);
        fpnk.i = -fpnk.i;
        const cf f1k = cadd(fpk, fpnk), f2k = csub(fpk, fpnk);
        const cf twv = cmul(f2k, to_cf(stw[k - 1]));
        cf lo, hi;
        lo.r = (f1k.r + twv.r) * 0.5f;
        lo.i = (f1k.i + twv.i) * 0.5f;
        hi.r = (f1k.r - twv.r) * 0.5f;
        hi.i = (twv.i - f1k.i) * 0.5f;
        g_st(spec, k, lo);
        g_st(spec, ncfft - k, hi);
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
        g_rfft(w_in, w_tmp, w_spec, fft, P.fft_fac, P.fft_levels, P.tw, P.stw);
        float energy = 0.0f;
        for (int k = 0; k < nbins; k++) {
            const cf v = g_ld(w_spec, k);
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
        g_rfft(w_in, w_tmp, w_spec, NF, P.dct_fac, P.dct_levels, P.dct_tw, P.dct_stw);
        float *orow = mfcc_out + (size_t)clip * out_stride + (size_t)ring_out_row(P, f) * ncep;
        for (int i = 0; i < ncep; i++) {
            float d;
            if (i <= half) {
                const cf r = g_ld(w_spec, i);
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

// ---------------------------------------------------------------------------------------------------------
size_t kws_generic_ws_bytes(const KwsDspPlan &P, int grid) { return (size_t)grid * g_ws_floats(P.fft_len, P.n_filters) * GL * sizeof(float); }

int kws_launch_spectral_generic(const KwsDspPlan &P, const void *pcm, int pcm_is_float, int n_clips, float *mfcc_out, const float *wrap,
                                int out_stride, float *ws, int grid, hipStream_t stream)
{
    (void)hipGetLastError();
    if (n_clips <= 0) return 0;
    if (out_stride == 0) out_stride = P.n_frames * (P.mfe_mel ? P.n_filters : P.n_cepstral);
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
    const size_t n = (size_t)n_clips * P.n_frames * P.n_cepstral;
    const int grid = (int)std::min<size_t>((n + 255) / 256, 65536);
    hipLaunchKernelGGL(kws_cmvn_generic_kernel, dim3(grid), dim3(256), 0, stream, P, mfcc, n, features, q_out, in_scale, in_zp);
    return (int)hipGetLastError();
}
