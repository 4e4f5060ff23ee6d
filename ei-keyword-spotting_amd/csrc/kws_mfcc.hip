// kws_mfcc.hip -- kws_mfcc_kernel: one wavefront owns one 1 s / 16 kHz clip.  Replaces the reference's
// extract_mfcc_features() (SDK/classifier/ei_run_dsp.h:256-308): coalesced 16-byte int16 loads, pre-emphasis in registers,
// the 256-point real FFT staged in LDS, power spectrum, sparse mel gather, fast log, DCT, windowed CMVN and the int8
// quantisation of ei_run_classifier.h:436-444.  See kws_device.h for the bit-exactness contract.
// Two shapes of the same code: the throughput shape (one wave per clip, persistent grid) and, for calls with a handful of
// windows, the latency shape (LW waves per clip: frames and cmvnw tasks dealt out over a workgroup).  Built without the SLP
// vectoriser (see the Makefile); the DCT constants are literals (kws_dct_tables.h).
// kws_mfcc8_kernel (further down) is the throughput shape for int16 PCM on the spectral layout of kws_fast_kernel -- eight lanes
// per frame, one LDS exchange per FFT instead of three -- with the same arithmetic, operation by operation; kws_mfcc_kernel keeps
// the float-sample calls (the SDK's signal_t callback), the latency shape, and stays selectable for int16 PCM through the
// development switch KWS_DEV_MFCC_OLD_LAYOUT (tools/gpu_mfcc_layout_check.py compares the two bit for bit, stage by stage).
#include <stdlib.h>

#include "kws_device.h"
#include "kws_dct_tables.h"

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
__device__ __forceinline__ void dct_spectrum(const float (&v)[NF], Emit emit)   // emit(i, R[i]), i = 0..NF/2
{
    constexpr int NC = NF / 2;
    // twiddles as literals (kws_dct_tables.h): every index below is a compile-time constant once the loops are unrolled.
    // From the plan's tables they are scalar loads whose registers get spilled and reloaded around every output.
    typedef KwsDctTab<NF> T;
    auto tw = [](int i) { cf c; c.r = T::tw_r[i]; c.i = T::tw_i[i]; return c; };
    auto stw = [](int i) { cf c; c.r = T::stw_r[i]; c.i = T::stw_i[i]; return c; };
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
        const cf d0 = tw(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) bfly4(F[4 * q], F[4 * q + 1], F[4 * q + 2], F[4 * q + 3], d0, d0, d0);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            bfly4(F[k], F[k + 4], F[k + 8], F[k + 12], tw(k), tw(2 * k), tw(3 * k));
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
        const cf d0 = tw(0), ya = tw(4), yb = tw(8);   // tw[fstride*m], tw[2*fstride*m]
#pragma unroll
        for (int q = 0; q < 4; ++q) bfly5(F[5 * q], F[5 * q + 1], F[5 * q + 2], F[5 * q + 3], F[5 * q + 4], d0, d0, d0, d0, ya, yb);
#pragma unroll
        for (int k = 0; k < 5; ++k)
            bfly4(F[k], F[k + 5], F[k + 10], F[k + 15], tw(k), tw(2 * k), tw(3 * k));
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
        cf twv = cmul(f2k, stw(k - 1));
        cf lo, hi;
        lo.r = (f1k.r + twv.r) * 0.5f;
        lo.i = (f1k.i + twv.i) * 0.5f;
        hi.r = (f1k.r - twv.r) * 0.5f;
        hi.i = (twv.i - f1k.i) * 0.5f;
        if (k != NC - k) emit(k, lo);                              // k == ncfft/2: overwritten by the "ncfft-k" store
        emit(NC - k, hi);
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

// Latency mode (LW > 0 waves per clip, one workgroup per clip, for calls with a handful of windows such as run_classifier()):
// every wave transforms its own 2 * CHP frames with private FFT / power-spectrum buffers; the log-mel / cepstra matrix, the
// frame energies and the pad map are shared, and cmvnw's (row group, column block) tasks are dealt out over the waves.
template <int CHP, int NF, int LW>
struct alignas(16) MfccSmemLat {
    static constexpr int CHF = 2 * CHP + 2;           // row stride of the power-spectrum buffer (two spare slots: the 40-filter mel
                                                      // stage reads frame slots in threes)
    static constexpr int MELS = NF + 1;
    float z[LW][2][KWS_ZF];
    float p[LW][KWS_NBINS * CHF];
    float dcny[LW][2 * CHF];
    int map[KWS_MAXPROW];
    float mel[kws_mel_rows(NF) * MELS];
    float energy[kws_mel_rows(NF)];
};
template <int CHP, int NF, int LW> struct MfccSmemSel { typedef MfccSmemLat<CHP, NF, LW> type; };
template <int CHP, int NF> struct MfccSmemSel<CHP, NF, 0> { typedef MfccSmem<CHP, NF> type; };

template <int CHP, bool F32IN, bool WITH_CMVN, int NZ, int NF = 32, bool PROF = false, bool WIDE = false, int LW = 0>
__global__ __launch_bounds__(LW ? KWS_WAVE * LW : KWS_WAVE, LW ? 1 : 2) void kws_mfcc_kernel(KwsDspPlan P, const void *__restrict__ pcm_v, int n_clips,
                                                            float *__restrict__ features, int8_t *__restrict__ q_out,
                                                            float in_scale, int in_zp, const float *__restrict__ wrap, int out_stride,
                                                            long long *prof_out = nullptr, const int *__restrict__ sel = nullptr)
{
    constexpr int CHF = 2 * CHP + (LW ? 2 : 0);          // frame slots per power-spectrum row (stride)
    constexpr int MELS = NF + 1, NCEPT = NF / 2 + 1;     // DCT only produces outputs 0..NF/2 (fast-dct-fft.cpp:71)
    __shared__ typename MfccSmemSel<CHP, NF, LW>::type sm;
    const int lane = threadIdx.x & (KWS_WAVE - 1), wave = LW ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
    const int half = lane >> 5, t = lane & 31;
    // this wave's view of the LDS block (one wave per workgroup unless LW > 0)
    float *zw, *sm_p, *sm_dcny, *sm_mel, *sm_energy;
    int *sm_map;
    if constexpr (LW > 0) {
        zw = &sm.z[wave][0][0]; sm_p = sm.p[wave]; sm_dcny = sm.dcny[wave]; sm_map = sm.map;
    } else {
        zw = &sm.z[0][0]; sm_p = sm.u.p; sm_dcny = sm.dcny; sm_map = sm.u.map;
    }
    sm_mel = sm.mel; sm_energy = sm.energy;

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
    float *zb = zw + half * KWS_ZF;
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
    if constexpr (LW > 0) {
        for (int i = threadIdx.x; i < prow; i += blockDim.x) sm_map[i] = P.pad_map[i];
        __syncthreads();
    }
    const int fp0 = LW ? wave * CHP : 0;                    // first frame pair of this wave

    const int n_sel = sel_count(sel, n_clips);
    for (int ci = blockIdx.x; ci < n_sel; ci += gridDim.x) {
        const int clip = sel_clip(sel, ci);
        const void *xbase = F32IN ? (const void *)((const float *)pcm_v + (size_t)clip * P.n_samples)
                                  : (const void *)((const int16_t *)pcm_v + (size_t)clip * P.n_samples);
        // software prefetch, two frame pairs deep: the samples of pair p+2 are requested before pair p is transformed
        RawSamples<F32IN> nxt = fetch_samples<F32IN>(xbase, min(2 * fp0 + half, nfr - 1) * P.frame_stride + 8 * t, P.n_samples);
        RawSamples<F32IN> nxt2 = fetch_samples<F32IN>(xbase, min(2 * fp0 + 2 + half, nfr - 1) * P.frame_stride + 8 * t, P.n_samples);
        const bool has_wrap = wrap != nullptr;
        const float wrapv = has_wrap ? wrap[clip] : 0.0f;

        for (int pair0 = fp0; pair0 < n_pairs; pair0 += (LW ? n_pairs : CHP)) {      // latency mode: this wave's one chunk
            const int pair1 = min(pair0 + CHP, n_pairs);
            cf pend[4] = {};                                  // split outputs of the previous pair: lo0, hi0, lo1, hi1
            bool pend_any = false, pend_live = false;
            float *pend_pcol = sm_p;
            // The power spectrum of the PREVIOUS pair (4 bins per lane, ~35 fp64-heavy instructions each, operands in
            // registers) is evaluated one bin at a time right after each stage's LDS reads are issued: that arithmetic
            // needs no memory, so it runs in the shadow of the round trip the butterflies would otherwise wait out.
            auto power_of_pending = [&](int which) {
                if (pend_any) {                                                    // uniform: no pair pending at chunk start
                    const int k = t + 1 + 32 * (which >> 1);
                    const cf v = pend[which];
                    const float pw = bin_power(v, P.inv_fft);
                    if (pend_live) {
                        if (which & 1) pend_pcol[(KWS_NC - k) * CHF] = pw;          // hi: the "ncfft-k" store
                        else if (k != KWS_NC / 2) pend_pcol[k * CHF] = pw;          // lo; k == 64 is overwritten by hi
                    }
                }
            };
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
                {
                    cf la[4], lb[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) { la[i] = ld_cf(zb, n0 + 16 * i); lb[i] = ld_cf(zb, n0 + 16 * i + 64); }
                    power_of_pending(0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) u[i] = k01 ? csub(la[i], lb[i]) : cadd(la[i], lb[i]);       // b * tw[0], tw[0] = (1, -0)
                }
                bfly4(u[0], u[1], u[2], u[3], a1, a2, a3);
#pragma unroll
                for (int i = 0; i < 4; ++i) st_cf(zb, 8 * g01 + k01 + 2 * i, u[i]);
                WAVE_SYNC();
                // ---- kf_bfly4 m=8, fstride=4 ----------------------------------------------------------------
#pragma unroll
                for (int i = 0; i < 4; ++i) u[i] = ld_cf(zb, 32 * G2 + K2 + 8 * i);
                power_of_pending(1);
                bfly4(u[0], u[1], u[2], u[3], b1, b2, b3);
#pragma unroll
                for (int i = 0; i < 4; ++i) st_cf(zb, 32 * G2 + K2 + 8 * i, u[i]);
                WAVE_SYNC();
                // ---- kf_bfly4 m=32, fstride=1 ---------------------------------------------------------------
#pragma unroll
                for (int i = 0; i < 4; ++i) u[i] = ld_cf(zb, t + 32 * i);
                power_of_pending(2);
                bfly4(u[0], u[1], u[2], u[3], c1, c2, c3);
#pragma unroll
                for (int i = 0; i < 4; ++i) st_cf(zb, t + 32 * i, u[i]);
                WAVE_SYNC();

                PH(1);
                // ---- kiss_fftr split (kiss_fftr.cpp:84-119); its power spectrum is left pending for the next pair -----
                const int fr = f - 2 * pair0;                 // frame slot in the chunk
                const bool live = f < nfr;
                cf fpk[2], fq[2];
#pragma unroll
                for (int rep = 0; rep < 2; ++rep) {
                    const int k = t + 1 + 32 * rep;            // 1..64
                    fpk[rep] = ld_cf(zb, k);
                    fq[rep] = ld_cf(zb, KWS_NC - k);
                }
                power_of_pending(3);
#pragma unroll
                for (int rep = 0; rep < 2; ++rep) {
                    const cf stw = rep ? st2 : st1;
                    cf fpnk; fpnk.r = fq[rep].r; fpnk.i = -fq[rep].i;
                    cf f1k = cadd(fpk[rep], fpnk), f2k = csub(fpk[rep], fpnk);
                    cf twv = cmul(f2k, stw);
                    cf lo, hi;
                    lo.r = (f1k.r + twv.r) * 0.5f;             // HALF_OF
                    lo.i = (f1k.i + twv.i) * 0.5f;
                    hi.r = (f1k.r - twv.r) * 0.5f;
                    hi.i = (twv.i - f1k.i) * 0.5f;
                    pend[2 * rep] = lo;
                    pend[2 * rep + 1] = hi;
                }
                pend_any = true;
                pend_live = live;
                pend_pcol = sm_p + fr;
                // DC / Nyquist bins (kiss_fftr.cpp:84-96) need tmp[0] only: parked per frame, evaluated once per chunk
                // with one frame per lane instead of one lane per wave here
                if (t == 0 && live) *(float2 *)(sm_dcny + 2 * fr) = *(const float2 *)zb;
                WAVE_SYNC();
                PH(2);
            }
            // the chunk's last pair
#pragma unroll
            for (int w = 0; w < 4; ++w) power_of_pending(w);
            pend_any = false;
            WAVE_SYNC();

            // ---- per chunk: frame energy (sequential fp32 sum, numpy.hpp:88-94) -------------------------------
            const int f_base = 2 * pair0;
            const int nfc = min(2 * pair1, nfr) - f_base;
            if (lane < nfc) {
                {
                    const float2 d = *(const float2 *)(sm_dcny + 2 * lane);
                    cf dc, ny;
                    dc.r = d.x + d.y; dc.i = 0.0f;
                    ny.r = d.x - d.y; ny.i = 0.0f;
                    sm_p[lane] = bin_power(dc, P.inv_fft);
                    sm_p[KWS_NC * CHF + lane] = bin_power(ny, P.inv_fft);
                }
                // 129 ordered adds; the operands arrive 16 at a time, one batch ahead of the adds (the chain would otherwise
                // wait for an LDS round trip per batch with only two waves per SIMD to cover it)
                float e = 0.0f;
                const float *pl = sm_p + lane;
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
                sm_energy[f_base + lane] = e;
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
                        float prod = sm_p[fbin[n] + fr] * fwt[n];   // finite, so they add an exact +0
                        acc += prod;
                    }
                    if (acc == 0.0f) acc = FLT_EPSILON;                                // functions.hpp:63-69
                    if constexpr (!WITH_CMVN)
                        if (P.mfe_mel) P.mfe_mel[(size_t)clip * out_stride + (size_t)ring_out_row(P, f_base + fr) * NF + t] = acc;
                    sm_mel[(f_base + fr) * MELS + t] = fast_log(acc);
                }
            } else {
                // one frame per pass, lane = filter (taps in registers as above; walking the CSR table from memory
                // instead cost a third of the kernel)
                if (lane < NF) {
                    // three frames per trip: their 3 x NZ operands are requested together (one frame at a time the loop is
                    // a chain of LDS round trips with only two waves per SIMD to cover them)
                    for (int fr0 = 0; fr0 < nfc; fr0 += 3) {
                        float acc[3] = { 0.0f, 0.0f, 0.0f };
                        float xv[3][NZ];
#pragma unroll
                        for (int u = 0; u < 3; ++u) {
                            // no clamp on the frame slot (one address per tap, the frames at immediate offsets): slots
                            // fr0 + u <= 3 * ((nfc - 1) / 3) + 2 < CHF hold a finite power of an earlier chunk at worst,
                            // and their sums are dropped below
                            static_assert(3 * ((2 * CHP - 1) / 3) + 2 < CHF, "frame slots of a trip stay inside a bin's row");
#pragma unroll
                            for (int n = 0; n < NZ; ++n) xv[u][n] = sm_p[fbin[n] + fr0 + u];
                        }
#pragma unroll
                        for (int u = 0; u < 3; ++u) {
#pragma unroll
                            for (int n = 0; n < NZ; ++n) {
                                float prod = xv[u][n] * fwt[n];
                                acc[u] += prod;
                            }
                        }
#pragma unroll
                        for (int u = 0; u < 3; ++u) {
                            const int fr = fr0 + u;
                            if (fr < nfc) {
                                float a = acc[u];
                                if (a == 0.0f) a = FLT_EPSILON;
                                if constexpr (!WITH_CMVN)
                                    if (P.mfe_mel) P.mfe_mel[(size_t)clip * out_stride + (size_t)ring_out_row(P, f_base + fr) * NF + lane] = a;
                                sm_mel[(f_base + fr) * MELS + lane] = fast_log(a);
                            }
                        }
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
        if constexpr (LW == 0) {
#pragma unroll
            for (int i = 0; i < KWS_MAXPROW / KWS_WAVE; ++i) sm_map[lane + i * KWS_WAVE] = mapreg[i];
        }
        const int drow = LW ? 2 * fp0 + lane : lane;               // latency mode: the frames this wave produced
        if (LW ? (lane < 2 * CHP && drow < nfr) : (lane < nfr)) {
            float v[NF];
            float *mrow = sm_mel + drow * MELS;
#pragma unroll
            for (int i = 0; i < NF; ++i) v[i] = mrow[i];
            float *orow = WITH_CMVN ? mrow : features + (size_t)clip * out_stride + ring_out_row(P, drow) * ncep;
            typedef KwsDctTab<NF> T;
            auto put = [&](int i, cf R) {
                // in place (WITH_CMVN) every output is stored: columns >= ncep of the row are never read again and the
                // row is MELS > NCEPT wide; the packed HBM rows of the stage API only take the first ncep
                if (WITH_CMVN || i < ncep) {
                    float a = R.r * T::cs[i];
                    float b = R.i * T::sn[i];
                    float d = (a + b) * 2.0f;
                    d = d * (i == 0 ? T::s0 : T::s1);
                    orow[i] = d;
                }
            };
            // coefficients above N/2 are never written by the transform: they keep the log-mel input (x2, scaled)
            if constexpr (NF == 32) {            // 17 spectrum points fit the register budget: scale + store after the split
                cf R[NCEPT];
                dct_spectrum<NF>(v, [&](int i, cf r) { R[i] = r; });
#pragma unroll
                for (int i = 0; i < NCEPT; ++i) put(i, R[i]);
#pragma unroll
                for (int i = NCEPT; i < NF; ++i)
                    if (i < ncep) orow[i] = (v[i] * 2.0f) * T::s1;
            } else {                              // 40 filters: hand every point on as soon as it exists (no spills);
                // in place: element i >= NCEPT is read, then written, by this lane only
                for (int i = NCEPT; i < ncep; ++i) orow[i] = (mrow[i] * 2.0f) * T::s1;
                dct_spectrum<NF>(v, put);
            }
            orow[0] = fast_log(sm_energy[drow]);                                       // feature.hpp:425-429
        }
        WAVE_SYNC();
        if constexpr (LW > 0) __syncthreads();                     // every wave's rows are in place
        PH(5);
        if constexpr (!WITH_CMVN) continue;

        // ---- cmvnw (processing.hpp:326-389) + input quantisation ---------------------------------------------
        {
            float *fout = features ? features + (size_t)clip * (nfr * ncep) : nullptr;
            int8_t *qclip = q_out ? q_out + (size_t)clip * (nfr * ncep) : nullptr;
            int *offt = (int *)zw;                                // the FFT buffers are dead by now
            // WIDE (more than 16 cepstra, chosen at launch): 20 columns x 3 row groups of 17 rows per pass instead of
            // 16 x 4 x 13 -- 40 cepstra take 2 passes instead of 3.  One layout per instantiation keeps the registers.
            auto emit = [&](int row, int c, float o) {
                const int idx = row * ncep + c;
                if (fout) fout[idx] = o;                      // optional output (extract_mfcc_features' matrix)
                if (qclip) qclip[idx] = quantize_feature(o, in_scale, in_zp);
            };
            if constexpr (LW > 0) {
                // short chains instead of few lanes: 5 rows per lane, 16 columns x 4 row groups per wave and task
                constexpr int LCR = 5, LCG = 16, LNG = KWS_WAVE / LCG;
                const int nrg = (nfr + LCR * LNG - 1) / (LCR * LNG), ncb = (ncep + LCG - 1) / LCG;
                for (int task = wave; task < nrg * ncb; task += LW) {
                    const int rg = task % nrg, cbk = task / nrg;
                    cmvn_columns<LCR, LCG>(sm_mel, MELS, sm_map, offt, lane, nfr, ncep, prow, P.win_size, emit, rg * LNG, cbk * LCG, cbk * LCG + LCG);
                    WAVE_SYNC();
                }
            } else if constexpr (WIDE) cmvn_columns<17, 20>(sm_mel, MELS, sm_map, offt, lane, nfr, ncep, prow, P.win_size, emit);
            else cmvn_columns<13, 16>(sm_mel, MELS, sm_map, offt, lane, nfr, ncep, prow, P.win_size, emit);
        }
        WAVE_SYNC();
        if constexpr (LW > 0) __syncthreads();                     // the shared matrix is rewritten by the next clip
        PH(7);
    }
    if (PROF && blockIdx.x == 0 && threadIdx.x == 0 && prof_out)
        for (int i = 0; i < KWS_NPHASE; ++i) prof_out[i] = ph[i];
}

// ---------------------------------------------------------------------------------------------------------
//  Kernel 1b: the same function -- same arithmetic, operation by operation -- on the spectral layout kws_fast_kernel introduced
//  (round 3): eight lanes own a frame and sixteen of its 128 complex points each, eight frames per pass, so that kf_bfly2 (m = 1),
//  kf_bfly4 (m = 2) and, after ONE exchange through LDS, kf_bfly4 m = 8 and m = 32 all run in registers (kiss_fft.cpp:15-84,
//  232-296); kws_mfcc_kernel's layout (32 lanes per frame, four points per lane) takes three exchanges per frame pair.  The lanes
//  of a frame swap the upper halves of their points for kiss_fftr's split (kiss_fftr.cpp:84-119); the power spectrum goes through
//  fp64 as in the reference (bin_power); a pass's eight power rows feed the frame energies (129 ordered additions, one lane per
//  frame) and the mel stage (lane = filter, its taps read from LDS).  DCT and cmvnw are kws_mfcc_kernel's.  int16 PCM, windows of 16
//  frames or more, one wave per window; float samples, short windows and the latency shape stay with kws_mfcc_kernel.
//  13 KB of LDS per wave (the exchange buffer, the power rows, the tail pass's buffers and cmvnw's tables share one region) + the mel taps.
// ---------------------------------------------------------------------------------------------------------
constexpr int KWS_M8_CHUNK = 8;      // frames per pass
constexpr int KWS_M8_XS = 144;       // floats per frame of the exchange buffer: 64 positions + 2 floats of padding per 8
constexpr int KWS_M8_PS = 136;       // floats per power row (129 bins): = 8 mod 64, the eight frames' stores of one bin cover the banks
constexpr int KWS_M8_MAP = 640;      // cmvnw's pad map sits behind its offset table (kws_mfcc_max_win: at most 640 offsets)
template <int NF, int NZ>
struct alignas(16) Mfcc8Smem {
    static constexpr int MELS = NF + 1;
    float r1[KWS_M8_CHUNK * KWS_M8_XS];
    float mel[kws_mel_rows(NF) * MELS];
    float energy[kws_mel_rows(NF)];
    // the mel filters' ascending-bin taps [tap][filter]: weight and bin (as an offset into a power row); taps beyond a filter's end have
    // weight 0 and read bin 0.  In registers they would stay live through the FFT (2 x 2 NZ values per lane) and be spilled.
    float tap_w[NZ * NF];
    int tap_b[NZ * NF];
};
static_assert(KWS_M8_CHUNK * KWS_M8_PS <= KWS_M8_CHUNK * KWS_M8_XS && 2 * KWS_M8_PS + 2 * KWS_ZF <= KWS_M8_CHUNK * KWS_M8_XS &&
              KWS_M8_MAP + KWS_MAXPROW <= KWS_M8_CHUNK * KWS_M8_XS, "everything that shares Mfcc8Smem::r1 fits");
static_assert(sizeof(Mfcc8Smem<32, KWS_MAXNZ>) <= 20 * 1024 && sizeof(Mfcc8Smem<40, KWS_MAXNZ>) <= 20 * 1024, "8 waves per CU need <= 20 KB LDS each");

template <bool WITH_CMVN, int NZ, int NF, bool PROF, bool WIDE, int OCC>
__global__ __launch_bounds__(KWS_WAVE, OCC) void kws_mfcc8_kernel(KwsDspPlan P, const int16_t *__restrict__ pcm, int n_clips,
                                                                  float *__restrict__ features, int8_t *__restrict__ q_out,
                                                                  float in_scale, int in_zp, const float *__restrict__ wrap, int out_stride,
                                                                  long long *prof_out = nullptr, const int *__restrict__ sel = nullptr)
{
    constexpr int MELS = NF + 1, NCEPT = NF / 2 + 1, PS = KWS_M8_PS, XS = KWS_M8_XS;
    __shared__ Mfcc8Smem<NF, NZ> sm;
    const int lane = threadIdx.x;
    float *const xw = sm.r1, *const pw = sm.r1;                        // the FFT's exchange buffer; the power rows reuse it
    int *const offt = (int *)sm.r1, *const sm_map = (int *)sm.r1 + KWS_M8_MAP;   // cmvnw's tables: the spectral buffers are dead by then
    const int nfr = P.n_frames, ncep = P.n_cepstral, prow = nfr + 2 * P.pad;
    // a remainder of one or two frames (the 49th of the standard window) would cost a whole eight-frame pass: it gets a tail pass
    // with 32 lanes per frame (kws_mfcc_kernel's layout)
    const int n_tail = (nfr >= KWS_M8_CHUNK && (nfr & 7) != 0 && (nfr & 7) <= 2) ? (nfr & 7) : 0;
    const int n_pass = n_tail ? nfr / KWS_M8_CHUNK : (nfr + KWS_M8_CHUNK - 1) / KWS_M8_CHUNK;
    const float pre_cof = P.pre_cof, inv_fft = P.inv_fft;
    const int frame_stride = P.frame_stride, n_samples = P.n_samples;
    long long ph[KWS_NPHASE] = { 0 }, tlast = PROF ? clock64() : 0;
    if (lane < NF) {
        const int s0 = P.filt_start[lane], e0 = P.filt_start[lane + 1];
#pragma unroll
        for (int n = 0; n < NZ; ++n) {
            const bool on = s0 + n < e0;
            sm.tap_b[n * NF + lane] = on ? P.filt_bin[s0 + n] : 0;
            sm.tap_w[n * NF + lane] = on ? P.filt_w[s0 + n] : 0.0f;
        }
    }
    WAVE_SYNC();

    const int n_sel = sel_count(sel, n_clips);
    int touched_next = 0;
    for (int ci = blockIdx.x; ci < n_sel; ci += gridDim.x) {
        const int clip = sel_clip(sel, ci);
        // per-lane constants of the spectral phase, re-derived per clip from a lane index the compiler cannot see through: hoisted out
        // of the clip loop they would stay live through cmvnw (three dozen registers) for ~100 L2-resident loads per clip
        int lane_c = lane;
        asm volatile("" : "+v"(lane_c));
        const int fl = lane_c & 7, fg = lane_c >> 3;
        // blocks fl (output positions 8 fl ..) and fl + 8: block j = 4 i1 + i2 reads input points i1 + 4 i2 + 16 i3 + 64 i4
        const int nbA = (fl >> 2) + 4 * (fl & 3);
        const cf a1 = to_cf(P.tw[16]), a2 = to_cf(P.tw[32]), a3 = to_cf(P.tw[48]);
        const cf b1 = to_cf(P.tw[4 * fl]), b2 = to_cf(P.tw[8 * fl]), b3 = to_cf(P.tw[12 * fl]);
        cf c1[4], c2[4], c3[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) { c1[a] = to_cf(P.tw[fl + 8 * a]); c2[a] = to_cf(P.tw[2 * (fl + 8 * a)]); c3[a] = to_cf(P.tw[3 * (fl + 8 * a)]); }
        // split twiddles of this lane's eight bin pairs (k, 128 - k), k = fl + 8 a + 32 b for b < 2; lane 0's first pair is (64, 64)
        cf stw[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = fl + 8 * (q & 3) + 32 * (q >> 2);
            stw[q] = to_cf(P.stw[(k == 0 ? KWS_NC / 2 : k) - 1]);
        }
        const int xwr = fg * XS + 18 * fl;                              // exchange buffer: position p of a frame at 2 p + 2 (p / 8)
        const int xrd = fg * XS + 2 * fl;
        const int partner = (lane_c & ~7) | ((8 - fl) & 7);
        const int half = lane_c >> 5, t = lane_c & 31;
        const int16_t *xbase = pcm + (size_t)clip * n_samples;
        // x[-1] of the window's first sample: its last sample (processing.hpp:68, 104-106) unless the caller says otherwise (continuous mode)
        // (a chunk of a longer window that starts inside it -- KwsDspPlan::wrap_index < 0 --: the sample before the chunk)
        const float wrap_prev = P.wrap_index < 0 ? (float)xbase[P.wrap_index] * (1.0f / 32768.0f) : wrap ? wrap[clip] : (float)xbase[n_samples - 1] * (1.0f / 32768.0f);
        // a point's four samples x[2n - 2 .. 2n + 1] of frame f, requested one pass ahead
        auto fetch = [&](int q, fast_i2 (&raw)[2][8]) {
            const int f = min(KWS_M8_CHUNK * q + fg, nfr - 1);
            const int16_t *xf = xbase + (f * frame_stride + 2 * nbA - 2);
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int16_t *src = xf + (4 * blk + 32 * (i >> 1) + 128 * (i & 1));
                    if (blk == 0 && i == 0) src = src < xbase ? xbase : src;
                    raw[blk][i] = *(const fast_i2 *)src;
                }
        };
        // the frames of the pass after that: one 64-byte segment per lane, a pass before the real requests (see kws_fast_kernel)
        auto touch = [&](int q) {
            const int f = min(KWS_M8_CHUNK * q + fg, nfr - 1);
            return *(const int *)(xbase + (f * frame_stride + 32 * fl));
        };
        // ---- frame energy (feature.hpp:289-298): sequential fp32 sum over the 129 bins of a power row (numpy.hpp:88-94) ----------
        auto energy_of = [&](const float *pl, int f) {
            float e = 0.0f;
            float4 cur[8], nxt4[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) cur[u] = *(const float4 *)(pl + 4 * u);
#pragma unroll
            for (int k0 = 0; k0 < KWS_NBINS - 1; k0 += 32) {
                if (k0 + 32 < KWS_NBINS - 1) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) nxt4[u] = *(const float4 *)(pl + k0 + 32 + 4 * u);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) { e += cur[u].x; e += cur[u].y; e += cur[u].z; e += cur[u].w; }
#pragma unroll
                for (int u = 0; u < 8; ++u) cur[u] = nxt4[u];
            }
            e += pl[KWS_NBINS - 1];
            if (e == 0.0f) e = FLT_EPSILON;                                           // feature.hpp:296-298
            sm.energy[f] = e;
            if constexpr (!WITH_CMVN)
                if (P.mfe_energy) P.mfe_energy[(size_t)clip * nfr + f] = e;
        };
        // ---- mel filterbank for the frames of a pass: dot_by_row (numpy.hpp:183-211) as an ascending-bin gather, zero handling, log.
        //      pairs = pairs of frame slots a lane half walks (2: slots 4 h .. 4 h + 3 of an eight-frame pass; 1: the tail pass, slots 0, 1)
        auto mel_phase = [&](int fbase, int nfc, int pairs) {
            const float *p1 = pw + 4 * half * PS, *p2 = pw + fg * PS;
            float macc[5] = { 1.0f, 1.0f, 1.0f, 1.0f, 1.0f };
            {
                // filter lane & 31 for four of the pass's frame slots
                float w1[NZ];
                int fb1[NZ];
#pragma unroll
                for (int n = 0; n < NZ; ++n) { w1[n] = sm.tap_w[n * NF + t]; fb1[n] = sm.tap_b[n * NF + t]; }
#pragma unroll
                for (int s2 = 0; s2 < 4; s2 += 2) {
                    if (s2 >= 2 * pairs) break;
                    float xv[2][NZ];
#pragma unroll
                    for (int s = 0; s < 2; ++s)
#pragma unroll
                        for (int n = 0; n < NZ; ++n) xv[s][n] = p1[(s2 + s) * PS + fb1[n]];
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        float acc = 0.0f;
#pragma unroll
                        for (int n = 0; n < NZ; ++n) {
                            const float prod = xv[s][n] * w1[n];
                            acc += prod;
                        }
                        macc[s2 + s] = acc;
                    }
                }
            }
            if constexpr (NF > 32) {
                // with 40 filters also filter 32 + lane & 7 for one slot
                float w2[NZ], xv2[NZ];
#pragma unroll
                for (int n = 0; n < NZ; ++n) { w2[n] = sm.tap_w[n * NF + 32 + fl]; xv2[n] = p2[sm.tap_b[n * NF + 32 + fl]]; }
                float acc = 0.0f;
#pragma unroll
                for (int n = 0; n < NZ; ++n) {
                    const float prod = xv2[n] * w2[n];
                    acc += prod;
                }
                macc[4] = acc;
            }
            auto put = [&](int slot, int j, float a) {
                if (a == 0.0f) a = FLT_EPSILON;                                       // functions.hpp:63-69
                if constexpr (!WITH_CMVN)
                    if (P.mfe_mel) P.mfe_mel[(size_t)clip * out_stride + (size_t)ring_out_row(P, fbase + slot) * NF + j] = a;
                sm.mel[(fbase + slot) * MELS + j] = fast_log(a);
            };
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int slot = 4 * half + s;
                if (s < 2 * pairs && slot < nfc && t < NF) put(slot, t, macc[s]);
            }
            if constexpr (NF > 32)
                if (fg < nfc && 32 + fl < NF) put(fg, 32 + fl, macc[4]);
        };

        fast_i2 nxt[2][8];
        fetch(0, nxt);
        asm volatile("" : : "v"(touched_next));
        int touched = touch(1);
        for (int q = 0; q < n_pass; ++q) {
            const int fbase = KWS_M8_CHUNK * q;
            const int f = fbase + fg;
            const bool live = f < nfr;
            cf u[4][4];                                                  // after the exchange: u[a][b] = position fl + 8 a + 32 b
            {
                cf z[2][8];
#pragma unroll
                for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                    for (int i = 0; i < 8; ++i) z[blk][i] = exact_point(nxt[blk][i], pre_cof);
                if (f == 0 && fl == 0) {                                 // the clip's first sample: its predecessor wraps
                    const fast_i2 v = nxt[0][0];                         // (the request was clamped to the window's start: v.x = x[0], x[1])
                    const float lo = (float)(short)(v.x & 0xffff) * (1.0f / 32768.0f), hi = (float)(v.x >> 16) * (1.0f / 32768.0f);
                    const float pl = pre_cof * wrap_prev;
                    z[0][0].r = lo - pl;
                    const float ph_ = pre_cof * lo;
                    z[0][0].i = hi - ph_;
                }
                // unconditional (the frame index is clamped): a conditional request makes the compiler copy all sixteen register
                // pairs around the branch
                fetch(q + 1, nxt);
                asm volatile("" : : "v"(touched));
                touched = touch(q + 2);
                PH(0);
                // kf_bfly2 (m = 1, twiddle 1) on the (i4 = 0, 1) pairs, then kf_bfly4 (m = 2) on the sums (k = 0) and the
                // differences (k = 1): outputs 8 j + k + 2 i
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    cf sv[4], df[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) { sv[i] = cadd(z[blk][2 * i], z[blk][2 * i + 1]); df[i] = csub(z[blk][2 * i], z[blk][2 * i + 1]); }
                    bfly4_unit(sv[0], sv[1], sv[2], sv[3]);
                    bfly4(df[0], df[1], df[2], df[3], a1, a2, a3);
#pragma unroll
                    for (int i = 0; i < 4; ++i) { z[blk][2 * i] = sv[i]; z[blk][2 * i + 1] = df[i]; }
                }
                // the exchange, half a frame at a time (64 positions per frame fit the buffer): block fl feeds b = 0, 1
#pragma unroll
                for (int rnd = 0; rnd < 2; ++rnd) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) *(float2 *)(xw + xwr + 2 * r) = make_float2(z[rnd][r].r, z[rnd][r].i);
                    WAVE_SYNC();
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int a = 0; a < 4; ++a) {
                            const float2 v = *(const float2 *)(xw + xrd + 18 * a + 72 * b);
                            u[a][2 * rnd + b].r = v.x; u[a][2 * rnd + b].i = v.y;
                        }
                    WAVE_SYNC();
                }
            }
            // kf_bfly4 m = 8 (k = fl) inside every block of 32, then m = 32 (k = fl + 8 a) across them
#pragma unroll
            for (int b = 0; b < 4; ++b) bfly4(u[0][b], u[1][b], u[2][b], u[3][b], b1, b2, b3);
#pragma unroll
            for (int a = 0; a < 4; ++a) bfly4(u[a][0], u[a][1], u[a][2], u[a][3], c1[a], c2[a], c3[a]);
            PH(1);
            // ---- kiss_fftr split (kiss_fftr.cpp:84-119) and the power spectrum (bin_power).  Bin pair (k, 128 - k) needs positions
            //      k and 128 - k: the second lives in lane (8 - fl) % 8 at (3 - a, 3 - b) -- in lane 0 itself, one position further --
            //      so the lanes swap their upper halves.
            {
                float *prw = pw + fg * PS;                           // (a row of a dead frame slot is written too, never used)
                const bool lane0 = fl == 0;
#pragma unroll
                for (int qq = 0; qq < 8; ++qq) {
                    const int a = qq & 3, b = qq >> 2;
                    cf other;
                    other.r = __int_as_float(__builtin_amdgcn_ds_bpermute(partner << 2, __float_as_int(u[3 - a][3 - b].r)));
                    other.i = __int_as_float(__builtin_amdgcn_ds_bpermute(partner << 2, __float_as_int(u[3 - a][3 - b].i)));
                    cf fpk = u[a][b];
                    int k = fl + 8 * a + 32 * b;
                    {
                        // lane 0: 128 - k = 8 (16 - a - 4 b) is position index 16 - qq of the lane itself; its pair 0 is (64, 64)
                        const int o = qq == 0 ? 8 : 16 - qq;
                        other.r = lane0 ? u[o & 3][o >> 2].r : other.r;
                        other.i = lane0 ? u[o & 3][o >> 2].i : other.i;
                        if (qq == 0) { fpk.r = lane0 ? u[0][2].r : fpk.r; fpk.i = lane0 ? u[0][2].i : fpk.i; k = lane0 ? KWS_NC / 2 : k; }
                    }
                    cf fpnk; fpnk.r = other.r; fpnk.i = -other.i;
                    const cf f1k = cadd(fpk, fpnk), f2k = csub(fpk, fpnk);
                    const cf twv = cmul(f2k, stw[qq]);
                    cf lo, hi;
                    lo.r = (f1k.r + twv.r) * 0.5f;                   // HALF_OF
                    lo.i = (f1k.i + twv.i) * 0.5f;
                    hi.r = (f1k.r - twv.r) * 0.5f;
                    hi.i = (twv.i - f1k.i) * 0.5f;
                    // bin 64 is written twice by the reference and the second store (the "ncfft - k" one) wins: same order here
                    prw[k] = bin_power(lo, inv_fft);
                    prw[KWS_NC - k] = bin_power(hi, inv_fft);
                }
                if (lane0) {                                         // tmp[0]: DC and Nyquist bins (kiss_fftr.cpp:84-96)
                    cf dc, ny;
                    dc.r = u[0][0].r + u[0][0].i; dc.i = 0.0f;
                    ny.r = u[0][0].r - u[0][0].i; ny.i = 0.0f;
                    prw[0] = bin_power(dc, inv_fft);
                    prw[KWS_NC] = bin_power(ny, inv_fft);
                }
            }
            WAVE_SYNC();
            PH(2);
            if (fl == 0 && live) energy_of(pw + fg * PS, f);
            PH(3);
            mel_phase(fbase, min(KWS_M8_CHUNK, nfr - fbase), 2);
            WAVE_SYNC();                                             // the next pass's exchange overwrites the power rows
            PH(4);
        }

        if (n_tail) {
            // ---- tail pass: frame 8 n_pass + h on lane half h, a lane transforms four of its frame's 128 points per stage (kws_mfcc_kernel's
            //      butterflies): kf_bfly2 (m = 1) fused with kf_bfly4 (m = 2), then kf_bfly4 m = 8 and m = 32, each through an in-place,
            //      padded buffer behind the two power rows it feeds
            const int ft = KWS_M8_CHUNK * n_pass + half;
            const bool live_t = ft < nfr;
            const int s0 = min(ft, nfr - 1) * frame_stride + 8 * t;
            const int4 rawv = *(const int4 *)(xbase + s0);
            // (s0 = 0 needs a window of one frame: no tail pass then)
            const float rawp = s0 == 0 ? wrap_prev : (float)xbase[s0 - 1] * (1.0f / 32768.0f);
            const int k01 = t & 1, g01 = t >> 1, n0 = (g01 >> 2) + 4 * (g01 & 3), K2 = t & 7, G2 = t >> 3;
            const cf ta1 = to_cf(P.tw[16 * k01]), ta2 = to_cf(P.tw[32 * k01]), ta3 = to_cf(P.tw[48 * k01]);
            const cf tb1 = to_cf(P.tw[4 * K2]), tb2 = to_cf(P.tw[8 * K2]), tb3 = to_cf(P.tw[12 * K2]);
            const cf tc1 = to_cf(P.tw[t]), tc2 = to_cf(P.tw[2 * t]), tc3 = to_cf(P.tw[3 * t]);
            float *zb = sm.r1 + 2 * PS + half * KWS_ZF;
            {
                float y[8];
                float prev = rawp;
                const int w[4] = { rawv.x, rawv.y, rawv.z, rawv.w };
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float lo = (float)(short)(w[i] & 0xffff) * (1.0f / 32768.0f);   // numpy::int16_to_float
                    const float hi = (float)(short)(w[i] >> 16) * (1.0f / 32768.0f);
                    const float pl = pre_cof * prev;
                    y[2 * i] = lo - pl;
                    const float ph_ = pre_cof * lo;
                    y[2 * i + 1] = hi - ph_;
                    prev = hi;
                }
                *(float4 *)(zb + 2 * zi(4 * t)) = make_float4(y[0], y[1], y[2], y[3]);
                *(float4 *)(zb + 2 * zi(4 * t) + 4) = make_float4(y[4], y[5], y[6], y[7]);
            }
            WAVE_SYNC();
            PH(0);
            cf v[4];
            {
                cf la[4], lb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { la[i] = ld_cf(zb, n0 + 16 * i); lb[i] = ld_cf(zb, n0 + 16 * i + 64); }
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = k01 ? csub(la[i], lb[i]) : cadd(la[i], lb[i]);
            }
            bfly4(v[0], v[1], v[2], v[3], ta1, ta2, ta3);
            WAVE_SYNC();                                              // every lane has read its inputs
#pragma unroll
            for (int i = 0; i < 4; ++i) st_cf(zb, 8 * g01 + k01 + 2 * i, v[i]);
            WAVE_SYNC();
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = ld_cf(zb, 32 * G2 + K2 + 8 * i);
            bfly4(v[0], v[1], v[2], v[3], tb1, tb2, tb3);
#pragma unroll
            for (int i = 0; i < 4; ++i) st_cf(zb, 32 * G2 + K2 + 8 * i, v[i]);
            WAVE_SYNC();
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = ld_cf(zb, t + 32 * i);
            bfly4(v[0], v[1], v[2], v[3], tc1, tc2, tc3);
#pragma unroll
            for (int i = 0; i < 4; ++i) st_cf(zb, t + 32 * i, v[i]);
            WAVE_SYNC();
            PH(1);
            {
                const cf st1 = to_cf(P.stw[t]), st2 = to_cf(P.stw[t + 32]);
                cf fpk[2], fq[2];
#pragma unroll
                for (int rep = 0; rep < 2; ++rep) {
                    const int k = t + 1 + 32 * rep;
                    fpk[rep] = ld_cf(zb, k);
                    fq[rep] = ld_cf(zb, KWS_NC - k);
                }
                const float2 d0 = *(const float2 *)zb;                // tmp[0]: DC and Nyquist bins (kiss_fftr.cpp:84-96)
                float *prw = pw + half * PS;
#pragma unroll
                for (int rep = 0; rep < 2; ++rep) {
                    const int k = t + 1 + 32 * rep;
                    const cf stw_ = rep ? st2 : st1;
                    cf fpnk; fpnk.r = fq[rep].r; fpnk.i = -fq[rep].i;
                    const cf f1k = cadd(fpk[rep], fpnk), f2k = csub(fpk[rep], fpnk);
                    const cf twv = cmul(f2k, stw_);
                    cf lo, hi;
                    lo.r = (f1k.r + twv.r) * 0.5f;
                    lo.i = (f1k.i + twv.i) * 0.5f;
                    hi.r = (f1k.r - twv.r) * 0.5f;
                    hi.i = (twv.i - f1k.i) * 0.5f;
                    const float plo = bin_power(lo, inv_fft), phi = bin_power(hi, inv_fft);
                    if (k != KWS_NC / 2) prw[k] = plo;                // bin 64 is written twice by the reference: the second store wins
                    prw[KWS_NC - k] = phi;
                }
                if (t == 0) {
                    cf dc, ny;
                    dc.r = d0.x + d0.y; dc.i = 0.0f;
                    ny.r = d0.x - d0.y; ny.i = 0.0f;
                    prw[0] = bin_power(dc, inv_fft);
                    prw[KWS_NC] = bin_power(ny, inv_fft);
                }
            }
            WAVE_SYNC();
            PH(2);
            if (t == 0 && live_t) energy_of(pw + half * PS, ft);
            PH(3);
            mel_phase(KWS_M8_CHUNK * n_pass, n_tail, 1);
            WAVE_SYNC();
            PH(4);
        }

        if constexpr (!WITH_CMVN)
            if (P.mfe_mel) { WAVE_SYNC(); continue; }                                   // MFE block: no log / DCT output
        // ---- DCT-II via NF-point kiss_fftr, one frame per lane (numpy.hpp:378-401, fast-dct-fft.cpp:37-80): the cepstra of a frame
        //      replace its log-mel row in place (row stride MELS); cmvnw's pad map moves into the dead spectral buffers
        // (lane-derived constants of the DCT and of cmvnw are re-derived per clip as well: hoisted out of the clip loop they are spilled)
        int lane_d = lane;
        asm volatile("" : "+v"(lane_d));
        for (int i = lane_d; i < prow; i += KWS_WAVE) sm_map[i] = P.pad_map[i];
        // the wave's next clip: its first pass's samples are warmed in the cache while cmvnw runs
        if (ci + (int)gridDim.x < n_sel) {
            const int16_t *xn = pcm + (size_t)sel_clip(sel, ci + gridDim.x) * n_samples;
            touched_next = *(const int *)(xn + (min(fg, nfr - 1) * frame_stride + 32 * fl));
        }
        if (lane_d < nfr) {
            float v[NF];
            float *mrow = sm.mel + lane_d * MELS;
#pragma unroll
            for (int i = 0; i < NF; ++i) v[i] = mrow[i];
            float *orow = WITH_CMVN ? mrow : features + (size_t)clip * out_stride + ring_out_row(P, lane_d) * ncep;
            typedef KwsDctTab<NF> T;
            auto put = [&](int i, cf R) {
                if (WITH_CMVN || i < ncep) {
                    float a = R.r * T::cs[i];
                    float b = R.i * T::sn[i];
                    float d = (a + b) * 2.0f;
                    d = d * (i == 0 ? T::s0 : T::s1);
                    orow[i] = d;
                }
            };
            // coefficients above N/2 are never written by the transform: they keep the log-mel input (x2, scaled)
            if constexpr (NF == 32) {
                cf R[NCEPT];
                dct_spectrum<NF>(v, [&](int i, cf r) { R[i] = r; });
#pragma unroll
                for (int i = 0; i < NCEPT; ++i) put(i, R[i]);
#pragma unroll
                for (int i = NCEPT; i < NF; ++i)
                    if (i < ncep) orow[i] = (v[i] * 2.0f) * T::s1;
            } else {
                for (int i = NCEPT; i < ncep; ++i) orow[i] = (mrow[i] * 2.0f) * T::s1;
                dct_spectrum<NF>(v, put);
            }
            orow[0] = fast_log(sm.energy[lane_d]);                                       // feature.hpp:425-429
        }
        WAVE_SYNC();
        PH(5);
        if constexpr (!WITH_CMVN) continue;

        // ---- cmvnw (processing.hpp:326-389) + input quantisation ---------------------------------------------
        {
            float *fout = features ? features + (size_t)clip * (nfr * ncep) : nullptr;
            int8_t *qclip = q_out ? q_out + (size_t)clip * (nfr * ncep) : nullptr;
            auto emit = [&](int row, int c, float o) {
                const int idx = row * ncep + c;
                if (fout) fout[idx] = o;                      // optional output (extract_mfcc_features' matrix)
                if (qclip) qclip[idx] = quantize_feature(o, in_scale, in_zp);
            };
            if constexpr (WIDE) cmvn_columns<17, 20>(sm.mel, MELS, sm_map, offt, lane_d, nfr, ncep, prow, P.win_size, emit);
            else cmvn_columns<13, 16>(sm.mel, MELS, sm_map, offt, lane_d, nfr, ncep, prow, P.win_size, emit);
        }
        WAVE_SYNC();
        PH(7);
    }
    asm volatile("" : : "v"(touched_next));
    if (PROF && blockIdx.x == 0 && threadIdx.x == 0 && prof_out)
        for (int i = 0; i < KWS_NPHASE; ++i) prof_out[i] = ph[i];
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
// latency mode: 7 waves x 8 frames per window (56 >= KWS_MAXF), taken for float-sample calls of at most 16 windows
constexpr int KWS_LAT_CHP = 4, KWS_LAT_WAVES = 7, KWS_LAT_MAX_CLIPS = 16;

// One row per instantiation of the tuned kernel: the first row whose limits cover the model's DSP block is launched.  Shapes
// outside every row run on the general kernels (kws_generic.hip; KwsDspPlan::generic is set by build_dsp_plan from the same limits).
struct MfccLaunchArgs {
    dim3 grid, block;
    hipStream_t stream;
    KwsDspPlan P;
    const void *pcm;
    int n_clips;
    float *out;
    int8_t *q_out;
    float in_scale;
    int in_zp;
    const float *wrap;
    int out_stride;
    long long *prof;
    const int *sel;
};
template <int CHP, bool F32IN, bool WITH_CMVN, int NZ, int NF, bool PROF, bool WIDE, int LW>
static void mfcc_launch(const MfccLaunchArgs &a)
{
    hipLaunchKernelGGL((kws_mfcc_kernel<CHP, F32IN, WITH_CMVN, NZ, NF, PROF, WIDE, LW>), a.grid, a.block, 0, a.stream, a.P, a.pcm, a.n_clips, a.out,
                       a.q_out, a.in_scale, a.in_zp, a.wrap, a.out_stride, a.prof, a.sel);
}
template <bool WITH_CMVN, int NZ, int NF, bool PROF, bool WIDE, int OCC>
static void mfcc8_launch(const MfccLaunchArgs &a)
{
    hipLaunchKernelGGL((kws_mfcc8_kernel<WITH_CMVN, NZ, NF, PROF, WIDE, OCC>), a.grid, a.block, 0, a.stream, a.P, (const int16_t *)a.pcm, a.n_clips, a.out,
                       a.q_out, a.in_scale, a.in_zp, a.wrap, a.out_stride, a.prof, a.sel);
}
struct MfccVariant {
    int n_filters, max_nz;      // mel filters; longest filter the variant keeps in registers
    int min_cepstra;            // WIDE cmvnw layout (20 columns x 3 row groups): only worth it above 16 cepstra, needs WITH_CMVN
    bool latency;               // one workgroup of KWS_LAT_WAVES waves per window (a handful of float-sample windows)
    void (*launch)(const MfccLaunchArgs &);
};

template <bool F32IN, bool WITH_CMVN, bool PROF>
static int launch_mfcc_t(const KwsDspPlan &P, const void *pcm, int n_clips, float *out, int8_t *q_out, float in_scale, int in_zp,
                         const float *wrap, int out_stride, int grid_cap, long long *prof, hipStream_t stream, const int *sel = nullptr)
{
    (void)hipGetLastError();      // the status returned below is this launch's, not a stale error of an earlier call
    if (n_clips <= 0) return 0;
    // development switch (same-box A/B of the two spectral layouts, tools/gpu_mfcc_layout_check.py): KWS_DEV_MFCC_OLD_LAYOUT keeps
    // kws_mfcc_kernel for int16 PCM too
    // Short windows (the 250 ms slices of continuous mode: 12 frames) stay with kws_mfcc_kernel: two passes cannot amortise what
    // kws_mfcc8_kernel re-derives per window (measured: 65 536 streams advance a slice in 0.790 vs 0.814 ms); KWS_DEV_MFCC8_MIN_FRAMES moves
    // the threshold (tests: 1 = every frame count on the new layout)
    const char *const min_env = KWS_DEV_ENV("KWS_DEV_MFCC8_MIN_FRAMES");
    const int min_frames8 = min_env ? atoi(min_env) : 16;
    const bool old_layout = KWS_DEV_ENV("KWS_DEV_MFCC_OLD_LAYOUT") != nullptr || P.n_frames < min_frames8;
    const MfccVariant table[] = {
        // latency shape first (float samples + cmvnw only): run_classifier() and other calls of at most KWS_LAT_MAX_CLIPS windows
        { 40, 8, 0, true, (F32IN && WITH_CMVN && !PROF) ? mfcc_launch<KWS_LAT_CHP, true, true, 8, 40, false, false, KWS_LAT_WAVES> : nullptr },
        { 40, KWS_MAXNZ, 0, true, (F32IN && WITH_CMVN && !PROF) ? mfcc_launch<KWS_LAT_CHP, true, true, KWS_MAXNZ, 40, false, false, KWS_LAT_WAVES> : nullptr },
        { 32, 4, 0, true, (F32IN && WITH_CMVN && !PROF) ? mfcc_launch<KWS_LAT_CHP, true, true, 4, 32, false, false, KWS_LAT_WAVES> : nullptr },
        { 32, KWS_MAXNZ, 0, true, (F32IN && WITH_CMVN && !PROF) ? mfcc_launch<KWS_LAT_CHP, true, true, KWS_MAXNZ, 32, false, false, KWS_LAT_WAVES> : nullptr },
        // throughput shape, int16 PCM: one wave per clip on the eight-lanes-per-frame spectral layout (kws_mfcc8_kernel)
        { 40, 8, 17, false, (!F32IN && WITH_CMVN && !old_layout) ? mfcc8_launch<WITH_CMVN, 8, 40, PROF, WITH_CMVN, 2> : nullptr },
        { 40, 8, 0, false, (!F32IN && !old_layout) ? mfcc8_launch<WITH_CMVN, 8, 40, PROF, false, 2> : nullptr },
        { 40, KWS_MAXNZ, 0, false, (!F32IN && !old_layout) ? mfcc8_launch<WITH_CMVN, KWS_MAXNZ, 40, PROF, false, 2> : nullptr },
        { 32, 4, 0, false, (!F32IN && !old_layout) ? mfcc8_launch<WITH_CMVN, 4, 32, PROF, false, 2> : nullptr },
        { 32, KWS_MAXNZ, 0, false, (!F32IN && !old_layout) ? mfcc8_launch<WITH_CMVN, KWS_MAXNZ, 32, PROF, false, 2> : nullptr },
        // throughput shape, float samples: one wave per clip, persistent grid
        { 40, 8, 17, false, WITH_CMVN ? mfcc_launch<KWS_CHP, F32IN, WITH_CMVN, 8, 40, PROF, WITH_CMVN, 0> : nullptr },
        { 40, 8, 0, false, mfcc_launch<KWS_CHP, F32IN, WITH_CMVN, 8, 40, PROF, false, 0> },
        { 40, KWS_MAXNZ, 0, false, mfcc_launch<KWS_CHP, F32IN, WITH_CMVN, KWS_MAXNZ, 40, PROF, false, 0> },
        { 32, 4, 0, false, mfcc_launch<KWS_CHP, F32IN, WITH_CMVN, 4, 32, PROF, false, 0> },
        { 32, KWS_MAXNZ, 0, false, mfcc_launch<KWS_CHP, F32IN, WITH_CMVN, KWS_MAXNZ, 32, PROF, false, 0> },
    };
    const bool few = n_clips <= KWS_LAT_MAX_CLIPS && P.n_frames <= 2 * KWS_LAT_CHP * KWS_LAT_WAVES;
    for (const MfccVariant &v : table) {
        if (!v.launch || v.n_filters != P.n_filters || P.max_nz > v.max_nz || P.n_cepstral < v.min_cepstra || (v.latency && !few)) continue;
        MfccLaunchArgs a = { v.latency ? dim3(n_clips) : dim3(n_clips < grid_cap ? n_clips : grid_cap),
                             dim3(KWS_WAVE * (v.latency ? KWS_LAT_WAVES : 1)), stream, P, pcm, n_clips, out, q_out, in_scale, in_zp, wrap,
                             out_stride, prof, sel };
        v.launch(a);
        return (int)hipGetLastError();
    }
    return (int)hipErrorInvalidValue;         // no instantiation: build_dsp_plan routes such shapes to the general kernels
}

// extract_mfcc_features (+ quantisation) for n_clips windows in one launch
int kws_launch_mfcc_fused(const KwsDspPlan &P, const void *pcm, int pcm_is_float, int n_clips, float *features, int8_t *q_out,
                          float in_scale, int in_zp, int grid_cap, hipStream_t stream, const int *sel)
{
    (void)hipGetLastError();      // the status returned below is this launch's, not a stale error of an earlier call
    return pcm_is_float ? launch_mfcc_t<true, true, false>(P, pcm, n_clips, features, q_out, in_scale, in_zp, nullptr, 0, grid_cap, nullptr, stream, sel)
                        : launch_mfcc_t<false, true, false>(P, pcm, n_clips, features, q_out, in_scale, in_zp, nullptr, 0, grid_cap, nullptr, stream, sel);
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
                        int out_stride, int grid_cap, hipStream_t stream, const int *sel)
{
    (void)hipGetLastError();      // the status returned below is this launch's, not a stale error of an earlier call
    if (out_stride == 0) out_stride = P.n_frames * P.n_cepstral;
    return pcm_is_float ? launch_mfcc_t<true, false, false>(P, pcm, n_clips, mfcc_out, nullptr, 0.f, 0, wrap, out_stride, grid_cap, nullptr, stream, sel)
                        : launch_mfcc_t<false, false, false>(P, pcm, n_clips, mfcc_out, nullptr, 0.f, 0, wrap, out_stride, grid_cap, nullptr, stream, sel);
}

// speechpy::feature::mfe (feature.hpp:193-318) for n_clips windows: mel energies + frame energies.  out_stride: floats between
// consecutive windows' mel matrices (0 = packed, n_frames * n_filters); wrap as in kws_launch_spectral.
int kws_launch_mfe(const KwsDspPlan &P0, const void *pcm, int pcm_is_float, int n_clips, float *mel_out, float *energy_out, const float *wrap,
                   int out_stride, int grid_cap, hipStream_t stream)
{
    (void)hipGetLastError();      // the status returned below is this launch's, not a stale error of an earlier call
    KwsDspPlan P = P0;
    P.mfe_mel = mel_out;
    P.mfe_energy = energy_out;
    if (out_stride == 0) out_stride = P.n_frames * P.n_filters;
    return pcm_is_float ? launch_mfcc_t<true, false, false>(P, pcm, n_clips, nullptr, nullptr, 0.f, 0, wrap, out_stride, grid_cap, nullptr, stream)
                        : launch_mfcc_t<false, false, false>(P, pcm, n_clips, nullptr, nullptr, 0.f, 0, wrap, out_stride, grid_cap, nullptr, stream);
}
