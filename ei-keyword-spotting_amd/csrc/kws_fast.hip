// kws_fast.hip -- kws_fast_kernel: KWS_MODE_FAST, the tolerance-mode form of run_classifier()'s hot path (see kws_fast.h for
// what is relaxed and what is not).  One wavefront owns one clip from the int16 PCM in HBM to its scores: extract_mfcc_features
// (SDK/classifier/ei_run_dsp.h:256-308) with the FFT in KissFFT's order and everything behind it in plain fp32, then -- for a
// float32 graph -- the CONV_2D / ADD / MAX_POOL_2D / FULLY_CONNECTED / SOFTMAX chain (TFL/kernels/internal/reference/conv.h:28-99,
// add.h:179-215, pooling.h:189-237, fully_connected.h:26-60, softmax.h:31-63) with the convolutions on v_mfma_f32_16x16x4_f32.
// Eight waves of a workgroup share the weights in LDS; nothing but the PCM and the scores crosses HBM.
#include "kws_device.h"
#include "kws_fast.h"

typedef float v4f __attribute__((ext_vector_type(4)));
static_assert(KWS_FAST_ZF == KWS_ZF && KWS_FAST_WAVE == KWS_WAVE, "kws_fast.h mirrors kws_device.h");

template <int CTRL> __device__ __forceinline__ float dpp_mov(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// sum over the 32 lanes of a half-wave; every lane of the half receives it
__device__ __forceinline__ float half_wave_sum(float v)
{
    v += dpp_mov<0xB1>(v);        // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);        // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);       // row_half_mirror: lanes i <-> 7 - i
    v += dpp_mov<0x140>(v);       // row_mirror: lanes i <-> 15 - i
    v += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F));   // lane ^ 16
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, KWS_WAVE));
    return v;
}
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, KWS_WAVE);
    return v;
}

struct FastRaw { int4 v; short prev; };
__device__ __forceinline__ FastRaw fast_fetch(const int16_t *x, int s0, int n_samples)
{
    FastRaw r;
    r.v = *(const int4 *)(x + s0);
    r.prev = x[s0 == 0 ? n_samples - 1 : s0 - 1];     // x[-1] is the window's last sample (processing.hpp:68, 104-106)
    return r;
}

// ---------------------------------------------------------------------------------------------------------
//  One CONV_2D block on the matrix cores: out[m][n] = sum_{tap, c} in[m + tap][c] * w[n][tap][c]   (the image's rows start
//  at time -pad_left and its padding rows hold zeros, so SAME padding needs no predicate).  Tiles of 16 rows x 16 channels,
//  k-steps of 4 (tap, channel) pairs; a lane fetches two k-steps' operands with one 8-byte read each:
//      A: image[(16 mt + l % 16 + tap) * stride + 8 cg + 2 (l / 16) + {0, 1}]
//      B: w[tap][4 cg + l / 16][n][{0, 1}]                          (layout built by kws_fast_plan.cpp)
//  Every accumulator (<= 4 x 2 tiles) stays in registers until the contraction is complete: only then is the input image dead
//  and may be overwritten by the un-pooled staging image.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void fast_conv_block(const KwsFastBlock &k, float *__restrict__ in, float *__restrict__ out,
                                                const float *__restrict__ shared, int lane, int out_stride, int out_halo,
                                                int out_rows, int out_cp)
{
    const int lm = lane & 15, lq = lane >> 4;
    const int MT = k.m_tiles, NT = k.n_tiles;
    v4f acc[4][2];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = v4f{ 0.f, 0.f, 0.f, 0.f };
    const int ncg = k.in_cp >> 3;
    const float *arow = in + lm * k.in_stride + 2 * lq;
    const float *wl = shared + k.w_off;
    const int n0 = min(lm, k.out_c - 1), n1 = min(16 + lm, k.out_c - 1);
    const int mstep = 16 * k.in_stride;
    for (int tap = 0; tap < k.taps; ++tap) {
        const float *at = arow + tap * k.in_stride;
        const float *wt = wl + ((size_t)(tap * (k.in_cp >> 1) + lq) * k.out_c) * 2;
        for (int cg = 0; cg < ncg; ++cg) {
            float2 a[4], b[2];
            b[0] = *(const float2 *)(wt + (size_t)(4 * cg * k.out_c + n0) * 2);
            if (NT > 1) b[1] = *(const float2 *)(wt + (size_t)(4 * cg * k.out_c + n1) * 2);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                if (mt < MT) a[mt] = *(const float2 *)(at + mt * mstep + 8 * cg);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                if (mt < MT) {
                    acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].x, b[0].x, acc[mt][0], 0, 0, 0);
                    if (NT > 1) acc[mt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].x, b[1].x, acc[mt][1], 0, 0, 0);
                }
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                if (mt < MT) {
                    acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].y, b[0].y, acc[mt][0], 0, 0, 0);
                    if (NT > 1) acc[mt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].y, b[1].y, acc[mt][1], 0, 0, 0);
                }
            }
        }
    }
    // ---- epilogue: bias, fused activation, ADD(constant) + activation (conv.h:88-93, add.h:200-212) ---------------------
    const bool pooled = k.pool > 1 || k.pool_stride > 1;
    float *stage = pooled ? in : out;                        // the input image is dead: every A operand is in a register
    const int sstride = pooled ? k.stage_stride : out_stride, shalo = pooled ? 0 : out_halo;
    if (!pooled) {
        // SAME-padding rows and the k-padding columns of the next block's image
        for (int i = lane; i < out_rows * out_stride; i += KWS_WAVE) {
            const int r = i / out_stride, c = i - r * out_stride;
            if (r < out_halo || r >= out_halo + k.out_w || (c >= k.out_c && c < out_cp)) out[i] = 0.0f;
        }
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        if (nt < NT) {
            const int n = 16 * nt + lm;
            const int nc = min(n, k.out_c - 1);
            const float bias = shared[k.bias_off + nc], addc = shared[k.addc_off + nc];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                if (mt < MT) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = 16 * mt + 4 * lq + i;
                        float v = acc[mt][nt][i] + bias;
                        v = fminf(fmaxf(v, k.conv_min), k.conv_max);
                        if (k.has_add) { v = v + addc; v = fminf(fmaxf(v, k.add_min), k.add_max); }
                        if (row < k.out_w && n < k.out_c) stage[(shalo + row) * sstride + n] = v;
                    }
                }
            }
        }
    }
    WAVE_SYNC();
    if (pooled) {
        // MAX_POOL_2D over time (pooling.h:189-237): windows clipped to the image, then the activation clamp
        for (int i = lane; i < out_rows * out_stride; i += KWS_WAVE) {
            const int r = i / out_stride, c = i - r * out_stride;
            const int p = r - out_halo;
            float v = 0.0f;
            bool wr = c < out_cp;
            if (p >= 0 && p < k.pool_w && c < k.out_c) {
                const int r0 = p * k.pool_stride;
                float m = -FLT_MAX;
                for (int j = 0; j < k.pool; ++j)
                    if (r0 + j < k.out_w) m = fmaxf(m, stage[(r0 + j) * sstride + c]);
                v = fminf(fmaxf(m, k.pool_min), k.pool_max);
            } else if (p >= 0 && p < k.pool_w && c >= out_cp) {
                wr = false;
            }
            if (wr) out[i] = v;
        }
        WAVE_SYNC();
    }
}

// ---------------------------------------------------------------------------------------------------------
//  cmvnw (processing.hpp:326-389) in place over the cepstra image, O(1) per (row, column): the window of padded row r + 1 is
//  the window of r minus padded row r plus padded row r + win, so running sums of d = x - pivot and d * d (pivot = the column's
//  first row: the sums stay small, var = Q/n - (S/n)^2 does not cancel) replace two win-term walks.  A lane owns one column and
//  CR consecutive rows; the first window of a row group is sum_j cnt[g][j] d_j with the multiplicities tabulated by the host.
//  Statistics stay in registers until every lane has read what it needs: only then are the rows overwritten.
// ---------------------------------------------------------------------------------------------------------
template <int CR, int CG, typename Emit>
__device__ __forceinline__ bool fast_cmvn(float *__restrict__ F, const float *__restrict__ shared, const KwsFastPlan &FP, int lane,
                                          int nfr, int ncep, Emit emit)
{
    constexpr int NG = KWS_WAVE / CG;
    const int cgrp = min(lane / CG, NG - 1), cl = lane - (lane / CG) * CG;
    const bool lane_on = lane < NG * CG;
    const int r0 = cgrp * CR;
    const float *cnt = shared + FP.cnt_off + cgrp * nfr;
    const int2 *upd = (const int2 *)(shared + FP.upd_off);
    const int fs = FP.fs;
    float *img = F + FP.f_halo * fs;
    bool bad = false;
    for (int cb = 0; cb < ncep; cb += CG) {
        const int c = cb + cl;
        const bool act = lane_on && c < ncep && r0 < nfr;
        float *col = img + min(c, ncep - 1);
        const float piv = col[0];
        float S = 0.0f, Q = 0.0f;
        for (int j = 0; j < nfr; ++j) {
            const float d = col[j * fs] - piv;
            const float w = cnt[j];
            const float wd = w * d;
            S += wd;
            Q = __fmaf_rn(wd, d, Q);
        }
        float mean[CR], rstd[CR];
#pragma unroll
        for (int i = 0; i < CR; ++i) {
            const int r = r0 + i;
            const float m = S * FP.inv_win;
            float var = __fmaf_rn(-m, m, Q * FP.inv_win);
            var = fmaxf(var, 0.0f);
            const float sd = __builtin_amdgcn_sqrtf(var);
            mean[i] = m;
            rstd[i] = __builtin_amdgcn_rcpf(sd + FLT_EPSILON);
            if (act && r < nfr) bad |= sd < FP.guard * fmaxf(1.0f, fabsf(m + piv));
            if (i + 1 < CR) {
                const int2 u = upd[min(r, nfr - 1)];
                const float dl = col[u.x] - piv, da = col[u.y] - piv;
                S = (S + da) - dl;
                Q = __fmaf_rn(da, da, Q);
                Q = __fmaf_rn(-dl, dl, Q);
            }
        }
        WAVE_SYNC();                                  // every lane's reads of this column block are done
#pragma unroll
        for (int i = 0; i < CR; ++i) {
            const int r = r0 + i;
            if (act && r < nfr) {
                const float o = ((col[r * fs] - piv) - mean[i]) * rstd[i];
                col[r * fs] = o;
                emit(r, c, o);
            }
        }
    }
    WAVE_SYNC();
    return bad;
}

// ---------------------------------------------------------------------------------------------------------
template <int NZ>
__global__ __launch_bounds__(512, 2) void kws_fast_kernel(KwsDspPlan P, const KwsFastPlan *__restrict__ FPp, const int16_t *__restrict__ pcm, int n_clips,
                                                          float *__restrict__ scores, float *__restrict__ features,
                                                          int8_t *__restrict__ q_out, float in_scale, int in_zp,
                                                          int *__restrict__ flag_count, int *__restrict__ flag_list)
{
    // the plan is read from memory (scalar loads, any block index); by value in the kernel arguments the compiler copies it to
    // scratch to index its blocks
    const KwsFastPlan &FP = *FPp;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & (KWS_WAVE - 1), wave = threadIdx.x >> 6;
    const int half = lane >> 5, t = lane & 31;
    float *shared = lds;
    float *F = lds + FP.shared_floats + wave * FP.wave_floats;       // log-mel -> cepstra -> features (block 0's input image)
    float *R1 = F + FP.f_floats;                                      // FFT buffers + power rows; later block 1's input image
    float *zw = R1, *pw = R1 + 2 * KWS_ZF;
    for (int i = threadIdx.x; i < FP.shared_floats; i += blockDim.x) shared[i] = FP.shared_init[i];
    __syncthreads();

    // ---- per-lane constants, fixed for the whole launch (the FFT is kws_mfcc_kernel's: KissFFT's order, bit for bit) ------
    const int k01 = t & 1, g01 = t >> 1;
    const int n0 = (g01 >> 2) + 4 * (g01 & 3);
    const cf a1 = to_cf(P.tw[16 * k01]), a2 = to_cf(P.tw[32 * k01]), a3 = to_cf(P.tw[48 * k01]);
    const int K2 = t & 7, G2 = t >> 3;
    const cf b1 = to_cf(P.tw[4 * K2]), b2 = to_cf(P.tw[8 * K2]), b3 = to_cf(P.tw[12 * K2]);
    const cf c1 = to_cf(P.tw[t]), c2 = to_cf(P.tw[2 * t]), c3 = to_cf(P.tw[3 * t]);
    const cf st1 = to_cf(P.stw[t]), st2 = to_cf(P.stw[t + 32]);
    const int nfr = P.n_frames, ncep = P.n_cepstral, NF = P.n_filters;
    const int n_pairs = (nfr + 1) >> 1;
    const int fs = FP.fs, halo = FP.f_halo;
    float *zb = zw + half * KWS_ZF;
    int off1[NZ], off2[NZ];
    float w1[NZ], w2[NZ];
#pragma unroll
    for (int n = 0; n < NZ; ++n) {
        const float2 v1 = FP.taps1[lane * NZ + n], v2 = FP.taps2[lane * NZ + n];
        off1[n] = __float_as_int(v1.x); w1[n] = v1.y;
        off2[n] = __float_as_int(v2.x); w2[n] = v2.y;
    }
    float dB[KWS_FAST_DCT_GROUPS][2][2];
#pragma unroll
    for (int g = 0; g < KWS_FAST_DCT_GROUPS; ++g)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                dB[g][i][nt] = (g < FP.dct_groups && nt < FP.dct_nt) ? FP.dct_frag[((g * 2 + i) * FP.dct_nt + nt) * KWS_WAVE + lane] : 0.0f;
    const int bmin = FP.bmin, bmax = FP.bmin + FP.nbins - 1, pstride = FP.pstride;
    const int n_waves = blockDim.x >> 6;

    for (int clip = blockIdx.x * n_waves + wave; clip < n_clips; clip += gridDim.x * n_waves) {
        const int16_t *xbase = pcm + (size_t)clip * P.n_samples;
        FastRaw nxt = fast_fetch(xbase, min(half, nfr - 1) * P.frame_stride + 8 * t, P.n_samples);
        FastRaw nxt2 = fast_fetch(xbase, min(2 + half, nfr - 1) * P.frame_stride + 8 * t, P.n_samples);
        // SAME-padding rows of block 0's image (the previous clip's staging image overwrote them)
        if (FP.fuse) {
            const int top = halo * fs, bot0 = (halo + nfr) * fs, bot = FP.f_rows * fs;
            for (int i = lane; i < top + (bot - bot0); i += KWS_WAVE) F[i < top ? i : bot0 + (i - top)] = 0.0f;
        }

        for (int pr = 0; pr < n_pairs; ++pr) {
            // ---- 8 samples per lane (16 B, coalesced), int16 -> float, pre-emphasis (numpy.hpp:1289, processing.hpp:104) ---
            const int f = 2 * pr + half;
            const FastRaw cur = nxt;
            nxt = nxt2;
            if (pr + 2 < n_pairs) nxt2 = fast_fetch(xbase, min(f + 4, nfr - 1) * P.frame_stride + 8 * t, P.n_samples);
            float y[8];
            {
                float prev = (float)cur.prev * (1.0f / 32768.0f);
                const int w[4] = { cur.v.x, cur.v.y, cur.v.z, cur.v.w };
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float lo = (float)(short)(w[j] & 0xffff) * (1.0f / 32768.0f);
                    const float hi = (float)(short)(w[j] >> 16) * (1.0f / 32768.0f);
                    const float pl = P.pre_cof * prev;
                    y[2 * j] = lo - pl;
                    const float ph = P.pre_cof * lo;
                    y[2 * j + 1] = hi - ph;
                    prev = hi;
                }
            }
            *(float4 *)(zb + 2 * zi(4 * t)) = make_float4(y[0], y[1], y[2], y[3]);
            *(float4 *)(zb + 2 * zi(4 * t) + 4) = make_float4(y[4], y[5], y[6], y[7]);
            WAVE_SYNC();
            // ---- kf_bfly2 (m = 1) fused with kf_bfly4 (m = 2), then kf_bfly4 m = 8 and m = 32 (kiss_fft.cpp:15-84, 232-296) ---
            cf u[4];
            {
                cf la[4], lb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { la[i] = ld_cf(zb, n0 + 16 * i); lb[i] = ld_cf(zb, n0 + 16 * i + 64); }
#pragma unroll
                for (int i = 0; i < 4; ++i) u[i] = k01 ? csub(la[i], lb[i]) : cadd(la[i], lb[i]);
            }
            bfly4(u[0], u[1], u[2], u[3], a1, a2, a3);
#pragma unroll
            for (int i = 0; i < 4; ++i) st_cf(zb, 8 * g01 + k01 + 2 * i, u[i]);
            WAVE_SYNC();
#pragma unroll
            for (int i = 0; i < 4; ++i) u[i] = ld_cf(zb, 32 * G2 + K2 + 8 * i);
            bfly4(u[0], u[1], u[2], u[3], b1, b2, b3);
#pragma unroll
            for (int i = 0; i < 4; ++i) st_cf(zb, 32 * G2 + K2 + 8 * i, u[i]);
            WAVE_SYNC();
#pragma unroll
            for (int i = 0; i < 4; ++i) u[i] = ld_cf(zb, t + 32 * i);
            bfly4(u[0], u[1], u[2], u[3], c1, c2, c3);
#pragma unroll
            for (int i = 0; i < 4; ++i) st_cf(zb, t + 32 * i, u[i]);
            WAVE_SYNC();
            // ---- kiss_fftr split (kiss_fftr.cpp:84-119) and the power spectrum, fp32: |X|^2 / fft_length ------------------
            const bool live = f < nfr;
            float *prow = pw + ((2 * pr + half) & (KWS_FAST_MEL_CHUNK - 1)) * pstride - bmin;
            float esum = 0.0f;
            cf fpk[2], fq[2];
#pragma unroll
            for (int rep = 0; rep < 2; ++rep) {
                const int k = t + 1 + 32 * rep;
                fpk[rep] = ld_cf(zb, k);
                fq[rep] = ld_cf(zb, KWS_NC - k);
            }
            const float2 d0 = *(const float2 *)zb;                  // tmp[0]: DC and Nyquist bins (kiss_fftr.cpp:84-96)
#pragma unroll
            for (int rep = 0; rep < 2; ++rep) {
                const int k = t + 1 + 32 * rep;
                const cf stw = rep ? st2 : st1;
                cf fpnk; fpnk.r = fq[rep].r; fpnk.i = -fq[rep].i;
                const cf f1k = cadd(fpk[rep], fpnk), f2k = csub(fpk[rep], fpnk);
                const cf twv = cmul(f2k, stw);
                cf lo, hi;
                lo.r = (f1k.r + twv.r) * 0.5f;
                lo.i = (f1k.i + twv.i) * 0.5f;
                hi.r = (f1k.r - twv.r) * 0.5f;
                hi.i = (twv.i - f1k.i) * 0.5f;
                const float plo = __fmaf_rn(lo.r, lo.r, lo.i * lo.i) * P.inv_fft;
                const float phi = __fmaf_rn(hi.r, hi.r, hi.i * hi.i) * P.inv_fft;
                if (k != KWS_NC / 2) {                               // bin 64 is written twice by the reference: the second store wins
                    esum += plo;
                    if (live && k >= bmin && k <= bmax) prow[k] = plo;
                }
                esum += phi;
                if (live && KWS_NC - k >= bmin && KWS_NC - k <= bmax) prow[KWS_NC - k] = phi;
            }
            if (t == 0) {
                const float dc = d0.x + d0.y, ny = d0.x - d0.y;
                const float pdc = (dc * dc) * P.inv_fft, pny = (ny * ny) * P.inv_fft;
                esum += pdc + pny;
                if (live && bmin == 0) prow[0] = pdc;
            }
            // frame energy (feature.hpp:289-298): its log is parked in the image's last column until the DCT has run
            esum = half_wave_sum(esum);
            if (t == 0 && live) F[(halo + f) * fs + fs - 1] = fast_log(esum == 0.0f ? FLT_EPSILON : esum);
            if ((pr & (KWS_FAST_MEL_CHUNK / 2 - 1)) != KWS_FAST_MEL_CHUNK / 2 - 1 && pr != n_pairs - 1) continue;

            // ---- mel filterbank for the buffered frames: dot_by_row as a register-tap gather, zero handling, log ----------
            WAVE_SYNC();
            const int fbase = (2 * pr) & ~(KWS_FAST_MEL_CHUNK - 1);
            const int nfc = min(KWS_FAST_MEL_CHUNK, nfr - fbase);
#pragma unroll
            for (int s = 0; s < KWS_FAST_MEL_CHUNK / 2; ++s) {       // filters 0..31: two frames per pass
                const int slot = 2 * s + half;
                const float *pr_ = pw + slot * pstride;
                float acc = 0.0f;
#pragma unroll
                for (int n = 0; n < NZ; ++n) acc = __fmaf_rn(pr_[off1[n]], w1[n], acc);
                if (acc == 0.0f) acc = FLT_EPSILON;                  // functions.hpp:63-69
                if (slot < nfc && t < NF) F[(halo + fbase + slot) * fs + t] = fast_log(acc);
            }
            if (FP.nf2p) {                                            // filters 32..NF-1: nf2p per frame slot
                const int fpp = KWS_WAVE / FP.nf2p;                   // frames per pass
                const int j2 = 32 + (lane & (FP.nf2p - 1)), sl0 = lane / FP.nf2p;
                for (int q = 0; q * fpp < KWS_FAST_MEL_CHUNK; ++q) {
                    const int slot = sl0 + q * fpp;
                    const float *pr_ = pw + slot * pstride;
                    float acc = 0.0f;
#pragma unroll
                    for (int n = 0; n < NZ; ++n) acc = __fmaf_rn(pr_[off2[n]], w2[n], acc);
                    if (acc == 0.0f) acc = FLT_EPSILON;
                    if (slot < nfc && j2 < NF) F[(halo + fbase + slot) * fs + j2] = fast_log(acc);
                }
            }
            WAVE_SYNC();
        }

        // ---- DCT-II (numpy.hpp:378-401) as [frames x NF] x [NF x NF/2+1] on the matrix cores, in place ------------------
        {
            const int lm = lane & 15, lq = lane >> 4;
            const int mtiles = (nfr + 15) >> 4;
            float *img = F + halo * fs;
            for (int mt = 0; mt < mtiles; ++mt) {
                const float *arow = img + (16 * mt + lm) * fs + 2 * lq;
                float2 a[KWS_FAST_DCT_GROUPS];
#pragma unroll
                for (int g = 0; g < KWS_FAST_DCT_GROUPS; ++g)
                    if (g < FP.dct_groups) a[g] = *(const float2 *)(arow + 8 * g);
                v4f acc[2] = { v4f{ 0.f, 0.f, 0.f, 0.f }, v4f{ 0.f, 0.f, 0.f, 0.f } };
#pragma unroll
                for (int g = 0; g < KWS_FAST_DCT_GROUPS; ++g) {
                    if (g < FP.dct_groups) {
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {
                            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g].x, dB[g][0][nt], acc[nt], 0, 0, 0);
                            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g].y, dB[g][1][nt], acc[nt], 0, 0, 0);
                        }
                    }
                }
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int n = 16 * nt + lm;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = 16 * mt + 4 * lq + i;
                        if (r < nfr && n <= NF / 2 && n > 0) img[r * fs + n] = acc[nt][i];   // column 0 is replaced below
                    }
                }
            }
            WAVE_SYNC();
            // c0 <- log(frame energy) (feature.hpp:425-429); coefficients above NF/2 are never written by the reference's
            // transform: they keep the log-mel input, doubled and scaled (fast-dct-fft.cpp:71-74, numpy.hpp:392-397)
            if (lane < nfr) img[lane * fs] = img[lane * fs + fs - 1];
            const int nst = ncep - (NF / 2 + 1);
            if (nst > 0)
                for (int i = lane; i < nfr * nst; i += KWS_WAVE) {
                    const int r = i / nst, c = NF / 2 + 1 + (i - r * nst);
                    img[r * fs + c] = (img[r * fs + c] * 2.0f) * FP.stale_scale;
                }
            WAVE_SYNC();
        }

        // ---- cmvnw + optional outputs (extract_mfcc_features' matrix, the int8 input tensor) --------------------------------
        float *fout = features ? features + (size_t)clip * (nfr * ncep) : nullptr;
        int8_t *qclip = q_out ? q_out + (size_t)clip * (nfr * ncep) : nullptr;
        auto emit = [&](int row, int c, float o) {
            const int idx = row * ncep + c;
            if (fout) fout[idx] = o;
            if (qclip) qclip[idx] = quantize_feature(o, in_scale, in_zp);
        };
        bool bad;
        if (FP.cr == 13) bad = fast_cmvn<13, 16>(F, shared, FP, lane, nfr, ncep, emit);
        else bad = fast_cmvn<17, 20>(F, shared, FP, lane, nfr, ncep, emit);
        if (__any(bad)) {
            if (lane == 0) flag_list[atomicAdd(flag_count, 1)] = clip;
        }
        if (!FP.fuse) continue;

        // ---- the float32 graph: CONV_2D blocks ping-pong between the two images, then FULLY_CONNECTED and SOFTMAX ----------
        float *cur = F, *oth = R1;
        for (int b = 0; b < FP.n_blocks; ++b) {
            const KwsFastBlock &k = FP.blk[b];
            const bool last = b + 1 == FP.n_blocks;
            const int o_stride = last ? k.out_c : FP.blk[b + 1].in_stride;
            const int o_halo = last ? 0 : FP.blk[b + 1].pad_left;
            const int o_rows = last ? k.pool_w : FP.blk[b + 1].in_rows;
            const int o_cp = last ? k.out_c : FP.blk[b + 1].in_cp;
            fast_conv_block(k, cur, oth, shared, lane, o_stride, o_halo, o_rows, o_cp);
            float *tmp = cur; cur = oth; oth = tmp;
        }
        {
            // FULLY_CONNECTED (fully_connected.h:26-60): lane = output unit; SOFTMAX (softmax.h:31-63)
            const float *wfc = shared + FP.fc_w_off + min(lane, FP.fc_out - 1) * FP.fc_in;
            float tot = 0.0f;
            for (int i = 0; i < FP.fc_in; ++i) tot = __fmaf_rn(cur[i], wfc[i], tot);
            tot += shared[FP.fc_b_off + min(lane, FP.fc_out - 1)];
            tot = fminf(fmaxf(tot, FP.fc_min), FP.fc_max);
            const bool on = lane < FP.fc_out;
            const float mx = wave_max(on ? tot : -FLT_MAX);
            const float e = on ? expf((tot - mx) * FP.beta) : 0.0f;
            const float sum = wave_sum(e);
            if (on) scores[(size_t)clip * FP.n_labels + lane] = e / sum;
        }
        WAVE_SYNC();
    }
}

// ---------------------------------------------------------------------------------------------------------
//  launcher (called from kws_api.cpp)
// ---------------------------------------------------------------------------------------------------------
template <int NZ>
static int launch_fast_t(const KwsDspPlan &P, const KwsFastPlan &FP, const KwsFastPlan *d_plan, const int16_t *pcm, int n_clips, float *scores, float *features,
                         int8_t *q_out, float in_scale, int in_zp, int *flag_count, int *flag_list, int n_cu, hipStream_t stream)
{
    const size_t smem = ((size_t)FP.shared_floats + (size_t)FP.n_waves * FP.wave_floats) * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void *)kws_fast_kernel<NZ>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return (int)hipGetLastError();
        attr_done = true;
    }
    const int per_wg = FP.n_waves;
    int grid = (n_clips + per_wg - 1) / per_wg;
    if (grid > n_cu) grid = n_cu;
    hipLaunchKernelGGL(kws_fast_kernel<NZ>, dim3(grid), dim3(KWS_WAVE * FP.n_waves), smem, stream, P, d_plan, pcm, n_clips, scores, features,
                       q_out, in_scale, in_zp, flag_count, flag_list);
    return (int)hipGetLastError();
}

int kws_launch_fast(const KwsDspPlan &P, const KwsFastPlan &FP, const KwsFastPlan *d_plan, const int16_t *pcm, int n_clips, float *scores, float *features,
                    int8_t *q_out, float in_scale, int in_zp, int *flag_count, int *flag_list, int n_cu, hipStream_t stream)
{
    (void)hipGetLastError();
    if (n_clips <= 0) return 0;
    if (FP.nz <= 4) return launch_fast_t<4>(P, FP, d_plan, pcm, n_clips, scores, features, q_out, in_scale, in_zp, flag_count, flag_list, n_cu, stream);
    if (FP.nz <= 8) return launch_fast_t<8>(P, FP, d_plan, pcm, n_clips, scores, features, q_out, in_scale, in_zp, flag_count, flag_list, n_cu, stream);
    return launch_fast_t<KWS_FAST_NZ_MAX>(P, FP, d_plan, pcm, n_clips, scores, features, q_out, in_scale, in_zp, flag_count, flag_list, n_cu, stream);
}
