// kws_fast.hip -- kws_fast_kernel: KWS_MODE_FAST, the tolerance-mode form of run_classifier()'s hot path (see kws_fast.h for
// what is relaxed and what is not).  One wavefront owns one clip from the int16 PCM in HBM to its scores: extract_mfcc_features
// (SDK/classifier/ei_run_dsp.h:256-308) with the FFT in KissFFT's order and everything behind it in plain fp32 (the DCT on
// v_mfma_f32_16x16x4_f32), then -- for a float32 graph -- the CONV_2D / ADD / MAX_POOL_2D / FULLY_CONNECTED / SOFTMAX chain
// (TFL/kernels/internal/reference/conv.h:28-99, add.h:179-215, pooling.h:189-237, fully_connected.h:26-60, softmax.h:31-63) with the
// contractions on v_mfma_f32_16x16x32_f16: every fp32 operand carried as two fp16 halves (22 significant bits), three products per pair,
// fp32 accumulation (fast_conv_tiles_h; blocks whose image does not fit the in-place split keep v_mfma_f32_16x16x4_f32, fast_conv_tiles).
// Eight waves of a workgroup share the weights in LDS; nothing but the PCM and the scores crosses HBM.
//
// With 13.6 KB of LDS per wave next to the shared 46 KB of weight fragments and tables, and 256 VGPRs, two waves fit a SIMD, so every
// phase is written to keep its own memory operations in flight: loads of a phase are issued as one batch before the arithmetic that
// consumes them, tile counts are template parameters (no predicated code inside the contraction loops), and the convolution loops
// fetch their operands ahead of the matrix instructions.  DESIGN.md 4.4 has the phase table, the counters and what was tried and rejected.
//
// Round 6: this file is compiled TWICE (csrc/Makefile).  The second compilation, -DKWS_FAST_WPS=3 -> kws_fast_w3.o, builds the PCM-entry float32-network
// forms for three waves per SIMD (<= 168 registers, twelve / eleven waves per workgroup) under the names *_w3; what it does differently sits behind
// KWS_FAST_WPS >= 3 below -- twiddles of the pass loop from an LDS table, no register reloaded from scratch inside a loop that prefetches, fragments
// from L2 in three rotating register sets, one sink per workgroup, clips dealt out to the waves by ticket -- and in profiles/r06_occupancy.md.  A plan
// says which build runs it (KwsFastPlan::wps).
#include <atomic>

#include "kws_device.h"
#include "kws_fast.h"
// The second compilation of this file (kws_fast_w3.o, -DKWS_FAST_WPS=3: three waves per SIMD, <= 168 registers, the float32-network forms only) goes into the
// same library under names of its own; the launchers of the first one hand a plan laid out for three waves (KwsFastPlan::wps) over to them.
#if KWS_FAST_WPS >= 3
#define kws_fast_kernel kws_fast_kernel_w3
#define kws_launch_fast kws_launch_fast_w3
#define kws_launch_fast_prof kws_launch_fast_prof_w3
#endif
#include "kws_nn_int8_dev.h"

typedef float v4f __attribute__((ext_vector_type(4)));
// The three-waves-per-SIMD build (KWS_FAST_WPS = 3: 168 registers; an experiment, profiles/r06_occupancy.md): a lane index is made opaque again at the
// entry of every phase, so that the dozens of lane-derived constants of a LATER phase are not worked out ahead of an earlier one's loops and spilled.
// Expands to nothing in the product build (two waves per SIMD), whose code it must not perturb.
#if KWS_FAST_WPS >= 3
#define KWS_OPAQUE3(v) asm volatile("" : "+v"(v))
#define KWS_FAST_SINK (shared + FP.sink_off)          // one sink per workgroup (KwsFastPlan::sink_off is relative to the shared block in this build)
#else
#define KWS_OPAQUE3(v) do { } while (0)
#define KWS_FAST_SINK (F + FP.sink_off)               // a sink per wave, behind its image
#endif
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
// whole-wave reductions, every lane receives the result: four DPP steps inside a row of 16 (no LDS round trip), then lane ^ 16
// and lane ^ 32 through the LDS crossbar
__device__ __forceinline__ float wave_max(float v)
{
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    v = fmaxf(v, dpp_mov<0x140>(v));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F)));
    return fmaxf(v, __shfl_xor(v, 32, KWS_WAVE));
}
__device__ __forceinline__ float wave_sum(float v)
{
    v = half_wave_sum(v);
    return v + __shfl_xor(v, 32, KWS_WAVE);
}
// sum over aligned groups of S lanes (S a power of two, wave-uniform); every lane of a group receives it
__device__ __forceinline__ float group_sum(float v, int S)
{
    if (S > 1) v += dpp_mov<0xB1>(v);
    if (S > 2) v += dpp_mov<0x4E>(v);
    if (S > 4) v += dpp_mov<0x141>(v);
    if (S > 8) v += dpp_mov<0x140>(v);
    if (S > 16) v += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F));
    if (S > 32) v += __shfl_xor(v, 32, KWS_WAVE);
    return v;
}

// The int16 -> float scale 2^-15 (numpy.hpp:1289) is not applied here: pre-emphasis, the FFT and the split are linear and every
// one of their roundings commutes with a power of two, so the factor is folded -- exactly -- into the power spectrum's scale.
__device__ __forceinline__ cf fast_point(fast_i2 v, float pre_cof)
{
    const float prev = (float)(v.x >> 16);
    const float lo = (float)(short)(v.y & 0xffff);
    const float hi = (float)(v.y >> 16);
    cf z;
    const float pl = pre_cof * prev;
    z.r = lo - pl;
    const float ph_ = pre_cof * lo;
    z.i = hi - ph_;
    return z;
}
// sum over the eight lanes of a frame group; every lane of the group receives it
__device__ __forceinline__ float oct_sum(float v)
{
    v += dpp_mov<0xB1>(v);        // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);        // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);       // row_half_mirror: lanes i <-> 7 - i
    return v;
}

// ---------------------------------------------------------------------------------------------------------
//  What follows a block's contraction, whichever matrix instruction ran it: the accumulators acc[MT][NT] (tile mt, nt: rows
//  16 mt + 4 (lane / 16) + i, channel 16 nt + lane % 16) and the vector-ALU rows vout hold  sum x w / scale  (scale = 1 for the fp32
//  instruction; the split-operand form carries the two powers of two its operands were multiplied by).
// ---------------------------------------------------------------------------------------------------------
template <int MT, int NT>
__device__ __forceinline__ void fast_conv_finish(const KwsFastBlock &k, v4f (&acc)[MT][NT], const float (&vout)[2], float *__restrict__ stage, int sstride,
                                                 const float *__restrict__ shared, int lane, float *__restrict__ sink, float scale)
{
    KWS_OPAQUE3(lane);
    const int lm = lane & 15, lq = lane >> 4;
    const int out_c = k.out_c;
    const int out_w = k.out_w;
    if (k.fpool) {
        // ---- MAX_POOL_2D with non-overlapping windows, taken on the raw accumulators: bias, the activation clamps and the ADD are
        //      non-decreasing in the accumulator, so max and epilogue commute (same values, bit for bit) and only pool_w x out_c
        //      values go through the epilogue (fast_pool_finish).  A lane folds its four rows of a tile into at most two window
        //      maxima in registers and merges them into pm[window][32] (the dead input image) with LDS float-max atomics.
        float *pm = stage;
        WAVE_SYNC();                                                  // every lane has read its last operands from the image
        for (int i = lane; i < k.pool_w * 32; i += KWS_WAVE) pm[i] = -FLT_MAX;
        WAVE_SYNC();
        const unsigned pinv = k.inv_pool16;                             // r / pool for r < 64 (the plan's reciprocal)
        const int pool_w = k.pool_w;
        float *const psink = pm + pool_w * 32 + lane;                  // windows past the last one (VALID pooling), channels past out_c
        auto merge = [&](int p, int n, float m, bool ok) {
            __builtin_amdgcn_ds_fmaxf((__attribute__((address_space(3))) float *)((ok && p < pool_w && n < out_c) ? pm + p * 32 + n : psink), m, __ATOMIC_RELAXED, __MEMORY_SCOPE_WRKGRP, false);
        };
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = 16 * nt + lm;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int r0 = 16 * mt + 4 * lq;
                int pw[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) pw[i] = (int)(__umul24((unsigned)(r0 + i), pinv) >> 16);
                float mlo = acc[mt][nt][0], mhi = acc[mt][nt][3];      // pool >= 4: the four rows touch at most two windows
#pragma unroll
                for (int i = 1; i < 4; ++i) {
                    const bool row_ok = r0 + i < out_w;
                    mlo = (row_ok && pw[i] == pw[0]) ? fmaxf(mlo, acc[mt][nt][i]) : mlo;
                    if (i < 3) mhi = (row_ok && pw[i] == pw[3]) ? fmaxf(mhi, acc[mt][nt][i]) : mhi;
                }
                merge(pw[0], n, mlo, r0 < out_w);
                merge(pw[3], n, mhi, r0 + 3 < out_w && pw[3] != pw[0]);
            }
        }
        for (int vr = 0; vr < k.vrows; ++vr) {
            const int r = 16 * MT + vr;
            merge((int)(((unsigned)r * pinv) >> 16), lane & 31, vout[vr], lane < 32);
        }
        return;
    }
    // ---- epilogue: bias, fused activation, ADD(constant) + activation (conv.h:88-93, add.h:200-212) ---------------------
    const float cmin = k.conv_min, cmax = k.conv_max, amin = k.add_min, amax = k.add_max;
    const bool has_add = k.has_add != 0;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = 16 * nt + lm;
        const int nc = min(n, out_c - 1);
        const float bias = shared[k.bias_off + nc], addc = shared[k.addc_off + nc];
        float *sp = stage + n;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 16 * mt + 4 * lq + i;
                float v = acc[mt][nt][i] * scale + bias;
                v = fminf(fmaxf(v, cmin), cmax);
                if (has_add) { v = v + addc; v = fminf(fmaxf(v, amin), amax); }
                *((row < out_w && n < out_c) ? sp + __mul24(row, sstride) : sink) = v;
            }
        }
    }
    for (int vr = 0; vr < k.vrows; ++vr) {
        const int n = lane & 31, nc = min(n, out_c - 1);
        float v = vout[vr] * scale + shared[k.bias_off + nc];
        v = fminf(fmaxf(v, cmin), cmax);
        if (has_add) { v = v + shared[k.addc_off + nc]; v = fminf(fmaxf(v, amin), amax); }
        if (lane < 32 && n < out_c) stage[(16 * MT + vr) * sstride + n] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------
//  One CONV_2D block on the matrix cores: out[m][n] = sum_{tap, c} in[m + tap - pad_left][c] * w[n][tap][c].  The image holds the
//  in_w real rows only: an operand whose row falls outside it (SAME padding, or the rows a 16-row tile has beyond the image) is
//  replaced by zero in the register -- padding rows in LDS would cost 1 KB per wave.  Tiles of 16 rows x 16 channels, k-steps of
//  4 (tap, channel) pairs; a lane fetches two k-steps' operands with one 8-byte read each:
//      A: image[(16 mt + l % 16 + tap - pad_left) * stride + 8 cg + 2 (l / 16) + {0, 1}]
//      B: w[tap][4 cg + l / 16][n][{0, 1}]  =  w2[(4 it + l / 16) * out_c + n],  it = tap * (in_cp / 8) + cg
//  Every accumulator (MT x NT tiles) stays in registers until the contraction is complete: only then is the input image dead
//  and may be overwritten by the un-pooled staging image.  The operands of step it + 1 are requested before the MFMAs of
//  step it are issued.
// ---------------------------------------------------------------------------------------------------------
template <int MT, int NT>
__device__ __forceinline__ float fast_conv_tiles(const KwsFastBlock &k, const float *__restrict__ in, float *__restrict__ stage,
                                                int sstride, const float *__restrict__ shared, int lane, float *__restrict__ sink,
                                                long long *t_loop = nullptr, long long *t_pre = nullptr)
{
    const int lm = lane & 15, lq = lane >> 4;
    v4f acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = v4f{ 0.f, 0.f, 0.f, 0.f };
    const int in_stride = k.in_stride, out_c = k.out_c, in_w = k.in_w;
    const int n_it = k.taps * (k.in_cp >> 3);
    // operand addresses as offsets from one LDS pointer each (pointers kept in an array lose their address space and every
    // weight read becomes a flat_load): a lane part in a VGPR + a step part the scalar unit advances
    const float *abase = in + (lm - k.pad_left) * in_stride + 2 * lq;
    const float *wbase = shared + k.w_off + 2 * (lq * out_c);
    int bl[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bl[nt] = 2 * min(16 * nt + lm, out_c - 1);
    const int mstep = 16 * in_stride;
    const int row0 = lm - k.pad_left;                       // image row of this lane's operand for tile 0, tap 0
    // the zeroing is applied when a fetched operand is handed to the MFMAs, not at the load: the loads of a step then go out
    // back to back and are only waited for after the previous step's MFMAs have been issued
    auto clip_rows = [&](int tap, float2 (&v)[MT]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const bool in_img = (unsigned)(row0 + tap + 16 * mt) < (unsigned)in_w;
            v[mt].x = in_img ? v[mt].x : 0.0f;
            v[mt].y = in_img ? v[mt].y : 0.0f;
        }
    };
    // ---- the rows beyond the tiles (k.vrows <= 2) on the vector ALU, before anything is written over the input image: lane =
    //      output channel x half of every tap's channel pairs; the input row is read as a broadcast, taps that fall into the
    //      SAME padding are skipped
    float vout[2] = { 0.0f, 0.0f };
    for (int vr = 0; vr < k.vrows; ++vr) {
        const int row = 16 * MT + vr, n = min(lane & 31, out_c - 1), hf = lane >> 5;
        const int npair = k.in_cp >> 2;                               // channel pairs per tap and half
        float s0 = 0.0f, s1 = 0.0f;
        for (int tap = 0; tap < k.taps; ++tap) {
            const int r = row + tap - k.pad_left;
            if ((unsigned)r >= (unsigned)in_w) continue;
            const float *ap = in + r * in_stride + 2 * hf * npair;
            const float *wp = shared + k.w_off + 2 * ((tap * (k.in_cp >> 1) + hf * npair) * out_c + n);
            int p = 0;
            // five double pairs at a time (a 40-channel image: a whole tap of this lane's half): fifteen requests go out before the
            // first product is formed, so the tap pays one LDS round trip instead of five
            for (; p + 9 < npair; p += 10) {
                float4 av[5];
                float2 wv[10];
#pragma unroll
                for (int i = 0; i < 5; ++i) av[i] = *(const float4 *)(ap + 2 * p + 4 * i);
#pragma unroll
                for (int i = 0; i < 10; ++i) wv[i] = *(const float2 *)(wp + 2 * (p + i) * out_c);
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    s0 = __fmaf_rn(av[i].x, wv[2 * i].x, s0); s1 = __fmaf_rn(av[i].y, wv[2 * i].y, s1);
                    s0 = __fmaf_rn(av[i].z, wv[2 * i + 1].x, s0); s1 = __fmaf_rn(av[i].w, wv[2 * i + 1].y, s1);
                }
            }
            for (; p + 1 < npair; p += 2) {
                const float4 av = *(const float4 *)(ap + 2 * p);
                const float2 w0 = *(const float2 *)(wp + 2 * p * out_c), w1 = *(const float2 *)(wp + 2 * (p + 1) * out_c);
                s0 = __fmaf_rn(av.x, w0.x, s0); s1 = __fmaf_rn(av.y, w0.y, s1);
                s0 = __fmaf_rn(av.z, w1.x, s0); s1 = __fmaf_rn(av.w, w1.y, s1);
            }
            if (p < npair) {
                const float2 av = *(const float2 *)(ap + 2 * p), w0 = *(const float2 *)(wp + 2 * p * out_c);
                s0 = __fmaf_rn(av.x, w0.x, s0); s1 = __fmaf_rn(av.y, w0.y, s1);
            }
        }
        const float part = s0 + s1;
        vout[vr] = part + __shfl_xor(part, 32, KWS_WAVE);
    }
    // Where a k-step's operands sit -- (image offset, tap, weight offset) -- comes from a table in the workgroup's LDS block (one
    // broadcast 16-byte read per step, requested a step before it is needed): advanced on the scalar unit, the same bookkeeping
    // was two dozen dependent scalar instructions per step, and a wave pays issue time for every one of them.
    // The loop is unrolled by two over two operand sets (no register moves between steps); a set's operands are requested
    // before the other set's MFMAs are issued.
    const int4 *steps = (const int4 *)(shared + k.st_off);
    auto fetch = [&](const int4 &d, float2 (&aa)[MT], float2 (&bb)[NT]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) aa[mt] = *(const float2 *)(abase + d.x + mt * mstep);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bb[nt] = *(const float2 *)(wbase + d.z + bl[nt]);
    };
    auto mfmas = [&](const float2 (&aa)[MT], const float2 (&bb)[NT]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[mt].x, bb[nt].x, acc[mt][nt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[mt].y, bb[nt].y, acc[mt][nt], 0, 0, 0);
    };
    float2 a0[MT], b0[NT], a1[MT], b1[NT];
    int4 d0 = steps[0], d1 = steps[1];
    fetch(d0, a0, b0);
    if (t_pre) *t_pre = clock64();
    // The contraction runs at raised wave priority: when both waves of a SIMD have an instruction ready, the one feeding the matrix pipe
    // goes first -- its MFMAs then execute while the other wave's vector instructions issue, instead of waiting behind them.  Measured
    // (same-box A/B, 65 536 clips): 1.669 -> 1.637 ms for the 49x40 graph, 1.246 -> 1.228 ms for the 49x13 twin; priority 3: the same.
    __builtin_amdgcn_s_setprio(1);
    // per half: request the other set's operands, zero this set's out-of-image rows (they were requested a half earlier: no wait),
    // issue this set's MFMAs
    for (int it = 0; it < n_it; it += 2) {
        const int4 d2 = steps[it + 2];                                   // n_it + 3 entries: the last step is repeated
        fetch(d1, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        clip_rows(d0.y, a0);
        mfmas(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        if (it + 1 >= n_it) break;                                        // odd step count
        const int tap1 = d1.y;
        d1 = steps[it + 3];
        fetch(d2, a0, b0);
        d0 = d2;
        __builtin_amdgcn_sched_barrier(0);
        clip_rows(tap1, a1);
        mfmas(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
    if (t_loop) *t_loop = clock64();
    fast_conv_finish<MT, NT>(k, acc, vout, stage, sstride, shared, lane, sink, 1.0f);
    return 1.0f;
}

// ---------------------------------------------------------------------------------------------------------
//  The same contraction on v_mfma_f32_16x16x32_f16 with SPLIT OPERANDS (KwsFastBlock::hconv).  An fp32 value x is carried as two
//  halves of x s (s a power of two): hi = half(x s), lo = half(x s - hi) -- 22 significant bits --, and a product as
//  hi hi + hi lo + lo hi, each term exact in the fp32 accumulator: what is dropped (lo lo, and the 2^-22 the two halves miss) is the
//  size of fp32's own rounding of the product, and the matrix pipe runs the three instructions in 3/16 of the time one
//  v_mfma_f32_16x16x4_f32 sequence of the same depth takes (tools/split_operand_study.py: the scores of the shipped float graphs move
//  by <= 6e-7, less than the reference's own summation order does).
//    fast_split_image  converts a block's input image IN PLACE: row r = [in_cp halves hi][in_cp halves lo] inside the row's
//                      in_stride floats; s = 2^14 / 2^ceil(log2 max|x|) of this clip's image (values below 2^-10 s^-1 would land in
//                      half's subnormals: with max|x| s >= 2^13 they are 2^-23 of the largest operand -- nothing).
//    the weights       are split once on the host (kws_fast_plan.cpp) into per-lane fragments [k-step][hi, lo][n tile][lane] x 16 bytes.
//  K runs over groups of eight channels of one tap, g = tap * (in_cp / 8) + cg; k-step s gives group 4 s + (lane / 16) to the lanes
//  with that quotient: A = the eight halves at image row (16 mt + lane % 16 + tap - pad_left), channels 8 cg .. 8 cg + 7 (one
//  ds_read_b128; rows outside the image read a block of zeros instead), B = this lane's 16 bytes of the fragment.  With eight
//  output tiles in registers a 49-row image runs its fourth row tile for one row: 72 of 216 instructions of 16 cycles -- the fp32
//  form paid 140 of 560 of 32 cycles for it, or a vector-ALU row (KwsFastBlock::vrows: 4.9 k clocks per clip).
// ---------------------------------------------------------------------------------------------------------
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
// a pointer the plan hands over as void * is a flat address to the compiler: flat loads count on the LDS counter too and return out of order, so every
// wait near them becomes "everything"; device memory said as such gives global loads (round 6)
typedef const __attribute__((address_space(1))) char *kws_gptr;
typedef const __attribute__((address_space(1))) v8h *kws_gv8h;
#define KWS_FAST_HP 16            // channel pairs a lane converts: images of up to 64 x 16 pairs (the plan checks)

// T trips of the wave, every one of them whole (pairs past the last one: the lane re-reads the last pair and stores the same halves to the same
// place).  The reads of all T trips go out as ONE batch, with nothing conditional around them: a trip in a block of its own ("if this trip has
// items: read") makes the compiler wait for every read before it issues the next one -- sixteen exposed LDS round trips, measured as 34 % of
// the first convolution block's clocks (profiles/r05_fast_subphase.txt).
template <int T>
__device__ __forceinline__ float fast_split_trips(float *__restrict__ img, int items, int in_c, int in_cp, int in_stride, int lane, unsigned inv, int ppr)
{
    float2 v[T];
    int off[T];                                                     // byte offset of the pair's hi halves (kept: the second loop needs no division)
    int pc[T];                                                      // the pair's first channel
#pragma unroll
    for (int u = 0; u < T; ++u) {
        const int i = min(lane + KWS_WAVE * u, items - 1);
        // (24-bit multiplies: v_mul_lo_u32 is a quarter-rate instruction)
        const int r = (int)(__umul24((unsigned)i, inv) >> 20), p2 = 2 * (i - __mul24(r, ppr));
        const int ro = __mul24(r, in_stride);
        v[u] = *(const float2 *)(img + ro + p2);
        off[u] = 4 * ro + 2 * p2;
        pc[u] = p2;
    }
    // (the values are pinned here: the selects below must not turn the reads above into predicated reads, which would wait one by one)
#pragma unroll
    for (int u = 0; u < T; ++u) asm volatile("" : "+v"(v[u].x), "+v"(v[u].y));
    float mx = 0.0f;
#pragma unroll
    for (int u = 0; u < T; ++u) {
        // the k-padding channels (in_c .. in_cp - 1) meet zero weights, but what sits there need not survive the scaling: zeros
        v[u].x = pc[u] < in_c ? v[u].x : 0.0f;
        v[u].y = pc[u] + 1 < in_c ? v[u].y : 0.0f;
        mx = fmaxf(mx, fmaxf(fabsf(v[u].x), fabsf(v[u].y)));
    }
    mx = wave_max(mx);
    // max|x| < 2^e; e kept where both s and 1 / s are normal numbers whatever the image holds
    const int e = __builtin_amdgcn_readfirstlane(min(max(__builtin_amdgcn_frexp_expf(mx), -100), 100));
    const float s = ldexpf(1.0f, 14 - e);
    WAVE_SYNC();                                                    // every lane has read its values: the rows may be overwritten
    char *const ib = (char *)img;
    const int lo_off = 2 * in_cp;
#pragma unroll
    for (int u = 0; u < T; ++u) {
        const float y0 = v[u].x * s, y1 = v[u].y * s;
        const _Float16 h0 = (_Float16)y0, h1 = (_Float16)y1;
        const float d0 = y0 - (float)h0, d1 = y1 - (float)h1;       // exact: the difference has at most 13 significant bits
        const v2h hi = { h0, h1 }, lo = { (_Float16)d0, (_Float16)d1 };
        *(v2h *)(ib + off[u]) = hi;
        *(v2h *)(ib + off[u] + lo_off) = lo;
    }
    WAVE_SYNC();
    return ldexpf(1.0f, e - 14);
}

// returns 1 / s
__device__ __forceinline__ float fast_split_image(float *__restrict__ img, int in_w, int in_c, int in_cp, int in_stride, int lane, unsigned ppr_inv20)
{
    const int ppr = in_cp >> 1, items = in_w * ppr;                 // i / ppr for i < 1024, ppr <= 32: the plan's reciprocal
    // the trip count is wave-uniform: a later block's small image takes a few trips, the first block's up to sixteen
    if (items <= 4 * KWS_WAVE) return fast_split_trips<4>(img, items, in_c, in_cp, in_stride, lane, ppr_inv20, ppr);
    if (items <= 8 * KWS_WAVE) return fast_split_trips<8>(img, items, in_c, in_cp, in_stride, lane, ppr_inv20, ppr);
    return fast_split_trips<KWS_FAST_HP>(img, items, in_c, in_cp, in_stride, lane, ppr_inv20, ppr);
}

// A small block whose weight fragments did not fit the workgroup's LDS block (one output-channel tile, at most eight k-steps): every
// fragment is requested from device memory BEFORE the image is split -- the round trips hide behind the conversion -- and the loop
// then only reads the image.
template <int MT>
__device__ __forceinline__ float fast_conv_small_h(const KwsFastBlock &k, float *__restrict__ in, float *__restrict__ stage, int sstride,
                                                   const float *__restrict__ shared, int zero_off, int lane, float *__restrict__ sink)
{
    constexpr int KSM = 8;
    const int lm = lane & 15, lq = lane >> 4, n_ks = k.h_ks;
    const kws_gptr bg = (kws_gptr)k.h_b_global + lane * 16;
    v8h bh[KSM], blo[KSM];
#pragma unroll
    for (int ks = 0; ks < KSM; ++ks) {
        const int kc = min(ks, n_ks - 1);
        bh[ks] = *(kws_gv8h)(bg + (kc * 2) * (KWS_WAVE * 16));
        blo[ks] = *(kws_gv8h)(bg + (kc * 2 + 1) * (KWS_WAVE * 16));
    }
    const float inv_s = fast_split_image(in, k.in_w, k.in_c, k.in_cp, k.in_stride, lane, k.inv_ppr20);
    v4f acc[MT][1];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt][0] = v4f{ 0.f, 0.f, 0.f, 0.f };
    const int rowb = k.in_stride * 4, in_w = k.in_w, lo_off = 2 * k.in_cp;
    const int row0 = lm - k.pad_left;
    const char *const abase = (const char *)in + row0 * rowb;
    const char *const zb = (const char *)(shared + zero_off);
    const int2 *const tab = (const int2 *)(shared + k.h_tab_off) + lq;
    // every operand of the image in one batch too (2 MT reads per k-step)
    v8h ah[KSM][MT], al[KSM][MT];
#pragma unroll
    for (int ks = 0; ks < KSM; ++ks) {
        const int2 d = tab[4 * min(ks, n_ks)];                            // row n_ks: zeros
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const bool in_img = (unsigned)(row0 + d.y + 16 * mt) < (unsigned)in_w;
            const char *p = in_img ? abase + d.x + mt * (16 * rowb) : zb;
            ah[ks][mt] = *(const v8h *)p;
            al[ks][mt] = *(const v8h *)(p + lo_off);
        }
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < KSM; ++ks) {
        if (ks < n_ks) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[ks][mt], bh[ks], acc[mt][0], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ks][mt], blo[ks], acc[mt][0], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ks][mt], bh[ks], acc[mt][0], 0, 0, 0);
        }
    }
    __builtin_amdgcn_s_setprio(0);
    const float scale = inv_s * k.h_inv_wscale;
    const float vout[2] = { 0.0f, 0.0f };
    fast_conv_finish<MT, 1>(k, acc, vout, stage, sstride, shared, lane, sink, scale);
    return scale;
}

// BG: the weight fragments are read from device memory (the workgroup's LDS block has no room for them) instead of LDS
// ROT (three-waves-per-SIMD build): fragments in LDS take the rotating-set loop of the BG form too.  The loop with two whole operand sets needs 128 registers
// for MT NT = 8 and spills an accumulator at 168 -- and because every tile shape is compiled into the one kernel, its pressure costs the kernel 19 more
// spilled registers around the block loop whichever shape runs (49x40 fp32, fragments from L2: 1.273 -> 1.254 ms same-box without it).  Where the fragments
// ARE in LDS the two-set loop is the faster one (49x13 fp32: 1.020 against 1.051 ms), so the choice follows the kernel instantiation: the 40-filter
// front end's plans (13.6 KB per wave: fragments of a 49-row block in L2) take ROT, the 32-filter one's (fragments in LDS) do not.
template <int MT, int NT, bool BG, bool ROT = false>
__device__ __forceinline__ float fast_conv_tiles_h(const KwsFastBlock &k, float *__restrict__ in, float *__restrict__ stage, int sstride,
                                                   const float *__restrict__ shared, int zero_off, int lane, float *__restrict__ sink)
{
    const int lm = lane & 15, lq = lane >> 4;
    const float inv_s = fast_split_image(in, k.in_w, k.in_c, k.in_cp, k.in_stride, lane, k.inv_ppr20);
    v4f acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = v4f{ 0.f, 0.f, 0.f, 0.f };
    const int rowb = k.in_stride * 4, in_w = k.in_w, lo_off = 2 * k.in_cp, n_ks = k.h_ks;
    const int row0 = lm - k.pad_left;                       // image row of this lane's operand for tile 0, tap 0
    const char *const zb = (const char *)(shared + zero_off);          // 16 bytes of zeros, and again lo_off bytes further
    const int2 *const tab = (const int2 *)(shared + k.h_tab_off) + lq;  // [k-step][lane / 16] { byte offset of the group, tap }
    const char *const bl = (const char *)(shared + (BG ? 0 : k.h_b_off)) + lane * 16;
    const kws_gptr bg = (kws_gptr)k.h_b_global + lane * 16;
    auto fetch = [&](int ks, const int2 &d, v8h (&ah)[MT], v8h (&al)[MT], v8h (&bh)[NT], v8h (&blo)[NT]) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int off = ((ks * 2) * NT + nt) * (KWS_WAVE * 16);
            if constexpr (BG) { bh[nt] = *(kws_gv8h)(bg + off); blo[nt] = *(kws_gv8h)(bg + off + NT * (KWS_WAVE * 16)); }
            else { bh[nt] = *(const v8h *)(bl + off); blo[nt] = *(const v8h *)(bl + off + NT * (KWS_WAVE * 16)); }
        }
        // (three waves per SIMD: the row tiles' 2 MT row numbers and addresses are formed here from two registers -- hoisted out of the loop as 2 MT
        // registers they are spilled, and a reload per tile waits inside the loop)
        int r0v = row0, abv = row0 * rowb;
        KWS_OPAQUE3(r0v); KWS_OPAQUE3(abv);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const bool in_img = (unsigned)(r0v + d.y + 16 * mt) < (unsigned)in_w;
            const char *p = in_img ? (const char *)in + (abv + d.x + mt * (16 * rowb)) : zb;
            ah[mt] = *(const v8h *)p;
            al[mt] = *(const v8h *)(p + lo_off);
        }
    };
    auto mfmas = [&](const v8h (&ah)[MT], const v8h (&al)[MT], const v8h (&bh)[NT], const v8h (&blo)[NT]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], blo[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
    };
#if KWS_FAST_WPS >= 3
    if constexpr (BG || (ROT && MT * NT >= 8)) {
        // Three waves per SIMD with the fragments in device memory (the LDS block holds eleven waves and no 36 KB of fragments): an L2 round trip is longer
        // than a k-step's 3 MT NT matrix instructions, so the fragments rotate through THREE register sets, requested two k-steps ahead, and the image
        // operands make do with ONE set refilled in two halves -- the lo halves after the products that read them, the hi halves after the rest -- :
        // 16 (1 + MT) + 16 * 3 NT... = 112 registers with the accumulators for MT = 4, NT = 2, where two whole sets of both took 128 and spilled an accumulator.
        v8h ah[MT], al[MT], bh[3][NT], blo[3][NT];
        auto fetch_b = [&](int ks, v8h (&h_)[NT], v8h (&l_)[NT]) {
            const int kc = ks < n_ks ? ks : n_ks - 1;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int off = ((kc * 2) * NT + nt) * (KWS_WAVE * 16);
                if constexpr (BG) { h_[nt] = *(kws_gv8h)(bg + off); l_[nt] = *(kws_gv8h)(bg + off + NT * (KWS_WAVE * 16)); }
                else { h_[nt] = *(const v8h *)(bl + off); l_[nt] = *(const v8h *)(bl + off + NT * (KWS_WAVE * 16)); }
            }
        };
        auto fetch_a = [&](const int2 &d, v8h (&dst)[MT], int plus) {
            int r0v = row0, abv = row0 * rowb;
            KWS_OPAQUE3(r0v); KWS_OPAQUE3(abv);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const bool in_img = (unsigned)(r0v + d.y + 16 * mt) < (unsigned)in_w;
                const char *p = in_img ? (const char *)in + (abv + d.x + mt * (16 * rowb)) : zb;
                dst[mt] = *(const v8h *)(p + plus);
            }
        };
        auto step = [&](int ks, const v8h (&bh_)[NT], const v8h (&bl_)[NT], v8h (&nh_)[NT], v8h (&nl_)[NT]) {
            fetch_b(ks + 2, nh_, nl_);
            const int2 dn = tab[4 * min(ks + 1, n_ks)];                  // (row n_ks reads zeros)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mt], bh_[nt], acc[mt][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            fetch_a(dn, al, lo_off);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bl_[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bh_[nt], acc[mt][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            fetch_a(dn, ah, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        fetch_b(0, bh[0], blo[0]);
        fetch_b(1, bh[1], blo[1]);
        { const int2 d = tab[0]; fetch_a(d, al, lo_off); fetch_a(d, ah, 0); }
        __builtin_amdgcn_s_setprio(1);
        // (whole trips of three: a k-step past the last one multiplies the zero block by the last fragments)
        for (int ks = 0; ks < n_ks; ks += 3) {
            step(ks, bh[0], blo[0], bh[2], blo[2]);
            step(ks + 1, bh[1], blo[1], bh[0], blo[0]);
            step(ks + 2, bh[2], blo[2], bh[1], blo[1]);
        }
        __builtin_amdgcn_s_setprio(0);
        const float scale = inv_s * k.h_inv_wscale;
        const float vout[2] = { 0.0f, 0.0f };
        fast_conv_finish<MT, NT>(k, acc, vout, stage, sstride, shared, lane, sink, scale);
        return scale;
    }
#endif
    // two operand sets, the loop unrolled by two: a set's operands are requested before the other set's matrix instructions are issued
    // (round 6: three sets with requests two steps ahead -- 48 more live registers -- measured 1.9 .. 3.9 % SLOWER same-box, profiles/r06_ab_variants.txt)
    v8h ah0[MT], al0[MT], bh0[NT], bl0[NT], ah1[MT], al1[MT], bh1[NT], bl1[NT];
    int2 d0 = tab[0], d1 = tab[4];
    fetch(0, d0, ah0, al0, bh0, bl0);
    __builtin_amdgcn_s_setprio(1);
    for (int ks = 0; ks < n_ks; ks += 2) {
        d0 = tab[4 * (ks + 2)];                                           // n_ks + 2 rows: the rows past the last one read zeros
        fetch(ks + 1 < n_ks ? ks + 1 : ks, d1, ah1, al1, bh1, bl1);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(ah0, al0, bh0, bl0);
        __builtin_amdgcn_sched_barrier(0);
        if (ks + 1 >= n_ks) break;                                        // odd step count
        d1 = tab[4 * (ks + 3)];
        fetch(ks + 2 < n_ks ? ks + 2 : ks, d0, ah0, al0, bh0, bl0);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(ah1, al1, bh1, bl1);
        __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
    const float scale = inv_s * k.h_inv_wscale;
    const float vout[2] = { 0.0f, 0.0f };
    fast_conv_finish<MT, NT>(k, acc, vout, stage, sstride, shared, lane, sink, scale);
    return scale;
}

// Pooled-in-flight blocks (KwsFastBlock::fpool): the window maxima of the raw accumulators sit in pm[pool_w][32]; bias, activation,
// ADD + activation, pooling clamp -- the reference's order per value (conv.h:88-93, add.h:200-212, pooling.h:231-233) -- and the
// zeroed k-padding channels of the next image.
__device__ __forceinline__ void fast_pool_finish(const KwsFastBlock &k, const float *__restrict__ pm, float *__restrict__ img,
                                                 const float *__restrict__ shared, int lane, int out_stride, int out_cp, float scale)
{
    const int items = k.pool_w * out_cp, out_c = k.out_c;
    const unsigned inv = k.inv_ocp20;
    const float cmin = k.conv_min, cmax = k.conv_max, amin = k.add_min, amax = k.add_max, pmin = k.pool_min, pmax = k.pool_max;
    const bool has_add = k.has_add != 0;
    for (int i = lane; i < items; i += KWS_WAVE) {
        const int p = (int)(((unsigned)i * inv) >> 20), c = i - p * out_cp, cc = min(c, out_c - 1);
        float v = pm[p * 32 + cc] * scale + shared[k.bias_off + cc];
        v = fminf(fmaxf(v, cmin), cmax);
        if (has_add) { v = v + shared[k.addc_off + cc]; v = fminf(fmaxf(v, amin), amax); }
        v = fminf(fmaxf(v, pmin), pmax);
        img[p * out_stride + c] = c < out_c ? v : 0.0f;
    }
}

// Block epilogue shared by every tile shape: either the k-padding channels of the next image are zeroed (the tiles were written
// straight into it) or MAX_POOL_2D over time runs from the staging image (pooling.h:189-237: windows clipped to the image, then
// the activation clamp).  Items = (pooled row, channel incl. the k-padding channels, which receive zeros).
__device__ __forceinline__ void fast_pool(const KwsFastBlock &k, const float *__restrict__ stage, float *__restrict__ img, int lane,
                                          int out_stride, int out_cp, bool pooled)
{
    const int items = k.pool_w * out_cp;
    const unsigned inv = k.inv_ocp20;                               // i / out_cp == (i * inv) >> 20 for i < 4096, out_cp <= 64
    const int sstride = k.stage_stride, out_w = k.out_w, out_c = k.out_c, pool = k.pool, pstr = k.pool_stride;
    const float pmin = k.pool_min, pmax = k.pool_max;
    if (!pooled) {
        // the tiles were written straight into the next image: only its k-padding channels (at most seven per row) are left
        const int npad = out_cp - out_c;
        if (npad == 0) return;
        const unsigned pinv = k.inv_npad20;
        for (int i = lane; i < k.pool_w * npad; i += KWS_WAVE) {
            const int p = (int)(((unsigned)i * pinv) >> 20), c = out_c + (i - p * npad);
            img[p * out_stride + c] = 0.0f;
        }
        return;
    }
    for (int i0 = 0; i0 < items; i0 += 2 * KWS_WAVE) {
        float v[2][8];
        int p[2], c[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = min(i0 + u * KWS_WAVE + lane, items - 1);
            p[u] = (int)(((unsigned)i * inv) >> 20);
            c[u] = i - p[u] * out_cp;
            const float *sp = stage + p[u] * pstr * sstride + min(c[u], out_c - 1);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                // a clipped or unused slot re-reads a row of the window: the maximum is unchanged
                const int r = (j < pool) ? min(p[u] * pstr + j, out_w - 1) : p[u] * pstr;
                v[u][j] = sp[(r - p[u] * pstr) * sstride];
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float m = v[u][0];
#pragma unroll
            for (int j = 1; j < 8; ++j) m = fmaxf(m, v[u][j]);
            m = fminf(fmaxf(m, pmin), pmax);
            if (i0 + u * KWS_WAVE + lane < items) img[p[u] * out_stride + c[u]] = c[u] < out_c ? m : 0.0f;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
//  One DEPTHWISE_CONV_2D block (reference/depthwiseconv_float.h:25: out[t][n] = sum_tap in[t + tap - pad_left][n / mult] * w[tap][n],
//  taps that fall into the SAME padding skipped) + bias + activation [+ ADD + activation] + MAX_POOL_2D over time (pooling.h:189-237:
//  windows clipped to the image) on the vector ALU, straight from the input image in LDS into the next block's image.  A few taps per
//  output: nothing to contract -- the pointwise 1x1 convolution that follows a depthwise block is the GEMM, and runs on the matrix cores
//  (fast_conv_tiles with taps = 1).  Items = (pooled row, channel incl. the next image's k-padding channels, which receive zeros);
//  consecutive lanes take consecutive channels of a row: every LDS access of a step is a contiguous run of words.
// ---------------------------------------------------------------------------------------------------------
//  fast_dw_t<TAPS>: a lane owns a SEGMENT of consecutive output rows of one channel -- a pooling window, or (no pooling) one of
//  64 / out_cp chunks of the time axis -- and slides a TAPS-deep register window along it: per row one LDS read (requested a row
//  ahead), TAPS multiply-adds on weights held in registers, the clamps, and a store or a running maximum.
template <int TAPS>
__device__ __forceinline__ void fast_dw_t(const KwsFastBlock &k, const float *__restrict__ in, float *__restrict__ img,
                                          const float *__restrict__ shared, int lane, int out_stride, int out_cp)
{
    const int out_c = k.out_c, in_w = k.in_w, out_w = k.out_w, pad_left = k.pad_left, in_stride = k.in_stride, mult = k.mult;
    const bool pooled = k.pool > 1 || k.pool_stride > 1;
    const int n_seg = pooled ? k.pool_w : k.dw_nseg;                // (max(1, 64 / out_cp), from the plan)
    const int seg_rows = pooled ? k.pool : k.dw_seg_rows, seg_step = pooled ? k.pool_stride : seg_rows;
    const int items = n_seg * out_cp;
    const unsigned inv = k.inv_ocp20;                               // i / out_cp for i < 4096, out_cp <= 64
    const float cmin = k.conv_min, cmax = k.conv_max, amin = k.add_min, amax = k.add_max, pmin = k.pool_min, pmax = k.pool_max;
    const bool has_add = k.has_add != 0;
    const float *wt = shared + k.w_off;
    for (int i = lane; i < items; i += KWS_WAVE) {
        const int seg = (int)(((unsigned)i * inv) >> 20), c = i - seg * out_cp, cc = min(c, out_c - 1);
        const float *col = in + (mult == 1 ? cc : cc / mult);
        float w[TAPS];
#pragma unroll
        for (int t = 0; t < TAPS; ++t) w[t] = wt[t * out_c + cc];
        const float bias = shared[k.bias_off + cc], addc = shared[k.addc_off + cc];
        const int r0 = seg * seg_step, r1 = min(r0 + seg_rows, out_w);
        auto fetch = [&](int row) { const float v = col[min(max(row, 0), in_w - 1) * in_stride]; return (unsigned)row < (unsigned)in_w ? v : 0.0f; };   // SAME padding
        // the window x[0 .. TAPS + 2]: TAPS - 1 rows of history + four new rows per trip, requested together (one LDS round trip per
        // four outputs; indices are compile-time constants after unrolling, so sliding the window costs no moves)
        float x[TAPS + 3];
#pragma unroll
        for (int t = 0; t + 1 < TAPS; ++t) x[t] = fetch(r0 + t - pad_left);
        float m = -FLT_MAX;
        float *dst = img + c;
        for (int r = r0; r < r1; r += 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) x[TAPS - 1 + e] = fetch(r + e + TAPS - 1 - pad_left);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float acc = 0.0f;
#pragma unroll
                for (int t = 0; t < TAPS; ++t) acc = __fmaf_rn(x[e + t], w[t], acc);
                float v = fminf(fmaxf(acc + bias, cmin), cmax);
                if (has_add) v = fminf(fmaxf(v + addc, amin), amax);
                const bool live = r + e < r1;
                if (pooled) m = live ? fmaxf(m, v) : m;
                else if (live) dst[(r + e) * out_stride] = c < out_c ? v : 0.0f;
            }
#pragma unroll
            for (int t = 0; t + 1 < TAPS; ++t) x[t] = x[t + 4];
        }
        if (pooled) dst[seg * out_stride] = c < out_c ? fminf(fmaxf(m, pmin), pmax) : 0.0f;
    }
}

// any tap count (more than eight taps: the general, unblocked form)
__device__ __forceinline__ void fast_dwconv_any(const KwsFastBlock &k, const float *__restrict__ in, float *__restrict__ img,
                                                const float *__restrict__ shared, int lane, int out_stride, int out_cp)
{
    const int items = k.pool_w * out_cp, out_c = k.out_c, in_w = k.in_w, out_w = k.out_w, taps = k.taps, pad_left = k.pad_left;
    const int in_stride = k.in_stride, pool = k.pool, pstr = k.pool_stride, mult = k.mult;
    const unsigned inv = k.inv_ocp20;                               // i / out_cp for i < 4096, out_cp <= 64
    const float cmin = k.conv_min, cmax = k.conv_max, amin = k.add_min, amax = k.add_max, pmin = k.pool_min, pmax = k.pool_max;
    const bool has_add = k.has_add != 0;
    const float *wt = shared + k.w_off;
    for (int i = lane; i < items; i += KWS_WAVE) {
        const int p = (int)(((unsigned)i * inv) >> 20), c = i - p * out_cp, cc = min(c, out_c - 1);
        const float *col = in + (mult == 1 ? cc : cc / mult);
        const float bias = shared[k.bias_off + cc], addc = shared[k.addc_off + cc];
        float m = -FLT_MAX;
        for (int j = 0; j < pool; ++j) {
            const int r = p * pstr + j;
            if (r >= out_w) break;                                  // ragged last window (SAME pooling): clipped as pooling.h clips it
            float acc = 0.0f;
            for (int tap = 0; tap < taps; ++tap) {
                const int row = r + tap - pad_left;
                if ((unsigned)row < (unsigned)in_w) acc = __fmaf_rn(col[row * in_stride], wt[tap * out_c + cc], acc);
            }
            float v = fminf(fmaxf(acc + bias, cmin), cmax);
            if (has_add) v = fminf(fmaxf(v + addc, amin), amax);
            m = fmaxf(m, v);
        }
        m = fminf(fmaxf(m, pmin), pmax);
        img[p * out_stride + c] = c < out_c ? m : 0.0f;
    }
}

__device__ __forceinline__ void fast_dwconv(const KwsFastBlock &k, const float *__restrict__ in, float *__restrict__ img,
                                            const float *__restrict__ shared, int lane, int out_stride, int out_cp)
{
    switch (k.taps) {
    case 1: fast_dw_t<1>(k, in, img, shared, lane, out_stride, out_cp); break;
    case 2: fast_dw_t<2>(k, in, img, shared, lane, out_stride, out_cp); break;
    case 3: fast_dw_t<3>(k, in, img, shared, lane, out_stride, out_cp); break;
    case 4: fast_dw_t<4>(k, in, img, shared, lane, out_stride, out_cp); break;
    case 5: fast_dw_t<5>(k, in, img, shared, lane, out_stride, out_cp); break;
    case 6: fast_dw_t<6>(k, in, img, shared, lane, out_stride, out_cp); break;
    case 7: fast_dw_t<7>(k, in, img, shared, lane, out_stride, out_cp); break;
    case 8: fast_dw_t<8>(k, in, img, shared, lane, out_stride, out_cp); break;
    default: fast_dwconv_any(k, in, img, shared, lane, out_stride, out_cp); break;
    }
}

// ---------------------------------------------------------------------------------------------------------
//  cmvnw (processing.hpp:326-389) in place over the cepstra image, O(1) per (row, column): the window of padded row r + 1 is
//  the window of r minus padded row r plus padded row r + win, so running sums of d = x - pivot and d * d (pivot = the column's
//  first row: the sums stay small, var = Q/n - (S/n)^2 does not cancel) replace two win-term walks.  A lane owns one column and
//  CR consecutive rows; the first window of a row group is sum_j cnt[g][j] d_j with the multiplicities tabulated by the host.
//  Three batches of loads per column block (the column, the update table, the rows the updates name); statistics and results
//  stay in registers until every lane has read what it needs, only then are the rows overwritten.  The features stay in the LDS
//  image: what has to leave the chip (feature matrix, int8 tensor) is written by the caller in one coalesced pass.
// ---------------------------------------------------------------------------------------------------------
//  ext_tab != NULL (win_size > 2 n_frames, the usual shapes): the first window counts every row m0 times and at most
//  KWS_FAST_CMVN_EXT rows more, so it is m0 x (the column's plain sums, gathered from the row groups' own rows with
//  ds_bpermute) + those few rows, instead of a walk over every row.
//  Returns this lane's share of the guard's variance estimate (kws_fast.h): sum over its windows of
//  ((abs_scale * (g.abs + g.lev * level) + g.rel * |mean|) / (deviation + eps))^2 -- the caller reduces it over the wave.
//  DEFER: the guard's terms are kept in registers and summed in the store loop, under the predicate the stores need anyway -- a select
//  of its own per window (compare, scalar and, conditional move: 17 x 2 per clip) costs 2.6 % of the whole kernel; the forms with the int8
//  network behind them sit at 256 registers and would spill the CR extra values (+7 % there): they sum in place.
template <int CR, int CG, bool DEFER>
__device__ __forceinline__ float fast_cmvn(float *__restrict__ img, const float *__restrict__ cnt_tab, const int *__restrict__ upd, int fs,
                                           float inv_win, const float *__restrict__ guard_tab, float level, float abs_scale, bool silent, float sys_t2, int piv_row, const float *__restrict__ mref, bool c0_exact, int lane, int nfr, int ncep,
                                           const float *__restrict__ ext_tab, float *__restrict__ sink)
{
    constexpr int NG = KWS_WAVE / CG;
    const int cgrp = min(lane / CG, NG - 1), cl = lane - (lane / CG) * CG;
    const bool lane_on = lane < NG * CG;
    const int r0 = cgrp * CR;
    const int nfr8 = (nfr + 7) & ~7;
    const float *cnt = cnt_tab + cgrp * nfr8;          // rows padded to a multiple of 8 with zeros
    float vacc = 0.0f;
    // column-independent table entries: the update table of this lane's rows, the first window's extra rows
    int u[CR - 1];                                    // leaving row offset | entering row offset << 16 (floats)
    int xo[KWS_FAST_CMVN_EXT];
    float we[KWS_FAST_CMVN_EXT], m0 = 0.0f;
#if KWS_FAST_WPS < 3
#pragma unroll
    for (int i = 0; i < CR - 1; ++i) u[i] = upd[min(r0 + i, nfr - 1)];
    if (ext_tab) {
        const float *ext = ext_tab + cgrp * (1 + 2 * KWS_FAST_CMVN_EXT);
        m0 = ext[0];
#pragma unroll
        for (int e = 0; e < KWS_FAST_CMVN_EXT; ++e) { xo[e] = __float_as_int(ext[1 + 2 * e]); we[e] = ext[2 + 2 * e]; }
    }
#endif
    for (int cb = 0; cb < ncep; cb += CG) {
#if KWS_FAST_WPS >= 3
        // (three waves per SIMD: the tables are re-read from LDS in every column block -- kept across the blocks, their ~20 registers are spilled and come back one by one)
        {
            int cg_ = cgrp;
            KWS_OPAQUE3(cg_);
            const int r0_ = cg_ * CR;
#pragma unroll
            for (int i = 0; i < CR - 1; ++i) u[i] = upd[min(r0_ + i, nfr - 1)];
            if (ext_tab) {
                const float *ext = ext_tab + cg_ * (1 + 2 * KWS_FAST_CMVN_EXT);
                m0 = ext[0];
#pragma unroll
                for (int e = 0; e < KWS_FAST_CMVN_EXT; ++e) { xo[e] = __float_as_int(ext[1 + 2 * e]); we[e] = ext[2 + 2 * e]; }
            }
        }
#endif
        const int c = cb + cl;
        const bool act = lane_on && c < ncep && r0 < nfr;
        float *col = img + min(c, ncep - 1);
        // every read of the column block goes out in one batch: the pivot, the lane's own rows, the rows the updates name.
        // The pivot (round 6): row piv_row of the column -- the clip's first digitally silent frame when it has one, else its first row.  With the
        // first row always, a clip that STARTS with a burst in otherwise silent audio had every d = x - pivot large, and var = Q/n - (S/n)^2
        // cancelled (pivot - mean)^2 / var digits away: a deviation off by 2e-6 relative, the largest logit errors of that family
        // (profiles/r06_guard_fit.txt).  A window shorter than the column (no ext_tab: every lane sums its own first window) takes the row in the
        // middle of the lane's group instead: a short window's mean follows the column's drift, and any clip-wide pivot leaves (pivot - mean)^2 /
        // var of 1e3 .. 1e4 where the window itself is quiet (a random configuration with win_size 15: a deviation off by 5e-4 relative).
        // (The median of three rows a third of the column apart does the same for any clip, for 1 % of the kernel's time: profiles/r06_ab_variants.txt.)
        const float piv = col[(ext_tab ? piv_row : min(r0 + CR / 2, nfr - 1)) * fs];
        const float4 gcol = ((const float4 *)guard_tab)[cb + cl];   // (absolute, per level, per |window mean|, its alternative); padded to a multiple of CG columns
        // column 0 (the log frame energy, |mean| ~ 10): when its deviation is small against its level, its window means have been summed
        // in the reference's own order (c0_exact) and the window-mean part of its guard does not apply.  The other columns: a clip with
        // digitally silent frames has runs of identical values in every column, and the reference's sequential window sums then round
        // systematically instead of randomly: the alternative coefficient is the larger one measured on such clips.
        const bool is_c0 = c0_exact && cb + cl == 0;
        // (level arrives multiplied by abs_scale = sqrt(live rows / rows): the rows of digitally silent frames carry the reference's own values)
        const float g_abs = __fmaf_rn(gcol.y, level, gcol.x * abs_scale);
        // the per-|window mean| coefficient: gcol.w where the reference's sequential window sums round SYSTEMATICALLY (runs of identical or nearly identical
        // values: a clip with silent frames; a near-constant column, below), gcol.z where they round at random.  Column 0: gcol.w = its means replayed.
        float g_rel = (cb + cl == 0 ? c0_exact : silent) ? gcol.w : gcol.z;
        float mr[CR];
#pragma unroll
        for (int i = 0; i < CR; ++i) mr[i] = (c0_exact && cb == 0) ? mref[min(r0 + i, nfr - 1)] : 0.0f;
        float own[CR];
#pragma unroll
        for (int i = 0; i < CR; ++i) own[i] = col[min(r0 + i, nfr - 1) * fs];
        float dl[CR - 1], da[CR - 1];
#pragma unroll
        for (int i = 0; i < CR - 1; ++i) { dl[i] = col[u[i] & 0xffff]; da[i] = col[(unsigned)u[i] >> 16]; }
        // first window of the row group: sum_j cnt[j] d_j, eight rows per batch, two partial sums
        float S0 = 0.0f, S1 = 0.0f, Q0 = 0.0f, Q1 = 0.0f;
        if (ext_tab) {
            float xe[KWS_FAST_CMVN_EXT];
#pragma unroll
            for (int e = 0; e < KWS_FAST_CMVN_EXT; ++e) xe[e] = col[xo[e]];
            float T0 = 0.0f, T1 = 0.0f, U0 = 0.0f, U1 = 0.0f;          // this lane's rows
#pragma unroll
            for (int i = 0; i < CR; ++i) {
                const float d = (lane_on && r0 + i < nfr) ? own[i] - piv : 0.0f;
                if (i & 1) { T1 += d; U1 = __fmaf_rn(d, d, U1); } else { T0 += d; U0 = __fmaf_rn(d, d, U0); }
            }
            float T = T0 + T1, U = U0 + U1;
            float Tt = T, Ut = U;
#pragma unroll
            for (int g = 1; g < NG; ++g) {                               // the other row groups' share of the column
                const int src = (cl + CG * ((cgrp + g) % NG)) << 2;
                Tt += __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(T)));
                Ut += __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(U)));
            }
            S0 = m0 * Tt; Q0 = m0 * Ut;
#pragma unroll
            for (int e = 0; e < KWS_FAST_CMVN_EXT; ++e) {
                const float d = xe[e] - piv, wd = we[e] * d;
                S1 += wd;
                Q1 = __fmaf_rn(wd, d, Q1);
            }
        } else
        for (int j0 = 0; j0 < nfr8; j0 += 8) {
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = col[min(j0 + e, nfr - 1) * fs];
            const float4 wa = *(const float4 *)(cnt + j0), wb = *(const float4 *)(cnt + j0 + 4);
            const float w[8] = { wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w };
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const float d0 = x[e] - piv, d1 = x[e + 1] - piv;
                const float wd0 = w[e] * d0, wd1 = w[e + 1] * d1;
                S0 += wd0; S1 += wd1;
                Q0 = __fmaf_rn(wd0, d0, Q0); Q1 = __fmaf_rn(wd1, d1, Q1);
            }
        }
#pragma unroll
        for (int i = 0; i < CR - 1; ++i) { dl[i] -= piv; da[i] -= piv; }
        float S = S0 + S1, Q = Q0 + Q1;
        // A near-constant column -- deviation below sys_t x |mean| in the lane's first window (every window of the usual shapes holds every row, so one look
        // per lane and column block will do; round 6, profiles/r06_guard_fit.txt: 0.6e-6 .. 1.0e-6 |mean| there against 0.15e-6 .. 0.3e-6) -- takes gcol.w too.
        // float32 graphs only (sys_t2 is 0 for an int8 graph, whose guard keeps rounds 4 - 5's constants instead, kws_fast_plan.cpp): the forms with the
        // int8 network behind them sit at 256 registers, and the test cost them 3 - 5 % (profiles/r06_ab_variants.txt).
        if constexpr (DEFER) {
            const float m0 = S * inv_win, v0 = fmaxf(__fmaf_rn(-m0, m0, Q * inv_win), 0.0f), am0 = m0 + piv;
            g_rel = (cb + cl != 0 && v0 < sys_t2 * (am0 * am0)) ? gcol.w : g_rel;
        }
        float o[CR], gq[DEFER ? CR : 1];
#pragma unroll
        for (int i = 0; i < CR; ++i) {
            const float m = S * inv_win;
            float var = __fmaf_rn(-m, m, Q * inv_win);
            var = fmaxf(var, 0.0f);
            const float sd = __builtin_amdgcn_sqrtf(var);
            const float rstd = __builtin_amdgcn_rcpf(sd + FLT_EPSILON);
            const float gterm = __fmaf_rn(g_rel, fabsf(m + piv), g_abs) * rstd;       // the guard's term of this window
            if constexpr (DEFER) gq[i] = gterm;
            else vacc = (act && r0 + i < nfr) ? __fmaf_rn(gterm, gterm, vacc) : vacc;
            o[i] = (is_c0 ? own[i] - mr[i] : (own[i] - piv) - m) * rstd;
            if (i + 1 < CR) {
                S = (S + da[i]) - dl[i];
                Q = __fmaf_rn(da[i], da[i], Q);
                Q = __fmaf_rn(-dl[i], dl[i], Q);
            }
        }
        WAVE_SYNC();                                  // every lane's reads of this column block are done
#pragma unroll
        for (int i = 0; i < CR; ++i) {
            const int r = r0 + i;
            const bool live = act && r < nfr;
            *(live ? col + r * fs : sink) = o[i];                  // no branch per value: rows / columns outside the matrix go to the sink
            if constexpr (DEFER) vacc = live ? __fmaf_rn(gq[i], gq[i], vacc) : vacc;
        }
    }
    WAVE_SYNC();
    return vacc;
}

// PROF: development aid -- shader-clock totals per phase of wave 0 of workgroup 0 (tools/gpu_fast_phase_profile.py)
#define KWS_FAST_NPHASE 20       // 0..8 phases, 9..11 detail of block 0, 12 + b: block b >= 1 on its own
#define FPH(i) do { if (PROF) { const long long now_ = clock64(); ph[i] += now_ - tlast; tlast = now_; } } while (0)

// ---------------------------------------------------------------------------------------------------------
// NZ: taps of mel filters 0..31 kept in registers (filters 32..39, when there are 40, keep KWS_FAST_NZ2); DG: DCT k-groups = filters / 8
// FROM_CEP: the windows arrive as cepstra before cmvnw [n_frames][n_cepstral] (continuous mode: the rolling buffers of kws_streams_*,
// ring-indexed per KwsDspPlan::ring_*) instead of PCM: the kernel starts at cmvnw.  feat_in (run-time, FROM_CEP forms): what arrives is
// extract_mfcc_features' matrix itself -- the exact kernels' bits --: cmvnw and its guard are skipped and only the network runs (the
// guard that is left is the network's own arithmetic, KwsFastPlan::v_net_feat).
// NET: the float32 network follows in the same launch (no feature / int8 outputs); !NET: the features / the int8 tensor leave for
// HBM and no network code is compiled in.  Two instantiations instead of run-time flags: each form's cmvnw stores are written
// for what it does (with a branch per value only where a global store hangs on it).
// MFE: the front end of the MFE block (extract_mfe_features of the newer SDK copy): speechpy::feature::mfe on the raw signal
// (the plan's pre-emphasis coefficient is 0) -- the kernel stops after the mel filterbank and its zero handling (no log, no DCT, no
// cmvnw: the block's own normalisation follows in kws_mfe_norm_kernel) and writes the [frames][filters] matrix to HBM.
// QCP: 16 / 64 = the int8 two-block network on the matrix cores follows in the same launch (with !NET): the quantised input tensor is
// written as the 16- / 64-byte activation rows of nn_mfma_clip (kws_nn_int8_dev.h) instead of -- or besides -- going to HBM, and the
// network is bit-exact from that tensor on, as in kws_nn_mfma_kernel; 0: no int8 network code.
template <int NZ, int DG, bool PROF = false, bool FROM_CEP = false, bool NET = true, int QCP = 0, bool MFE = false>
__global__ __launch_bounds__(256 * KWS_FAST_WPS, KWS_FAST_WPS) void kws_fast_kernel(KwsDspPlan P, const KwsFastPlan *__restrict__ FPp, const int16_t *__restrict__ pcm, int n_clips,
                                                          float *__restrict__ scores, float *__restrict__ features,
                                                          int8_t *__restrict__ q_out, float in_scale, int in_zp,
                                                          int *__restrict__ flag_count, int *__restrict__ flag_list,
                                                          long long *__restrict__ prof_out = nullptr, const float *__restrict__ cep = nullptr,
                                                          const KwsNnPlan *__restrict__ QNp = nullptr, const int *__restrict__ sel = nullptr,
                                                          float *__restrict__ tap_logits = nullptr, int feat_in = 0)
{
    static_assert(QCP == 0 || !NET, "the int8 network follows the feature-emitting form");
    static_assert(!MFE || (!NET && QCP == 0 && !FROM_CEP && !PROF), "the MFE form is the spectral prefix: mel energies to HBM");
    // the plan is read from memory (scalar loads, any block index); by value in the kernel arguments the compiler copies it to
    // scratch to index its blocks
    const KwsFastPlan &FP = *FPp;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & (KWS_WAVE - 1), wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));    // uniform: per-wave addresses stay in scalar registers
    // a workgroup none of whose waves has a clip (the usually empty list of the second tier, a short list) leaves before it stages
    // its 50 KB of tables
    if (FROM_CEP && (int)blockIdx.x * (int)(blockDim.x >> 6) >= sel_count(sel, n_clips)) return;
    float *shared = lds;
    float *F = lds + FP.shared_floats + FP.q_floats + wave * FP.wave_floats;       // log-mel -> cepstra -> features (block 0's input image)
    float *R1 = F + FP.f_floats;                                      // FFT buffers + power rows; later block 1's input image
    float *xw = R1, *pw = R1;                                         // the FFT's exchange buffer; the power rows reuse it
    for (int i = threadIdx.x; i < FP.shared_floats; i += blockDim.x) shared[i] = FP.shared_init[i];
    // the power rows are padded so that every filter can read its full tap count: the padding is only ever multiplied by zero
    // weights, but it must be finite (LDS is not cleared between kernels)
    for (int i = lane; i < FP.r1_floats; i += KWS_WAVE) R1[i] = 0.0f;
    // the int8 network's tables, shared by the workgroup, behind the float block: ADD + ReLU look-up tables, FULLY_CONNECTED weights and
    // softmax tables (nn_head_stage), weight fragments and requantisation words (nn_mfma_stage_lds)
    // (layout of that block: 32 x 256 + 16 x 256 table bytes, the head's tables, 16-byte weight fragments [KS1 + 4][64], 6 x 64 words;
    // its addresses are re-derived where they are used -- kept across the clip loop they would cost registers in every phase)
    constexpr int Q_HEAD = 48 * 256, Q_WB1 = Q_HEAD + ((KWS_HEAD_BYTES + 15) & ~15), Q_WB2 = Q_WB1 + (QCP == 16 ? 4 : 16) * KWS_WAVE * 16,
                  Q_RQ = Q_WB2 + 4 * KWS_WAVE * 16;
    if constexpr (QCP != 0) {
        unsigned char *const qs = (unsigned char *)(lds + FP.shared_floats);
        const KwsNnPlan &QN = *QNp;
        for (int i = threadIdx.x * 4; i < QN.blk[0].out_c * 256; i += blockDim.x * 4) *(int *)(qs + i) = *(const int *)(QN.blk[0].add_lut + i);
        for (int i = threadIdx.x * 4; i < QN.blk[1].out_c * 256; i += blockDim.x * 4) *(int *)(qs + 32 * 256 + i) = *(const int *)(QN.blk[1].add_lut + i);
        (void)nn_head_stage(QN, qs + Q_HEAD);
        nn_mfma_stage_lds<QCP == 0 ? 16 : QCP>(QN, (v4i *)(qs + Q_WB1), (v4i *)(qs + Q_WB2), (int *)(qs + Q_RQ));
    }
    // from cepstra nothing fills the images' channel padding before the first convolution reads it (times zero weights)
    if constexpr (FROM_CEP)
        for (int i = lane; i < FP.wave_floats; i += KWS_WAVE) F[i] = 0.0f;
    __syncthreads();

    const int nfr = P.n_frames, ncep = P.n_cepstral, NF = 8 * DG;
    // a remainder of one or two frames (the 49th of the standard window) would cost a whole eight-frame pass: it gets a tail pass
    // with 32 lanes per frame instead (four points per lane, three exchanges -- round 1's layout of the same butterflies)
    const int n_tail = (nfr >= KWS_FAST_MEL_CHUNK && (nfr & 7) != 0 && (nfr & 7) <= 2) ? (nfr & 7) : 0;
    const int n_pass = n_tail ? nfr / KWS_FAST_MEL_CHUNK : (nfr + KWS_FAST_MEL_CHUNK - 1) / KWS_FAST_MEL_CHUNK;
    const int fs = FP.fs;
    float *img = F;                                                   // [n_frames][fs]
    float *elog = F + nfr * fs;                                       // log frame energies, parked until the DCT has run
    constexpr int NZ2 = DG > 4 ? KWS_FAST_NZ2 : 1;
    const float *dct_frag = shared + FP.dct_off;
    const int pstride = FP.pstride;
    const int n_waves = blockDim.x >> 6;
    // 1 / fft_length, the int16 scale 2^-15 squared and the split's two halvings: powers of two (the plan checks fft_length)
    const float pre_cof = P.pre_cof, pscale = P.inv_fft * (1.0f / 1073741824.0f) * 0.25f;
    const int frame_stride = P.frame_stride, n_samples = P.n_samples;
    const float *cnt_tab = shared + FP.cnt_off;
    const int *upd_tab = (const int *)(shared + FP.upd_off);
    const float *ext_tab = FP.ext_off >= 0 ? shared + FP.ext_off : nullptr;
    const float inv_win = FP.inv_win;
    // cepstra that arrive from HBM are the exact kernels' (continuous mode's slices, the second tier of a batch call): the DCT term
    // of the guard does not apply to them (KwsFastPlan::guard_cep_off)
    const float *guard_tab = shared + (FROM_CEP ? FP.guard_cep_off : FP.guard_off);
    const int *pad_idx = (const int *)(shared + FP.pad_off);
    const int win_size = P.win_size, prow = nfr + 2 * P.pad;
    const int cr = FP.cr, n_blocks = FP.n_blocks, n_labels = FP.n_labels;
    long long ph[KWS_FAST_NPHASE] = { 0 }, tlast = PROF ? clock64() : 0;

    // A window of 8 k + 1 frames leaves one frame for the tail pass, which has room for two: the tail pass of every other clip also
    // transforms the last frame of the wave's NEXT clip and parks its log-mel row and log energy in the stash (per wave, behind the sink).
    const int clip_stride = gridDim.x * n_waves;
    float *const stash = F + FP.stash_off;                             // [0 .. NF): log-mel row, [47]: log frame energy
    bool have_stash = false;
    // sel: optional clip-selection list (sel[0] = count, sel[1 + i] = clip) -- the clips an earlier launch handed back
    // (only the forms that start from cepstra are launched over a list: the PCM forms see every clip, and their paired tail pass needs
    // no list look-ups)
    const int n_sel = FROM_CEP ? sel_count(sel, n_clips) : n_clips;
#if KWS_FAST_WPS >= 3
    // Three waves per SIMD, clips by TICKET: a workgroup of eleven waves leaves one SIMD with two, whose waves run faster -- with the static split below they
    // finish early and the launch waits for the SIMDs that hold three (vector pipe 80 % occupied where twelve waves reach 89 %, profiles/r06_occupancy.md).
    // A wave's first clip is its own number; every further one is drawn from a counter in device memory, two clips ahead (the paired tail pass and the
    // prefetches want the NEXT clip when a clip starts): the draw's round trip ends long before its value is looked at.  The counter cleans up after
    // itself -- a wave that leaves counts itself out, and the last one to do so zeroes both words (below the loop) --, so a launch needs nothing from the
    // host: no argument that changes from launch to launch, no memset.  Launches of a handle are serial (its flag
    // lists are too).
    int *const tk = FP.tickets;
    int tv = 0;                                                       // lane 0: the ticket drawn last
    int ni = 0;
    // (a draw only while the clip it would follow exists: when the wave leaves the loop, every draw it made has been waited for)
    auto draw = [&]() { if (ni < n_sel && lane == 0) tv = __hip_atomic_fetch_add(tk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    draw();
    ni = clip_stride + __builtin_amdgcn_readfirstlane(tv);
    draw();
    for (int ci = blockIdx.x * n_waves + wave; ci < n_sel; ci = ni, ni = ni < n_sel ? clip_stride + __builtin_amdgcn_readfirstlane(tv) : ni, draw()) {
        const int clip = FROM_CEP ? sel_clip(sel, ci) : ci;
        const int next_ci = ni;
        const int next_clip = ni < n_sel ? ni : -1;                                    // the wave's next clip (paired tail pass; PCM forms only)
#else
    for (int ci = blockIdx.x * n_waves + wave; ci < n_sel; ci += clip_stride) {
        const int clip = FROM_CEP ? sel_clip(sel, ci) : ci;
        const int next_ci = ci + clip_stride;
        const int next_clip = ci + clip_stride < n_sel ? ci + clip_stride : -1;        // the wave's next clip (paired tail pass; PCM forms only)
#endif
        // ---- per-lane constants of the spectral phase (the FFT is kws_mfcc_kernel's: KissFFT's order, bit for bit).  They are
        //      re-derived per clip from a lane index the compiler cannot see through: hoisted out of the clip loop, the FFT's
        //      twiddles and the three dozen LDS addresses of the pair loop stay live through the DCT, cmvnw and convolution phases
        //      and push those into scratch; re-deriving them costs ~60 L2-resident loads per clip.
        int lane_c = lane;
        asm volatile("" : "+v"(lane_c));
        // Eight lanes own a frame (eight frames per pass), a lane owns sixteen of its 128 complex points: kf_bfly2 (m = 1) and
        // kf_bfly4 (m = 2) run on two blocks of eight output positions in registers, ONE exchange through LDS re-deals the
        // points so that kf_bfly4 m = 8 and m = 32 run in registers too (kiss_fft.cpp:15-84, 232-296; every butterfly keeps the
        // reference's operation order, only its lane changes).
        const int fl = lane_c & 7, fg = lane_c >> 3;
        // blocks fl (output positions 8 fl ..) and fl + 8: block j = 4 i1 + i2 reads input points i1 + 4 i2 + 16 i3 + 64 i4
        const int nbA = (fl >> 2) + 4 * (fl & 3);
        const cf a1 = to_cf(P.tw[16]), a2 = to_cf(P.tw[32]), a3 = to_cf(P.tw[48]);
        const cf b1 = to_cf(P.tw[4 * fl]), b2 = to_cf(P.tw[8 * fl]), b3 = to_cf(P.tw[12 * fl]);
#if KWS_FAST_WPS >= 3
        // (three waves per SIMD: the last level's twiddles and the split's are read from the workgroup's LDS table in every pass, KwsFastPlan::twl_off)
        const float *const twl_c = shared + FP.twl_off + 24 * fl, *const twl_s = shared + FP.twl_off + 192 + 16 * fl;
#else
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
#endif
        const int xwr = fg * KWS_FAST_XS + 18 * fl;                     // exchange buffer: position p of a frame at 2 p + 2 (p / 8)
        const int xrd = fg * KWS_FAST_XS + 2 * fl;
        const int partner = (lane_c & ~7) | ((8 - fl) & 7);
        const int half = lane_c >> 5, t = lane_c & 31;
        // a mel filter's taps are consecutive bins: first bin (as an offset into a frame's power row) + NZ weights, zero beyond its end
        const int start1 = FP.tap_start1[lane_c], start2 = FP.tap_start2[lane_c];
        float lvl_sum = 0.0f;                                            // the guard's level: this lane's share of sum_frames |sum_filters log-mel| / sqrt(NF)
        float w1[NZ], w2[NZ2];
    #pragma unroll
        for (int n = 0; n < NZ; ++n) w1[n] = FP.tap_w1[lane_c * KWS_FAST_NZ_MAX + n];
    #pragma unroll
        for (int n = 0; n < NZ2; ++n) w2[n] = FP.tap_w2[lane_c * KWS_FAST_NZ2 + n];
        int touched_a = 0, touched_b = 0;                                   // (FROM_CEP) the next window's cache lines, see below
        if constexpr (FROM_CEP) {
            // the window's matrix in batches of requests (round 5: a loop of load -> store pairs exposed a round trip per trip); windows of up to
            // 52 x 40 values: 33 per lane
            const float *src = cep + (size_t)clip * (nfr * ncep);
            const int n_val = nfr * ncep;
            const unsigned inv = (1u << 20) / (unsigned)ncep + 1u;           // i / ncep for i < 4096
            // (a rolled loop, four requests per trip: the fully unrolled forms -- 33 or 3 x 11 words per lane -- cost this kernel 150 - 180 spilled registers)
            const int rr = P.ring_rows, rh = P.ring_head;
            auto ring = [&](int r) { const int t = r + rh; return (rr != 0 && r < rr) ? (t >= rr ? t - rr : t) : r; };     // ring_in_row without its division
            auto trips = [&](int i_from, int i_to) {
                for (int i0 = i_from + lane; i0 - lane < i_to; i0 += 4 * KWS_WAVE) {
                    float v[4];
                    int dst[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int i = min(i0 + KWS_WAVE * u, n_val - 1);
                        const int r = (int)(((unsigned)i * inv) >> 20), c = i - r * ncep;
                        v[u] = src[ring(r) * ncep + c];
                        dst[u] = r * fs + c;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) if (i0 + KWS_WAVE * u < n_val) img[dst[u]] = v[u];
                }
            };
            trips(0, 4 * KWS_WAVE);
            // ... and the NEXT window of this wave: one touch per 64-byte line (two per lane cover 8 KB), so that its requests find their lines
            // in the cache a window later; the values are only handed to an empty asm statement at the end of this window
            if (next_ci < n_sel) {
                const char *nsrc = (const char *)(cep + (size_t)sel_clip(sel, next_ci) * n_val);
                const int nbytes = n_val * 4 - 4;
                touched_a = *(const int *)(nsrc + (min(128 * lane, nbytes) & ~3));
                touched_b = *(const int *)(nsrc + (min(128 * lane + 64, nbytes) & ~3));
            }
            trips(4 * KWS_WAVE, n_val);
            WAVE_SYNC();
        } else {
        const int16_t *xbase = pcm + (size_t)clip * n_samples;
        const int wrap_prev = (int)xbase[n_samples - 1];                 // x[-1] is the window's last sample (processing.hpp:68, 104-106)
        // a point's four samples x[2n - 2 .. 2n + 1] of frame f, requested one pass ahead
        auto fetch = [&](int q, fast_i2 (&raw)[2][8]) {
            const int f = min(KWS_FAST_MEL_CHUNK * q + fg, nfr - 1);
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
        // the frames of the pass after that: one 64-byte segment per lane, requested a pass before the real requests so that
        // those find their lines in the cache (a pass is shorter than an HBM round trip under load); the value is only handed to
        // an empty asm statement a pass later, which is what keeps its register reserved until the data has arrived
        auto touch = [&](int q) {
            const int f = min(KWS_FAST_MEL_CHUNK * q + fg, nfr - 1);
            return *(const int *)(xbase + (f * frame_stride + 32 * fl));
        };
        // ---- mel filterbank for the frames of a pass: dot_by_row as a register-tap gather, zero handling, log.  pairs = pairs of
        //      frame slots a lane half walks (2: slots 4 h .. 4 h + 3 of an eight-frame pass; 1: the tail pass, slots 0, 1)
        // adj1: added to the image offset of frame slot 1's stores (the tail pass parks the NEXT clip's last frame in the stash)
        auto mel_phase = [&](int fbase, int nfc, int pairs, int adj1) {
            WAVE_SYNC();
            // filters 0..31: lane half h takes frame slots 4 h .. 4 h + 3; filters 32..39 (40 filters only): one frame slot each
            const int j2 = 32 + (lane_c & 7), sl2 = lane_c >> 3;
            const float *p1 = pw + 4 * half * pstride + start1, *p2 = pw + sl2 * pstride + start2;
            float macc[5] = { 1.0f, 1.0f, 1.0f, 1.0f, 1.0f };
#pragma unroll
            for (int s2 = 0; s2 < 4; s2 += 2) {
                if (s2 >= 2 * pairs) break;
                float xv[2][NZ];
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int n = 0; n < NZ; ++n) xv[s][n] = p1[(s2 + s) * pstride + n];
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    float acc = 0.0f;
#pragma unroll
                    for (int n = 0; n < NZ; ++n) acc = __fmaf_rn(xv[s][n], w1[n], acc);
                    macc[s2 + s] = acc;
                }
            }
            float acc2 = 0.0f;
            if (DG > 4) {
                float xv2[NZ2];
#pragma unroll
                for (int n = 0; n < NZ2; ++n) xv2[n] = p2[n];
#pragma unroll
                for (int n = 0; n < NZ2; ++n) acc2 = __fmaf_rn(xv2[n], w2[n], acc2);
            }
            macc[4] = acc2;
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                const float a = macc[s] == 0.0f ? FLT_EPSILON : macc[s];                               // functions.hpp:63-69
                macc[s] = MFE ? a : fast_log(a);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int slot = 4 * half + s;
                if (slot < nfc && t < NF) img[(fbase + slot) * fs + t + (slot == 1 ? adj1 : 0)] = macc[s];
            }
            if (DG > 4 && sl2 < nfc && j2 < NF) img[(fbase + sl2) * fs + j2 + (sl2 == 1 ? adj1 : 0)] = macc[4];
        };
        fast_i2 nxt[2][8];
        fetch(0, nxt);
        int touched = touch(1);
        const bool pair_tail = n_tail == 1 && !have_stash && next_clip >= 0;
        // the next clip's last frame: 256 samples = 512 bytes, warmed now so that the tail pass finds them in the cache
        int touched_tail = 0;
        const int16_t *const xnext = pcm + (size_t)max(next_clip, 0) * n_samples;
        if (pair_tail && lane_c < 8) touched_tail = *(const int *)(xnext + (nfr - 1) * frame_stride + 32 * lane_c);
        for (int q = 0; q < n_pass; ++q) {
            const int fbase = KWS_FAST_MEL_CHUNK * q;
            const int f = fbase + fg;
            const bool live = f < nfr;
            cf u[4][4];                                                  // after the exchange: u[a][b] = position fl + 8 a + 32 b
            {
                cf z[2][8];
#pragma unroll
                for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                    for (int i = 0; i < 8; ++i) z[blk][i] = fast_point(nxt[blk][i], pre_cof);
                if (f == 0 && fl == 0) {                                 // the clip's first sample: its predecessor wraps
                    fast_i2 v = nxt[0][0];
                    v.y = v.x;
                    v.x = wrap_prev << 16;
                    z[0][0] = fast_point(v, pre_cof);
                }
                // unconditional (the frame index is clamped): a conditional request makes the compiler copy all sixteen register
                // pairs around the branch
                fetch(q + 1, nxt);
                asm volatile("" : : "v"(touched));
                touched = touch(q + 2);
                FPH(0);
                // kf_bfly2 (m = 1, twiddle 1) on the (i4 = 0, 1) pairs, then kf_bfly4 (m = 2) on the sums (k = 0) and the
                // differences (k = 1): outputs 8 j + k + 2 i
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    cf sm[4], df[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) { sm[i] = cadd(z[blk][2 * i], z[blk][2 * i + 1]); df[i] = csub(z[blk][2 * i], z[blk][2 * i + 1]); }
                    bfly4_unit(sm[0], sm[1], sm[2], sm[3]);
                    bfly4(df[0], df[1], df[2], df[3], a1, a2, a3);
#pragma unroll
                    for (int i = 0; i < 4; ++i) { z[blk][2 * i] = sm[i]; z[blk][2 * i + 1] = df[i]; }
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
#if KWS_FAST_WPS >= 3
            for (int a = 0; a < 4; ++a) {
                const float4 c12 = *(const float4 *)(twl_c + 6 * a);
                const float2 c3v = *(const float2 *)(twl_c + 6 * a + 4);
                cf t1, t2, t3;
                t1.r = c12.x; t1.i = c12.y; t2.r = c12.z; t2.i = c12.w; t3.r = c3v.x; t3.i = c3v.y;
                bfly4(u[a][0], u[a][1], u[a][2], u[a][3], t1, t2, t3);
            }
            cf stw[8];
#pragma unroll
            for (int q = 0; q < 8; q += 2) {
                const float4 v = *(const float4 *)(twl_s + 2 * q);
                stw[q].r = v.x; stw[q].i = v.y; stw[q + 1].r = v.z; stw[q + 1].i = v.w;
            }
#else
            for (int a = 0; a < 4; ++a) bfly4(u[a][0], u[a][1], u[a][2], u[a][3], c1[a], c2[a], c3[a]);
#endif
            FPH(1);
            // ---- kiss_fftr split (kiss_fftr.cpp:84-119) and the power spectrum, fp32: |X|^2 / fft_length.  Bin pair (k, 128 - k)
            //      needs positions k and 128 - k: the second lives in lane (8 - fl) % 8 at (3 - a, 3 - b) -- in lane 0 itself, one
            //      position further -- so the lanes swap their upper halves.
            {
                float *prow = pw + fg * pstride;                     // bins 0 .. 127 (a row of a dead frame slot is written too, never read)
                const bool lane0 = fl == 0;
                float esum = 0.0f;
#pragma unroll
                for (int qq = 0; qq < 8; ++qq) {
                    const int a = qq & 3, b = qq >> 2;
                    cf other;
                    // (round 6: all sixteen requests of the loop issued before the first use -- sixteen more live registers -- measured 0.3 .. 1.3 % SLOWER
                    // same-box, profiles/r06_ab_variants.txt)
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
                    cf lo, hi;                                       // twice the reference's: the halving is part of pscale
                    lo.r = f1k.r + twv.r;
                    lo.i = f1k.i + twv.i;
                    hi.r = f1k.r - twv.r;
                    hi.i = twv.i - f1k.i;
                    const float plo = __fmaf_rn(lo.r, lo.r, lo.i * lo.i) * pscale;
                    const float phi = __fmaf_rn(hi.r, hi.r, hi.i * hi.i) * pscale;
                    // bin 64 is written twice by the reference and the second store wins: same order here; its first value is not
                    // part of the frame energy
                    prow[k] = plo;
                    esum += (qq == 0 && lane0) ? 0.0f : plo;
                    esum += phi;
                    prow[KWS_NC - k] = phi;
                }
                if (fl == 0) {                                       // tmp[0]: DC and Nyquist bins (kiss_fftr.cpp:84-96)
                    const float dc = u[0][0].r + u[0][0].i, ny = u[0][0].r - u[0][0].i;
#if KWS_FAST_WPS >= 3
                    // (4 pscale would live in a vector register through the pass loop and come back from scratch here; both factors are powers of two: the same bits)
                    const float pdc = ((dc * dc) * pscale) * 4.0f, pny = ((ny * ny) * pscale) * 4.0f;
#else
                    const float pdc = (dc * dc) * (4.0f * pscale), pny = (ny * ny) * (4.0f * pscale);
#endif
                    esum += pdc + pny;
                    prow[0] = pdc;
                }
                // frame energy (feature.hpp:289-298): its log is parked until the DCT has run
                esum = oct_sum(esum);
                if (fl == 0 && live) elog[f] = fast_log(esum == 0.0f ? FLT_EPSILON : esum);
            }
            FPH(2);

            mel_phase(fbase, min(KWS_FAST_MEL_CHUNK, nfr - fbase), 2, 0);
            WAVE_SYNC();
            FPH(3);
        }

        asm volatile("" : : "v"(touched_tail));
        if (n_tail == 1 && have_stash) {
            // this clip's last frame was transformed by the previous clip's tail pass
            if (lane_c < NF) img[(nfr - 1) * fs + lane_c] = stash[lane_c];
            if (lane_c == 47) elog[nfr - 1] = stash[47];
            have_stash = false;
            WAVE_SYNC();
        } else if (n_tail) {
            // ---- tail pass: frames 8 n_pass + h on lane half h -- or, when one frame is left and the wave has another clip to come, this
            //      clip's last frame on half 0 and the next clip's on half 1 --, a lane transforms four of its frame's 128 points per
            //      stage: kf_bfly2 (m = 1) fused with kf_bfly4 (m = 2), then kf_bfly4 m = 8 and m = 32, each through an in-place,
            //      padded buffer behind the two power rows it feeds
            const int16_t *const xb_t = (pair_tail && half == 1) ? xnext : xbase;
            const int ft = pair_tail ? nfr - 1 : KWS_FAST_MEL_CHUNK * n_pass + half;
            const bool live_t = ft < nfr;
            const int s0 = min(ft, nfr - 1) * frame_stride + 8 * t;
            const int4 rawv = *(const int4 *)(xb_t + s0);
            const int rawp = s0 == 0 ? wrap_prev : (int)xb_t[s0 - 1];       // (s0 = 0 needs a window of one frame: no tail pass then)
            const int k01 = t & 1, g01 = t >> 1, n0 = (g01 >> 2) + 4 * (g01 & 3), K2 = t & 7, G2 = t >> 3;
            const cf ta1 = to_cf(P.tw[16 * k01]), ta2 = to_cf(P.tw[32 * k01]), ta3 = to_cf(P.tw[48 * k01]);
            const cf tb1 = to_cf(P.tw[4 * K2]), tb2 = to_cf(P.tw[8 * K2]), tb3 = to_cf(P.tw[12 * K2]);
            const cf tc1 = to_cf(P.tw[t]), tc2 = to_cf(P.tw[2 * t]), tc3 = to_cf(P.tw[3 * t]);
            float *zb = R1 + 2 * pstride + half * KWS_ZF;
            {
                float y[8];
                float prev = (float)rawp;                              // unscaled, like the eight-frame passes (see pscale)
                const int w[4] = { rawv.x, rawv.y, rawv.z, rawv.w };
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float lo = (float)(short)(w[i] & 0xffff), hi = (float)(w[i] >> 16);
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
                float *prow = pw + half * pstride;
                float esum = 0.0f;
#pragma unroll
                for (int rep = 0; rep < 2; ++rep) {
                    const int k = t + 1 + 32 * rep;
                    const cf stw_ = rep ? st2 : st1;
                    cf fpnk; fpnk.r = fq[rep].r; fpnk.i = -fq[rep].i;
                    const cf f1k = cadd(fpk[rep], fpnk), f2k = csub(fpk[rep], fpnk);
                    const cf twv = cmul(f2k, stw_);
                    cf lo, hi;                                       // twice the reference's: the halving is part of pscale
                    lo.r = f1k.r + twv.r;
                    lo.i = f1k.i + twv.i;
                    hi.r = f1k.r - twv.r;
                    hi.i = twv.i - f1k.i;
                    const float plo = __fmaf_rn(lo.r, lo.r, lo.i * lo.i) * pscale;
                    const float phi = __fmaf_rn(hi.r, hi.r, hi.i * hi.i) * pscale;
                    if (k != KWS_NC / 2) {                           // bin 64 is written twice by the reference: the second store wins
                        esum += plo;
                        prow[k] = plo;
                    }
                    esum += phi;
                    prow[KWS_NC - k] = phi;
                }
                if (t == 0) {
                    const float dc = d0.x + d0.y, ny = d0.x - d0.y;
                    const float pdc = (dc * dc) * (4.0f * pscale), pny = (ny * ny) * (4.0f * pscale);
                    esum += pdc + pny;
                    prow[0] = pdc;
                }
                esum = half_wave_sum(esum);
                if (t == 0 && live_t) *((pair_tail && half == 1) ? stash + 47 : elog + ft) = fast_log(esum == 0.0f ? FLT_EPSILON : esum);
            }
            // frame slot 1 of a paired pass belongs to the next clip: its row goes to the stash instead of image row n_frames
            mel_phase(KWS_FAST_MEL_CHUNK * n_pass, pair_tail ? 2 : n_tail, 1, pair_tail ? (int)(stash - img) - nfr * fs : 0);
            have_stash = pair_tail;
            WAVE_SYNC();
        }

        // ---- DCT-II (numpy.hpp:378-401) as [frames x NF] x [NF x NF/2+1] on the matrix cores, in place: two rounds of two
        //      16-frame tiles.  The transform's operand fragments are re-read per clip (L2-resident, 4 DG values per lane): kept in
        //      registers across the clip loop they push the FFT's constants into scratch.
        if constexpr (!MFE) {
            int lane_l = lane;
            asm volatile("" : "+v"(lane_l));                 // not loop-invariant as far as the compiler can tell
            const int lm = lane_l & 15, lq = lane_l >> 4;
            float dB[DG][2][2];
#pragma unroll
            for (int g = 0; g < DG; ++g)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) dB[g][i][nt] = dct_frag[((g * 2 + i) * 2 + nt) * KWS_WAVE + lane_l];
            const float *arow = img + lm * fs + 2 * lq;
            float e0 = 0.0f;
            if (lane_l < nfr) e0 = elog[lane_l];
            v4f acc[4][2];
            // raised wave priority for the transform's 16 DG MFMAs, as for the convolution's contraction loop (fast_conv_tiles)
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int rnd = 0; rnd < 2; ++rnd) {
                float2 a[2][DG];
#pragma unroll
                for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                    for (int g = 0; g < DG; ++g) a[m2][g] = *(const float2 *)(arow + 16 * (2 * rnd + m2) * fs + 8 * g);
#pragma unroll
                for (int m2 = 0; m2 < 2; ++m2) { acc[2 * rnd + m2][0] = v4f{ 0.f, 0.f, 0.f, 0.f }; acc[2 * rnd + m2][1] = v4f{ 0.f, 0.f, 0.f, 0.f }; }
#pragma unroll
                for (int g = 0; g < DG; ++g) {
#pragma unroll
                    for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
                            acc[2 * rnd + m2][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m2][g].x, dB[g][0][nt], acc[2 * rnd + m2][nt], 0, 0, 0);
#pragma unroll
                    for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
                            acc[2 * rnd + m2][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m2][g].y, dB[g][1][nt], acc[2 * rnd + m2][nt], 0, 0, 0);
                }
            }
            __builtin_amdgcn_s_setprio(0);
            WAVE_SYNC();                                     // every operand of the transform is in a register
            // Coefficients above NF/2 are never written by the reference's transform: they keep the log-mel input, doubled and
            // scaled (fast-dct-fft.cpp:71-74, numpy.hpp:392-397).  cmvnw normalises every column by its own mean and deviation,
            // so a constant factor on a column does not reach the features (it only rescales the epsilon added to the
            // deviation, 1.2e-7): those columns are left as they are.  c0 <- log(frame energy) (feature.hpp:425-429).
            // Stores without a branch per value: rows past the last frame and coefficients that are not this tile's go to a
            // per-lane sink.
            float *const sink = KWS_FAST_SINK + lane_l;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int n = 16 * nt + lm;
                    const bool col_ok = n <= NF / 2 && n > 0;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = 16 * mt + 4 * lq + i;
                        *((col_ok && r < nfr) ? img + r * fs + n : sink) = acc[mt][nt][i];
                    }
                }
            if (lane_l < nfr) img[lane_l * fs] = e0;
            // The transform's coefficient 0 -- sum_k log-mel[r][k] / sqrt(NF), replaced by the log frame energy above -- is the frame's mean
            // log-mel level x sqrt(NF): what the fp32 rounding of the spectral phase scales with (the guard's `level`, kws_fast.h).  The
            // lanes of column 0 (lm == 0) hold it for sixteen rows each.
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int i = 0; i < 4; ++i) lvl_sum += (16 * mt + 4 * lq + i < nfr) ? fabsf(acc[mt][0][i]) : 0.0f;      // (the row predicate of the stores above)
            lvl_sum = lm == 0 ? lvl_sum : 0.0f;
            WAVE_SYNC();
        }
        FPH(4);
        }   // !FROM_CEP

        // ---- cmvnw + optional outputs (extract_mfcc_features' matrix, the int8 input tensor) --------------------------------
        float *fout = features ? features + (size_t)clip * (nfr * ncep) : nullptr;
        int8_t *qclip = q_out ? q_out + (size_t)clip * (nfr * ncep) : nullptr;
        int8_t *const act1 = (int8_t *)R1;                              // [KWS_A1_ROWS][QCP]: row = time + tap, padding = the input zero point
        float vlane = 0.0f;
        int lane_m = lane;
        asm volatile("" : "+v"(lane_m));
        float *const csink = KWS_FAST_SINK + lane_m;
        // ---- column 0's window means in the reference's own order (processing.hpp:326-389 over numpy::mean_axis0, numpy.hpp:746-784:
        //      a sequential fp32 sum of win_size padded rows, then a division).  The log frame energy sits near -10 for quiet audio, so
        //      that sum rounds at ~6e-5 per step and the mean carries ~1e-6 |mean| of rounding noise of the reference's own making:
        //      for a stationary background (deviation of the log energy ~0.05) that would be the largest error of the whole feature
        //      matrix -- unless the sum is simply replayed.  One column only: a lane per row, win_size additions over a padded copy of
        //      the column in the dead exchange buffer (conflict-free: consecutive lanes read consecutive words).
        // Only when it matters: every window holds every row at least c0_mult times (plan), so a window's deviation is at least
        // sqrt(c0_mult n_frames / win_size) x the column's plain deviation over the n_frames rows; if that already clears the relative
        // guard at the column's largest magnitude, the running-sum mean is good enough for every window (error kappa |mean| / deviation
        // below the feature tolerance) and the pass is skipped -- the usual case for audio whose loudness changes over the second.
        bool c0_exact = false, silent = false;
        int n_silent = 0;                                               // frames whose rows were replaced by the reference's silent row (wave-uniform)
        int piv_row = 0;                                                // cmvnw's pivot row (fast_cmvn)
        const bool feats_given = FROM_CEP && feat_in != 0;              // wave-uniform
        if constexpr (!MFE) if (!feats_given) {
            const bool on = lane_m < nfr;
            const float x0 = img[min(lane_m, nfr - 1) * fs];
            // a frame without any energy: zero handling put FLT_EPSILON there, and its log (bit-identical in both tiers: the reference's own
            // polynomial) sits in column 0
            const unsigned long long smask = __ballot(on && x0 == fast_log(FLT_EPSILON));      // bit r: frame r is silent (lane r looked at row r)
            silent = smask != 0ull;
            piv_row = silent ? __builtin_ctzll(smask) : 0;
            if constexpr (!FROM_CEP) {
                // Those frames' DCT outputs are the reference's own (KwsFastPlan::sil_off): the transform of a constant row is a handful of
                // rounding residues that the reference and the matrix cores do not share, and a column that is otherwise quiet would be decided
                // by them.  A rare path (no clip of a noise-floored recording has such a frame): kept out of the way of the others.
                if (silent && FP.sil_off >= 0) {
                    // (a scalar walk over the silent frames, one store of NF/2 lanes per frame: one vector register)
                    const int half_nf = min(NF / 2, ncep - 1);
                    const float sv = shared[FP.sil_off + min(lane_m + 1, 31)];
                    float *const dst = img + 1 + lane_m;
                    for (unsigned long long mm = smask; mm != 0ull; mm &= mm - 1ull) {
                        const int r = __builtin_ctzll(mm);
                        if (lane_m < half_nf) dst[r * fs] = sv;
                    }
                    n_silent = __popcll(smask);
                    WAVE_SYNC();
                }
            }
            const float mu = wave_sum(on ? x0 : 0.0f) * FP.c0_inv_rows;
            const float d0 = on ? x0 - mu : 0.0f;
            const float sd0 = __builtin_amdgcn_sqrtf(wave_sum(d0 * d0) * FP.c0_inv_rows);
            const float top = wave_max(on ? fabsf(x0) : 0.0f);
            // ... and always for a clip with digitally silent frames: their runs of identical log energies make the reference's sequential window
            // sums round the same way add after add (0.45e-6 .. 0.75e-6 |mean| measured), which no running sum reproduces.  (A larger coefficient
            // for such clips instead of the replay -- 3.4 % of the kernel's time on word-then-silence input -- cost every other input 0.7 % (float
            // form) to 4.6 % (int8 forms) in the register allocation of cmvnw: profiles/r06_ab_variants.txt.)
            c0_exact = silent || !(FP.c0_factor * sd0 >= __fmaf_rn(FP.c0_rel, top, FP.c0_abs));
        }
        if (c0_exact) {
            // four copies of the padded column, copy s shifted by s rows: lane r's window pad[r ..] then starts at a 16-byte aligned slot
            // of copy r & 3 and arrives as ds_read_b128s, four of them (16 terms) requested before the first addition of a trip.  No
            // predicate per term: whole trips, then whole 16-byte groups, then the last win_size % 4 terms.
            constexpr int PS = 260;                                          // floats per copy: up to 4 x 64 slots + the reads past the window
            float *pc = R1;
            for (int p = lane_m; p < prow; p += KWS_WAVE) {                  // a rare path: kept small rather than batched (registers)
                int prw[4];
                float pvv[4];
#pragma unroll
                for (int sft = 0; sft < 4; ++sft) prw[sft] = pad_idx[min(p + sft, prow - 1)];
#pragma unroll
                for (int sft = 0; sft < 4; ++sft) pvv[sft] = img[prw[sft] * fs];
#pragma unroll
                for (int sft = 0; sft < 4; ++sft) pc[sft * PS + p] = pvv[sft];
            }
            WAVE_SYNC();
            const int r = min(lane_m, nfr - 1);
            const float4 *pv = (const float4 *)(pc + (r & 3) * PS + (r & ~3));
            float sum = 0.0f;
            int q = 0;
            const int nq = win_size >> 2;
            for (; q + 4 <= nq; q += 4) {
                float4 v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = pv[q + e];
#pragma unroll
                for (int e = 0; e < 4; ++e) { sum += v[e].x; sum += v[e].y; sum += v[e].z; sum += v[e].w; }
            }
            for (; q < nq; ++q) { const float4 v = pv[q]; sum += v.x; sum += v.y; sum += v.z; sum += v.w; }
            {
                const float4 v = pv[q];                                      // the slots past the window hold later rows (finite), not used
                const int rem = win_size & 3;
                if (rem > 0) sum += v.x;
                if (rem > 1) sum += v.y;
                if (rem > 2) sum += v.z;
            }
            if (lane_m < nfr) elog[lane_m] = __fdiv_rn(sum, (float)win_size);
            WAVE_SYNC();
        }
        if (PROF) { const long long now_ = clock64(); ph[19] += now_ - tlast; }
        if constexpr (QCP != 0) {
            // the exchange buffer is dead from here to the next clip: it becomes the network's first activation image
            const int z1 = (int)((unsigned)(QNp->blk[0].in_zp & 0xff) * 0x01010101u);
            for (int i = lane_m; i < KWS_A1_ROWS * (QCP / 4); i += KWS_WAVE) ((int *)act1)[i] = z1;
            WAVE_SYNC();
        }
        // the guard (kws_fast.h): V = variance estimate of a logit difference's error; the clip stays iff V max(c1 P^2, c2) <= 1
        float gV = 0.0f;
        if constexpr (!MFE) {                             // (MFE: nothing is divided by a deviation: the mel energies leave as they are)
          if (feats_given) gV = FP.v_net_feat;
          else {
            // the level of the LIVE rows (a silent row's own level is |log FLT_EPSILON|, and it carries no spectral error), and the share of
            // the rows that carry one: the absolute / per-level terms are an rms over a column's rows
            float level = FROM_CEP ? 0.0f : wave_sum(lvl_sum) * FP.lvl_inv, abs_scale = 1.0f;
            if (!FROM_CEP && n_silent > 0) {
                const float live = (float)(nfr - n_silent), rows = (float)nfr;
                level = live > 0.0f ? fmaxf(__fmaf_rn(level, rows, (float)n_silent * fast_log(FLT_EPSILON)), 0.0f) / live : 0.0f;     // (log FLT_EPSILON < 0)
                abs_scale = __builtin_amdgcn_sqrtf(live / rows);
                level *= abs_scale;
            }
            if (KWS_FAST_WPS >= 3 || cr == 13) vlane = fast_cmvn<13, 16, QCP == 0>(img, cnt_tab, upd_tab, fs, inv_win, guard_tab, level, abs_scale, silent, FP.sys_t2, piv_row, elog, c0_exact, lane_m, nfr, ncep, ext_tab, csink);
            else vlane = fast_cmvn<17, 20, QCP == 0>(img, cnt_tab, upd_tab, fs, inv_win, guard_tab, level, abs_scale, silent, FP.sys_t2, piv_row, elog, c0_exact, lane_m, nfr, ncep, ext_tab, csink);
            gV = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(wave_sum(vlane) + FP.v_net)));      // wave-uniform: a scalar register
          }
        }
        if constexpr (!NET) {
            // the scores are another kernel's (or an int8 network's): P = 1/4, the largest p (1 - p) there is.  !(x <= 1): a NaN hands the clip on
            if (!MFE && !(gV * fmaxf(FP.g_c1 * 0.0625f, FP.g_c2) <= 1.0f)) {
                if (lane == 0) flag_list[atomicAdd(flag_count, 1)] = clip;
            }
        }
        if constexpr (!NET) {
            // ---- what leaves the chip, and the network's quantised input: one pass over the feature image, consecutive lanes take
            //      consecutive values of extract_mfcc_features' [frame][coefficient] order (coalesced stores)
            const unsigned inv = (1u << 20) / (unsigned)ncep + 1u;           // i / ncep for i < 4096
            const int q_pad = QCP != 0 ? QNp->blk[0].pad_left : 0;
            for (int i = lane; i < nfr * ncep; i += KWS_WAVE) {
                const int r = (int)(((unsigned)i * inv) >> 20), c = i - r * ncep;
                const float o = img[r * fs + c];
                if (fout) fout[i] = o;
                if (QCP != 0 || qclip) {
                    const int8_t qb = quantize_feature(o, in_scale, in_zp);
                    if (qclip) qclip[i] = qb;
                    if constexpr (QCP != 0) act1[(r + q_pad) * QCP + c] = qb;
                }
            }
            WAVE_SYNC();
        }
        FPH(5);
        if constexpr (QCP != 0) {
            // ---- the int8 graph (two CONV_2D blocks on v_mfma_i32_*_i8, FULLY_CONNECTED, SOFTMAX): the image is dead, its first words
            //      hold the second activation image and the head's vectors
            const KwsNnPlan &QN = *QNp;
            int8_t *const act2 = (int8_t *)F;                           // [KWS_A2_ROWS][32]
            int *const vec = (int *)(F + KWS_A2_ROWS * 8);              // 64 + 16 ints: FULLY_CONNECTED input and logits
            const int z2 = (int)((unsigned)(QN.blk[1].in_zp & 0xff) * 0x01010101u);
            for (int i = lane; i < KWS_A2_ROWS * 8; i += KWS_WAVE) ((int *)act2)[i] = z2;
            WAVE_SYNC();
            unsigned char *const qs = (unsigned char *)(lds + FP.shared_floats);
            const NnMfmaLds<QCP> qctx = nn_mfma_lds_ctx<QCP>(QN, (const v4i *)(qs + Q_WB1), (const v4i *)(qs + Q_WB2), (const int *)(qs + Q_RQ), lane);
            const NnHeadTab q_head = nn_head_tab(qs + Q_HEAD);
            const NnTaps no_taps = { nullptr, 0, nullptr, nullptr, nullptr, nullptr };
            __builtin_amdgcn_s_setprio(1);                   // the matrix-core sections of the int8 network go first, like the float contraction loop
            nn_mfma_clip<QCP>(qctx, QN, q_head, act1, act2, vec, (const int8_t *)qs, (const int8_t *)qs + 32 * 256, lane, clip, scores, no_taps);
            __builtin_amdgcn_s_setprio(0);
            WAVE_SYNC();
        }
        if constexpr (FROM_CEP) asm volatile("" : : "v"(touched_a), "v"(touched_b));
        if constexpr (!NET) continue;
        else {

        // ---- the float32 graph: CONV_2D blocks ping-pong between the two images, then FULLY_CONNECTED and SOFTMAX ----------
        float *cur = F, *oth = R1;
        float *const wsink = KWS_FAST_SINK + lane;
        int lane_n = lane;
        asm volatile("" : "+v"(lane_n));
        for (int b = 0; b < n_blocks; ++b) {
            const KwsFastBlock &k = FP.blk[b];
            int lane_b = lane_n;
            KWS_OPAQUE3(lane_b);
            const bool last = b + 1 == n_blocks;
            const int o_stride = last ? k.out_c : FP.blk[b + 1].in_stride;
            const int o_cp = last ? k.out_c : FP.blk[b + 1].in_cp;
            const bool pooled = k.pool > 1 || k.pool_stride > 1;
            // un-pooled: straight into the next image; pooled: staged in this block's own (dead) input image
            float *stage = pooled ? cur : oth;
            const int sstride = pooled ? k.stage_stride : o_stride;
            long long t_loop = 0, t_pre = 0, *tl = (PROF && b == 0) ? &t_loop : nullptr, *tp = (PROF && b == 0) ? &t_pre : nullptr;
            if (k.dw) {
                fast_dwconv(k, cur, oth, shared, lane_b, o_stride, o_cp);
                WAVE_SYNC();
                float *tmp = cur; cur = oth; oth = tmp;
                if (PROF && b > 0) { const long long now_ = clock64(); ph[12 + b] += now_ - tlast; }
                FPH(6 + (b > 0));
                continue;
            }
            float cscale = 1.0f;                                         // what the accumulators still carry (split operands: two powers of two)
            if (k.hconv) {
                // split operands on v_mfma_f32_16x16x32_f16: row tiles 1 / 2 / 4 (an idle tile costs 16 cycles per instruction there)
                const int zo = FP.zero_off;
                const bool bgl = k.h_b_off < 0;
                if (bgl && k.n_tiles == 1 && k.m_tiles == 1 && k.h_ks <= 8) cscale = fast_conv_small_h<1>(k, cur, stage, sstride, shared, zo, lane_b, wsink);
                else
                switch ((k.m_tiles > 2 ? 4 : k.m_tiles) * 4 + k.n_tiles) {
                case 4 * 4 + 2: cscale = bgl ? fast_conv_tiles_h<4, 2, true>(k, cur, stage, sstride, shared, zo, lane_b, wsink) : fast_conv_tiles_h<4, 2, false, (DG > 4)>(k, cur, stage, sstride, shared, zo, lane_b, wsink); break;
                case 4 * 4 + 1: cscale = bgl ? fast_conv_tiles_h<4, 1, true>(k, cur, stage, sstride, shared, zo, lane_b, wsink) : fast_conv_tiles_h<4, 1, false>(k, cur, stage, sstride, shared, zo, lane_b, wsink); break;
                case 2 * 4 + 2: cscale = bgl ? fast_conv_tiles_h<2, 2, true>(k, cur, stage, sstride, shared, zo, lane_b, wsink) : fast_conv_tiles_h<2, 2, false>(k, cur, stage, sstride, shared, zo, lane_b, wsink); break;
                case 2 * 4 + 1: cscale = bgl ? fast_conv_tiles_h<2, 1, true>(k, cur, stage, sstride, shared, zo, lane_b, wsink) : fast_conv_tiles_h<2, 1, false>(k, cur, stage, sstride, shared, zo, lane_b, wsink); break;
                case 1 * 4 + 2: cscale = bgl ? fast_conv_tiles_h<1, 2, true>(k, cur, stage, sstride, shared, zo, lane_b, wsink) : fast_conv_tiles_h<1, 2, false>(k, cur, stage, sstride, shared, zo, lane_b, wsink); break;
                default: cscale = bgl ? fast_conv_tiles_h<1, 1, true>(k, cur, stage, sstride, shared, zo, lane_b, wsink) : fast_conv_tiles_h<1, 1, false>(k, cur, stage, sstride, shared, zo, lane_b, wsink); break;
                }
            } else
            switch (k.m_tiles * 4 + k.n_tiles) {
            case 4 * 4 + 2: fast_conv_tiles<4, 2>(k, cur, stage, sstride, shared, lane_b, wsink, tl, tp); break;
            case 4 * 4 + 1: fast_conv_tiles<4, 1>(k, cur, stage, sstride, shared, lane_b, wsink); break;
            case 3 * 4 + 2: fast_conv_tiles<3, 2>(k, cur, stage, sstride, shared, lane_b, wsink, tl, tp); break;
            case 3 * 4 + 1: fast_conv_tiles<3, 1>(k, cur, stage, sstride, shared, lane_b, wsink); break;
            case 2 * 4 + 2: fast_conv_tiles<2, 2>(k, cur, stage, sstride, shared, lane_b, wsink); break;
            case 2 * 4 + 1: fast_conv_tiles<2, 1>(k, cur, stage, sstride, shared, lane_b, wsink); break;
            case 1 * 4 + 2: fast_conv_tiles<1, 2>(k, cur, stage, sstride, shared, lane_b, wsink); break;
            default: fast_conv_tiles<1, 1>(k, cur, stage, sstride, shared, lane_b, wsink); break;
            }
            WAVE_SYNC();
            // (block 0's sub-phases: only the fp32-instruction form takes these clocks; the split-operand form's are tools/gpu_fast_subphase.py's)
            if (PROF && b == 0 && t_loop != 0) { ph[9] += t_loop - t_pre; ph[10] += clock64() - t_loop; ph[11] += t_pre - tlast; }
            if (k.fpool) fast_pool_finish(k, stage, oth, shared, lane_b, o_stride, o_cp, cscale);
            else fast_pool(k, stage, oth, lane_b, o_stride, o_cp, pooled);
            WAVE_SYNC();
            float *tmp = cur; cur = oth; oth = tmp;
            if (PROF && b > 0) { const long long now_ = clock64(); ph[12 + b] += now_ - tlast; }
            FPH(6 + (b > 0));
        }
        {
            // FULLY_CONNECTED (fully_connected.h:26-60): an output unit is served by an aligned group of S = 64 / 2^ceil(log2 fc_out)
            // lanes, each summing every S-th product (a lane per unit walked fc_in dependent LDS round trips with four lanes busy);
            // SOFTMAX (softmax.h:31-63)
            const int fc_in = FP.fc_in, fc_out = FP.fc_out;
            const int sh = fc_out > 1 ? 32 - __builtin_clz(fc_out - 1) : 0;
            const int S = KWS_WAVE >> sh, unit = lane_n >> (6 - sh), sl = lane_n & (S - 1), uc = min(unit, fc_out - 1);
            const float *wfc = shared + FP.fc_w_off + uc * fc_in;
            float tot = 0.0f, tot1 = 0.0f;
            int i = sl;
            for (; i + S < fc_in; i += 2 * S) { tot = __fmaf_rn(cur[i], wfc[i], tot); tot1 = __fmaf_rn(cur[i + S], wfc[i + S], tot1); }
            if (i < fc_in) tot = __fmaf_rn(cur[i], wfc[i], tot);
            tot = group_sum(tot + tot1, S) + shared[FP.fc_b_off + uc];
            tot = fminf(fmaxf(tot, FP.fc_min), FP.fc_max);
            const bool on = unit < fc_out;
            if (tap_logits && on && sl == 0) tap_logits[(size_t)clip * n_labels + unit] = tot;       // kws_set_logits_tap
            const float mx = wave_max(on ? tot : -FLT_MAX);
            const float e = expf((tot - mx) * FP.beta);
            const float sum = wave_sum(on && sl == 0 ? e : 0.0f);
            const float pr = e / sum;
            if (on && sl == 0) scores[(size_t)clip * n_labels + unit] = pr;
            // the guard, a posteriori: |d score_i| <= p_i (1 - p_i) max_j |d(z_i - z_j)|, so the clip's own scores say how much of the
            // logit error's k sigma reaches a score (a saturated softmax passes next to nothing).  !(x <= 1): a NaN hands the clip on
            const float pq = wave_max(on && sl == 0 ? pr * (1.0f - pr) : 0.0f);
            if (!(gV * fmaxf(FP.g_c1 * pq * pq, FP.g_c2) <= 1.0f)) {
                if (lane_n == 0) flag_list[atomicAdd(flag_count, 1)] = clip;
            }
        }
        WAVE_SYNC();
        FPH(8);
        }   // NET
    }
#if KWS_FAST_WPS >= 3
    // the ticket counter's clean-up: every draw of this wave has landed (its value was waited for: the clip loop), so the wave counts itself out; the last
    // wave of the launch finds every other wave's count -- hence every draw -- behind it and zeroes both words for the next launch
    if (lane == 0) {
        // (relaxed: the counters are only ever touched by atomics, which meet in one place, and each wave has waited for its own; an acquire / release
        // pair at agent scope writes back and invalidates the caches under the waves that are still running: +5 % on the whole launch)
        const int gone = __hip_atomic_fetch_add(FP.tickets + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gone == (int)(gridDim.x * (blockDim.x >> 6)) - 1) {
            __hip_atomic_store(FP.tickets, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(FP.tickets + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
#endif
    if (PROF && blockIdx.x == 0 && threadIdx.x == 0 && prof_out)
        for (int i = 0; i < KWS_FAST_NPHASE; ++i) prof_out[i] = ph[i];
}

#ifndef KWS_FAST_NO_LAUNCHERS      // (tools/fast_one_form.sh: a scratch translation unit that instantiates ONE form of the kernel, for register / spill experiments)
// ---------------------------------------------------------------------------------------------------------
//  launchers (called from kws_api.cpp)
// ---------------------------------------------------------------------------------------------------------
#if KWS_FAST_WPS < 3
// the same launchers of kws_fast.hip's second compilation (three waves per SIMD, float32-network forms; kws_fast.h): reached through the ones above
int kws_launch_fast_w3(const KwsDspPlan &P, const KwsFastPlan &FP, const KwsFastPlan *d_plan, const int16_t *pcm, int n_clips, float *scores, float *features,
                       int8_t *q_out, float in_scale, int in_zp, int *flag_count, int *flag_list, int n_cu, hipStream_t stream, const KwsNnPlan *d_nn, float *tap_logits);
int kws_launch_fast_prof_w3(const KwsDspPlan &P, const KwsFastPlan &FP, const KwsFastPlan *d_plan, const int16_t *pcm, int n_clips, float *scores,
                            int *flag_count, int *flag_list, int n_cu, long long *prof_out, hipStream_t stream);
#endif
template <int NZ, int DG, bool PROF, bool FROM_CEP = false, bool NET = true, int QCP = 0, bool MFE = false>
static int launch_fast_t(const KwsDspPlan &P, const KwsFastPlan &FP, const KwsFastPlan *d_plan, const int16_t *pcm, int n_clips, float *scores,
                         float *features, int8_t *q_out, float in_scale, int in_zp, int *flag_count, int *flag_list, int n_cu,
                         long long *prof_out, hipStream_t stream, const float *cep = nullptr, const KwsNnPlan *d_nn = nullptr, const int *sel = nullptr,
                         float *tap_logits = nullptr, int feat_in = 0)
{
    const size_t smem = ((size_t)FP.shared_floats + FP.q_floats + (size_t)FP.n_waves * FP.wave_floats) * sizeof(float);
    // the opt-in for more than 64 KB of dynamic LDS is per device (and this instantiation): one bit per device, set once
    static std::atomic<unsigned long long> attr_done{ 0 };
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return (int)hipGetLastError();
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_done.load(std::memory_order_acquire) & bit)) {
        if (hipFuncSetAttribute((const void *)kws_fast_kernel<NZ, DG, PROF, FROM_CEP, NET, QCP, MFE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return (int)hipGetLastError();
        attr_done.fetch_or(bit, std::memory_order_release);
    }
    const int per_wg = FP.n_waves;
    int grid = (n_clips + per_wg - 1) / per_wg;
    if (grid > n_cu) grid = n_cu;
#if KWS_FAST_WPS >= 3
    if (!FP.tickets) return (int)hipErrorInvalidValue;
#endif
    hipLaunchKernelGGL((kws_fast_kernel<NZ, DG, PROF, FROM_CEP, NET, QCP, MFE>), dim3(grid), dim3(KWS_WAVE * FP.n_waves), smem, stream, P, d_plan, pcm, n_clips, scores,
                       features, q_out, in_scale, in_zp, flag_count, flag_list, prof_out, cep, d_nn, sel, tap_logits, feat_in);
    return (int)hipGetLastError();
}

int kws_launch_fast(const KwsDspPlan &P, const KwsFastPlan &FP, const KwsFastPlan *d_plan, const int16_t *pcm, int n_clips, float *scores,
                    float *features, int8_t *q_out, float in_scale, int in_zp, int *flag_count, int *flag_list, int n_cu, hipStream_t stream,
                    const KwsNnPlan *d_nn, float *tap_logits)
{
    (void)hipGetLastError();
    if (n_clips <= 0) return 0;
#define KWS_FAST_ARGS P, FP, d_plan, pcm, n_clips, scores, features, q_out, in_scale, in_zp, flag_count, flag_list, n_cu, nullptr, stream
#define KWS_FAST_NARGS KWS_FAST_ARGS, nullptr, nullptr, nullptr, tap_logits
#if KWS_FAST_WPS >= 3
    if (FP.wps != KWS_FAST_WPS || FP.qnet || FP.mfe || !FP.fuse) return (int)hipErrorInvalidValue;        // this build: the float32-network forms
#define KWS_FAST_PLAIN(...) ((int)hipErrorInvalidValue)
#else
#define KWS_FAST_PLAIN(...) (__VA_ARGS__)
    if (FP.wps >= 3) return kws_launch_fast_w3(P, FP, d_plan, pcm, n_clips, scores, features, q_out, in_scale, in_zp, flag_count, flag_list, n_cu, stream, d_nn, tap_logits);
    if (FP.qnet) {
        // int8 graph fused: 16-byte activation rows go with the 32-filter front end, 64-byte rows with the 40-filter one (the plan checks)
#define KWS_FAST_QARGS P, FP, d_plan, pcm, n_clips, scores, features, q_out, in_scale, in_zp, flag_count, flag_list, n_cu, nullptr, stream, nullptr, d_nn
        if (FP.qnet == 16 && FP.dct_groups == 4)
            return FP.nz <= 4 ? launch_fast_t<4, 4, false, false, false, 16>(KWS_FAST_QARGS) : FP.nz <= 8 ? launch_fast_t<8, 4, false, false, false, 16>(KWS_FAST_QARGS)
                                                                                                          : launch_fast_t<KWS_FAST_NZ_MAX, 4, false, false, false, 16>(KWS_FAST_QARGS);
        if (FP.qnet == 64 && FP.dct_groups == 5)
            return FP.nz <= 4 ? launch_fast_t<4, 5, false, false, false, 64>(KWS_FAST_QARGS) : FP.nz <= 8 ? launch_fast_t<8, 5, false, false, false, 64>(KWS_FAST_QARGS)
                                                                                                          : launch_fast_t<KWS_FAST_NZ_MAX, 5, false, false, false, 64>(KWS_FAST_QARGS);
        return (int)hipErrorInvalidValue;
    }
    if (FP.mfe) {
        if (scores || q_out || !features) return (int)hipErrorInvalidValue;          // the MFE form has one output: the mel matrix
        if (FP.dct_groups == 4)
            return FP.nz <= 4 ? launch_fast_t<4, 4, false, false, false, 0, true>(KWS_FAST_ARGS) : FP.nz <= 8 ? launch_fast_t<8, 4, false, false, false, 0, true>(KWS_FAST_ARGS)
                                                                                                            : launch_fast_t<KWS_FAST_NZ_MAX, 4, false, false, false, 0, true>(KWS_FAST_ARGS);
        if (FP.dct_groups == 5)
            return FP.nz <= 4 ? launch_fast_t<4, 5, false, false, false, 0, true>(KWS_FAST_ARGS) : FP.nz <= 8 ? launch_fast_t<8, 5, false, false, false, 0, true>(KWS_FAST_ARGS)
                                                                                                            : launch_fast_t<KWS_FAST_NZ_MAX, 5, false, false, false, 0, true>(KWS_FAST_ARGS);
        return (int)hipErrorInvalidValue;
    }
#endif
    if (FP.dct_groups == 4)
        return FP.fuse ? (FP.nz <= 4 ? launch_fast_t<4, 4, false>(KWS_FAST_NARGS) : FP.nz <= 8 ? launch_fast_t<8, 4, false>(KWS_FAST_NARGS)
                                                                                              : launch_fast_t<KWS_FAST_NZ_MAX, 4, false>(KWS_FAST_NARGS))
                       : KWS_FAST_PLAIN(FP.nz <= 4 ? launch_fast_t<4, 4, false, false, false>(KWS_FAST_ARGS) : FP.nz <= 8 ? launch_fast_t<8, 4, false, false, false>(KWS_FAST_ARGS)
                                                                                              : launch_fast_t<KWS_FAST_NZ_MAX, 4, false, false, false>(KWS_FAST_ARGS));
    if (FP.dct_groups == 5)
        return FP.fuse ? (FP.nz <= 4 ? launch_fast_t<4, 5, false>(KWS_FAST_NARGS) : FP.nz <= 8 ? launch_fast_t<8, 5, false>(KWS_FAST_NARGS)
                                                                                              : launch_fast_t<KWS_FAST_NZ_MAX, 5, false>(KWS_FAST_NARGS))
                       : KWS_FAST_PLAIN(FP.nz <= 4 ? launch_fast_t<4, 5, false, false, false>(KWS_FAST_ARGS) : FP.nz <= 8 ? launch_fast_t<8, 5, false, false, false>(KWS_FAST_ARGS)
                                                                                              : launch_fast_t<KWS_FAST_NZ_MAX, 5, false, false, false>(KWS_FAST_ARGS));
    return (int)hipErrorInvalidValue;
}

#if KWS_FAST_WPS < 3
// bytes of the workgroup's shared LDS block the fused int8 network's tables take (the layout of the kernel's prologue)
size_t kws_fast_qnet_bytes(int qcp)
{
    return (size_t)48 * 256 + ((KWS_HEAD_BYTES + 15) & ~15) + (size_t)((qcp == 16 ? 4 : 16) + 4) * KWS_WAVE * 16 + 6 * KWS_WAVE * 4;
}
#endif

#if KWS_FAST_WPS < 3
// cmvnw + (fused float network | features / int8 tensor) from cepstra in HBM, ring-indexed per P.ring_* (continuous mode)
int kws_launch_fast_from_cepstra(const KwsDspPlan &P, const KwsFastPlan &FP, const KwsFastPlan *d_plan, const float *cep, int n_clips, float *scores,
                                 float *features, int8_t *q_out, float in_scale, int in_zp, int *flag_count, int *flag_list, int n_cu, hipStream_t stream,
                                 const int *sel, float *tap_logits, int feat_in)
{
    (void)hipGetLastError();
    if (n_clips <= 0) return 0;
    if (feat_in && !FP.fuse) return (int)hipErrorInvalidValue;          // features in: only the fused network is left to run
    if (FP.wps != KWS_FAST_WPS) return (int)hipErrorInvalidValue;       // (the forms that start from cepstra exist at two waves per SIMD only: kws_internal.h, fast_fused_cep)
    // mel taps / DCT are not part of this variant: one instantiation serves every model
    return FP.fuse ? launch_fast_t<4, 4, false, true>(P, FP, d_plan, nullptr, n_clips, scores, features, q_out, in_scale, in_zp, flag_count, flag_list,
                                                      n_cu, nullptr, stream, cep, nullptr, sel, tap_logits, feat_in)
                   : KWS_FAST_PLAIN((launch_fast_t<4, 4, false, true, false>(P, FP, d_plan, nullptr, n_clips, scores, features, q_out, in_scale, in_zp, flag_count,
                                                             flag_list, n_cu, nullptr, stream, cep, nullptr, sel)));
}
#endif

// development aid: phase clocks (<= 4-tap builds only)
int kws_launch_fast_prof(const KwsDspPlan &P, const KwsFastPlan &FP, const KwsFastPlan *d_plan, const int16_t *pcm, int n_clips, float *scores,
                         int *flag_count, int *flag_list, int n_cu, long long *prof_out, hipStream_t stream)
{
    (void)hipGetLastError();
    if (FP.nz > 4) return (int)hipErrorInvalidValue;
#if KWS_FAST_WPS >= 3
    if (FP.wps != KWS_FAST_WPS) return (int)hipErrorInvalidValue;
#else
    if (FP.wps >= 3) return kws_launch_fast_prof_w3(P, FP, d_plan, pcm, n_clips, scores, flag_count, flag_list, n_cu, prof_out, stream);
#endif
    if (FP.dct_groups == 5)
        return launch_fast_t<4, 5, true>(P, FP, d_plan, pcm, n_clips, scores, nullptr, nullptr, 1.0f, 0, flag_count, flag_list, n_cu, prof_out, stream);
    if (FP.dct_groups == 4)
        return launch_fast_t<4, 4, true>(P, FP, d_plan, pcm, n_clips, scores, nullptr, nullptr, 1.0f, 0, flag_count, flag_list, n_cu, prof_out, stream);
    return (int)hipErrorInvalidValue;
}
#endif      // KWS_FAST_NO_LAUNCHERS
