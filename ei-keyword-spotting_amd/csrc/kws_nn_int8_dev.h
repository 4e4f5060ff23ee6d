// kws_nn_int8_dev.h -- device code of the int8 network shared by kws_nn_int8.hip (the network kernels) and kws_fast.hip (the same
// network fused behind the fast MFCC block): gemmlowp / TFLite fixed-point helpers, FULLY_CONNECTED + SOFTMAX head, per-channel
// requantisation, and the matrix-core path of the two-block graph shape.  Internal to libkws_mi355x.so.
#pragma once
#include "kws_device.h"

// ---------------------------------------------------------------------------------------------------------
//  gemmlowp / TFLite fixed-point helpers (fixedpoint.h:329-368, TFL/kernels/internal/common.h:138-162)
// ---------------------------------------------------------------------------------------------------------
// gemmlowp SaturatingRoundingDoublingHighMul (fixedpoint.h:329-339): nudge = ab >= 0 ? 2^30 : 1 - 2^30, then a TRUNCATING
// division of ab + nudge by 2^31.  With ab = q * 2^31 + r (0 <= r < 2^31) both signs give q + (r >= 2^30), i.e. the
// arithmetic shift (ab + 2^30) >> 31: for ab < 0 the numerator ab + 1 - 2^30 is negative with remainder r + 1 - 2^30 (mod
// 2^31), and truncation adds 1 exactly when that remainder is non-zero after the wrap, which is again r >= 2^30.
__device__ __forceinline__ int srdhm(int a, int b)
{
    const bool overflow = (a == b) && (a == (int)0x80000000);
    const long long ab = (long long)a * (long long)b;
    const int hi = (int)((ab + (1ll << 30)) >> 31);
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
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
constexpr int KWS_NN_WAVES = 4;
constexpr int KWS_NN_WAVES_MAX = 16;     // generic kernel: as many waves per workgroup as the LDS allows (they share weights and tables)
constexpr int KWS_POOL_MAX = 8;
// rows of a block's padded int8 input image in the generic kernel: un-pooled blocks are walked KWS_POOL_MAX time steps at
// a time, so reads reach up to ceil(out_w / 8) * 8 + taps - 1
// (matrix-core blocks read whole 32-row tiles: up to ceil(out_w / 32) * 32 + taps rows, the odd tap of a 2-tap k-step included)
__host__ __device__ inline int nn_rows(const KwsConvBlock &k)
{
    const int walk = k.mfma ? ((k.out_w + 31) & ~31) + 1 : (k.out_w + KWS_POOL_MAX - 1) & ~(KWS_POOL_MAX - 1);
    return max(k.in_w, walk) + k.taps;
}

struct NnTaps {            // optional debug outputs for the parity tests (all int8, per clip)
    int8_t *pooled;        // concatenation of every block's pooled output [pool_w][out_c]
    int pooled_stride;
    int8_t *fc;            // [fc_out]
    int8_t *out_q;         // [n_labels]
    long long *prof;       // development aid: shader-clock totals per phase of wave 0 of workgroup 0 (generic kernel), or NULL
    const int *sel;        // optional clip selection (kws_device.h sel_count / sel_clip), or NULL
};

// FULLY_CONNECTED (integer_ops/fully_connected.h:23-63) + SOFTMAX int8->int8 (reference/softmax.h:66-144) for one clip.
// xin: the last block's pooled output (the FC input vector, int8, in LDS); lg: 64 ints of LDS for the logits.
// The FC weights / bias and the softmax tables are read from LDS copies (NnHeadTab, staged once per workgroup): from global
// memory every step of these short dependent loops is an L2 round trip.
constexpr int KWS_HEAD_FCW = 48 * 64;       // the two-block kernels: fc_out <= 48, fc_in <= 16; the generic kernel sizes its copy
constexpr int KWS_HEAD_REST = 48 * 4 + 256 * 4 + 256;
constexpr int KWS_HEAD_BYTES = KWS_HEAD_FCW + KWS_HEAD_REST;
__host__ __device__ inline int nn_head_fcw_bytes(const KwsNnPlan &N) { return (N.fc_out * N.fc_in + 15) & ~15; }   // <= KWS_FC_W_MAX (plan)
__host__ __device__ inline int nn_fcx_bytes(const KwsNnPlan &N) { return max(64, (N.fc_in + 15) & ~15); }         // per wave: the FC input vector
struct NnHeadTab { const int8_t *fc_w; const int *fc_bias; const int *sm_exp; const uint8_t *sm_valid; };
__device__ __forceinline__ NnHeadTab nn_head_stage(const KwsNnPlan &N, unsigned char *lds, int fcw_bytes = KWS_HEAD_FCW)   // 16-byte aligned
{
    int8_t *fw = (int8_t *)lds;
    int *fb = (int *)(lds + fcw_bytes), *se = fb + 48;
    uint8_t *sv = (uint8_t *)(se + 256);
    for (int i = threadIdx.x; i < N.fc_out * N.fc_in; i += blockDim.x) fw[i] = N.fc_w[i];
    for (int i = threadIdx.x; i < N.fc_out; i += blockDim.x) fb[i] = N.fc_bias[i];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) { se[i] = N.sm_exp[i]; sv[i] = N.sm_valid[i]; }
    NnHeadTab t = { fw, fb, se, sv };
    return t;                                                                                // caller: __syncthreads()
}

// the addresses of an already staged block (same layout as nn_head_stage writes)
__device__ __forceinline__ NnHeadTab nn_head_tab(unsigned char *lds, int fcw_bytes = KWS_HEAD_FCW)
{
    const int *fb = (const int *)(lds + fcw_bytes), *se = fb + 48;
    NnHeadTab t = { (const int8_t *)lds, fb, se, (const uint8_t *)(se + 256) };
    return t;
}

// whole-wave reductions: four DPP steps inside a row of 16 lanes, then lane ^ 16 and lane ^ 32 through the LDS crossbar
template <typename T, typename Op>
__device__ __forceinline__ T wave_reduce(T v, Op op)
{
    v = op(v, (T)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true));     // quad_perm [1,0,3,2]
    v = op(v, (T)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true));     // quad_perm [2,3,0,1]
    v = op(v, (T)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true));    // row_half_mirror
    v = op(v, (T)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true));    // row_mirror
    v = op(v, (T)__builtin_amdgcn_ds_swizzle((int)v, 0x401F));                      // lane ^ 16
    v = op(v, (T)__shfl_xor((int)v, 32, KWS_WAVE));
    return v;
}
__device__ __forceinline__ int wave_reduce_max(int v) { return wave_reduce<int>(v, [](int a, int b) { return max(a, b); }); }
__device__ __forceinline__ unsigned wave_reduce_add(unsigned v) { return wave_reduce<unsigned>(v, [](unsigned a, unsigned b) { return a + b; }); }

__device__ __forceinline__ void nn_head(const KwsNnPlan &N, const NnHeadTab &H, const int8_t *xin, int *lg, int lane, int clip,
                                        float *__restrict__ scores, const NnTaps &taps)
{
    // ---- FULLY_CONNECTED (integer_ops/fully_connected.h:23-63): input = last pooled vector ------------------
    // int32 sums are exact, so the fc_in terms of an output are split over n_seg lanes (a power of two, 64 / fc_out or less)
    // and reduced with shuffles; fc_in can be hundreds (conv -> pool 2 -> conv -> pool 2 -> Dense exports)
    int n_seg = 1;
    while (2 * n_seg * N.fc_out <= KWS_WAVE && 2 * n_seg <= N.fc_in) n_seg *= 2;
    const int out = lane / n_seg, seg = lane - out * n_seg;
    int acc = 0;
    if (out < N.fc_out) {
        const int8_t *wr = H.fc_w + out * N.fc_in;
        for (int d = seg; d < N.fc_in; d += n_seg)
            acc += ((int)wr[d] + N.fc_w_off) * ((int)xin[d] + N.fc_in_off);
    }
    for (int sft = 1; sft < n_seg; sft *= 2) acc += __shfl_xor(acc, sft);
    if (out < N.fc_out && seg == 0) {
        acc += H.fc_bias[out];
        acc = mbqm(acc, N.fc_mult, N.fc_shift) + N.fc_out_zp;
        const int lgt = min(max(acc, N.fc_act_min), N.fc_act_max);
        lg[out] = lgt;
        if (taps.fc) taps.fc[(size_t)clip * N.fc_out + out] = (int8_t)lgt;
    }
    WAVE_SYNC();
    const bool on = lane < N.fc_out;
    const int logit = on ? lg[lane] : 0;
    // ---- SOFTMAX int8 -> int8 (reference/softmax.h:66-144): lane = class; the maximum and the sum of exponentials are wave
    //      reductions (the reference's unsigned wrap-around additions commute) ----------------------------------------------
    const int mx = wave_reduce_max(on ? logit : -128);
    const int d = on ? mx - logit : 0;
    const bool valid = on && H.sm_valid[d] != 0;
    const int ex = H.sm_exp[d];
    const int sum = (int)wave_reduce_add(valid ? (unsigned)rdivpot(ex, 12) : 0u);
    if (on) {
        const int hp1 = sum ? __clz(sum) : 32;                                  // GetReciprocal, common.h:530-546
        const int nbits = 12 - hp1;
        const int ssm1 = (int)(((unsigned)sum << hp1) - (1u << 31));
        const int scale = one_over_one_plus_x(ssm1);
        int o = -128;
        if (valid) {
            const int unsat = rdivpot(srdhm(scale, ex), nbits + 31 - 8);
            o = min(max(unsat - 128, -128), 127);
        }
        if (taps.out_q) taps.out_q[(size_t)clip * N.fc_out + lane] = (int8_t)o;
        scores[(size_t)clip * N.fc_out + lane] = (float)(o - N.out_zp) * N.out_scale;   // ei_run_classifier.h:470
    }
    WAVE_SYNC();
}

// per-channel requantisation constants (MultiplyByQuantizedMultiplier, common.h:138-162), staged in LDS once per workgroup
struct NnRq { int bias, mult, ls, rs, mask, half; };
__device__ __forceinline__ NnRq nn_rq_of(const int4 *tab, int oc)
{
    const int4 v = tab[oc];                                   // bias_eff, multiplier, shift, -
    NnRq q;
    q.bias = v.x; q.mult = v.y;
    q.ls = max(v.z, 0); q.rs = max(-v.z, 0);
    q.mask = (int)((1ll << q.rs) - 1); q.half = q.mask >> 1;
    return q;
}
// bias + requantise + output offset + activation clamp of one accumulator (integer_ops/conv.h:111-116).  srdhm() without its
// overflow branch: that needs a == b == INT_MIN, and the plan refuses negative multipliers.
__device__ __forceinline__ int nn_requant(int m, const NnRq &q, int out_zp, int act_min, int act_max)
{
    m += q.bias;
    const int x = (int)((unsigned)m << q.ls);
    const long long p = (long long)x * (long long)q.mult + (1ll << 30);
    const int hi = (int)(p >> 31);
    const int rem = hi & q.mask, thr = q.half + (hi < 0 ? 1 : 0);
    const int r = (hi >> q.rs) + (rem > thr ? 1 : 0) + out_zp;
    return min(max(r, act_min), act_max);
}

// ---------------------------------------------------------------------------------------------------------
//  Kernel 2, matrix-core path for the shipped graph shape (two conv blocks: time<=64 x 16ch x <=8 taps -> <=32 ch,
//  pool 7/7; then time<=16 x 32ch x <=8 taps -> <=16 ch, global pool).  The 1xK convolutions are genuine
//  contractions: per clip  [time x (taps*16)] x [(taps*16) x out_c]  on v_mfma_i32_32x32x32_i8 (8 per clip) and
//  [time x (taps*32)] x [(taps*32) x out_c] on v_mfma_i32_16x16x64_i8 (4 per clip).  int32 accumulation is exact, so
//  the result is bit-identical to the reference's scalar loops whatever the summation order.  One wave per clip;
//  weight fragments stay in registers for the whole launch; activations are read from LDS as aligned 16-byte rows.
// ---------------------------------------------------------------------------------------------------------

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

// The same constants with the weight fragments in LDS (kws_fast_kernel: no registers to spare across its phases): wb1 [KS1][64],
// wb2 [4][64] 16-byte fragments staged once per workgroup by nn_mfma_stage_lds; the six requantisation words come per clip
template <int CP>
struct NnMfmaLds {
    static constexpr int KS1 = CP == 16 ? 4 : 16;
    const v4i *wb1, *wb2;
    int b1, m1, sh1, b2, m2, sh2;
    bool oc1_ok, oc2_ok;
};
template <int CP> __device__ __forceinline__ v4i nn_w1(const NnMfmaCtx<CP> &c, int s, int) { return c.wb1[s]; }
template <int CP> __device__ __forceinline__ v4i nn_w2(const NnMfmaCtx<CP> &c, int s, int) { return c.wb2[s]; }
template <int CP> __device__ __forceinline__ v4i nn_w1(const NnMfmaLds<CP> &c, int s, int lane) { return c.wb1[s * KWS_WAVE + lane]; }
template <int CP> __device__ __forceinline__ v4i nn_w2(const NnMfmaLds<CP> &c, int s, int lane) { return c.wb2[s * KWS_WAVE + lane]; }

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

// LDS form: the fragments of nn_mfma_init written to wb1 [KS1][64] / wb2 [4][64] by the whole workgroup, and the per-lane
// requantisation words to rq [6][64] (bias, multiplier, shift of block 1 for lane & 31, of block 2 for lane & 15)
template <int CP>
__device__ __forceinline__ void nn_mfma_stage_lds(const KwsNnPlan &N, v4i *wb1, v4i *wb2, int *rq)
{
    const KwsConvBlock &k1 = N.blk[0], &k2 = N.blk[1];
    constexpr int KS1 = NnMfmaLds<CP>::KS1;
    for (int i = threadIdx.x; i < KS1 * KWS_WAVE; i += blockDim.x) {
        const int s = i >> 6, lane = i & 63, oc = lane & 31, h = lane >> 5;
        const int tap = CP == 16 ? 2 * s + h : s >> 1;
        const int ch = CP == 16 ? 0 : 32 * (s & 1) + 16 * h;
        v4i w = { 0, 0, 0, 0 };
        if (oc < k1.out_c && tap < k1.taps && ch < k1.in_cpad) w = *(const v4i *)(k1.w + ((size_t)oc * k1.taps + tap) * k1.in_cpad + ch);
        wb1[i] = w;
    }
    for (int i = threadIdx.x; i < 4 * KWS_WAVE; i += blockDim.x) {
        const int s = i >> 6, lane = i & 63, oc2 = lane & 15, g = lane >> 4;
        const int G = 4 * s + g, tap = G >> 1, ch = G & 1;
        v4i w = { 0, 0, 0, 0 };
        if (oc2 < k2.out_c && tap < k2.taps) w = *(const v4i *)(k2.w + ((size_t)oc2 * k2.taps + tap) * 32 + ch * 16);
        wb2[i] = w;
    }
    for (int lane = threadIdx.x; lane < KWS_WAVE; lane += blockDim.x) {
        const int oc = lane & 31, oc2 = lane & 15;
        const bool ok1 = oc < k1.out_c, ok2 = oc2 < k2.out_c;
        rq[0 * KWS_WAVE + lane] = ok1 ? k1.bias_eff[oc] : 0; rq[1 * KWS_WAVE + lane] = ok1 ? k1.mult[oc] : 0; rq[2 * KWS_WAVE + lane] = ok1 ? k1.shift[oc] : 0;
        rq[3 * KWS_WAVE + lane] = ok2 ? k2.bias_eff[oc2] : 0; rq[4 * KWS_WAVE + lane] = ok2 ? k2.mult[oc2] : 0; rq[5 * KWS_WAVE + lane] = ok2 ? k2.shift[oc2] : 0;
    }
}
template <int CP>
__device__ __forceinline__ NnMfmaLds<CP> nn_mfma_lds_ctx(const KwsNnPlan &N, const v4i *wb1, const v4i *wb2, const int *rq, int lane)
{
    NnMfmaLds<CP> c;
    c.wb1 = wb1; c.wb2 = wb2;
    c.b1 = rq[lane]; c.m1 = rq[KWS_WAVE + lane]; c.sh1 = rq[2 * KWS_WAVE + lane];
    c.b2 = rq[3 * KWS_WAVE + lane]; c.m2 = rq[4 * KWS_WAVE + lane]; c.sh2 = rq[5 * KWS_WAVE + lane];
    c.oc1_ok = (lane & 31) < N.blk[0].out_c; c.oc2_ok = (lane & 15) < N.blk[1].out_c;
    return c;
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
template <int CP, typename Ctx>
__device__ __forceinline__ void nn_mfma_clip(const Ctx &c, const KwsNnPlan &N, const NnHeadTab &head, const int8_t *act1, int8_t *act2, int *vec,
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
        const v4i wf = nn_w1<CP>(c, s, lane);
        acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, wf, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, wf, acc1, 0, 0, 0);
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
        c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, nn_w2<CP>(c, s, lane), c2, 0, 0, 0);
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
    nn_head(N, head, (const int8_t *)vec, vec + 16, lane, clip, scores, taps);
}


// the matrix-core kernel covers this graph shape; anything else runs on kws_nn_kernel
static bool nn_fits_mfma(const KwsNnPlan &N)
{
    if (N.n_blocks != 2) return false;
    const KwsConvBlock &a = N.blk[0], &b = N.blk[1];
    if (a.depthwise || b.depthwise) return false;
    return (a.in_cpad == 16 || a.in_cpad <= 64) && a.taps <= 8 && a.out_c <= 32 && a.in_w <= 64 && a.pool == KWS_MFMA_POOL && a.pool_stride == KWS_MFMA_POOL &&
           a.pool_w <= KWS_MFMA_POOL && a.out_w == a.pool_w * KWS_MFMA_POOL &&       // whole windows only: this kernel pools every row it computes
           b.in_cpad == 32 && b.taps <= 8 && b.out_c <= 16 && b.in_w <= 16 && b.pool_w == 1 &&
           b.pool >= b.out_w && N.fc_in == b.out_c;
}

