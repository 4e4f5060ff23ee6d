// kws_nn_int8.hip -- the int8 network: kws_nn_kernel (generic: any chain of conv blocks; un-pooled CONV_2D on the matrix
// cores, depthwise in registers, the rest on v_dot4), kws_nn_mfma_kernel (the two-block shape entirely on the matrix cores) and
// kws_cmvn_nn_kernel (cmvnw + quantise [+ network] for the stage API / continuous mode).  Replaces the EON-compiled
// TFLite-Micro graph (MODEL/tflite-model/trained_model_compiled.cpp:312-328).
#include "kws_device.h"
#include <cstdio>
#include <cstdlib>

#include "kws_nn_int8_dev.h"


// MAXW: waves per workgroup the build allows.  16 (four per SIMD) caps the kernel at 128 VGPRs, which it overruns by 20 (176 B of
// scratch per lane); 12 (three per SIMD) gives it 168 and no scratch -- the launcher picks by measurement (below).
template <int MAXW>
__global__ __launch_bounds__(KWS_WAVE * MAXW) void kws_nn_kernel(KwsNnPlan N, const int8_t *__restrict__ q_in, int n_clips,
                                                                         float *__restrict__ scores, NnTaps taps)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), n_waves = blockDim.x >> 6;

    // ---- shared: weights, ADD tables and requantisation constants of every block, in block order ---------------
    const int fcw_bytes = nn_head_fcw_bytes(N), fcx_bytes = nn_fcx_bytes(N);
    const NnHeadTab head = nn_head_stage(N, smem_raw, fcw_bytes);
    unsigned char *const blocks_base = smem_raw + fcw_bytes + ((KWS_HEAD_REST + 15) & ~15);
    unsigned char *sp = blocks_base;
    int act_b[2] = { 0, 0 };                           // block b reads buffer b & 1: each is sized for its own blocks
    for (int b = 0; b < N.n_blocks; ++b) {
        const KwsConvBlock &k = N.blk[b];
        const int wbytes = k.w_bytes;
        for (int i = threadIdx.x * 4; i < wbytes; i += blockDim.x * 4) *(int *)(sp + i) = *(const int *)(k.w + i);
        sp += (wbytes + 15) & ~15;
        const int lbytes = k.has_lut ? k.out_c * 256 : 0;
        for (int i = threadIdx.x * 4; i < lbytes; i += blockDim.x * 4) *(int *)(sp + i) = *(const int *)(k.add_lut + i);
        sp += lbytes;
        for (int i = threadIdx.x; i < k.out_c; i += blockDim.x) ((int4 *)sp)[i] = make_int4(k.bias_eff[i], k.mult[i], k.shift[i], 0);
        sp += k.out_c * 16;
        const int ab = nn_rows(k) * k.in_cpad;
        act_b[b & 1] = max(act_b[b & 1], ab);
    }
    act_b[0] = (act_b[0] + 15) & ~15; act_b[1] = (act_b[1] + 15) & ~15;
    // per wave: two activation buffers (ping-pong) + a small vector for FC/softmax
    int8_t *actA = (int8_t *)(sp + wave * (act_b[0] + act_b[1] + fcx_bytes + 64 * 4));
    int8_t *actB = actA + act_b[0];
    int8_t *fcx = actB + act_b[1];                    // the FULLY_CONNECTED input vector (last block's pooled output)
    int *lgv = (int *)(fcx + fcx_bytes);               // logits; until the head runs, the sink of stores that fall outside the image
    int8_t *const sink = (int8_t *)lgv + 4 * lane;
    __syncthreads();

    const int F = N.n_features;
    // i / in_c for i < n_features <= 4096 as a multiply + shift (exact: i * 1 < 2^20 / in_c for in_c <= 64)
    const unsigned inv_c = (1u << 20) / (unsigned)N.blk[0].in_c + 1u;
    const bool profiling = taps.prof != nullptr && blockIdx.x == 0 && wave == 0;
    long long ph[KWS_MAX_BLOCKS + 2] = { 0 }, tlast = profiling ? clock64() : 0;
    auto mark = [&](int i) { if (profiling) { const long long now = clock64(); ph[i] += now - tlast; tlast = now; } };
    const int n_sel = sel_count(taps.sel, n_clips);
    for (int ci = blockIdx.x * n_waves + wave; ci < n_sel; ci += gridDim.x * n_waves) {
        const int clip = sel_clip(taps.sel, ci);
        // ---- stage the int8 input as [pad_left + t][in_cpad], padding = zero point ((x + offset) == 0) -------
        {
            const KwsConvBlock &k = N.blk[0];
            const int rows = nn_rows(k);
            const int zp4 = (int)((unsigned)(k.in_zp & 0xff) * 0x01010101u);
            for (int i = lane * 16; i < rows * k.in_cpad; i += 64 * 16) *(int4 *)(actA + i) = make_int4(zp4, zp4, zp4, zp4);
            WAVE_SYNC();
            const int8_t *src = q_in + (size_t)clip * F;
            if ((k.in_c & 3) == 0) {                       // four channels per copy (feature vector and rows 4-byte aligned)
                for (int i = lane * 4; i < k.in_w * k.in_c; i += 64 * 4) {
                    const int tt = (int)(((unsigned)i * inv_c) >> 20), c = i - tt * k.in_c;
                    *(int *)(actA + (tt + k.pad_left) * k.in_cpad + c) = *(const int *)(src + i);
                }
            } else {
                for (int i = lane; i < k.in_w * k.in_c; i += 64) {
                    const int tt = (int)(((unsigned)i * inv_c) >> 20), c = i - tt * k.in_c;       // i / in_c
                    actA[(tt + k.pad_left) * k.in_cpad + c] = src[i];
                }
            }
            WAVE_SYNC();
        }
        mark(0);
        int8_t *cur = actA, *nxt = actB;
        int pooled_off = 0;
        const unsigned char *bp = blocks_base;
        for (int b = 0; b < N.n_blocks; ++b) {
            const KwsConvBlock &k = N.blk[b];
            // this block's share of the workgroup's LDS block (kept as running pointers: an array of them would be indexed
            // through the scalar unit per access)
            const int8_t *wb = (const int8_t *)bp;
            bp += (k.w_bytes + 15) & ~15;
            const int8_t *lutb = (const int8_t *)bp;
            bp += k.has_lut ? k.out_c * 256 : 0;
            const int4 *rqt = (const int4 *)bp;
            bp += k.out_c * 16;
            const bool last = (b + 1 == N.n_blocks);
            const int nrows = last ? 0 : nn_rows(N.blk[b + 1]);
            const int ncp = last ? k.out_c : N.blk[b + 1].in_cpad;
            const int npl = last ? 0 : N.blk[b + 1].pad_left;
            if (!last) {
                const int zp4 = (int)((unsigned)(N.blk[b + 1].in_zp & 0xff) * 0x01010101u);
                for (int i = lane * 16; i < nrows * ncp; i += 64 * 16) *(int4 *)(nxt + i) = make_int4(zp4, zp4, zp4, zp4);
                WAVE_SYNC();
            }
            const int out_c = k.out_c, out_w = k.out_w, n_out = k.pool_w * out_c;
            const int out_zp = k.out_zp, act_min = k.act_min, act_max = k.act_max;
            const bool has_lut = k.has_lut != 0;
            // output (pooled row pw, channel oc) lands at dbase + pw * dstride + oc; the optional debug tap mirrors it in HBM
            int8_t *const dbase = last ? fcx : nxt + npl * ncp;
            const int dstride = last ? out_c : ncp;
            int8_t *const tp = taps.pooled ? taps.pooled + (size_t)clip * taps.pooled_stride + pooled_off : nullptr;
            // one accumulator at a time (the paths below that are not worth batching)
            auto finish = [&](int m, int pw, int oc, const NnRq &q) {
                const int r = nn_requant(m, q, out_zp, act_min, act_max);
                const int8_t o = has_lut ? lutb[oc * 256 + (r + 128)] : (int8_t)r;
                dbase[pw * dstride + oc] = o;
                if (tp) tp[pw * out_c + oc] = o;
            };
            if (k.depthwise && k.depth_mult == 1 && (out_c & 3) == 0 && k.taps <= 8) {
                // integer_ops/depthwise_conv.h:64-103, one input channel per output.  A lane owns 4 consecutive channels of
                // one pooling window (or of 8 time steps when the block is not pooled): the tb + taps - 1 activation rows
                // it needs are read ONCE as 32-bit words (4 channels each), the 4 x taps weights as 8 words, everything
                // else is register arithmetic.  Outputs are requantised as a batch, looked up as a batch and stored four
                // channels per word.
                const bool pooled = k.pool > 1;
                const int tb = pooled ? k.pool : KWS_POOL_MAX, tstride = pooled ? k.pool_stride : KWS_POOL_MAX;
                const int n_tb = pooled ? k.pool_w : (out_w + KWS_POOL_MAX - 1) / KWS_POOL_MAX;
                const int n_cg = out_c >> 2, tp4 = (k.taps + 3) & ~3, nrow = tb + k.taps - 1;
                for (int item = lane; item < n_tb * n_cg; item += 64) {
                    const int pw = item / n_cg, oc0 = (item - pw * n_cg) * 4;
                    const int t0 = pw * tstride;
                    int xw[KWS_POOL_MAX + 7], ww[4][2];
#pragma unroll
                    for (int r = 0; r < KWS_POOL_MAX + 7; ++r)
                        xw[r] = r < nrow ? *(const int *)(cur + (t0 + r) * k.in_cpad + oc0) : 0;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int *wr = (const int *)(wb + (oc0 + c) * tp4);
                        ww[c][0] = wr[0];
                        ww[c][1] = tp4 > 4 ? wr[1] : 0;
                    }
                    NnRq rq[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) rq[c] = nn_rq_of(rqt, oc0 + c);
                    int acc[KWS_POOL_MAX][4];
#pragma unroll
                    for (int i = 0; i < KWS_POOL_MAX; ++i)
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[i][c] = 0;
#pragma unroll
                    for (int tap = 0; tap < 8; ++tap) {
                        if (tap < k.taps) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                const int wv = (ww[c][tap >> 2] << (24 - 8 * (tap & 3))) >> 24;      // sign-extended byte
#pragma unroll
                                for (int i = 0; i < KWS_POOL_MAX; ++i)
                                    if (i < tb) acc[i][c] += wv * ((xw[i + tap] << (24 - 8 * c)) >> 24);
                            }
                        }
                    }
                    if (pooled) {
                        int o[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            int m = (int)0x80000000;
#pragma unroll
                            for (int i = 0; i < KWS_POOL_MAX; ++i)
                                if (i < tb && t0 + i < out_w) m = max(m, acc[i][c]);
                            o[c] = nn_requant(m, rq[c], out_zp, act_min, act_max);
                        }
                        if (has_lut)
#pragma unroll
                            for (int c = 0; c < 4; ++c) o[c] = lutb[(oc0 + c) * 256 + (o[c] + 128)];
                        const int word = (o[0] & 0xff) | ((o[1] & 0xff) << 8) | ((o[2] & 0xff) << 16) | ((int)((unsigned)o[3] << 24));
                        *(int *)(dbase + pw * dstride + oc0) = word;
                        if (tp)
#pragma unroll
                            for (int c = 0; c < 4; ++c) tp[pw * out_c + oc0 + c] = (int8_t)o[c];
                    } else {
#pragma unroll
                        for (int i = 0; i < KWS_POOL_MAX; ++i) {               // a time step's four channels: one word
                            int o[4];
#pragma unroll
                            for (int c = 0; c < 4; ++c) o[c] = nn_requant(acc[i][c], rq[c], out_zp, act_min, act_max);
                            if (has_lut)
#pragma unroll
                                for (int c = 0; c < 4; ++c) o[c] = lutb[(oc0 + c) * 256 + (o[c] + 128)];
                            const int word = (o[0] & 0xff) | ((o[1] & 0xff) << 8) | ((o[2] & 0xff) << 16) | ((int)((unsigned)o[3] << 24));
                            const bool ok = t0 + i < out_w;
                            *(int *)(ok ? dbase + (t0 + i) * dstride + oc0 : sink) = word;
                            if (tp && ok)
#pragma unroll
                                for (int c = 0; c < 4; ++c) tp[(t0 + i) * out_c + oc0 + c] = (int8_t)o[c];
                        }
                    }
                }
            } else if (k.depthwise) {                    // any depth multiplier / channel count: one output per lane and pass
                const int tp4 = (k.taps + 3) & ~3;
                for (int idx = lane; idx < n_out; idx += 64) {
                    const int pw = idx / out_c, oc = idx - pw * out_c;
                    const int t0 = pw * k.pool_stride;
                    int acc[KWS_POOL_MAX];
#pragma unroll
                    for (int i = 0; i < KWS_POOL_MAX; ++i) acc[i] = 0;
                    const int8_t *wrow = wb + oc * tp4;
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
                        if (i < k.pool && t0 + i < out_w) m = max(m, acc[i]);
                    finish(m, pw, oc, nn_rq_of(rqt, oc));
                }
            } else if (k.mfma) {
                // CONV_2D without pooling on v_mfma_i32_32x32x32_i8: [time x (taps * in_cpad)] x [(taps * in_cpad) x out_c], one or
                // two 32-row tiles, k-steps of 32 bytes = two taps (16-byte rows), one tap (32) or half a tap (64).  A and B use
                // the same slot -> (tap, channel) map; int32 accumulation is exact, so the sums equal the reference's scalar loops
                const int cp = k.in_cpad, n = lane & 31, hh = lane >> 5;
                const int ks = cp == 16 ? (k.taps + 1) >> 1 : cp == 32 ? k.taps : 2 * k.taps;
                const bool two = out_w > 32;
                const int nc = min(n, out_c - 1);                                              // columns >= out_c: never stored
                v16i acc0 = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 }, acc1 = acc0;
                const int8_t *wrow = wb + (size_t)nc * k.taps * cp;
                const NnRq rq = nn_rq_of(rqt, nc);
                for (int s_ = 0; s_ < ks; ++s_) {
                    const int tap = cp == 16 ? 2 * s_ + hh : cp == 32 ? s_ : s_ >> 1;
                    const int ch = cp == 16 ? 0 : cp == 32 ? 16 * hh : 32 * (s_ & 1) + 16 * hh;
                    v4i wv = { 0, 0, 0, 0 };
                    if (tap < k.taps) wv = *(const v4i *)(wrow + tap * cp + ch);
                    const v4i a0 = *(const v4i *)(cur + (n + tap) * cp + ch);                  // row = time (lane & 31) + tap
                    acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, wv, acc0, 0, 0, 0);
                    if (two) {
                        const v4i a1 = *(const v4i *)(cur + (32 + n + tap) * cp + ch);
                        acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, wv, acc1, 0, 0, 0);
                    }
                }
                // accumulator register r of a tile holds row (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), column lane & 31: the lane
                // requantises its column's 16 (32) values as a batch, looks them up as a batch, then stores (rows / columns
                // outside the image go to the sink: no branch per value)
                const int8_t *lp = lutb + nc * 256 + 128;
                int8_t *const dcol = dbase + n;
                const bool col_ok = n < out_c;
                for (int tile = 0; tile < (two ? 2 : 1); ++tile) {
                    const v16i &acc = tile ? acc1 : acc0;
                    // registers 4 g .. 4 g + 3 hold rows 32 tile + 8 g + 4 hh + 0 .. 3: a group past the image is skipped whole
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if (32 * tile + 8 * g >= out_w) break;
                        int o[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) o[i] = nn_requant(acc[4 * g + i], rq, out_zp, act_min, act_max);
                        if (has_lut)
#pragma unroll
                            for (int i = 0; i < 4; ++i) o[i] = lp[o[i]];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int row = 32 * tile + 8 * g + 4 * hh + i;
                            const bool ok = col_ok && row < out_w;
                            *(ok ? dcol + row * dstride : sink) = (int8_t)o[i];
                            if (tp && ok) tp[row * out_c + n] = (int8_t)o[i];
                        }
                    }
                }
            } else {
                // a lane owns one pooling window of OB = 2 (or 1) output channels: every 16-byte activation read feeds
                // 4 * OB dot products, every 16-byte weight read `pool` of them
                // (an un-pooled block is walked in groups of KWS_POOL_MAX time steps, each stored on its own)
                const bool pooled = k.pool > 1;
                const int tb = pooled ? k.pool : KWS_POOL_MAX, tstride = pooled ? k.pool_stride : KWS_POOL_MAX;
                const int n_tb = pooled ? k.pool_w : (out_w + KWS_POOL_MAX - 1) / KWS_POOL_MAX;
                const int ob = (n_tb * out_c > 64 && (out_c & 1) == 0) ? 2 : 1;
                const int n_ocb = out_c / ob;
                const int c16n = k.in_cpad >> 4;
                for (int item = lane; item < n_tb * n_ocb; item += 64) {
                    const int pw = item / n_ocb, oc0 = (item - pw * n_ocb) * ob;
                    const int t0 = pw * tstride;
                    int acc[KWS_POOL_MAX][2];
#pragma unroll
                    for (int i = 0; i < KWS_POOL_MAX; ++i) acc[i][0] = acc[i][1] = 0;
                    const int8_t *w0 = wb + (size_t)oc0 * k.taps * k.in_cpad;
                    const int8_t *w1 = w0 + (ob == 2 ? k.taps * k.in_cpad : 0);
                    for (int tap = 0; tap < k.taps; ++tap) {
                        const int8_t *xrow = cur + (t0 + tap) * k.in_cpad;
                        for (int c16 = 0; c16 < c16n; ++c16) {
                            const int4 wa = *(const int4 *)(w0 + tap * k.in_cpad + 16 * c16);
                            const int4 wb4 = *(const int4 *)(w1 + tap * k.in_cpad + 16 * c16);
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
                                        c = __builtin_amdgcn_sdot4(wb4.x, xv.x, c, false);
                                        c = __builtin_amdgcn_sdot4(wb4.y, xv.y, c, false);
                                        c = __builtin_amdgcn_sdot4(wb4.z, xv.z, c, false);
                                        c = __builtin_amdgcn_sdot4(wb4.w, xv.w, c, false);
                                    }
                                    acc[i][0] = a; acc[i][1] = c;
                                }
                            }
                        }
                    }
                    for (int o = 0; o < ob; ++o) {
                        const NnRq rq = nn_rq_of(rqt, oc0 + o);
                        if (pooled) {
                            int m = (int)0x80000000;
#pragma unroll
                            for (int i = 0; i < KWS_POOL_MAX; ++i)
                                if (i < tb && t0 + i < out_w) m = max(m, acc[i][o]);
                            finish(m, pw, oc0 + o, rq);
                        } else {
#pragma unroll
                            for (int i = 0; i < KWS_POOL_MAX; ++i)
                                if (t0 + i < out_w) finish(acc[i][o], t0 + i, oc0 + o, rq);
                        }
                    }
                }
            }
            pooled_off += n_out;
            WAVE_SYNC();
            mark(1 + b);
            int8_t *tmp = cur; cur = nxt; nxt = tmp;
        }
        nn_head(N, head, fcx, lgv, lane, clip, scores, taps);
        mark(1 + KWS_MAX_BLOCKS);
    }
    if (profiling && lane == 0)
        for (int i = 0; i < KWS_MAX_BLOCKS + 2; ++i) taps.prof[i] = ph[i];
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
    __shared__ __attribute__((aligned(16))) unsigned char s_head[KWS_HEAD_BYTES];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));    // uniform: per-wave addresses stay in scalar registers
    const KwsConvBlock &k1 = N.blk[0], &k2 = N.blk[1];
    const NnHeadTab head = nn_head_stage(N, s_head);
    for (int i = threadIdx.x * 4; i < k1.out_c * 256; i += blockDim.x * 4) *(int *)(s_lut1 + i) = *(const int *)(k1.add_lut + i);
    for (int i = threadIdx.x * 4; i < k2.out_c * 256; i += blockDim.x * 4) *(int *)(s_lut2 + i) = *(const int *)(k2.add_lut + i);
    int8_t *act1 = s_act1[wave], *act2 = s_act2[wave];
    nn_mfma_fill_padding<CP>(N, act1, act2, lane);
    NnMfmaCtx<CP> ctx;
    nn_mfma_init<CP>(ctx, N, lane);
    __syncthreads();
    const int F = N.n_features;
    const unsigned inv_c = (1u << 20) / (unsigned)k1.in_c + 1u;          // i / in_c == (i * inv_c) >> 20 for i < n_features <= 4096
    const int n_sel = sel_count(taps.sel, n_clips);
    for (int ci = blockIdx.x * KWS_NN_WAVES + wave; ci < n_sel; ci += gridDim.x * KWS_NN_WAVES) {
        const int clip = sel_clip(taps.sel, ci);
        // ---- int8 input tensor [time][in_c] -> LDS rows of CP bytes at row (time + pad_left) --------------------
        const int8_t *src = q_in + (size_t)clip * F;
        if ((k1.in_c & 3) == 0) {                          // four channels per copy (feature vector and rows 4-byte aligned)
            for (int i = lane * 4; i < F; i += 64 * 4) {
                const int tt = (int)(((unsigned)i * inv_c) >> 20), c = i - tt * k1.in_c;
                *(int *)(act1 + (tt + k1.pad_left) * CP + c) = *(const int *)(src + i);
            }
        } else {
            for (int i = lane; i < F; i += 64) {
                const int tt = (int)(((unsigned)i * inv_c) >> 20), c = i - tt * k1.in_c;       // i / in_c
                act1[(tt + k1.pad_left) * CP + c] = src[i];
            }
        }
        WAVE_SYNC();
        nn_mfma_clip<CP>(ctx, N, head, act1, act2, s_vec[wave], s_lut1, s_lut2, lane, clip, scores, taps);
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
    __shared__ __attribute__((aligned(16))) unsigned char s_head[FUSE ? KWS_HEAD_BYTES : 16];
    NnHeadTab head = { nullptr, nullptr, nullptr, nullptr };
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));    // uniform: per-wave addresses stay in scalar registers
    const int nfr = P.n_frames, ncep = P.n_cepstral, nfeat = nfr * ncep;
    const int prow = nfr + 2 * P.pad;
    for (int i = threadIdx.x; i < prow; i += blockDim.x) s_map[i] = P.pad_map[i];
    NnMfmaCtx<16> ctx;
    int8_t *act1 = s_act1[wave], *act2 = s_act2[wave];
    if constexpr (FUSE) {
        head = nn_head_stage(N, s_head);
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

    const int n_sel = sel_count(taps.sel, n_clips);
    for (int ci = blockIdx.x * KWS_NN_WAVES + wave; ci < n_sel; ci += gridDim.x * KWS_NN_WAVES) {
        const int clip = sel_clip(taps.sel, ci);
        const float *src = mfcc + (size_t)clip * nfeat;
        if (P.ring_rows) {
            const unsigned inv = (1u << 20) / (unsigned)ncep + 1u;          // i / ncep for i < 4096, ncep <= 64
            for (int i = lane; i < nfeat; i += 64) {
                const int r = (int)(((unsigned)i * inv) >> 20);
                mf[i] = src[ring_in_row(P, r) * ncep + (i - r * ncep)];
            }
        } else {
            for (int i = lane; i < nfeat; i += 64) mf[i] = src[i];
        }
        WAVE_SYNC();
        cmvn_columns<13, 16>(mf, ncep, s_map, s_off[wave], lane, nfr, ncep, prow, win, [&](int row, int c, float o) {
            const int idx = row * ncep + c;
            if (features) features[(size_t)clip * nfeat + idx] = o;
            const int8_t qb = quantize_feature(o, in_scale, in_zp);
            if (q_out) q_out[(size_t)clip * nfeat + idx] = qb;
            if constexpr (FUSE) act1[(row + N.blk[0].pad_left) * 16 + c] = qb;
        });
        WAVE_SYNC();
        if constexpr (FUSE) nn_mfma_clip<16>(ctx, N, head, act1, act2, s_vec[wave], s_lut1, s_lut2, lane, clip, scores, taps);
    }
}

// ---------------------------------------------------------------------------------------------------------
//  launchers (called from kws_api.cpp)
// ---------------------------------------------------------------------------------------------------------

long long *kws_dev_nn_prof = nullptr;   // development aid (tools/gpu_nn_phase_profile.py)
int kws_force_scalar_nn = 0;   // tests: run the generic (dot4) kernel even when the matrix-core kernel applies
int kws_nn_uses_mfma(const KwsNnPlan &N) { return nn_fits_mfma(N) && !kws_force_scalar_nn; }

// cmvnw + quantise (+ the network when it fits the matrix-core path and scores != NULL).  Returns 1 in *ran_nn if the
// network ran inside this launch.
int kws_launch_cmvn_nn(const KwsDspPlan &P, const KwsNnPlan &N, const float *mfcc, int n_clips, float *features, int8_t *q_out,
                       float *scores, int8_t *tap_pooled, int pooled_stride, int8_t *tap_fc, int8_t *tap_out_q, int grid_cap,
                       int *ran_nn, hipStream_t stream, const int *sel)
{
    (void)hipGetLastError();      // the status returned below is this launch's, not a stale error of an earlier call
    *ran_nn = 0;
    if (n_clips <= 0) return 0;
    int grid = (n_clips + KWS_NN_WAVES - 1) / KWS_NN_WAVES;
    if (grid > grid_cap) grid = grid_cap;
    NnTaps taps = { tap_pooled, pooled_stride, tap_fc, tap_out_q, kws_dev_nn_prof, sel };
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
size_t kws_nn_smem_bytes(const KwsNnPlan &N, int n_waves)
{
    size_t s = (size_t)nn_head_fcw_bytes(N) + ((KWS_HEAD_REST + 15) & ~15);
    int act[2] = { 0, 0 };
    for (int b = 0; b < N.n_blocks; ++b) {
        const KwsConvBlock &k = N.blk[b];
        s += ((size_t)k.w_bytes + 15) & ~(size_t)15;
        s += k.has_lut ? (size_t)k.out_c * 256 : 0;
        s += (size_t)k.out_c * 16;                      // requantisation constants
        const int ab = nn_rows(k) * k.in_cpad;
        act[b & 1] = ab > act[b & 1] ? ab : act[b & 1];
    }
    return s + (size_t)n_waves * (((act[0] + 15) & ~15) + ((act[1] + 15) & ~15) + nn_fcx_bytes(N) + 64 * 4);
}

int kws_launch_nn(const KwsNnPlan &N, const int8_t *q_in, int n_clips, float *scores, int8_t *tap_pooled,
                  int pooled_stride, int8_t *tap_fc, int8_t *tap_out_q, int grid_cap, hipStream_t stream, const int *sel)
{
    (void)hipGetLastError();      // the status returned below is this launch's, not a stale error of an earlier call
    if (n_clips <= 0) return 0;
    int grid = (n_clips + KWS_NN_WAVES - 1) / KWS_NN_WAVES;
    if (grid > grid_cap) grid = grid_cap;
    NnTaps taps = { tap_pooled, pooled_stride, tap_fc, tap_out_q, kws_dev_nn_prof, sel };
    if (nn_fits_mfma(N) && !kws_force_scalar_nn) {
        if (N.blk[0].in_cpad == 16)
            hipLaunchKernelGGL(kws_nn_mfma_kernel<16>, dim3(grid), dim3(KWS_WAVE * KWS_NN_WAVES), 0, stream, N, q_in, n_clips, scores, taps);
        else
            hipLaunchKernelGGL(kws_nn_mfma_kernel<64>, dim3(grid), dim3(KWS_WAVE * KWS_NN_WAVES), 0, stream, N, q_in, n_clips, scores, taps);
        return (int)hipGetLastError();
    }
    // generic kernel (168 VGPRs: three waves per SIMD): one persistent workgroup per CU with as many waves as the LDS holds (they
    // share the weights and tables), up to twelve
    int nw = KWS_NN_WAVES_MAX, per_cu = 1;
    while (nw > KWS_NN_WAVES && kws_nn_smem_bytes(N, nw) > 158 * 1024) --nw;
    if (const char *ev = KWS_DEV_ENV("KWS_DEV_NN_WAVES")) { int a = 0, b2 = 0; if (sscanf(ev, "%d,%d", &a, &b2) == 2 && a >= 1 && a <= KWS_NN_WAVES_MAX && b2 >= 1) { nw = a; per_cu = b2; } }   // development aid (occupancy experiments)
    const size_t smem = kws_nn_smem_bytes(N, nw);
    grid = (n_clips + nw - 1) / nw;
    if (grid > (grid_cap / 4) * per_cu) grid = (grid_cap / 4) * per_cu;          // grid_cap = 4 workgroups per CU
    if (nw <= 12) {
        if (smem > 64 * 1024) {                    // wide models: opt in to more than the default 64 KB of dynamic LDS
            hipError_t e = hipFuncSetAttribute((const void *)kws_nn_kernel<12>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL(kws_nn_kernel<12>, dim3(grid), dim3(KWS_WAVE * nw), smem, stream, N, q_in, n_clips, scores, taps);
        return (int)hipGetLastError();
    }
    if (smem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kws_nn_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kws_nn_kernel<16>, dim3(grid), dim3(KWS_WAVE * nw), smem, stream, N, q_in, n_clips, scores, taps);
    return (int)hipGetLastError();
}

