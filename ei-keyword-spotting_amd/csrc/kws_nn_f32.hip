// kws_nn_f32.hip -- kws_nn_f32_kernel: float32 graphs (the reference's float TFLite-Micro kernels replayed).
#include <algorithm>

#include "kws_device.h"

// ---------------------------------------------------------------------------------------------------------
//  Kernel 2f: float32 models.  One wave per clip.  Every accumulation replays the reference's sequential
//  `total += input * filter` order (tap outer, channel inner; product and sum rounded separately -- the build has
//  -ffp-contract=off), so everything up to the logits is bit-identical to the float TFLite-Micro kernels; only softmax's
//  expf is the device's.  Zero-padded activation rows stand in for the reference's skipped out-of-image taps: they add
//  an exact 0 (x*0 = +-0, and total + (+-0) == total for every total the chain can hold, +0 included).
//
//  The multiply-adds of a clip are order-constrained only WITHIN one output's chain, so a lane runs TB x OB chains (TB
//  consecutive time steps x OB consecutive output channels) side by side: per chain step it reads TB activations and one
//  OB-wide weight vector from LDS for TB*OB independent mul+add pairs.  The plan picks (TB, OB) per conv block so that the
//  work items fill the 64 lanes (shipped shape: conv1 49x30 -> 7x4, 56 lanes; conv2 7x10 -> 1x2, 35 lanes); a lane that
//  owns exactly one pooling window max-pools in registers, an un-pooled block writes straight into the next image.
//  LDS: per workgroup the weights of every block transposed to [tap*in_c + c][out_c padded to 4]; per wave the ping-pong
//  input images A / B (zero-padded rows), Y only for blocks whose pooling cannot happen in registers, and 128 floats for
//  the FULLY_CONNECTED input / logits.  The plan is read from device memory (nnf_layout() is shared with the host).
//  LDS locations are carried as offsets from the extern __shared__ base (pointers kept in arrays degrade to flat loads); the
//  next clip's feature vector is requested after block 0 and placed at the next iteration's top (256-register build).
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
__host__ __device__ __forceinline__ int nnf_ntb_of(const KwsConvBlockF32 &k) { return k.fused_pool ? k.pool_w : (k.out_w + k.tb - 1) / k.tb; }
__host__ __device__ __forceinline__ int nnf_rows_of(const KwsConvBlockF32 &k)       // rows of the zero-padded input image
{
    const int a = k.pad_left + k.in_w, b = nnf_ntb_of(k) * k.tb + k.taps - 1;  // rows the blocked walk touches (k.tb is final)
    return a > b ? a : b;
}
// x / d for an item index: the plan's reciprocal where it is exact (x d < 2^20), a real division (~35 vector instructions) elsewhere
__device__ __forceinline__ int nnf_div(int x, int d, unsigned inv20) { return inv20 ? (int)(((unsigned)x * inv20) >> 20) : x / d; }
// ... as the plan holds them (kws_nn_f32_pick_blocking sets both with the blocking)
__host__ __device__ __forceinline__ int nnf_ntb(const KwsConvBlockF32 &k) { return k.ntb; }
__host__ __device__ __forceinline__ int nnf_rows(const KwsConvBlockF32 &k) { return k.rows; }
__host__ __device__ __forceinline__ int nnf_ocp(const KwsConvBlockF32 &k) { return (k.out_c + 3) & ~3; }
// LDS position of weight (step j, output channel oc) of a conv block: [j][oc] (a lane reads OB consecutive channels of one
// step) for register-blocked blocks; [oc block][j][OB] for one-time-step blocks, whose lanes stream their own weights
__device__ __forceinline__ int nnf_w_index(const KwsConvBlockF32 &k, int J, int j, int oc)
{
    if (k.depthwise || k.tb != 1) return j * nnf_ocp(k) + oc;
    return ((oc / k.ob) * J + j) * k.ob + (oc % k.ob);
}

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
        const int tb = nnf_div(item, n_ob, k.inv_item20), ob = item - tb * n_ob;
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
            // one time step per lane (small blocks): a chain step is a single multiply-add per output channel, so issue
            // slots decide.  The weights of such a block sit in LDS as [oc block][step][OB] (nnf_w_index), i.e. a lane's
            // weight stream is contiguous: 8 steps' operands are 16 * OB / 4 + 4 reads at immediate offsets from two running
            // pointers, full batches run without clamps or predicates, the J % 8 last steps one by one.  (No double
            // buffering: the rotating copies cost more issue slots than the exposed round trip, which the SIMD's other
            // wave -- usually inside a register-blocked block -- fills.)
            typedef float f2v __attribute__((ext_vector_type(2)));
            constexpr int U = 8, NV = (OB + 1) / 2;
            const float *wq = wt + (size_t)ob * J * OB;
            const int Jm = J & ~(U - 1);
            f2v a2[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) a2[v] = (f2v){ 0.0f, 0.0f };
            auto step1 = [&](const float *wj, float xs_) {        // products and sums rounded separately (-ffp-contract=off)
                if constexpr (OB == 1) { const float prod = xs_ * wj[0]; a2[0].x += prod; }
                else {
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        const f2v wv = { wj[2 * v], wj[2 * v + 1] };
                        const f2v prod = wv * xs_;
                        a2[v] += prod;
                    }
                }
            };
            for (int j0 = 0; j0 < Jm; j0 += U) {
                float w[U * OB], xs_[U];
#pragma unroll
                for (int i = 0; i < U * OB; ++i) w[i] = wq[j0 * OB + i];
#pragma unroll
                for (int u = 0; u < U; ++u) xs_[u] = xp[j0 + u];
#pragma unroll
                for (int u = 0; u < U; ++u) step1(w + u * OB, xs_[u]);
            }
            for (int j = Jm; j < J; ++j) {
                float w[OB];
#pragma unroll
                for (int o = 0; o < OB; ++o) w[o] = wq[j * OB + o];
                step1(w, xp[j]);
            }
#pragma unroll
            for (int o = 0; o < OB; ++o) acc[0][o] = (o & 1) ? a2[o >> 1].y : a2[o >> 1].x;
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
        const int tb = nnf_div(item, k.out_c, k.inv_outc20), oc = item - tb * k.out_c;
        const int t0 = tb * TB;
        const float *xp = x + t0 * k.in_c + (k.depth_mult == 1 ? oc : oc / k.depth_mult);
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
struct NnfLayout { int w_floats, a_floats, b_floats, y_floats, fc_floats, vec_floats; };
__host__ __device__ __forceinline__ NnfLayout nnf_layout(const KwsNnPlanF32 &N)
{
    NnfLayout L = { 0, 0, 0, 0, 0, 0 };
    for (int b = 0; b < N.n_blocks; ++b) {
        const KwsConvBlockF32 &k = N.blk[b];
        L.w_floats += (k.depthwise ? k.taps : k.taps * k.in_c) * nnf_ocp(k);
        const int img = nnf_rows(k) * k.in_c;
        if (b & 1) L.b_floats = img > L.b_floats ? img : L.b_floats;
        else L.a_floats = img > L.a_floats ? img : L.a_floats;
        if (nnf_staged(k)) { const int yf = k.out_w * k.out_c; L.y_floats = yf > L.y_floats ? yf : L.y_floats; }
    }
    L.a_floats = (L.a_floats + 3) & ~3; L.b_floats = (L.b_floats + 3) & ~3; L.y_floats = (L.y_floats + 3) & ~3;
    // FULLY_CONNECTED weights + bias, shared by the workgroup (a global read per chain step would be an L2 round trip each);
    // per wave: the FC input vector (the last block's pooled output, also reused for the exponentials) + 64 floats of logits
    L.fc_floats = (N.fc_out * N.fc_in + N.fc_out + 3) & ~3;
    L.vec_floats = (((N.fc_in > 64 ? N.fc_in : 64) + 3) & ~3) + 64;
    return L;
}

template <int MAXT>     // threads per workgroup the build allows: 1024 (<= 128 VGPRs) or 512 (<= 256 VGPRs, vectorised conv steps)
__global__ __launch_bounds__(MAXT) void kws_nn_f32_kernel(const KwsNnPlanF32 *__restrict__ Np, const float *__restrict__ features,
                                                          int n_clips, float *__restrict__ scores,
                                                          float *__restrict__ tap_logits, long long *__restrict__ prof,
                                                          const int *__restrict__ sel)
{
    // the plan is read from memory (scalar loads, any block index); by value in the kernel arguments the compiler copies it to
    // scratch as soon as a block is indexed dynamically
    const KwsNnPlanF32 &N = *Np;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), n_waves = blockDim.x >> 6;
    // development aid: shader-clock totals per phase of wave 0 of workgroup 0 (input, each block, head)
    const bool profiling = prof != nullptr && blockIdx.x == 0 && wave == 0;
    long long ph[KWS_MAX_BLOCKS + 2] = { 0 }, tlast = profiling ? clock64() : 0;
    auto mark = [&](int i) { if (profiling) { const long long now = clock64(); ph[i] += now - tlast; tlast = now; } };
    // a workgroup none of whose waves has a clip (the empty re-run list of a KWS_MODE_FAST call, a short list) leaves before
    // the weights are staged: 16.5 us -> launch overhead for the empty list
    const int n_sel = sel_count(sel, n_clips);
    if ((int)blockIdx.x * n_waves >= n_sel) return;
    float *sp = (float *)smem_raw;
    int s_w_off[KWS_MAX_BLOCKS];     // float offsets into the LDS block: pointers kept in an array lose their address space (flat loads)
    for (int b = 0; b < N.n_blocks; ++b) {
        const KwsConvBlockF32 &k = N.blk[b];
        const int J = k.depthwise ? k.taps : k.taps * k.in_c, ocp = nnf_ocp(k);
        for (int i = threadIdx.x; i < J * ocp; i += blockDim.x) {      // [oc][j] -> [j][oc], zero in the padding channels
            const int j = i / ocp, oc = i - j * ocp;
            sp[nnf_w_index(k, J, j, oc)] = oc < k.out_c ? (k.depthwise ? k.w[j * k.out_c + oc] : k.w[oc * J + j]) : 0.0f;
        }
        s_w_off[b] = (int)(sp - (float *)smem_raw);
        sp += J * ocp;
    }
    const NnfLayout L = nnf_layout(N);
    const float *s_fcw = sp, *s_fcb = sp + N.fc_out * N.fc_in;
    for (int i = threadIdx.x; i < N.fc_out * N.fc_in; i += blockDim.x) sp[i] = N.fc_w[i];
    for (int i = threadIdx.x; i < N.fc_out; i += blockDim.x) sp[N.fc_out * N.fc_in + i] = N.fc_bias[i];
    sp += L.fc_floats;
    float *A = sp + wave * (L.a_floats + L.b_floats + L.y_floats + L.vec_floats);
    float *B = A + L.a_floats;
    float *Y = B + L.b_floats;
    float *vec = Y + L.y_floats;
    __syncthreads();

    // next-clip prefetch registers (256-register build only): NNF_PF x 64 float4 cover the first block's padded input image
    constexpr int NNF_PF = 9;
    float4 pf[NNF_PF];
#pragma unroll
    for (int u = 0; u < NNF_PF; ++u) pf[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    bool have_pf = false;
    const bool pf_ok = [&]() {
        const KwsConvBlockF32 &k = N.blk[0];
        const int lo = k.pad_left * k.in_c, hi = lo + k.in_w * k.in_c, tot = nnf_rows(k) * k.in_c;
        return ((lo | hi | N.n_features) & 3) == 0 && ((tot + 3) >> 2) <= 64 * NNF_PF && N.n_blocks > 1;
    }();
    for (int ci = blockIdx.x * n_waves + wave; ci < n_sel; ci += gridDim.x * n_waves) {
        const int clip = sel_clip(sel, ci);
        {
            const KwsConvBlockF32 &k = N.blk[0];
            const int lo = k.pad_left * k.in_c, hi = lo + k.in_w * k.in_c, tot = nnf_rows(k) * k.in_c;
            const float *src = features + (size_t)clip * N.n_features;
            if (have_pf) {
                // the feature vector was requested while the previous clip's tail blocks ran (see below): it only has to be
                // placed, zero padding rows included
                float4 *A4 = (float4 *)A;
                const int lo4 = lo >> 2, hi4 = hi >> 2, tot4 = (tot + 3) >> 2;
#pragma unroll
                for (int u = 0; u < NNF_PF; ++u) {
                    const int i = lane + 64 * u;
                    if (i < tot4) A4[i] = (i >= lo4 && i < hi4) ? pf[u] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            } else if (((lo | hi | N.n_features) & 3) == 0) {
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
            // a COPY of the block's parameters (scalar registers): through the reference every epilogue step reloads its clamp
            // bounds and strides from memory, because the LDS stores in between might alias the plan
            const KwsConvBlockF32 k = N.blk[b];
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
            case 8: nnf_conv_ob<8, (MAXT <= 512)>(k, cur, (const float *)smem_raw + s_w_off[b], Y, dst, lane); break;
            case 7: nnf_conv_ob<7, (MAXT <= 512)>(k, cur, (const float *)smem_raw + s_w_off[b], Y, dst, lane); break;
            case 4: nnf_conv_ob<4, (MAXT <= 512)>(k, cur, (const float *)smem_raw + s_w_off[b], Y, dst, lane); break;
            case 2: nnf_conv_ob<2, (MAXT <= 512)>(k, cur, (const float *)smem_raw + s_w_off[b], Y, dst, lane); break;
            default: nnf_conv_ob<1, (MAXT <= 512)>(k, cur, (const float *)smem_raw + s_w_off[b], Y, dst, lane); break;
            }
            WAVE_SYNC();
            if (b == 0 && MAXT <= 512 && pf_ok) {
                // block 0 (most of the clip's time) is done: request the NEXT clip's feature vector now, so that it arrives
                // while the short tail blocks, FC and softmax run (phases with little VALU work and nothing to prefetch)
                const int nci = ci + gridDim.x * n_waves;
                have_pf = nci < n_sel;
                if (have_pf) {
                    const int nclip = sel_clip(sel, nci);
                    const KwsConvBlockF32 &k0 = N.blk[0];
                    const int lo4 = (k0.pad_left * k0.in_c) >> 2, hi4 = lo4 + ((k0.in_w * k0.in_c) >> 2);
                    const float4 *src4 = (const float4 *)(features + (size_t)nclip * N.n_features);
#pragma unroll
                    for (int u = 0; u < NNF_PF; ++u) {
                        const int i = lane + 64 * u;
                        if (i >= lo4 && i < hi4) pf[u] = src4[i - lo4];
                    }
                }
            }
            if (nnf_staged(k)) {
                // MAX_POOL_2D over time (pooling.h:189-237) from the staged conv output
                for (int idx = lane; idx < n_out; idx += 64) {
                    const int pw = nnf_div(idx, k.out_c, k.inv_outc20), oc = idx - pw * k.out_c;
                    float mx = -FLT_MAX;
                    for (int q = 0; q < k.pool && pw * k.pool_stride + q < k.out_w; ++q) {      // the last window may be ragged (SAME)
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
        float *lg = vec + (L.vec_floats - 64), *ex = vec;          // ex overwrites the FC input once every lane is done with it
        const int fc_in = N.fc_in, fc_out = N.fc_out;
        const float beta = N.beta;
        if (lane < fc_out) {
            const float *fw = s_fcw + lane * fc_in;
            float total = 0.0f;
            int d = 0;
            for (; d + 4 <= fc_in; d += 4) {                  // four weights in flight per round trip; the chain stays in order
                const float w0 = fw[d], w1 = fw[d + 1], w2 = fw[d + 2], w3 = fw[d + 3];
                const float x0 = vec[d], x1 = vec[d + 1], x2 = vec[d + 2], x3 = vec[d + 3];
                const float p0 = x0 * w0, p1 = x1 * w1, p2 = x2 * w2, p3 = x3 * w3;
                total += p0; total += p1; total += p2; total += p3;
            }
            for (; d < fc_in; ++d) {
                const float prod = vec[d] * fw[d];
                total += prod;
            }
            const float lgt = act_clamp(total + s_fcb[lane], N.fc_min, N.fc_max);
            lg[lane] = lgt;
            if (tap_logits) tap_logits[(size_t)clip * fc_out + lane] = lgt;
        }
        WAVE_SYNC();
        // softmax.h:31-63: max, then sum += exp((x - max) * beta) in class order, then exp(...) / sum.  Each lane evaluates
        // its own class's exponential once; the sum adds the same values in the same order
        float e_own = 0.0f;
        if (lane < fc_out) {
            float mx = -FLT_MAX;
            for (int c = 0; c < fc_out; ++c) mx = mx < lg[c] ? lg[c] : mx;
            e_own = expf((lg[lane] - mx) * beta);
            ex[lane] = e_own;
        }
        WAVE_SYNC();
        if (lane < fc_out) {
            float sum = 0.0f;
            for (int c = 0; c < fc_out; ++c) sum += ex[c];
            scores[(size_t)clip * fc_out + lane] = e_own / sum;
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
    return ((size_t)L.w_floats + L.fc_floats + (size_t)n_waves * (L.a_floats + L.b_floats + L.y_floats + L.vec_floats)) * sizeof(float);
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
    k->ntb = nnf_ntb_of(*k);
    k->rows = nnf_rows_of(*k);
    // the kernel's item splits: items = time blocks x channel blocks (conv) or x out_c (depthwise, the pooling pass: out_w x out_c at most)
    const long n_ob = k->depthwise ? k->out_c : (k->out_c + k->ob - 1) / k->ob;
    // a reciprocal is used only where x * d < 2^20 (the quotient is then exact) AND x * inv fits nnf_div's 32-bit product (x / d below ~4096:
    // a single channel block with thousands of time blocks would pass the first test and overflow the second -- ADVICE round 5)
    auto recip20 = [](long x_max, long d) -> unsigned {
        const unsigned long inv = (1ul << 20) / (unsigned long)d + 1ul;
        return (x_max * d < (1L << 20) && (unsigned long)x_max * inv < (1ul << 32)) ? (unsigned)inv : 0u;
    };
    k->inv_item20 = recip20((long)k->ntb * n_ob + 64, n_ob);
    k->inv_outc20 = recip20((long)std::max(k->ntb * std::max(k->tb, 1), k->out_w) * k->out_c + 64, k->out_c);
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
                      float *tap_logits, int n_cu, hipStream_t stream, const int *sel)
{
    (void)hipGetLastError();      // the status returned below is this launch's, not a stale error of an earlier call
    if (n_clips <= 0) return 0;
    const int n_waves = kws_nn_f32_waves(N), grid_mult = 1;
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
                           scores, tap_logits, kws_dev_f32_prof, sel);
    else
        hipLaunchKernelGGL(kws_nn_f32_kernel<1024>, dim3(grid), dim3(KWS_WAVE * n_waves), smem, stream, d_plan, features, n_clips,
                           scores, tap_logits, kws_dev_f32_prof, sel);
    return (int)hipGetLastError();
}
