// kws_plan.cpp -- execution plans.  Everything the reference recomputes per clip on the CPU but that does not depend on the
// audio is computed HERE once per model, with the reference's own formulas and precisions (each builder cites its source),
// and uploaded to HBM.
#include "kws_internal.h"
#include "kws_dct_tables.h"
#include <cstring>

template <int NF>
static bool dct_tables_match(const std::vector<float2> &tw, const std::vector<float2> &stw, const std::vector<float> &cs,
                             const std::vector<float> &sn, float s0, float s1)
{
    typedef KwsDctTab<NF> T;
    auto same = [](float a, float b) { return std::memcmp(&a, &b, sizeof(float)) == 0; };      // bit for bit (signed zeros too)
    if ((int)tw.size() != NF / 2 || (int)stw.size() != NF / 4 || (int)cs.size() != NF / 2 + 1 || (int)sn.size() != NF / 2 + 1) return false;
    for (int i = 0; i < NF / 2; ++i) if (!same(tw[i].x, T::tw_r[i]) || !same(tw[i].y, T::tw_i[i])) return false;
    for (int i = 0; i < NF / 4; ++i) if (!same(stw[i].x, T::stw_r[i]) || !same(stw[i].y, T::stw_i[i])) return false;
    for (int i = 0; i < NF / 2 + 1; ++i) if (!same(cs[i], T::cs[i]) || !same(sn[i], T::sn[i])) return false;
    return same(s0, T::s0) && same(s1, T::s1);
}

EI_IMPULSE_ERROR build_dsp_plan(kws_handle *h)
{
    const Model &m = h->model;
    const DspCfg &c = m.dsp;
    KwsDspPlan &P = h->dsp;
    const uint32_t fs = m.frequency;
    // framing: processing.hpp:194-284
    const int frame_len = (int)roundf((float)fs * c.frame_length);
    const float stride_f = roundf((float)fs * c.frame_stride);
    const int stride = (int)stride_f;
    if (frame_len < 1 || stride_f < 1.0f || (size_t)frame_len > (size_t)m.raw_sample_count || m.raw_sample_count > (1u << 30))
        return fail(KWS_ERROR_UNSUPPORTED_MODEL, "framing: %u samples, frame length %d, stride %g", m.raw_sample_count, frame_len, (double)stride_f);
    const size_t diff = (size_t)m.raw_sample_count - (size_t)frame_len;
    const int nfr = (int)floorf((float)diff / stride_f);
    P.n_samples = (int)m.raw_sample_count;
    P.n_frames = nfr;
    P.frame_stride = stride;
    P.frame_len = frame_len;
    P.fft_len = c.fft_length;
    P.n_bins = c.fft_length / 2 + 1;
    P.n_filters = c.num_filters;
    // columns of the feature matrix: cepstra, or -- MFE block (extract_mfe_features) -- the mel filters themselves
    const bool mfe = c.block == DSP_BLOCK_MFE;
    const int ncep = mfe ? c.num_filters : c.num_cepstral;
    P.n_cepstral = ncep;
    P.win_size = c.win_size;
    P.pad = (int)(uint16_t)((c.win_size - 1) / 2);
    P.pre_shift = c.pre_shift;
    P.pre_cof = mfe ? 0.0f : c.pre_cof;        // the MFE block hands the raw signal to feature::mfe (L432 ei_run_dsp.h:398-400)
    P.inv_fft = (float)(1.0 / (double)(float)c.fft_length);
    const int N = c.num_filters;
    P.dct_s0 = sqrtf(1.0f / (float)(4 * N));
    P.dct_s1 = sqrtf(1.0f / (float)(2 * N));

    // What the tuned kernel (kws_mfcc.hip) is instantiated for: fft 256, 32 or 40 filters, up to 52 frames, 16-byte aligned
    // frames, short mel filters.  Everything else the reference accepts goes to the general kernels (kws_generic.hip).
    if (c.axes != 1) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "MFCC block with %d axes", c.axes);
    const int used = std::min(frame_len, c.fft_length);
    if (c.fft_length < 4 || (c.fft_length & 1) || N < 2 || (N & 1) || N > 128 || c.pre_shift != 1 || (c.win_size & 1) == 0 || c.win_size < 1 ||
        nfr < 1 || stride < 1 || ncep < 1 || ncep > N || (long)(nfr - 1) * stride + used > (long)P.n_samples ||
        (size_t)nfr * ncep != m.nn_input_frame_size)
        return fail(KWS_ERROR_UNSUPPORTED_MODEL, "MFCC configuration outside the kernels (fft %d, filters %d, frames %d, frame_len %d, stride %d, "
                    "cepstra %d, win %d, shift %d)", c.fft_length, N, nfr, frame_len, stride, ncep, c.win_size, c.pre_shift);
    auto factor = [](int n, int *fac) {                         // kf_factor: 4s, then 2s, then 3, 5, 7 ...; returns levels or -1
        int p = 4, levels = 0;
        const double floor_sqrt = floor(sqrt((double)n));
        do {
            while (n % p) {
                switch (p) { case 4: p = 2; break; case 2: p = 3; break; default: p += 2; break; }
                if (p > floor_sqrt) p = n;
            }
            n /= p;
            if (p > 5 || levels >= 12) return -1;               // kf_bfly_generic (other primes) is not restated
            fac[2 * levels] = p; fac[2 * levels + 1] = n;
            levels++;
        } while (n > 1);
        return levels;
    };
    P.fft_levels = factor(c.fft_length / 2, P.fft_fac);
    P.dct_levels = factor(N / 2, P.dct_fac);
    if (P.fft_levels < 0 || P.dct_levels < 0)
        return fail(KWS_ERROR_UNSUPPORTED_MODEL, "fft_length %d / %d filters need a radix other than 2, 3, 4, 5", c.fft_length, N);
    const bool tuned = c.fft_length == 256 && (N == 32 || N == 40) && frame_len >= c.fft_length && nfr <= kws_mfcc_max_frames(N) &&
                       (stride * 2) % 16 == 0 && (P.n_samples * 2) % 16 == 0 && nfr + 2 * P.pad <= kws_mfcc_max_prow() &&
                       c.win_size <= kws_mfcc_max_win(ncep) && nfr <= kws_mfcc_max_frames_for(N, ncep) &&
                       c.win_size >= ((N == 40 && ncep > 16) ? 17 : 13) && nfr <= 4 * kws_mfcc_cmvn_rows();
    P.generic = tuned ? 0 : 1;
    P.spectral_tuned = (!tuned && !mfe && c.fft_length == 256 && (N == 32 || N == 40) && frame_len >= c.fft_length && (stride * 2) % 16 == 0 &&
                        (P.n_samples * 2) % 16 == 0) ? 1 : 0;
    P.wrap_index = P.n_samples - 1;
    if (mfe && (!tuned || nfr > (N > 16 ? 51 : 52) || c.win_size < (N > 16 ? 17 : 13)))
        return fail(KWS_ERROR_UNSUPPORTED_MODEL, "MFE block: %d frames x %d filters, window %d outside the normalisation kernel's limits (tuned "
                    "configurations only)", nfr, N, c.win_size);

    std::vector<float2> tw, stw, dtw, dstw;
    h_twiddles(c.fft_length / 2, tw);
    h_super_twiddles(c.fft_length / 2, stw);
    h_twiddles(N / 2, dtw);
    h_super_twiddles(N / 2, dstw);
    std::vector<float> dcos(N / 2 + 1), dsin(N / 2 + 1);
    for (int i = 0; i < N / 2 + 1; i++) {                       // fast-dct-fft.cpp:71-74
        float temp = (float)((double)i * M_PI / (double)(N * 2));
        dcos[i] = cosf(temp);
        dsin[i] = sinf(temp);
    }
    // the kernel carries the DCT constants as literals (kws_dct_tables.h, tools/gen_dct_tables.cpp): they must be exactly
    // what this host computes in the reference's way, or the features would silently differ
    if (!P.generic) {
        bool same;
        if (N == 32) same = dct_tables_match<32>(dtw, dstw, dcos, dsin, P.dct_s0, P.dct_s1);
        else same = dct_tables_match<40>(dtw, dstw, dcos, dsin, P.dct_s0, P.dct_s1);
        if (!same) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "DCT constants of the kernel build differ from this host's (%d filters)", N);
    }
    const uint32_t high = c.high_frequency == 0 ? fs / 2 : (uint32_t)c.high_frequency;   // feature.hpp:203-205
    std::vector<float> fb = h_filterbank(N, P.n_bins, fs, (uint32_t)c.low_frequency, high, c.quantize_fb != 0);
    std::vector<int> fstart(N + 1, 0), fbin;
    std::vector<float> fw;
    int max_nz = 0;
    for (int j = 0; j < N; j++) {
        fstart[j] = (int)fbin.size();
        for (int k = 0; k < P.n_bins; k++) {
            const float w = fb[(size_t)k * N + j];
            if (w != 0.0f) { fbin.push_back(k); fw.push_back(w); }     // zero weights add an exact +0: skipped
        }
        max_nz = std::max(max_nz, (int)fbin.size() - fstart[j]);
    }
    fstart[N] = (int)fbin.size();
    P.max_nz = max_nz;
    P.filt_nnz = (int)fbin.size();
    if (max_nz > kws_mfcc_max_nz()) { P.generic = 1; P.spectral_tuned = 0; }      // the tuned kernel keeps a filter's taps in registers
    std::vector<int> pmap;
    h_pad_map(nfr, P.pad, pmap);

    EI_IMPULSE_ERROR e;
    if ((e = h->upload(tw, &P.tw))) return e;
    if ((e = h->upload(stw, &P.stw))) return e;
    if ((e = h->upload(dtw, &P.dct_tw))) return e;
    if ((e = h->upload(dstw, &P.dct_stw))) return e;
    if ((e = h->upload(dcos, &P.dct_cos))) return e;
    if ((e = h->upload(dsin, &P.dct_sin))) return e;
    if ((e = h->upload(fstart, &P.filt_start))) return e;
    if ((e = h->upload(fbin, &P.filt_bin))) return e;
    if ((e = h->upload(fw, &P.filt_w))) return e;
    if ((e = h->upload(pmap, &P.pad_map))) return e;
    return EI_IMPULSE_OK;
}

int kws_frames_for_length(const kws_handle *h, size_t n)
{
    const KwsDspPlan &P = h->dsp;
    const float stride_f = roundf((float)h->model.frequency * h->model.dsp.frame_stride);          // as build_dsp_plan
    if (n < (size_t)P.frame_len || n > (1u << 30)) return 0;
    return (int)floorf((float)(n - (size_t)P.frame_len) / stride_f);
}

EI_IMPULSE_ERROR kws_plan_for_length(kws_handle *h, size_t n, KwsDspPlan *out)
{
    const int nfr = kws_frames_for_length(h, n);
    if (nfr < 1 || nfr > h->dsp.n_frames) return fail(KWS_ERROR_BAD_ARGUMENT, "a window of %zu samples has %d frames (the model's: %d)", n, nfr, h->dsp.n_frames);
    *out = h->dsp;
    out->n_samples = (int)n;
    out->n_frames = nfr;
    if (nfr != h->dsp.n_frames) {
        auto it = h->pad_maps_by_rows.find(nfr);
        if (it == h->pad_maps_by_rows.end()) {
            std::vector<int> pmap;
            h_pad_map(nfr, out->pad, pmap);
            const int *d = nullptr;
            if (EI_IMPULSE_ERROR e = h->upload(pmap, &d)) return e;
            it = h->pad_maps_by_rows.emplace(nfr, d).first;
        }
        out->pad_map = it->second;
    }
    return EI_IMPULSE_OK;
}

static EI_IMPULSE_ERROR build_nn_plan_f32(kws_handle *h);

// Recognise the Edge Impulse 1-D CNN family and fold its per-model constants.
EI_IMPULSE_ERROR build_nn_plan(kws_handle *h)
{
    const Model &m = h->model;
    KwsNnPlan &N = h->nn;
    memset(&N, 0, sizeof(N));
    const Tensor &tin = m.t[m.t_in], &tout = m.t[m.t_out];
    if (tin.type == TYPE_F32 && tout.type == TYPE_F32) return build_nn_plan_f32(h);
    if (tin.type != TYPE_I8 || tout.type != TYPE_I8 || tin.scale.empty() || tout.scale.empty())
        return fail(KWS_ERROR_UNSUPPORTED_MODEL, "only int8-quantised and float32 models are implemented");
    N.n_features = (int)m.nn_input_frame_size;
    N.in_scale = tin.scale[0]; N.in_zp = tin.zero[0];
    N.out_scale = tout.scale[0]; N.out_zp = tout.zero[0];
    N.n_labels = (int)m.labels.size();

    int cur = (int)m.t_in;       // tensor currently flowing through the graph
    int cur_w = 0, cur_c = 0;    // logical [time][channel] shape once known
    size_t i = 0;
    auto same_quant = [&](int a, int b) {
        return !m.t[a].scale.empty() && !m.t[b].scale.empty() && m.t[a].scale[0] == m.t[b].scale[0] && m.t[a].zero[0] == m.t[b].zero[0];
    };
    auto skip_reshapes = [&]() {
        while (i < m.n.size() && m.n[i].op == OP_RESHAPE && m.n[i].in[0] == cur) {
            if (!same_quant(cur, m.n[i].out[0])) return false;
            cur = m.n[i].out[0];
            i++;
        }
        return true;
    };
    if (!skip_reshapes()) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "reshape changes quantisation");
    while (i < m.n.size() && (m.n[i].op == OP_CONV_2D || m.n[i].op == OP_DEPTHWISE_CONV_2D)) {
        if (N.n_blocks >= KWS_MAX_BLOCKS) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "more than %d conv blocks", KWS_MAX_BLOCKS);
        const Node &cv = m.n[i];
        const bool dw = cv.op == OP_DEPTHWISE_CONV_2D;
        if (cv.in[0] != cur || cv.in.size() < 2) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "conv input is not the flowing tensor");
        const Tensor &x = m.t[cur], &w = m.t[cv.in[1]], &y = m.t[cv.out[0]];
        const Tensor *bias = (cv.in.size() > 2 && cv.in[2] >= 0) ? &m.t[cv.in[2]] : nullptr;
        const int in_h = x.dim4(1), in_w = x.dim4(2), in_c = x.dim4(3);
        const int out_c = dw ? w.dim4(3) : w.dim4(0), f_h = w.dim4(1), f_w = w.dim4(2);
        const int depth_mult = dw ? cv.p[6] : 1;
        const int padding = cv.p[0], stride_w = cv.p[1], stride_h = cv.p[2], act = cv.p[3], dil_w = cv.p[4], dil_h = cv.p[5];
        if (x.dim4(0) != 1 || in_h != 1 || f_h != 1 || stride_w != 1 || stride_h != 1 || dil_w != 1 || dil_h != 1 || !w.is_const ||
            (dw ? (w.dim4(0) != 1 || depth_mult < 1 || out_c != in_c * depth_mult) : (w.dim4(3) != in_c)) || (bias && !bias->is_const) ||
            f_w > 16 || out_c > 64 || x.dims.size() != 4 || (size_t)w.nbytes != (size_t)out_c * f_w * (dw ? 1 : in_c))
            return fail(KWS_ERROR_UNSUPPORTED_MODEL, "conv %zu is not a stride-1 1xK (depthwise) convolution over time", i);
        if (cur_w && (cur_w != in_w || cur_c != in_c)) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "shape mismatch into conv %zu", i);
        if (bias && (bias->type != TYPE_I32 || (int)bias->nbytes != out_c * 4)) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "conv %zu bias", i);
        if (w.type != TYPE_I8 || x.type != TYPE_I8 || y.type != TYPE_I8) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "conv %zu tensor types", i);
        const int out_w = h_out_size(padding, in_w, f_w, 1, 1);
        const int pad_left = h_pad_amount(1, 1, in_w, f_w, out_w);
        if (out_w != y.dim4(2) || y.dim4(3) != out_c) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "conv %zu output shape", i);
        KwsConvBlock &k = N.blk[N.n_blocks];
        k.in_w = in_w; k.in_c = in_c; k.in_cpad = (in_c + 15) & ~15; k.out_c = out_c; k.taps = f_w; k.pad_left = pad_left;
        k.out_w = out_w; k.in_zp = x.zero[0]; k.out_zp = y.zero[0];
        k.depthwise = dw ? 1 : 0; k.depth_mult = depth_mult;
        h_act_range(act, y.scale[0], y.zero[0], &k.act_min, &k.act_max);
        // the reference's int8 depthwise op clamps to the int8 range whatever its fused activation says
        // (TFL/micro/kernels/depthwise_conv.cc:618-620, "TODO(b/130439627)") -- pinned in tests/test_oracle_vs_reference.py
        if (dw) { k.act_min = -128; k.act_max = 127; }
        // weights -> [out_c][taps][in_cpad] (depthwise: [out_c][taps padded to 4], from the filter's [taps][out_c]);
        // bias_eff = bias + input_offset * sum(w)   (integer_ops/conv.h:64-113, depthwise_conv.h:64-106)
        const int tp4 = (f_w + 3) & ~3;
        std::vector<int8_t> wp(dw ? (size_t)out_c * tp4 : (size_t)out_c * f_w * k.in_cpad, 0);
        k.w_bytes = (int)wp.size();
        std::vector<int32_t> beff(out_c), mult(out_c), shift(out_c);
        const int8_t *wd = (const int8_t *)w.data.data();
        const int32_t in_off = -x.zero[0];
        const bool per_channel = w.scale.size() > 1;
        if (per_channel && (int)w.scale.size() != out_c) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "per-channel scale count");
        for (int oc = 0; oc < out_c; oc++) {
            int64_t wsum = 0;
            for (int tap = 0; tap < f_w; tap++) {
                if (dw) {
                    const int8_t v = wd[(size_t)tap * out_c + oc];
                    wp[(size_t)oc * tp4 + tap] = v;
                    wsum += v;
                    continue;
                }
                for (int c = 0; c < in_c; c++) {
                    const int8_t v = wd[((size_t)oc * f_w + tap) * in_c + c];
                    wp[((size_t)oc * f_w + tap) * k.in_cpad + c] = v;
                    wsum += v;
                }
            }
            beff[oc] = (int32_t)((bias ? ((const int32_t *)bias->data.data())[oc] : 0) + (int64_t)in_off * wsum);
            const double eff = (double)x.scale[0] * (double)(per_channel ? w.scale[oc] : w.scale[0]) / (double)y.scale[0];
            int sh;
            h_quantize_multiplier(eff, &mult[oc], &sh);          // kernel_util_lite.cc:89-103
            shift[oc] = sh;
            if (mult[oc] < 0) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "negative requantisation multiplier");
        }
        cur = cv.out[0]; cur_w = out_w; cur_c = out_c;
        i++;
        if (!skip_reshapes()) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "reshape changes quantisation");
        // optional ADD(const per-channel tensor) with fused activation -> 256-entry table per channel
        std::vector<int8_t> lut((size_t)out_c * 256);
        for (int c = 0; c < out_c; c++) for (int v = 0; v < 256; v++) lut[(size_t)c * 256 + v] = (int8_t)(v - 128);
        k.has_lut = 0;
        if (i < m.n.size() && m.n[i].op == OP_ADD) {
            k.has_lut = 1;
            const Node &ad = m.n[i];
            int a_id = ad.in[0], b_id = ad.in[1];
            if (a_id != cur && b_id == cur) std::swap(a_id, b_id);
            const Tensor &t1 = m.t[ad.in[0]], &t2 = m.t[ad.in[1]], &to = m.t[ad.out[0]];
            const Tensor &cb = m.t[b_id];
            if (a_id != cur || !cb.is_const || cb.type != TYPE_I8 || (int)cb.nbytes != out_c || cb.dims.back() != out_c ||
                t1.type != TYPE_I8 || t2.type != TYPE_I8 || to.type != TYPE_I8)
                return fail(KWS_ERROR_UNSUPPORTED_MODEL, "ADD %zu is not a per-channel constant int8 add", i);
            // CalculateOpData, add.cc:271-309
            const int left_shift = 20;
            const double twice_max = 2 * (double)std::max(t1.scale[0], t2.scale[0]);
            int32_t m1, m2, mo; int s1, s2, so;
            h_quantize_multiplier((double)t1.scale[0] / twice_max, &m1, &s1);
            h_quantize_multiplier((double)t2.scale[0] / twice_max, &m2, &s2);
            h_quantize_multiplier(twice_max / ((double)(1 << left_shift) * (double)to.scale[0]), &mo, &so);
            int32_t amin, amax;
            h_act_range(ad.p[0], to.scale[0], to.zero[0], &amin, &amax);
            const bool cur_is_first = (ad.in[0] == cur);
            for (int c = 0; c < out_c; c++) {
                const int8_t cval = ((const int8_t *)cb.data.data())[c];
                int prev = -129;
                for (int v = -128; v <= 127; v++) {
                    const int32_t x1 = cur_is_first ? v : cval, x2 = cur_is_first ? cval : v;
                    const int32_t v1 = -t1.zero[0] + x1, v2 = -t2.zero[0] + x2;           // integer_ops/add.h:108-131
                    const int32_t q1 = h_rdivpot(h_srdhm(v1 * (1 << left_shift), m1), -s1);
                    const int32_t q2 = h_rdivpot(h_srdhm(v2 * (1 << left_shift), m2), -s2);
                    int32_t o = h_rdivpot(h_srdhm(q1 + q2, mo), -so) + to.zero[0];
                    o = std::min(amax, std::max(amin, o));
                    if (o < prev) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "ADD table not monotonic");
                    prev = o;
                    lut[(size_t)c * 256 + (v + 128)] = (int8_t)o;
                }
            }
            cur = ad.out[0];
            i++;
            if (!skip_reshapes()) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "reshape changes quantisation");
        }
        // MAX_POOL_2D over time (optional: pool 1 = identity)
        k.pool = 1; k.pool_stride = 1; k.pool_w = out_w;
        if (i < m.n.size() && m.n[i].op == OP_MAX_POOL_2D) {
            const Node &pl = m.n[i];
            const Tensor &px = m.t[pl.in[0]], &py = m.t[pl.out[0]];
            if (pl.in[0] != cur || !same_quant(cur, pl.out[0])) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "pool %zu input", i);
            const int ph = px.dim4(1), pw = px.dim4(2);
            int f, s, out_n;
            if (pw == 1 && ph == cur_w) { f = pl.p[4]; s = pl.p[2]; if (pl.p[3] != 1) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "2-D pool"); out_n = py.dim4(1); }
            else if (ph == 1 && pw == cur_w) { f = pl.p[3]; s = pl.p[1]; if (pl.p[4] != 1) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "2-D pool"); out_n = py.dim4(2); }
            else return fail(KWS_ERROR_UNSUPPORTED_MODEL, "pool %zu is not over the time axis", i);
            const int po = h_out_size(pl.p[0], cur_w, f, s, 1);
            if (po != out_n || h_pad_amount(s, 1, cur_w, f, po) != 0 || (po - 1) * s >= cur_w || f > 8 || pl.p[5] != 0)
                return fail(KWS_ERROR_UNSUPPORTED_MODEL, "pool %zu window/padding outside the kernel's limits", i);
            k.pool = f; k.pool_stride = s; k.pool_w = po;
            cur = pl.out[0]; cur_w = po;
            i++;
            if (!skip_reshapes()) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "reshape changes quantisation");
        }
        // un-pooled CONV_2D blocks (e.g. the pointwise halves of a depthwise-separable CNN) run on the matrix cores in the
        // generic kernel: 16-byte k-groups need activation rows of 16, 32 or 64 bytes
        k.mfma = (!dw && k.pool == 1 && in_c <= 64 && out_c <= 32 && out_w <= 64 && f_w <= 8) ? 1 : 0;
        if (k.mfma) {
            const int cp = in_c <= 16 ? 16 : in_c <= 32 ? 32 : 64;
            if (cp != k.in_cpad) {
                std::vector<int8_t> w2((size_t)out_c * f_w * cp, 0);
                for (int r = 0; r < out_c * f_w; r++) std::memcpy(&w2[(size_t)r * cp], &wp[(size_t)r * k.in_cpad], (size_t)in_c);
                wp.swap(w2);
                k.in_cpad = cp;
                k.w_bytes = (int)wp.size();
            }
        }
        EI_IMPULSE_ERROR e;
        if ((e = h->upload(wp, &k.w))) return e;
        if ((e = h->upload(beff, &k.bias_eff))) return e;
        if ((e = h->upload(mult, &k.mult))) return e;
        if ((e = h->upload(shift, &k.shift))) return e;
        if ((e = h->upload(lut, &k.add_lut))) return e;
        h->pooled_tap_bytes += k.pool_w * k.out_c;
        N.n_blocks++;
    }
    if (N.n_blocks == 0) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "graph does not start with a convolution block");
    if (N.blk[0].in_w * N.blk[0].in_c != N.n_features) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "first conv does not consume the feature vector");
    for (int b = 0; b + 1 < N.n_blocks; b++)
        if (N.blk[b].pool_w != N.blk[b + 1].in_w || N.blk[b].out_c != N.blk[b + 1].in_c)
            return fail(KWS_ERROR_UNSUPPORTED_MODEL, "conv blocks do not chain");
    // FULLY_CONNECTED (fully_connected.cc:322-396)
    if (i < m.n.size() && m.n[i].op == OP_FULLY_CONNECTED && (m.t[m.n[i].in[1]].type != TYPE_I8 || m.t[m.n[i].out[0]].type != TYPE_I8))
        return fail(KWS_ERROR_UNSUPPORTED_MODEL, "FULLY_CONNECTED tensor types");
    if (i >= m.n.size() || m.n[i].op != OP_FULLY_CONNECTED || m.n[i].in[0] != cur)
        return fail(KWS_ERROR_UNSUPPORTED_MODEL, "expected FULLY_CONNECTED after the conv blocks");
    {
        const Node &fc = m.n[i];
        const Tensor &x = m.t[cur], &w = m.t[fc.in[1]], &y = m.t[fc.out[0]];
        const Tensor *bias = (fc.in.size() > 2 && fc.in[2] >= 0) ? &m.t[fc.in[2]] : nullptr;
        N.fc_in = w.dims.back(); N.fc_out = w.dims[0];
        const KwsConvBlock &lb = N.blk[N.n_blocks - 1];
        if (N.fc_in != lb.pool_w * lb.out_c || N.fc_in > KWS_FC_IN_MAX || N.fc_in * N.fc_out > KWS_FC_W_MAX || N.fc_out > 48 || N.fc_out != N.n_labels || !w.is_const ||
            w.type != TYPE_I8 || (int)w.nbytes != N.fc_in * N.fc_out || y.type != TYPE_I8 ||
            (bias && (!bias->is_const || bias->type != TYPE_I32 || (int)bias->nbytes != N.fc_out * 4)))
            return fail(KWS_ERROR_UNSUPPORTED_MODEL, "FULLY_CONNECTED shape %dx%d outside the kernel's limits", N.fc_out, N.fc_in);
        N.fc_in_off = -x.zero[0]; N.fc_w_off = -w.zero[0]; N.fc_out_zp = y.zero[0];
        const double in_prod = (double)(x.scale[0] * w.scale[0]);     // kernel_util_lite.cc:160-172 (float product)
        int32_t mult; int exponent;
        h_quantize_multiplier(in_prod / (double)y.scale[0], &mult, &exponent);
        N.fc_mult = mult; N.fc_shift = exponent;
        h_act_range(fc.p[0], y.scale[0], y.zero[0], &N.fc_act_min, &N.fc_act_max);
        std::vector<int8_t> wv((const int8_t *)w.data.data(), (const int8_t *)w.data.data() + w.nbytes);
        std::vector<int32_t> bv(N.fc_out, 0);
        if (bias) memcpy(bv.data(), bias->data.data(), sizeof(int32_t) * N.fc_out);
        EI_IMPULSE_ERROR e;
        if ((e = h->upload(wv, &N.fc_w))) return e;
        if ((e = h->upload(bv, &N.fc_bias))) return e;
        cur = fc.out[0];
        i++;
    }
    // SOFTMAX (softmax.cc:187-226): everything but the final reciprocal/rescale depends only on (max - x)
    if (i >= m.n.size() || m.n[i].op != OP_SOFTMAX || m.n[i].in[0] != cur || m.n[i].out[0] != (int)m.t_out || i + 1 != m.n.size())
        return fail(KWS_ERROR_UNSUPPORTED_MODEL, "expected a final SOFTMAX");
    {
        const Node &sn = m.n[i];
        const Tensor &x = m.t[cur];
        if (tout.scale[0] != 1.f / 256 || tout.zero[0] != -128) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "softmax output quantisation");
        double rm = (double)sn.beta * (double)x.scale[0] * (double)(1 << (31 - 5));
        rm = std::min(rm, (double)((1ll << 31) - 1.0));
        int32_t mult; int left_shift;
        h_quantize_multiplier(rm, &mult, &left_shift);
        const double max_in = 1.0 * ((1 << 5) - 1) * (double)(1ll << (31 - 5)) / (double)(1ll << left_shift);
        const int diff_min = (int)(-1.0 * (double)(int)floor(max_in));
        std::vector<int32_t> ex(256);
        std::vector<uint8_t> valid(256);
        for (int d = 0; d < 256; d++) {
            const int32_t diff = -d;
            valid[d] = diff >= diff_min;
            const int32_t resc = h_srdhm((int32_t)((uint32_t)diff * (1u << left_shift)), mult);
            ex[d] = valid[d] ? h_exp_neg_q5_26(resc) : 0;
        }
        EI_IMPULSE_ERROR e;
        if ((e = h->upload(ex, &N.sm_exp))) return e;
        if ((e = h->upload(valid, &N.sm_valid))) return e;
    }
    // the generic kernel keeps every block's weights and ADD tables in LDS: a model that does not fit fails here, at load time,
    // not at its first inference
    if (!kws_nn_uses_mfma(N) && kws_nn_smem_bytes(N, 4) > 158 * 1024)
        return fail(KWS_ERROR_UNSUPPORTED_MODEL, "int8 model needs %zu B of LDS in the generic network kernel (at most %d)",
                    kws_nn_smem_bytes(N, 4), 158 * 1024);
    return EI_IMPULSE_OK;
}

// The same graph family with float32 tensors (the "fp32" configuration of BASELINE.json): constants are uploaded as they
// are, the fused activations become clamp ranges (kernel_util_lite.h:79-97 CalculateActivationRange<float>).
static void h_act_range_f32(int act, float *lo, float *hi)
{
    *lo = -FLT_MAX; *hi = FLT_MAX;                       // kTfLiteActNone: numeric_limits lowest()/max()
    if (act == 1) { *lo = 0.f; }                         // kTfLiteActRelu
    else if (act == 2) { *lo = -1.f; *hi = 1.f; }        // kTfLiteActReluN1To1
    else if (act == 3) { *lo = 0.f; *hi = 6.f; }         // kTfLiteActRelu6
}

static EI_IMPULSE_ERROR build_nn_plan_f32(kws_handle *h)
{
    const Model &m = h->model;
    KwsNnPlanF32 &N = h->nnf;
    memset(&N, 0, sizeof(N));
    h->is_float = true;
    memset(&h->nn, 0, sizeof(h->nn));
    h->nn.in_scale = 1.0f;                               // never used for a result: float models have no int8 input tensor
    h->nn.n_features = (int)m.nn_input_frame_size;
    N.n_features = (int)m.nn_input_frame_size;
    N.n_labels = (int)m.labels.size();
    for (const Tensor &t : m.t)
        if (t.type != TYPE_F32 && !(t.type == TYPE_I32 && t.is_const))     // int32 constants: RESHAPE shape operands
            return fail(KWS_ERROR_UNSUPPORTED_MODEL, "mixed float/integer graphs are not implemented");

    int cur = (int)m.t_in, cur_w = 0, cur_c = 0;
    size_t i = 0;
    auto skip_reshapes = [&]() {
        while (i < m.n.size() && m.n[i].op == OP_RESHAPE && m.n[i].in[0] == cur) { cur = m.n[i].out[0]; i++; }
    };
    auto floats = [](const Tensor &t) { return std::vector<float>((const float *)t.data.data(), (const float *)t.data.data() + t.nbytes / 4); };
    skip_reshapes();
    while (i < m.n.size() && (m.n[i].op == OP_CONV_2D || m.n[i].op == OP_DEPTHWISE_CONV_2D)) {
        if (N.n_blocks >= KWS_MAX_BLOCKS) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "more than %d conv blocks", KWS_MAX_BLOCKS);
        const Node &cv = m.n[i];
        const bool dw = cv.op == OP_DEPTHWISE_CONV_2D;
        if (cv.in[0] != cur || cv.in.size() < 2) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "conv input is not the flowing tensor");
        const Tensor &x = m.t[cur], &w = m.t[cv.in[1]], &y = m.t[cv.out[0]];
        const Tensor *bias = (cv.in.size() > 2 && cv.in[2] >= 0) ? &m.t[cv.in[2]] : nullptr;
        const int in_h = x.dim4(1), in_w = x.dim4(2), in_c = x.dim4(3);
        const int out_c = dw ? w.dim4(3) : w.dim4(0), f_h = w.dim4(1), f_w = w.dim4(2);
        const int depth_mult = dw ? cv.p[6] : 1;
        const int padding = cv.p[0], stride_w = cv.p[1], stride_h = cv.p[2], act = cv.p[3], dil_w = cv.p[4], dil_h = cv.p[5];
        if (x.dim4(0) != 1 || in_h != 1 || f_h != 1 || stride_w != 1 || stride_h != 1 || dil_w != 1 || dil_h != 1 || !w.is_const ||
            (dw ? (w.dim4(0) != 1 || depth_mult < 1 || out_c != in_c * depth_mult) : (w.dim4(3) != in_c)) || (bias && !bias->is_const) ||
            f_w > 16 || out_c > 64 || x.dims.size() != 4 || (size_t)w.nbytes != sizeof(float) * out_c * f_w * (dw ? 1 : in_c))
            return fail(KWS_ERROR_UNSUPPORTED_MODEL, "conv %zu is not a stride-1 1xK (depthwise) convolution over time", i);
        if (cur_w && (cur_w != in_w || cur_c != in_c)) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "shape mismatch into conv %zu", i);
        const int out_w = h_out_size(padding, in_w, f_w, 1, 1);
        const int pad_left = h_pad_amount(1, 1, in_w, f_w, out_w);
        if (out_w != y.dim4(2) || y.dim4(3) != out_c) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "conv %zu output shape", i);
        KwsConvBlockF32 &k = N.blk[N.n_blocks];
        k.in_w = in_w; k.in_c = in_c; k.out_c = out_c; k.taps = f_w; k.pad_left = pad_left; k.out_w = out_w;
        k.depthwise = dw ? 1 : 0; k.depth_mult = depth_mult;
        h_act_range_f32(act, &k.conv_min, &k.conv_max);           // the float depthwise op honours its activation
        std::vector<float> wv = floats(w), bv(out_c, 0.0f), av(out_c, 0.0f);
        if ((int)wv.size() != out_c * f_w * (dw ? 1 : in_c)) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "conv %zu filter size", i);
        if (bias) { if ((int)bias->nbytes != out_c * 4) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "conv %zu bias size", i); bv = floats(*bias); }
        cur = cv.out[0]; cur_w = out_w; cur_c = out_c;
        i++;
        skip_reshapes();
        k.has_add = 0;
        h_act_range_f32(0, &k.add_min, &k.add_max);
        if (i < m.n.size() && m.n[i].op == OP_ADD) {
            const Node &ad = m.n[i];
            int a_id = ad.in[0], b_id = ad.in[1];
            if (a_id != cur && b_id == cur) std::swap(a_id, b_id);
            const Tensor &cb = m.t[b_id];
            if (a_id != cur || !cb.is_const || (int)cb.nbytes != out_c * 4 || cb.dims.back() != out_c)
                return fail(KWS_ERROR_UNSUPPORTED_MODEL, "ADD %zu is not a per-channel constant add", i);
            av = floats(cb);                              // x + c == c + x in IEEE arithmetic: operand order is immaterial
            k.has_add = 1;
            h_act_range_f32(ad.p[0], &k.add_min, &k.add_max);
            cur = ad.out[0];
            i++;
            skip_reshapes();
        }
        k.pool = 1; k.pool_stride = 1; k.pool_w = out_w;
        h_act_range_f32(0, &k.pool_min, &k.pool_max);
        if (i < m.n.size() && m.n[i].op == OP_MAX_POOL_2D) {
            const Node &pl = m.n[i];
            const Tensor &px = m.t[pl.in[0]], &py = m.t[pl.out[0]];
            if (pl.in[0] != cur) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "pool %zu input", i);
            const int ph = px.dim4(1), pw = px.dim4(2);
            int f, s, out_n;
            if (pw == 1 && ph == cur_w) { f = pl.p[4]; s = pl.p[2]; if (pl.p[3] != 1) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "2-D pool"); out_n = py.dim4(1); }
            else if (ph == 1 && pw == cur_w) { f = pl.p[3]; s = pl.p[1]; if (pl.p[4] != 1) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "2-D pool"); out_n = py.dim4(2); }
            else return fail(KWS_ERROR_UNSUPPORTED_MODEL, "pool %zu is not over the time axis", i);
            const int po = h_out_size(pl.p[0], cur_w, f, s, 1);
            if (po != out_n || h_pad_amount(s, 1, cur_w, f, po) != 0 || (po - 1) * s >= cur_w || f > 8)
                return fail(KWS_ERROR_UNSUPPORTED_MODEL, "pool %zu window/padding outside the kernel's limits", i);
            k.pool = f; k.pool_stride = s; k.pool_w = po;
            h_act_range_f32(pl.p[5], &k.pool_min, &k.pool_max);
            cur = pl.out[0]; cur_w = po;
            i++;
            skip_reshapes();
        }
        EI_IMPULSE_ERROR e;
        if ((e = h->upload(wv, &k.w))) return e;
        if ((e = h->upload(bv, &k.bias))) return e;
        if ((e = h->upload(av, &k.addc))) return e;
        h->hostf.w[N.n_blocks] = wv; h->hostf.bias[N.n_blocks] = bv; h->hostf.addc[N.n_blocks] = av;
        kws_nn_f32_pick_blocking(&k);
        N.n_blocks++;
    }
    if (N.n_blocks == 0) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "graph does not start with a convolution block");
    if (N.blk[0].in_w * N.blk[0].in_c != N.n_features) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "first conv does not consume the feature vector");
    for (int b = 0; b + 1 < N.n_blocks; b++)
        if (N.blk[b].pool_w != N.blk[b + 1].in_w || N.blk[b].out_c != N.blk[b + 1].in_c)
            return fail(KWS_ERROR_UNSUPPORTED_MODEL, "conv blocks do not chain");
    if (i >= m.n.size() || m.n[i].op != OP_FULLY_CONNECTED || m.n[i].in[0] != cur)
        return fail(KWS_ERROR_UNSUPPORTED_MODEL, "expected FULLY_CONNECTED after the conv blocks");
    {
        const Node &fc = m.n[i];
        const Tensor &w = m.t[fc.in[1]];
        const Tensor *bias = (fc.in.size() > 2 && fc.in[2] >= 0) ? &m.t[fc.in[2]] : nullptr;
        N.fc_in = w.dims.back(); N.fc_out = w.dims[0];
        const KwsConvBlockF32 &lb = N.blk[N.n_blocks - 1];
        if (N.fc_in != lb.pool_w * lb.out_c || N.fc_in > KWS_FC_IN_MAX || N.fc_in * N.fc_out * 4 > KWS_FC_W_MAX || N.fc_out > 48 || N.fc_out != N.n_labels || !w.is_const ||
            (int)w.nbytes != N.fc_in * N.fc_out * 4)
            return fail(KWS_ERROR_UNSUPPORTED_MODEL, "FULLY_CONNECTED shape %dx%d outside the kernel's limits", N.fc_out, N.fc_in);
        h_act_range_f32(fc.p[0], &N.fc_min, &N.fc_max);
        std::vector<float> wv = floats(w), bv(N.fc_out, 0.0f);
        if (bias) { if ((int)bias->nbytes != N.fc_out * 4) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "FULLY_CONNECTED bias size"); bv = floats(*bias); }
        EI_IMPULSE_ERROR e;
        if ((e = h->upload(wv, &N.fc_w))) return e;
        if ((e = h->upload(bv, &N.fc_bias))) return e;
        h->hostf.fc_w = wv; h->hostf.fc_b = bv;
        cur = fc.out[0];
        i++;
    }
    if (i >= m.n.size() || m.n[i].op != OP_SOFTMAX || m.n[i].in[0] != cur || m.n[i].out[0] != (int)m.t_out || i + 1 != m.n.size())
        return fail(KWS_ERROR_UNSUPPORTED_MODEL, "expected a final SOFTMAX");
    N.beta = m.n[i].beta;
    if (kws_nn_f32_smem_bytes(N, 4) > 150 * 1024) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "float model too large for the kernel's LDS budget");
    {
        void *d = nullptr;
        HIP_TRY(hipMalloc(&d, sizeof(KwsNnPlanF32)));
        h->dev_allocs.push_back(d);
        HIP_TRY(hipMemcpy(d, &N, sizeof(KwsNnPlanF32), hipMemcpyHostToDevice));
        h->d_nnf = (const KwsNnPlanF32 *)d;
    }
    return EI_IMPULSE_OK;
}

