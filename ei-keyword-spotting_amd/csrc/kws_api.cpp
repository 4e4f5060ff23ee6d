// kws_api.cpp -- the C ABI of include/kws/kws.h: model handles, batch and stage entry points.  The per-clip arithmetic runs
// in kws_mfcc.hip / kws_nn_int8.hip / kws_nn_f32.hip.  There is no CPU fallback: if no HIP device or code object is
// available every entry point fails with KWS_ERROR_HIP.
#include "kws_internal.h"

#pragma GCC visibility push(default)     // the library is built with -fvisibility=hidden: only the C ABI is exported
extern "C" {

EI_IMPULSE_ERROR kws_create(const void *blob, size_t nbytes, int device, kws_handle **out)
{
    if (!blob || !out) return fail(KWS_ERROR_BAD_ARGUMENT, "null argument");
    *out = nullptr;
    kws_handle *h = new kws_handle();
    h->device = device;
    // the blob is checked first (it may come from anywhere): a malformed one is KWS_ERROR_BAD_ARGUMENT on any machine
    if (!parse_model(blob, nbytes, h->model)) { delete h; return fail(KWS_ERROR_BAD_ARGUMENT, "not a valid .kwsm model blob"); }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        delete h;
        return fail(KWS_ERROR_HIP, "no HIP device available (libkws_mi355x has no CPU fallback)");
    }
    if (device < 0 || device >= ndev) { delete h; return fail(KWS_ERROR_HIP, "device %d out of range (%d devices)", device, ndev); }
    if (hipSetDevice(device) != hipSuccess) { delete h; return fail(KWS_ERROR_HIP, "hipSetDevice(%d) failed", device); }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) h->n_cu = prop.multiProcessorCount;
    EI_IMPULSE_ERROR e = build_dsp_plan(h);
    if (e == EI_IMPULSE_OK) e = build_nn_plan(h);
    if (e == EI_IMPULSE_OK) e = build_fast_plans(h);
    if (e != EI_IMPULSE_OK) { kws_destroy(h); return e; }
    h->maf.assign(h->model.labels.size(), ei_impulse_maf{});
    *out = h;
    return EI_IMPULSE_OK;
}

EI_IMPULSE_ERROR kws_create_from_file(const char *path, int device, kws_handle **out)
{
    if (!path) return fail(KWS_ERROR_BAD_ARGUMENT, "null path");
    FILE *f = fopen(path, "rb");
    if (!f) return fail(KWS_ERROR_NO_MODEL, "cannot open model file %s", path);
    std::vector<uint8_t> buf;
    uint8_t tmp[4096];
    size_t n;
    while ((n = fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    fclose(f);
    return kws_create(buf.data(), buf.size(), device, out);
}

void kws_destroy(kws_handle *h)
{
    if (!h) return;
    kws_sdk_forget_default(h);
    (void)hipSetDevice(h->device);
    if (h->scratch_ev) { (void)hipEventSynchronize(h->scratch_ev); (void)hipEventDestroy(h->scratch_ev); }
    for (void *p : h->dev_allocs) (void)hipFree(p);
    if (h->s_mfcc) (void)hipFree(h->s_mfcc);
    if (h->s_q) (void)hipFree(h->s_q);
    if (h->d_flags) (void)hipFree(h->d_flags);
    if (h->d_flags2) (void)hipFree(h->d_flags2);
    if (h->d_flags3) (void)hipFree(h->d_flags3);
    if (h->s_cep) (void)hipFree(h->s_cep);
    for (hipEvent_t ev : h->gen_tune.ev) if (ev) (void)hipEventDestroy(ev);
    for (auto &g : h->g_sets) for (void *p : { (void *)g.ws, (void *)g.mfcc, (void *)g.feat }) if (p) (void)hipFree(p);
    for (int k = 0; k < 2; ++k) {
        for (void *p : { (void *)h->pipe.pcm[k], (void *)h->pipe.s[k], (void *)h->pipe.f[k], (void *)h->pipe.q[k] }) if (p) (void)hipFree(p);
        if (h->pipe.st[k]) (void)hipStreamDestroy(h->pipe.st[k]);
    }
    for (void *p : { (void *)h->ws.d_x, (void *)h->ws.d_f, (void *)h->ws.d_s, (void *)h->ws.d_w, (void *)h->ws.d_q })
        if (p) (void)hipFree(p);
    for (void *p : { (void *)h->ws.h_x, (void *)h->ws.h_s, (void *)h->ws.h_f })
        if (p) (void)hipHostFree(p);
    for (hipEvent_t ev : h->ws.ev) if (ev) (void)hipEventDestroy(ev);
    if (h->ws.st) (void)hipStreamDestroy(h->ws.st);
    delete h;
}

int kws_label_count(const kws_handle *h) { return (int)h->model.labels.size(); }
const char *kws_label(const kws_handle *h, int i) { return h->model.labels[i].c_str(); }
int kws_feature_count(const kws_handle *h) { return (int)h->model.nn_input_frame_size; }
int kws_clip_samples(const kws_handle *h) { return (int)h->model.raw_sample_count; }
int kws_frame_count(const kws_handle *h) { return h->dsp.n_frames; }
int kws_filter_count(const kws_handle *h) { return h->dsp.n_filters; }
int kws_pooled_tap_bytes(const kws_handle *h) { return h->pooled_tap_bytes; }
int kws_model_is_float(const kws_handle *h) { return h->is_float ? 1 : 0; }
// (the tuned kernel of the int16 batch paths; float samples -- the SDK's signal_t callback -- run kws_mfcc_kernel, the same arithmetic on
// the older lane layout)
const char *kws_mfcc_kernel_name(const kws_handle *h)
{
    // (int16 batches of a general plan whose spectral stage fits the tuned kernel: launch_spectral_tuned_chunks; float samples stay on the general kernel)
    if (h->dsp.generic && h->dsp.spectral_tuned && !KWS_DEV_ENV("KWS_DEV_GENERIC_NO_TUNED_SPECTRAL")) return "kws_mfcc8_kernel (chunked)";
    if (h->dsp.generic) return kws_generic_uses_lds(h->dsp) ? "kws_spectral_lds_kernel" : "kws_spectral_generic_kernel";
    return h->dsp.n_frames < 16 ? "kws_mfcc_kernel" : "kws_mfcc8_kernel";
}
const char *kws_nn_kernel_name(const kws_handle *h)
{
    return h->is_float ? "kws_nn_f32_kernel" : kws_nn_uses_mfma(h->nn) ? "kws_nn_mfma_kernel" : "kws_nn_kernel";
}

EI_IMPULSE_ERROR ensure_scratch(kws_handle *h, size_t B)
{
    if (B <= h->s_cap) return EI_IMPULSE_OK;
    if (h->s_mfcc) (void)hipFree(h->s_mfcc);
    if (h->s_q) (void)hipFree(h->s_q);
    h->s_mfcc = nullptr; h->s_q = nullptr; h->s_cap = 0;
    const size_t F = h->model.nn_input_frame_size;
    HIP_TRY(hipMalloc((void **)&h->s_mfcc, B * F * sizeof(float)));
    HIP_TRY(hipMalloc((void **)&h->s_q, B * F));
    h->s_cap = B;
    return EI_IMPULSE_OK;
}

EI_IMPULSE_ERROR kws_set_mode(kws_handle *h, int mode)
{
    if (!h || (mode != KWS_MODE_EXACT && mode != KWS_MODE_FAST)) return fail(KWS_ERROR_BAD_ARGUMENT, "kws_set_mode: bad argument");
    if (mode == KWS_MODE_FAST && !h->fast_plain_ok) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "%s", h->fast_why.c_str());
    std::lock_guard<std::mutex> lk(h->mu);
    h->mode = mode;
    return EI_IMPULSE_OK;
}
int kws_get_mode(const kws_handle *h)
{
    std::lock_guard<std::mutex> lk(const_cast<kws_handle *>(h)->mu);
    return h->mode;
}
int kws_fast_is_fused(const kws_handle *h) { return (h->fast_fused_ok || h->fast_q_ok) ? 1 : 0; }
EI_IMPULSE_ERROR kws_fast_fallback_count(kws_handle *h, size_t *count)
{
    if (!h || !count) return fail(KWS_ERROR_BAD_ARGUMENT, "null argument");
    *count = 0;
    std::lock_guard<std::mutex> lk(h->mu);
    if (!h->d_flags) return EI_IMPULSE_OK;
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    int n = 0;
    HIP_TRY(hipMemcpy(&n, h->d_flags, sizeof(int), hipMemcpyDeviceToHost));
    *count = (size_t)n;
    return EI_IMPULSE_OK;
}

EI_IMPULSE_ERROR kws_fast_guard(const kws_handle *h, int tier, float *coef)
{
    if (!h || !coef || (tier != 1 && tier != 2)) return fail(KWS_ERROR_BAD_ARGUMENT, "kws_fast_guard: bad argument");
    if (!h->fast_plain_ok) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "%s", h->fast_why.c_str());
    const size_t n = h->fast_guard_coef[tier - 1][0].size();
    for (int k = 0; k < 4; k++)
        for (size_t c = 0; c < n; c++) coef[(size_t)k * n + c] = h->fast_guard_coef[tier - 1][k][c];
    return EI_IMPULSE_OK;
}

EI_IMPULSE_ERROR kws_fast_gain(const kws_handle *h, float *col_gain)
{
    if (!h || !col_gain) return fail(KWS_ERROR_BAD_ARGUMENT, "kws_fast_gain: bad argument");
    if (!h->is_float || !h->gain.calibrated) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "kws_fast_gain: float32 graphs of the tuned DSP shapes only");
    for (size_t c = 0; c < h->gain.col.size(); c++) col_gain[c] = h->gain.col[c];
    return EI_IMPULSE_OK;
}

EI_IMPULSE_ERROR kws_fast_tolerance_info(const kws_handle *h, kws_fast_tolerance *out)
{
    if (!h || !out) return fail(KWS_ERROR_BAD_ARGUMENT, "kws_fast_tolerance_info: bad argument");
    if (!h->fast_plain_ok) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "%s", h->fast_why.c_str());
    const KwsFastPlan &F = h->fast_plain;
    memset(out, 0, sizeof(*out));
    out->score_tol = 1.0e-4f;
    out->lin_margin = 1.1f;
    out->k_sigma = sqrtf(F.g_c1) * out->score_tol / out->lin_margin;
    out->logit_cap = out->k_sigma / sqrtf(F.g_c2);
    out->g_c1 = F.g_c1; out->g_c2 = F.g_c2;
    out->calibrated = (h->is_float && h->gain.calibrated) ? 1 : 0;
    out->n_columns = (int)h->fast_guard_coef[0][0].size();
    out->n_frames = h->dsp.n_frames;
    out->entry_tier = h->fast_entry_tier;
    out->dev_overrides = h->fast_dev_overrides;
    out->k_sigma_worst_column = (h->is_float && h->gain.calibrated) ? out->k_sigma / 1.3f : out->k_sigma;
    out->silent_rows_exact = F.sil_off >= 0 ? 1 : 0;
    out->systematic_ratio = sqrtf(F.sys_t2);
    if (h->fast_fused_ok) { out->fused_waves_per_simd = h->fast_fused.wps; out->fused_waves = h->fast_fused.n_waves; }
    out->sigma_net = sqrtf(F.v_net);       // the terms of V that do not depend on the clip: the fused network's re-ordering noise, the deviation's own error
    // sum of gain^2 over every feature: a feature error of rms size t on every feature gives V = sigma_net^2 + t^2 x this
    double g2 = 0.0;
    for (float g : h->fast_gain_used) g2 += (double)g * (double)g * (double)h->dsp.n_frames;
    out->total_gain = (float)sqrt(g2);
    const double vmax = std::min(16.0 / (double)F.g_c1, 1.0 / (double)F.g_c2) - (double)F.v_net;
    out->uniform_feature_tol = (vmax > 0.0 && g2 > 0.0) ? (float)sqrt(vmax / g2) : 0.0f;
    return EI_IMPULSE_OK;
}

EI_IMPULSE_ERROR kws_set_logits_tap(kws_handle *h, float *logits)
{
    if (!h) return fail(KWS_ERROR_BAD_ARGUMENT, "null argument");
    if (logits && !h->is_float) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "kws_set_logits_tap: float32 graphs only (an int8 graph's taps: kws_nn_batch_device)");
    std::lock_guard<std::mutex> lk(h->mu);
    h->tap_logits = logits;
    return EI_IMPULSE_OK;
}

EI_IMPULSE_ERROR kws_fast_exact_count(kws_handle *h, size_t *count)
{
    if (!h || !count) return fail(KWS_ERROR_BAD_ARGUMENT, "null argument");
    *count = 0;
    std::lock_guard<std::mutex> lk(h->mu);
    if (!h->d_flags2) return EI_IMPULSE_OK;
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    int n = 0;
    HIP_TRY(hipMemcpy(&n, h->d_flags2, sizeof(int), hipMemcpyDeviceToHost));
    *count = (size_t)n;
    return EI_IMPULSE_OK;
}

static EI_IMPULSE_ERROR ensure_flags(kws_handle *h, size_t B)
{
    if (B + 1 <= h->flags_cap) return EI_IMPULSE_OK;
    if (h->d_flags) (void)hipFree(h->d_flags);
    if (h->d_flags2) (void)hipFree(h->d_flags2);
    if (h->d_flags3) (void)hipFree(h->d_flags3);
    h->d_flags = h->d_flags2 = h->d_flags3 = nullptr; h->flags_cap = 0;
    HIP_TRY(hipMalloc((void **)&h->d_flags, (B + 1) * sizeof(int)));
    HIP_TRY(hipMalloc((void **)&h->d_flags2, (B + 1) * sizeof(int)));
    HIP_TRY(hipMalloc((void **)&h->d_flags3, (B + 1) * sizeof(int)));
    HIP_TRY(hipMemset(h->d_flags2, 0, sizeof(int)));
    h->flags_cap = B + 1;
    return EI_IMPULSE_OK;
}

static EI_IMPULSE_ERROR classify_fast_device(kws_handle *h, const int16_t *pcm, size_t B, float *scores, float *fx, bool want_f, int8_t *q,
                                             hipStream_t s);
static EI_IMPULSE_ERROR classify_device(kws_handle *h, const int16_t *pcm, size_t B, float *scores, float *f, int8_t *q, hipStream_t s);
int grid_cap_mfcc(const kws_handle *h) { return h->n_cu * 8; }
int grid_cap_nn(const kws_handle *h) { return h->n_cu * 4; }

// Buffers of the general MFCC kernels for a batch of B windows on stream s (B = 0: the transform scratch only).  A stream keeps
// its set; when more than kGenericSets streams have been seen, the oldest set is handed on after the device has drained.
// Which chunk length kws_spectral_lds_kernel runs with (8 or 4 frames, csrc/kws_generic.hip): measured, not guessed -- 4 is 15 - 22 % faster on
// some shapes and 8 % slower on others and no rule here predicts which (profiles/r04_generic_rate.txt).  The first calls of a handle that bring
// at least kGenTuneMinClips clips are timed between two events on their own stream -- 8, 4, 8, 4 -- and the one with the smaller time per clip is
// used from then on.  Both are bit-exact: nothing but the time depends on the choice.  While it is being measured a call waits for the previous
// timed call's kernel before it starts (four host waits in a handle's life).  KWS_DEV_GENERIC_LCH=4|8 pins the choice (A/B runs, tests).
static const size_t kGenTuneMinClips = 2048;
static int generic_chunk_begin(kws_handle *h, size_t B, hipStream_t s, bool *mine)
{
    *mine = false;
    const char *fe = KWS_DEV_ENV("KWS_DEV_GENERIC_LCH");          // (read per call: a test toggles it between handles)
    const int forced = fe ? atoi(fe) : 0;
    if (forced == 4 || forced == 8) return forced;
    hipEvent_t wait_for[2] = { nullptr, nullptr };
    {
        std::lock_guard<std::mutex> lk(h->g_mu);
        kws_handle::GenericTune &T = h->gen_tune;
        if (T.choice) return T.choice;
        if (T.armed == 1) return 8;                           // another call's timed sample is in flight: do not disturb it
        if (T.armed == 2) { T.armed = 3; wait_for[0] = T.ev[0]; wait_for[1] = T.ev[1]; }      // this call collects it -- outside the lock
        else if (T.armed == 3) return 8;                      // somebody else is collecting
    }
    if (wait_for[1]) {
        float ms = 0.0f;
        const bool ok = hipEventSynchronize(wait_for[1]) == hipSuccess && hipEventElapsedTime(&ms, wait_for[0], wait_for[1]) == hipSuccess;
        std::lock_guard<std::mutex> lk(h->g_mu);
        kws_handle::GenericTune &T = h->gen_tune;
        if (ok && T.clips) { T.ms_per_clip[T.phase & 1] += (double)ms / (double)T.clips; T.phase++; }
        T.armed = 0;
        if (T.phase >= 4) { T.choice = T.ms_per_clip[0] <= T.ms_per_clip[1] ? 8 : 4; return T.choice; }
    }
    std::lock_guard<std::mutex> lk(h->g_mu);
    kws_handle::GenericTune &T = h->gen_tune;
    if (T.choice) return T.choice;
    if (T.armed != 0 || B < kGenTuneMinClips) return 8;       // (too small to time: the default, not a measurement)
    const int lch = (T.phase & 1) ? 4 : 8;
    // the first launch of a chunk length pays one-time host costs (function attributes, the code object's lazy load) between the events:
    // it runs un-timed, the sample is taken from the next call of that length
    if (!(T.warm & lch)) { T.warm |= lch; return lch; }
    if (!T.ev[0] && (hipEventCreate(&T.ev[0]) != hipSuccess || hipEventCreate(&T.ev[1]) != hipSuccess)) { T.choice = 8; return 8; }
    T.armed = 1;
    T.owner = s; T.owned = true;
    T.clips = B;
    *mine = true;
    (void)hipEventRecord(T.ev[0], s);
    return lch;
}
// only the call that armed the sample closes it, on its own stream (another thread's call on another stream used to)
static void generic_chunk_end(kws_handle *h, hipStream_t s, bool mine)
{
    if (!mine) return;
    std::lock_guard<std::mutex> lk(h->g_mu);
    kws_handle::GenericTune &T = h->gen_tune;
    if (T.armed == 1 && T.owned && T.owner == s) { (void)hipEventRecord(T.ev[1], s); T.armed = 2; T.owned = false; }
}
// the general-shape spectral launch with the handle's chunk length (the scratch kernel ignores it)
// A general-shape plan whose spectral stage fits the tuned kernel (KwsDspPlan::spectral_tuned: fft 256, 32 / 40 filters; general because of its frame
// count or its cmvnw window): int16 windows go through kws_mfcc8_kernel in chunks of frames -- 0.43 ns per frame instead of the cooperative kernel's
// 0.95 at the same transform -- bit for bit the same cepstra: frames are independent but for pre-emphasis' predecessor of a
// chunk's first sample, which is the sample before it (wrap_index = -1) where the chunk starts inside the window.  Returns -1 where the path does not apply.
static int launch_spectral_tuned_chunks(kws_handle *h, const KwsDspPlan &P, const void *pcm, int is_float, size_t B, float *mfcc, const float *wrap,
                                        int out_stride, hipStream_t s)
{
    if (!P.spectral_tuned || is_float || P.mfe_mel || P.ring_rows != 0 || !mfcc || ((uintptr_t)pcm & 15) != 0 || KWS_DEV_ENV("KWS_DEV_GENERIC_NO_TUNED_SPECTRAL")) return -1;
    const int nfr = P.n_frames, maxf = std::min(kws_mfcc_max_frames(P.n_filters), 49);      // (49 = six passes of eight frames + the tail pass: 98 frames are two such chunks)
    const int k = (nfr + maxf - 1) / maxf, base = nfr / k, rem = nfr % k;
    if (k > 1 && base < 16) return -1;                                    // (chunks below sixteen frames would take kws_mfcc_kernel, which has no wrap_index)
    const int stride_out = out_stride ? out_stride : nfr * P.n_cepstral;
    int f0 = 0;
    for (int c = 0; c < k; c++) {
        const int n = base + (c < rem ? 1 : 0);
        KwsDspPlan Pc = P;
        Pc.generic = 0;
        Pc.n_frames = n;
        Pc.pad = 0; Pc.win_size = 1;                                        // (the kernel stages its cmvnw pad map even when it stops at the cepstra)
        Pc.wrap_index = c == 0 ? P.n_samples - 1 : -1;
        const int rc = kws_launch_spectral(Pc, (const int16_t *)pcm + (size_t)f0 * P.frame_stride, 0, (int)B, mfcc + (size_t)f0 * P.n_cepstral,
                                           c == 0 ? wrap : nullptr, stride_out, grid_cap_mfcc(h), s);
        if (rc) return rc;
        f0 += n;
    }
    return 0;
}

static int launch_spectral_generic_for(kws_handle *h, const KwsDspPlan &P, const void *pcm, int is_float, size_t B, float *mfcc, const float *wrap,
                                       int out_stride, float *ws, hipStream_t s)
{
    const int rt = launch_spectral_tuned_chunks(h, P, pcm, is_float, B, mfcc, wrap, out_stride, s);
    if (rt >= 0) return rt;
    const bool lds = kws_generic_uses_lds(P);
    bool mine = false;
    const int lch = lds ? generic_chunk_begin(h, B, s, &mine) : 8;
    const int rc = kws_launch_spectral_generic(P, pcm, is_float, (int)B, mfcc, wrap, out_stride, ws, grid_cap_mfcc(h), lch, s);
    if (lds) generic_chunk_end(h, s, mine);
    return rc;
}

EI_IMPULSE_ERROR generic_for(kws_handle *h, hipStream_t s, size_t B, kws_handle::GenericBuf **out)
{
    std::lock_guard<std::mutex> lk(h->g_mu);
    kws_handle::GenericBuf *g = nullptr;
    for (auto &c : h->g_sets) if (c.used && c.s == s) { g = &c; break; }
    if (!g) for (auto &c : h->g_sets) if (!c.used) { g = &c; break; }
    if (!g) {
        g = &h->g_sets[h->g_next];
        h->g_next = (h->g_next + 1) % kws_handle::kGenericSets;
        HIP_TRY(hipDeviceSynchronize());          // its stream may be gone by now: wait for everything
    }
    g->used = true; g->s = s;
    const size_t need_ws = kws_generic_ws_bytes(h->dsp, grid_cap_mfcc(h));
    if (need_ws > g->ws_bytes) {
        if (g->ws) (void)hipFree(g->ws);
        g->ws = nullptr; g->ws_bytes = 0;
        HIP_TRY(hipMalloc((void **)&g->ws, need_ws));
        g->ws_bytes = need_ws;
    }
    if (B > g->cap) {
        for (void *p : { (void *)g->mfcc, (void *)g->feat }) if (p) (void)hipFree(p);
        g->mfcc = g->feat = nullptr; g->cap = 0;
        const size_t F = h->model.nn_input_frame_size;
        HIP_TRY(hipMalloc((void **)&g->mfcc, std::max<size_t>(B * F, 1) * sizeof(float)));
        HIP_TRY(hipMalloc((void **)&g->feat, std::max<size_t>(B * F, 1) * sizeof(float)));
        g->cap = B;
    }
    *out = g;
    return EI_IMPULSE_OK;
}

// speechpy::feature::mfcc for B windows (kernel 1)
EI_IMPULSE_ERROR spectral_device(kws_handle *h, const KwsDspPlan &P, const void *pcm, int is_float, size_t B, float *mfcc,
                                        const float *wrap, hipStream_t s, int out_stride)
{
    if (B > 0x7fffffff) return fail(KWS_ERROR_BAD_ARGUMENT, "batch too large");
    if (h->model.dsp.block == DSP_BLOCK_MFE && !P.mfe_mel) {
        // the MFE block's per-window stage is speechpy::feature::mfe itself (extract_mfe_per_slice_features, L432 ei_run_dsp.h:420-470)
        if (((uintptr_t)pcm & 15) != 0) return fail(KWS_ERROR_BAD_ARGUMENT, "pcm must be 16-byte aligned");
        int rc = kws_launch_mfe(P, pcm, is_float, (int)B, mfcc, nullptr, wrap, out_stride, grid_cap_mfcc(h), s);
        if (rc) return fail(KWS_ERROR_HIP, "MFE kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
        return EI_IMPULSE_OK;
    }
    if (h->dsp.generic) {
        kws_handle::GenericBuf *g = nullptr;
        EI_IMPULSE_ERROR e = generic_for(h, s, 0, &g);
        if (e) return e;
        int rc = launch_spectral_generic_for(h, P, pcm, is_float, B, mfcc, wrap, out_stride, g->ws, s);
        if (rc) return fail(KWS_ERROR_HIP, "MFCC kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
        return EI_IMPULSE_OK;
    }
    if (((uintptr_t)pcm & 15) != 0) return fail(KWS_ERROR_BAD_ARGUMENT, "pcm must be 16-byte aligned");
    int rc = kws_launch_spectral(P, pcm, is_float, (int)B, mfcc, wrap, out_stride, grid_cap_mfcc(h), s);
    if (rc) return fail(KWS_ERROR_HIP, "MFCC kernel launch failed: %s (is the gfx950 code object present?)", hipGetErrorString((hipError_t)rc));
    return EI_IMPULSE_OK;
}

// extract_mfcc_features + quantisation in one launch (fused kernel)
EI_IMPULSE_ERROR mfcc_fused_device_plan(kws_handle *h, const KwsDspPlan &P, const void *pcm, int is_float, size_t B, float *features, int8_t *q, hipStream_t s)
{
    if (B > 0x7fffffff) return fail(KWS_ERROR_BAD_ARGUMENT, "batch too large");
    if (h->model.dsp.block == DSP_BLOCK_MFE) {
        // extract_mfe_features (L432 classifier/ei_run_dsp.h:369-418): feature::mfe, cmvnw(win, false, true) + numpy::normalize, then
        // the input quantisation of an int8 graph.  The float feature matrix is needed either way.
        if (!features) return fail(KWS_ERROR_BAD_ARGUMENT, "the MFE block needs a float feature buffer");
        if (((uintptr_t)pcm & 15) != 0) return fail(KWS_ERROR_BAD_ARGUMENT, "pcm must be 16-byte aligned");
        int rc = kws_launch_mfe(P, pcm, is_float, (int)B, features, nullptr, nullptr, 0, grid_cap_mfcc(h), s);
        if (!rc) rc = kws_launch_mfe_norm(features, (int)B, P.n_frames, P.n_filters, P.win_size, P.pad_map, P.n_frames + 2 * P.pad, grid_cap_nn(h), s);
        if (!rc && q) rc = kws_launch_quantize(features, q, B * h->model.nn_input_frame_size, h->nn.in_scale, h->nn.in_zp, s);
        if (rc) return fail(KWS_ERROR_HIP, "MFE block launch failed: %s", hipGetErrorString((hipError_t)rc));
        return EI_IMPULSE_OK;
    }
    if (P.generic) {
        // cepstra -> g_mfcc, then cmvnw + quantisation (the general kernels are two launches; the cepstra go through HBM)
        kws_handle::GenericBuf *g = nullptr;
        EI_IMPULSE_ERROR e = generic_for(h, s, B, &g);
        if (e) return e;
        int rc = launch_spectral_generic_for(h, P, pcm, is_float, B, g->mfcc, nullptr, 0, g->ws, s);
        if (!rc) rc = kws_launch_cmvn_generic(P, g->mfcc, (int)B, features, q, h->nn.in_scale, h->nn.in_zp, s);
        if (rc) return fail(KWS_ERROR_HIP, "MFCC kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
        return EI_IMPULSE_OK;
    }
    if (((uintptr_t)pcm & 15) != 0) return fail(KWS_ERROR_BAD_ARGUMENT, "pcm must be 16-byte aligned");
    int rc = kws_launch_mfcc_fused(P, pcm, is_float, (int)B, features, q, h->nn.in_scale, h->nn.in_zp, h->n_cu * 8, s);
    if (rc) return fail(KWS_ERROR_HIP, "MFCC kernel launch failed: %s (is the gfx950 code object present?)", hipGetErrorString((hipError_t)rc));
    return EI_IMPULSE_OK;
}
EI_IMPULSE_ERROR mfcc_fused_device(kws_handle *h, const void *pcm, int is_float, size_t B, float *features, int8_t *q, hipStream_t s)
{
    return mfcc_fused_device_plan(h, h->dsp, pcm, is_float, B, features, q, s);
}

// float32 models: the network reads the feature matrix itself (ei_run_classifier.h:447-452 copies it into the input tensor)
EI_IMPULSE_ERROR nn_f32_device(kws_handle *h, const float *features, size_t B, float *scores, float *tap_logits, hipStream_t s)
{
    if (B > 0x7fffffff) return fail(KWS_ERROR_BAD_ARGUMENT, "batch too large");
    int rc = kws_launch_nn_f32(h->nnf, h->d_nnf, features, (int)B, scores, tap_logits, h->n_cu, s);
    if (rc) return fail(KWS_ERROR_HIP, "float NN kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    return EI_IMPULSE_OK;
}
#define KWS_INT8_ONLY(h) do { if ((h)->is_float) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "%s takes an int8 tensor; the loaded model is float32", __func__); } while (0)

// cmvnw + quantise + (optionally) the network (kernel 2; the generic NN kernel follows when the graph does not fit
// the matrix-core path)
EI_IMPULSE_ERROR cmvn_nn_device(kws_handle *h, const float *mfcc, size_t B, float *features, int8_t *q, float *scores,
                                       int8_t *tap_pooled, int8_t *tap_fc, int8_t *tap_out, hipStream_t s, int ring_rows, int ring_head)
{
    // ring_rows != 0: mfcc holds ring-indexed rolling buffers (continuous mode, kws_streams_*)
    KwsDspPlan PR = h->dsp;
    PR.ring_rows = ring_rows; PR.ring_head = ring_rows > 0 ? ((ring_head % ring_rows) + ring_rows) % ring_rows : 0;      // (the fast kernel's ring() wraps with one subtraction: head < rows)
    if (B > 0x7fffffff) return fail(KWS_ERROR_BAD_ARGUMENT, "batch too large");
    int ran_nn = 0;
    if (h->model.dsp.block == DSP_BLOCK_MFE) {
        // calc_cepstral_mean_and_var_normalization_mfe on a copy of the mel matrices (L432 classifier/ei_run_classifier.h:745-775)
        if (h->is_float && (q || tap_pooled || tap_fc || tap_out)) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "int8 outputs requested from a float32 model");
        const KwsDspPlan &P = h->dsp;
        const size_t F = h->model.nn_input_frame_size;
        float *f = features ? features : h->s_mfcc;
        int8_t *qq = h->is_float ? nullptr : (q ? q : h->s_q);
        int rc = kws_launch_unring(mfcc, f, (int)B, P.n_frames, P.n_filters, ring_rows, ring_head, s);      // the reference normalises a COPY
        if (rc) return fail(KWS_ERROR_HIP, "copy kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
        rc = kws_launch_mfe_norm(f, (int)B, P.n_frames, P.n_filters, P.win_size, P.pad_map, P.n_frames + 2 * P.pad, grid_cap_nn(h), s);
        if (!rc && qq) rc = kws_launch_quantize(f, qq, B * F, h->nn.in_scale, h->nn.in_zp, s);
        if (rc) return fail(KWS_ERROR_HIP, "MFE normalisation launch failed: %s", hipGetErrorString((hipError_t)rc));
        if (!scores) return EI_IMPULSE_OK;
        if (h->is_float) return nn_f32_device(h, f, B, scores, nullptr, s);
        rc = kws_launch_nn(h->nn, qq, (int)B, scores, tap_pooled, h->pooled_tap_bytes, tap_fc, tap_out, grid_cap_nn(h), s);
        if (rc) return fail(KWS_ERROR_HIP, "NN kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
        return EI_IMPULSE_OK;
    }
    if (h->dsp.generic) {
        if (h->is_float && (q || tap_pooled || tap_fc || tap_out)) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "int8 outputs requested from a float32 model");
        kws_handle::GenericBuf *g = nullptr;
        EI_IMPULSE_ERROR e = generic_for(h, s, B, &g);
        if (e) return e;
        float *f = features ? features : (h->is_float ? g->feat : nullptr);
        int8_t *qq = h->is_float ? nullptr : (q ? q : h->s_q);
        int rc = 0;
        if (ring_rows) {
            rc = kws_launch_unring(mfcc, g->mfcc, (int)B, h->dsp.n_frames, h->dsp.n_cepstral, ring_rows, ring_head, s);
            mfcc = g->mfcc;
        }
        if (!rc) rc = kws_launch_cmvn_generic(h->dsp, mfcc, (int)B, f, qq, h->nn.in_scale, h->nn.in_zp, s);
        if (rc) return fail(KWS_ERROR_HIP, "CMVN kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
        if (!scores) return EI_IMPULSE_OK;
        if (h->is_float) return nn_f32_device(h, f, B, scores, nullptr, s);
        rc = kws_launch_nn(h->nn, qq, (int)B, scores, tap_pooled, h->pooled_tap_bytes, tap_fc, tap_out, grid_cap_nn(h), s);
        if (rc) return fail(KWS_ERROR_HIP, "NN kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
        return EI_IMPULSE_OK;
    }
    if (h->is_float) {
        if (q || tap_pooled || tap_fc || tap_out) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "int8 outputs requested from a float32 model");
        float *f = features ? features : h->s_mfcc;
        int rc = kws_launch_cmvn_nn(PR, h->nn, mfcc, (int)B, f, nullptr, nullptr, nullptr, 0, nullptr, nullptr, grid_cap_nn(h), &ran_nn, s);
        if (rc) return fail(KWS_ERROR_HIP, "CMVN kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
        return scores ? nn_f32_device(h, f, B, scores, h->tap_logits, s) : EI_IMPULSE_OK;
    }
    int8_t *qq = q;
    if (scores && !qq) qq = h->s_q;           // the generic NN kernel reads the quantised tensor from HBM
    int rc = kws_launch_cmvn_nn(PR, h->nn, mfcc, (int)B, features, qq, scores, tap_pooled, h->pooled_tap_bytes, tap_fc, tap_out,
                                grid_cap_nn(h), &ran_nn, s);
    if (rc) return fail(KWS_ERROR_HIP, "CMVN kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    if (scores && !ran_nn) {
        rc = kws_launch_nn(h->nn, qq, (int)B, scores, tap_pooled, h->pooled_tap_bytes, tap_fc, tap_out, grid_cap_nn(h), s);
        if (rc) return fail(KWS_ERROR_HIP, "NN kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    }
    return EI_IMPULSE_OK;
}

// KWS_MODE_FAST for windows that arrive as cepstra (continuous mode): O(1) cmvnw + the network (fused for float graphs), then the
// exact kernels over the windows the fast kernel listed as ill-conditioned.  Uses the handle's scratch (h->mu held by the caller).
EI_IMPULSE_ERROR cmvn_nn_fast_device(kws_handle *h, const float *mfcc, size_t B, float *scores, hipStream_t s, int ring_rows, int ring_head,
                                     float *features, int8_t *q_out)
{
    if (B > 0x7fffffff) return fail(KWS_ERROR_BAD_ARGUMENT, "batch too large");
    // MFE block: what arrives here are the exact kernels' mel matrices, and the block's normalisation has no fast form (nothing to gain:
    // it divides by the matrix's range, not by a deviation): the exact path serves both modes
    if (h->model.dsp.block == DSP_BLOCK_MFE)
        return cmvn_nn_device(h, mfcc, B, features, h->is_float ? nullptr : q_out, scores, nullptr, nullptr, nullptr, s, ring_rows, ring_head);
    EI_IMPULSE_ERROR e = ensure_flags(h, B);
    if (e) return e;
    HIP_TRY(hipMemsetAsync(h->d_flags, 0, sizeof(int), s));
    HIP_TRY(hipMemsetAsync(h->d_flags2, 0, sizeof(int), s));      // kws_fast_exact_count describes the LAST fast call: this path has one fast tier
    KwsDspPlan PR = h->dsp;
    PR.ring_rows = ring_rows; PR.ring_head = ring_rows > 0 ? ((ring_head % ring_rows) + ring_rows) % ring_rows : 0;      // (the fast kernel's ring() wraps with one subtraction: head < rows)
    const bool fused = scores && h->is_float && h->fast_fused_ok;       // scores == NULL: features / int8 tensor only (extract_mfcc_features)
    float *fx = features ? features : h->s_mfcc;
    int8_t *q = h->is_float ? nullptr : (q_out ? q_out : h->s_q);
    int rc = 0;
    if (fused && features) {
        // the fused form keeps the feature matrix on chip: a caller who also wants it gets it from the feature-emitting form first
        rc = kws_launch_fast_from_cepstra(PR, h->fast_plain, h->d_fast_plain, mfcc, (int)B, nullptr, fx, nullptr, h->nn.in_scale, h->nn.in_zp, h->d_flags,
                                          h->d_flags + 1, h->n_cu, s);
        if (rc) return fail(KWS_ERROR_HIP, "fast kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
        HIP_TRY(hipMemsetAsync(h->d_flags3, 0, sizeof(int), s));
    }
    const KwsFastPlan &FP = fused ? h->fast_fused_cep : h->fast_plain;
    int *const fl = (fused && features) ? h->d_flags3 : h->d_flags;      // the feature-emitting launch's list stands (kws_internal.h: d_flags3)
    rc = kws_launch_fast_from_cepstra(PR, FP, fused ? h->d_fast_fused_cep : h->d_fast_plain, mfcc, (int)B, scores, fused ? nullptr : fx, q,
                                      h->nn.in_scale, h->nn.in_zp, fl, fl + 1, h->n_cu, s, nullptr, h->tap_logits);
    if (rc) return fail(KWS_ERROR_HIP, "fast kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    if (!fused && scores) {
        if (h->is_float) { if ((e = nn_f32_device(h, fx, B, scores, h->tap_logits, s))) return e; }
        else {
            rc = kws_launch_nn(h->nn, q, (int)B, scores, nullptr, h->pooled_tap_bytes, nullptr, nullptr, grid_cap_nn(h), s);
            if (rc) return fail(KWS_ERROR_HIP, "NN kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
        }
    }
    // exact re-run of the listed windows (indexed by their own numbers)
    int ran_nn = 0;
    rc = kws_launch_cmvn_nn(PR, h->nn, mfcc, (int)B, (h->is_float || features) ? fx : nullptr, q, h->is_float ? nullptr : scores, nullptr, h->pooled_tap_bytes,
                            nullptr, nullptr, grid_cap_nn(h), &ran_nn, s, h->d_flags);
    if (rc) return fail(KWS_ERROR_HIP, "CMVN kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    HIP_TRY(hipMemcpyAsync(h->d_flags2, h->d_flags, sizeof(int), hipMemcpyDeviceToDevice, s));     // handed on = finished by the exact kernels
    if (!scores) return EI_IMPULSE_OK;
    if (h->is_float) rc = kws_launch_nn_f32(h->nnf, h->d_nnf, fx, (int)B, scores, h->tap_logits, h->n_cu, s, h->d_flags);
    else if (!ran_nn) rc = kws_launch_nn(h->nn, q, (int)B, scores, nullptr, h->pooled_tap_bytes, nullptr, nullptr, grid_cap_nn(h), s, h->d_flags);
    if (rc) return fail(KWS_ERROR_HIP, "NN kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    return EI_IMPULSE_OK;
}

EI_IMPULSE_ERROR kws_mfcc_batch_device(kws_handle *h, const int16_t *pcm, size_t B, float *mfcc, void *stream)
{
    if (!h || !pcm || !mfcc) return fail(KWS_ERROR_BAD_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    return spectral_device(h, h->dsp, pcm, 0, B, mfcc, nullptr, (hipStream_t)stream);
}

EI_IMPULSE_ERROR kws_mfe_batch_device(kws_handle *h, const int16_t *pcm, size_t B, float *mel, float *energy, void *stream)
{
    if (!h || !pcm || !mel) return fail(KWS_ERROR_BAD_ARGUMENT, "null argument");
    if (B > 0x7fffffff) return fail(KWS_ERROR_BAD_ARGUMENT, "batch too large");
    HIP_TRY(hipSetDevice(h->device));
    if (h->dsp.generic) {
        KwsDspPlan P = h->dsp;
        P.mfe_mel = mel; P.mfe_energy = energy;
        return spectral_device(h, P, pcm, 0, B, nullptr, nullptr, (hipStream_t)stream);
    }
    if (((uintptr_t)pcm & 15) != 0) return fail(KWS_ERROR_BAD_ARGUMENT, "pcm must be 16-byte aligned");
    int rc = kws_launch_mfe(h->dsp, pcm, 0, (int)B, mel, energy, nullptr, 0, grid_cap_mfcc(h), (hipStream_t)stream);
    if (rc) return fail(KWS_ERROR_HIP, "MFE kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    return EI_IMPULSE_OK;
}

EI_IMPULSE_ERROR kws_extract_mfe_batch_device(kws_handle *h, const int16_t *pcm, size_t B, float *features, void *stream)
{
    if (!h || !pcm || !features) return fail(KWS_ERROR_BAD_ARGUMENT, "null argument");
    if (B > 0x7fffffff) return fail(KWS_ERROR_BAD_ARGUMENT, "batch too large");
    if (h->dsp.generic) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "the MFE block's normalisation kernel serves the tuned configurations only");
    if (((uintptr_t)pcm & 15) != 0) return fail(KWS_ERROR_BAD_ARGUMENT, "pcm must be 16-byte aligned");
    HIP_TRY(hipSetDevice(h->device));
    // the MFE block hands the raw signal to feature::mfe (ei_run_dsp.h:398-400; extract_mfcc_features wraps it in the
    // pre-emphasis class first): coefficient 0 makes the kernel's y = x - cof * prev the identity, bit for bit
    KwsDspPlan P = h->dsp;
    P.pre_cof = 0.0f;
    const int rows = P.n_frames, cols = P.n_filters;
    // cmvn_columns<17, 20> (more than 16 columns) walks 3 x 17 rows and needs a window of at least 17 rows; <13, 16>: 4 x 13, 13
    if (rows > (cols > 16 ? 51 : 52) || P.win_size < (cols > 16 ? 17 : 13))
        return fail(KWS_ERROR_UNSUPPORTED_MODEL, "%d frames x %d filters, window %d outside the MFE normalisation kernel's limits", rows, cols, P.win_size);
    int rc = kws_launch_mfe(P, pcm, 0, (int)B, features, nullptr, nullptr, 0, grid_cap_mfcc(h), (hipStream_t)stream);
    if (rc) return fail(KWS_ERROR_HIP, "MFE kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    rc = kws_launch_mfe_norm(features, (int)B, rows, cols, P.win_size, P.pad_map, rows + 2 * P.pad, grid_cap_nn(h), (hipStream_t)stream);
    if (rc) return fail(KWS_ERROR_HIP, "MFE normalisation kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    return EI_IMPULSE_OK;
}

EI_IMPULSE_ERROR kws_cmvn_inference_batch_device(kws_handle *h, const float *mfcc, size_t B, float *scores, float *features,
                                                 int8_t *q_in, void *stream)
{
    if (!h || !mfcc || !scores) return fail(KWS_ERROR_BAD_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    std::lock_guard<std::mutex> lk(h->mu);
    EI_IMPULSE_ERROR e = ensure_scratch(h, B);
    if (e) return e;
    ScratchUse use(h, (hipStream_t)stream);
    if (h->mode == KWS_MODE_FAST) {
        if (h->is_float && q_in) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "int8 input tensor requested from a float32 model");
        return cmvn_nn_fast_device(h, mfcc, B, scores, (hipStream_t)stream, 0, 0, features, q_in);
    }
    return cmvn_nn_device(h, mfcc, B, features, q_in, scores, nullptr, nullptr, nullptr, (hipStream_t)stream);
}

EI_IMPULSE_ERROR kws_extract_mfcc_batch_device(kws_handle *h, const int16_t *pcm, size_t B, float *features, int8_t *q_in, void *stream)
{
    if (!h || !pcm || !features) return fail(KWS_ERROR_BAD_ARGUMENT, "null argument");
    if (q_in) KWS_INT8_ONLY(h);
    HIP_TRY(hipSetDevice(h->device));
    std::lock_guard<std::mutex> lk(h->mu);
    EI_IMPULSE_ERROR e = ensure_scratch(h, B);
    if (e) return e;
    ScratchUse use(h, (hipStream_t)stream);
    if (h->mode == KWS_MODE_FAST) return classify_fast_device(h, pcm, B, nullptr, features, true, q_in, (hipStream_t)stream);
    return mfcc_fused_device(h, pcm, 0, B, features, q_in, (hipStream_t)stream);
}

// development / test aid (not in the public headers): kws_spectral_lds_kernel's chunk length on this handle: 0 while it is being measured
int kws_dev_generic_chunk(const kws_handle *h) { return h ? h->gen_tune.choice : 0; }

// development aid (not in the public headers): per-phase shader-clock totals of wave 0 of the float network kernel
extern long long *kws_dev_f32_prof;
void kws_dev_set_f32_prof(long long *dev_buf) { kws_dev_f32_prof = dev_buf; }
extern long long *kws_dev_nn_prof;
void kws_dev_set_nn_prof(long long *dev_buf) { kws_dev_nn_prof = dev_buf; }

// development/test aid (not in the public headers): force the generic dot4 NN kernel
void kws_dev_force_scalar_nn(int on) { kws_force_scalar_nn = on; }

// development aid (not in the public headers): per-phase shader-clock totals of wave 0 of workgroup 0 of the fast kernel
EI_IMPULSE_ERROR kws_dev_fast_phase_profile(kws_handle *h, const int16_t *pcm, size_t B, float *scores, long long *prof_dev)
{
    HIP_TRY(hipSetDevice(h->device));
    if (!h->fast_plain_ok) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "%s", h->fast_why.c_str());
    EI_IMPULSE_ERROR e = ensure_flags(h, B);
    if (e) return e;
    HIP_TRY(hipMemsetAsync(h->d_flags, 0, sizeof(int), nullptr));
    const bool fused = h->is_float && h->fast_fused_ok;
    int rc = kws_launch_fast_prof(h->dsp, fused ? h->fast_fused : h->fast_plain, fused ? h->d_fast_fused : h->d_fast_plain, pcm, (int)B, scores,
                                  h->d_flags, h->d_flags + 1, h->n_cu, prof_dev, nullptr);
    if (rc) return fail(KWS_ERROR_HIP, "launch failed");
    return EI_IMPULSE_OK;
}

// development aid (not in the public headers): per-phase shader-clock totals of workgroup 0 of kernel 1
EI_IMPULSE_ERROR kws_dev_mfcc_phase_profile(kws_handle *h, const int16_t *pcm, size_t B, float *features, long long *prof_dev)
{
    HIP_TRY(hipSetDevice(h->device));
    int rc = kws_launch_mfcc_fused_prof(h->dsp, pcm, (int)B, features, nullptr, h->nn.in_scale, h->nn.in_zp, h->n_cu * 8, prof_dev, nullptr);
    if (rc) return fail(KWS_ERROR_HIP, "launch failed");
    return EI_IMPULSE_OK;
}

EI_IMPULSE_ERROR kws_nn_batch_device(kws_handle *h, const int8_t *q_in, size_t B, float *scores, int8_t *tap_pooled, int8_t *tap_fc,
                                     int8_t *tap_out, void *stream)
{
    if (!h || !q_in || !scores) return fail(KWS_ERROR_BAD_ARGUMENT, "null argument");
    KWS_INT8_ONLY(h);
    if (B > 0x7fffffff) return fail(KWS_ERROR_BAD_ARGUMENT, "batch too large");
    HIP_TRY(hipSetDevice(h->device));
    int rc = kws_launch_nn(h->nn, q_in, (int)B, scores, tap_pooled, h->pooled_tap_bytes, tap_fc, tap_out, grid_cap_nn(h), (hipStream_t)stream);
    if (rc) return fail(KWS_ERROR_HIP, "NN kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    return EI_IMPULSE_OK;
}

EI_IMPULSE_ERROR kws_nn_f32_batch_device(kws_handle *h, const float *features, size_t B, float *scores, float *tap_logits, void *stream)
{
    if (!h || !features || !scores) return fail(KWS_ERROR_BAD_ARGUMENT, "null argument");
    if (!h->is_float) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "kws_nn_f32_batch_device needs a float32 model; the loaded model is int8");
    HIP_TRY(hipSetDevice(h->device));
    return nn_f32_device(h, features, B, scores, tap_logits, (hipStream_t)stream);
}

EI_IMPULSE_ERROR kws_run_inference_batch_device(kws_handle *h, const float *features, size_t B, float *scores, void *stream)
{
    if (!h || !features || !scores) return fail(KWS_ERROR_BAD_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    if (h->is_float) return nn_f32_device(h, features, B, scores, nullptr, (hipStream_t)stream);
    std::lock_guard<std::mutex> lk(h->mu);
    EI_IMPULSE_ERROR e = ensure_scratch(h, B);
    if (e) return e;
    ScratchUse use(h, (hipStream_t)stream);
    int rc = kws_launch_quantize(features, h->s_q, B * h->model.nn_input_frame_size, h->nn.in_scale, h->nn.in_zp, (hipStream_t)stream);
    if (rc) return fail(KWS_ERROR_HIP, "quantise kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    return kws_nn_batch_device(h, h->s_q, B, scores, nullptr, nullptr, nullptr, stream);
}

// extract_mfcc_features + the network for B windows in HBM.  float models: f must be a [B][n_features] buffer (the network
// reads it); int8 models: q must be a [B][n_features] buffer, f is optional (the float feature matrix only leaves the chip
// when somebody asks for it: the cepstra stay in LDS)
// KWS_MODE_FAST: the fast kernel over every clip, then the exact kernels over the clips it listed as ill-conditioned (the list and
// its length stay in HBM: nothing synchronises).  fx: feature scratch / output [B][n_features] (exact re-runs of float graphs and
// un-fused graphs read it), q: int8 tensor [B][n_features] for int8 graphs; want_f: the caller asked for the feature matrix.
// What follows the fast kernel of a batch call, on the clips it handed back (list d_flags; list and count stay in HBM, nothing
// synchronises).  Second tier: their cepstra from the exact kernels -- bit-identical to the reference's, so the DCT term of the guard is
// gone -- then the fast cmvnw + network from those cepstra (kws_fast_kernel<FROM_CEP> over the list), with the guard that is left
// (window mean and near-constant columns, KwsFastPlan::guard_cep_off).  Third tier: what that hands back (list d_flags2) goes through
// the exact cmvnw + network and comes out with the exact mode's bits.  A clip costs its tiers: ~0.4 x the exact path for the second.
static EI_IMPULSE_ERROR rerun_flagged_device(kws_handle *h, const int16_t *pcm, size_t B, float *scores, float *fx, bool want_f, int8_t *q, hipStream_t s)
{
    static const bool skip = KWS_DEV_ENV("KWS_DEV_FAST_NO_RERUN") != nullptr;     // development aid: what the (usually empty) re-run launches cost; results are wrong when set
    if (skip) return EI_IMPULSE_OK;
    const size_t F = h->model.nn_input_frame_size;
    if (B > h->cep_cap) {
        if (h->s_cep) (void)hipFree(h->s_cep);
        h->s_cep = nullptr; h->cep_cap = 0;
        HIP_TRY(hipMalloc((void **)&h->s_cep, B * F * sizeof(float)));
        h->cep_cap = B;
    }
    HIP_TRY(hipMemsetAsync(h->d_flags2, 0, sizeof(int), s));
    int rc = kws_launch_spectral(h->dsp, pcm, 0, (int)B, h->s_cep, nullptr, 0, grid_cap_mfcc(h), s, h->d_flags);
    if (rc) return fail(KWS_ERROR_HIP, "MFCC kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    const bool fused = scores && h->is_float && h->fast_fused_ok;
    if (!scores || (fused && want_f) || !fused) {
        // the feature-emitting form: features (float graphs: the network's input; or the caller's wish) and the int8 tensor
        rc = kws_launch_fast_from_cepstra(h->dsp, h->fast_plain, h->d_fast_plain, h->s_cep, (int)B, nullptr, (h->is_float || want_f) ? fx : nullptr, q,
                                          h->nn.in_scale, h->nn.in_zp, h->d_flags2, h->d_flags2 + 1, h->n_cu, s, h->d_flags);
        if (rc) return fail(KWS_ERROR_HIP, "fast kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
        if (fused) HIP_TRY(hipMemsetAsync(h->d_flags3, 0, sizeof(int), s));      // the feature-emitting launch's list stands (kws_internal.h: d_flags3)
    }
    if (fused) {
        int *const fl = want_f ? h->d_flags3 : h->d_flags2;
        rc = kws_launch_fast_from_cepstra(h->dsp, h->fast_fused_cep, h->d_fast_fused_cep, h->s_cep, (int)B, scores, nullptr, nullptr, h->nn.in_scale, h->nn.in_zp,
                                          fl, fl + 1, h->n_cu, s, h->d_flags, h->tap_logits);
        if (rc) return fail(KWS_ERROR_HIP, "fast kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    } else if (scores) {
        if (h->is_float) rc = kws_launch_nn_f32(h->nnf, h->d_nnf, fx, (int)B, scores, h->tap_logits, h->n_cu, s, h->d_flags);
        else rc = kws_launch_nn(h->nn, q, (int)B, scores, nullptr, h->pooled_tap_bytes, nullptr, nullptr, grid_cap_nn(h), s, h->d_flags);
        if (rc) return fail(KWS_ERROR_HIP, "NN kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    }
    // third tier: exact cmvnw (+ the int8 matrix-core network when it is fused there) from the same cepstra, then the exact network
    int ran_nn = 0;
    rc = kws_launch_cmvn_nn(h->dsp, h->nn, h->s_cep, (int)B, (h->is_float || want_f) ? fx : nullptr, q, (scores && !h->is_float) ? scores : nullptr, nullptr,
                            h->pooled_tap_bytes, nullptr, nullptr, grid_cap_nn(h), &ran_nn, s, h->d_flags2);
    if (rc) return fail(KWS_ERROR_HIP, "CMVN kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    if (!scores) return EI_IMPULSE_OK;
    if (h->is_float) rc = kws_launch_nn_f32(h->nnf, h->d_nnf, fx, (int)B, scores, h->tap_logits, h->n_cu, s, h->d_flags2);
    else if (!ran_nn) rc = kws_launch_nn(h->nn, q, (int)B, scores, nullptr, h->pooled_tap_bytes, nullptr, nullptr, grid_cap_nn(h), s, h->d_flags2);
    if (rc) return fail(KWS_ERROR_HIP, "NN kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    return EI_IMPULSE_OK;
}

static EI_IMPULSE_ERROR classify_fast_device(kws_handle *h, const int16_t *pcm, size_t B, float *scores, float *fx, bool want_f, int8_t *q,
                                             hipStream_t s)
{
    if (B > 0x7fffffff) return fail(KWS_ERROR_BAD_ARGUMENT, "batch too large");
    if (((uintptr_t)pcm & 15) != 0) return fail(KWS_ERROR_BAD_ARGUMENT, "pcm must be 16-byte aligned");
    EI_IMPULSE_ERROR e = ensure_flags(h, B);
    if (e) return e;
    HIP_TRY(hipMemsetAsync(h->d_flags, 0, sizeof(int), s));
    const bool fused = scores && h->is_float && h->fast_fused_ok;
    int rc = 0;
    if (h->model.dsp.block == DSP_BLOCK_MFE) {
        // extract_mfe_features (L432 classifier/ei_run_dsp.h:369-418) with the tolerance-mode front end: kws_fast_kernel<..., MFE> writes the
        // mel matrix (KissFFT-order FFT, fp32 power, fused mel products), then the block's own normalisation -- cmvnw(win, false, true) +
        // numpy::normalize, the exact kernel: it divides by the matrix's range, not by a window's deviation, so nothing is ill-conditioned
        // and no clip is handed back --, the input quantisation and the network's exact kernel
        const KwsDspPlan &P = h->dsp;
        const int rows = P.n_frames, cols = P.n_filters;
        if (rows > (cols > 16 ? 51 : 52) || P.win_size < (cols > 16 ? 17 : 13))
            return fail(KWS_ERROR_UNSUPPORTED_MODEL, "%d frames x %d filters, window %d outside the MFE normalisation kernel's limits", rows, cols, P.win_size);
        rc = kws_launch_fast(P, h->fast_plain, h->d_fast_plain, pcm, (int)B, nullptr, fx, nullptr, h->nn.in_scale, h->nn.in_zp, h->d_flags, h->d_flags + 1, h->n_cu, s);
        if (rc) return fail(KWS_ERROR_HIP, "fast kernel launch failed: %s (is the gfx950 code object present?)", hipGetErrorString((hipError_t)rc));
        rc = kws_launch_mfe_norm(fx, (int)B, rows, cols, P.win_size, P.pad_map, rows + 2 * P.pad, grid_cap_nn(h), s);
        if (!rc && q) rc = kws_launch_quantize(fx, q, B * h->model.nn_input_frame_size, h->nn.in_scale, h->nn.in_zp, s);
        if (rc) return fail(KWS_ERROR_HIP, "MFE block launch failed: %s", hipGetErrorString((hipError_t)rc));
        if (!scores) return EI_IMPULSE_OK;
        if (h->is_float) return nn_f32_device(h, fx, B, scores, h->tap_logits, s);
        rc = kws_launch_nn(h->nn, q, (int)B, scores, nullptr, h->pooled_tap_bytes, nullptr, nullptr, grid_cap_nn(h), s);
        if (rc) return fail(KWS_ERROR_HIP, "NN kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
        return EI_IMPULSE_OK;
    }
    if (h->fast_entry_tier >= 2 && (scores || want_f || q)) {
        // the graph's gain leaves the first tier no room (build_guard): every clip's cepstra come from the exact kernels, then the fast
        // cmvnw + network with the second tier's guard and the exact kernels for what that hands on (the continuous mode's path) -- or,
        // entry tier 3, the exact kernels throughout
        if (h->is_float && h->fast_fused_ok) {
            // float32 graph of the fused shapes: the DSP block by the exact kernels -- extract_mfcc_features' matrix, bit for bit: no feature
            // error, whatever the graph's gain -- and the network on the matrix cores from that matrix (kws_fast_kernel's feat_in form).
            // What is left of the guard is the network's own arithmetic (split 22-bit operands, KwsFastPlan::v_net_feat) against the clip's
            // own scores; a clip it does not clear is listed and goes through the exact network.  (configs[4].)
            HIP_TRY(hipMemsetAsync(h->d_flags2, 0, sizeof(int), s));
            if ((e = mfcc_fused_device(h, pcm, 0, B, fx, nullptr, s))) return e;
            if (!scores) return EI_IMPULSE_OK;                               // extract_mfcc_features only
            rc = kws_launch_fast_from_cepstra(h->dsp, h->fast_fused_cep, h->d_fast_fused_cep, fx, (int)B, scores, nullptr, nullptr, h->nn.in_scale, h->nn.in_zp,
                                              h->d_flags, h->d_flags + 1, h->n_cu, s, nullptr, h->tap_logits, 1);
            if (rc) return fail(KWS_ERROR_HIP, "fast kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
            rc = kws_launch_nn_f32(h->nnf, h->d_nnf, fx, (int)B, scores, h->tap_logits, h->n_cu, s, h->d_flags);
            if (rc) return fail(KWS_ERROR_HIP, "NN kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
            HIP_TRY(hipMemcpyAsync(h->d_flags2, h->d_flags, sizeof(int), hipMemcpyDeviceToDevice, s));      // those clips carry the exact mode's scores
            return EI_IMPULSE_OK;
        }
        if (h->fast_entry_tier >= 3) {
            HIP_TRY(hipMemsetAsync(h->d_flags2, 0, sizeof(int), s));
            if (!scores) return mfcc_fused_device(h, pcm, 0, B, fx, q, s);
            return classify_device(h, pcm, B, scores, fx, q, s);
        }
        const size_t F = h->model.nn_input_frame_size;
        if (B > h->cep_cap) {
            if (h->s_cep) (void)hipFree(h->s_cep);
            h->s_cep = nullptr; h->cep_cap = 0;
            HIP_TRY(hipMalloc((void **)&h->s_cep, B * F * sizeof(float)));
            h->cep_cap = B;
        }
        if ((e = spectral_device(h, h->dsp, pcm, 0, B, h->s_cep, nullptr, s))) return e;
        return cmvn_nn_fast_device(h, h->s_cep, B, scores, s, 0, 0, want_f ? fx : nullptr, q);
    }
    if (scores && !h->is_float && h->fast_q_ok) {
        // int8 graph of the matrix-core shape: the network runs in the same launch on the quantised tensor it has just produced in LDS;
        // the feature matrix / the tensor only go to HBM when the caller asked for them (q is the caller's buffer or the scratch the
        // exact re-run below needs anyway -- the fast kernel writes it only in the first case)
        rc = kws_launch_fast(h->dsp, h->fast_q, h->d_fast_q, pcm, (int)B, scores, want_f ? fx : nullptr, q != h->s_q ? q : nullptr, h->nn.in_scale, h->nn.in_zp,
                             h->d_flags, h->d_flags + 1, h->n_cu, s, h->d_nn);
        if (rc) return fail(KWS_ERROR_HIP, "fast kernel launch failed: %s (is the gfx950 code object present?)", hipGetErrorString((hipError_t)rc));
        return rerun_flagged_device(h, pcm, B, scores, fx, want_f, q, s);
    }
#ifdef KWS_DEV_SWITCHES
    if (fused && KWS_DEV_ENV("KWS_DEV_FAST_SPLIT")) {
        // development prototype ("build the two-kernel split and time it"): the spectral half + cmvnw as the
        // feature-emitting form, the features through HBM, the network half as the feat_in form of the from-cepstra kernel.  Timing only: the
        // first launch's guard assumes P = 1/4 and its list is not re-run here.
        rc = kws_launch_fast(h->dsp, h->fast_plain, h->d_fast_plain, pcm, (int)B, nullptr, fx, nullptr, h->nn.in_scale, h->nn.in_zp, h->d_flags, h->d_flags + 1, h->n_cu, s);
        if (rc) return fail(KWS_ERROR_HIP, "fast kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
        HIP_TRY(hipMemsetAsync(h->d_flags3, 0, sizeof(int), s));
        rc = kws_launch_fast_from_cepstra(h->dsp, h->fast_fused_cep, h->d_fast_fused_cep, fx, (int)B, scores, nullptr, nullptr, h->nn.in_scale, h->nn.in_zp,
                                          h->d_flags3, h->d_flags3 + 1, h->n_cu, s, nullptr, h->tap_logits, 1);
        if (rc) return fail(KWS_ERROR_HIP, "fast kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
        return EI_IMPULSE_OK;
    }
#endif
    if (fused && want_f) {
        // the fused kernel keeps the feature matrix on chip: a caller who also wants it gets it from the feature-emitting form first
        // (that launch knows no scores: its guard assumes the largest p (1 - p) there is, and its list decides for features and scores)
        rc = kws_launch_fast(h->dsp, h->fast_plain, h->d_fast_plain, pcm, (int)B, nullptr, fx, nullptr, h->nn.in_scale, h->nn.in_zp, h->d_flags, h->d_flags + 1,
                             h->n_cu, s);
        if (rc) return fail(KWS_ERROR_HIP, "fast kernel launch failed: %s (is the gfx950 code object present?)", hipGetErrorString((hipError_t)rc));
        HIP_TRY(hipMemsetAsync(h->d_flags3, 0, sizeof(int), s));
    }
    const KwsFastPlan &FP = fused ? h->fast_fused : h->fast_plain;
    int *const fl = (fused && want_f) ? h->d_flags3 : h->d_flags;          // the feature-emitting launch's list stands (kws_internal.h: d_flags3)
    rc = kws_launch_fast(h->dsp, FP, fused ? h->d_fast_fused : h->d_fast_plain, pcm, (int)B, scores, fused ? nullptr : fx, fused ? nullptr : q, h->nn.in_scale,
                         h->nn.in_zp, fl, fl + 1, h->n_cu, s, nullptr, h->tap_logits);
    if (rc) return fail(KWS_ERROR_HIP, "fast kernel launch failed: %s (is the gfx950 code object present?)", hipGetErrorString((hipError_t)rc));
    if (scores && !fused) {
        if (h->is_float) { if ((e = nn_f32_device(h, fx, B, scores, h->tap_logits, s))) return e; }
        else {
            rc = kws_launch_nn(h->nn, q, (int)B, scores, nullptr, h->pooled_tap_bytes, nullptr, nullptr, grid_cap_nn(h), s);
            if (rc) return fail(KWS_ERROR_HIP, "NN kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
        }
    }
    return rerun_flagged_device(h, pcm, B, scores, fx, want_f, q, s);
}

static EI_IMPULSE_ERROR classify_device(kws_handle *h, const int16_t *pcm, size_t B, float *scores, float *f, int8_t *q, hipStream_t s)
{
    EI_IMPULSE_ERROR e;
    if (h->is_float) {
        e = mfcc_fused_device(h, pcm, 0, B, f, nullptr, s);
        return e ? e : nn_f32_device(h, f, B, scores, h->tap_logits, s);
    }
    e = mfcc_fused_device(h, pcm, 0, B, (f || h->model.dsp.block != DSP_BLOCK_MFE) ? f : h->s_mfcc, q, s);
    if (e) return e;
    if (B > 0x7fffffff) return fail(KWS_ERROR_BAD_ARGUMENT, "batch too large");
    int rc = kws_launch_nn(h->nn, q, (int)B, scores, nullptr, h->pooled_tap_bytes, nullptr, nullptr, grid_cap_nn(h), s);
    if (rc) return fail(KWS_ERROR_HIP, "NN kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    return EI_IMPULSE_OK;
}

EI_IMPULSE_ERROR kws_run_classifier_batch_device(kws_handle *h, const int16_t *pcm, size_t B, float *scores, float *features,
                                                 int8_t *q_in, void *stream)
{
    if (!h || !pcm || !scores) return fail(KWS_ERROR_BAD_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    if (h->is_float && q_in) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "int8 input tensor requested from a float32 model");
    std::lock_guard<std::mutex> lk(h->mu);
    EI_IMPULSE_ERROR e = ensure_scratch(h, B);
    if (e) return e;
    ScratchUse use(h, (hipStream_t)stream);
    if (h->mode == KWS_MODE_FAST)
        return classify_fast_device(h, pcm, B, scores, features ? features : h->s_mfcc, features != nullptr,
                                    h->is_float ? nullptr : (q_in ? q_in : h->s_q), (hipStream_t)stream);
    return classify_device(h, pcm, B, scores, h->is_float ? (features ? features : h->s_mfcc) : features,
                           h->is_float ? nullptr : (q_in ? q_in : h->s_q), (hipStream_t)stream);
}

// Host buffers in, host buffers out.  The batch is cut into chunks that alternate between two streams, each with its own device
// buffers: while one chunk's kernels run, the next chunk's PCM crosses PCIe (pinned host memory: truly asynchronous; pageable
// memory: the runtime stages it, the other stream's kernels still overlap).  The device buffers live in the handle.
static const size_t kHostChunk = 8192;       // clips per chunk: 262 MB of PCM per buffer
static EI_IMPULSE_ERROR ensure_pipe(kws_handle *h, size_t chunk)
{
    kws_handle::HostPipe &p = h->pipe;
    const size_t n = h->model.raw_sample_count, F = h->model.nn_input_frame_size, C = h->model.labels.size();
    for (int k = 0; k < 2; ++k)
        if (!p.st[k] && hipStreamCreateWithFlags(&p.st[k], hipStreamNonBlocking) != hipSuccess) return fail(KWS_ERROR_HIP, "stream creation failed");
    if (chunk <= p.cap) return EI_IMPULSE_OK;
    for (int k = 0; k < 2; ++k) {
        for (void *q : { (void *)p.pcm[k], (void *)p.s[k], (void *)p.f[k], (void *)p.q[k] }) if (q) (void)hipFree(q);
        p.pcm[k] = nullptr; p.s[k] = nullptr; p.f[k] = nullptr; p.q[k] = nullptr;
    }
    p.cap = 0;
    for (int k = 0; k < 2; ++k) {
        if (hipMalloc((void **)&p.pcm[k], chunk * n * sizeof(int16_t)) != hipSuccess || hipMalloc((void **)&p.s[k], chunk * C * sizeof(float)) != hipSuccess ||
            hipMalloc((void **)&p.f[k], chunk * F * sizeof(float)) != hipSuccess || (!h->is_float && hipMalloc((void **)&p.q[k], chunk * F) != hipSuccess))
            return fail(EI_IMPULSE_ALLOC_FAILED, "device allocation failed");
    }
    p.cap = chunk;
    return EI_IMPULSE_OK;
}

EI_IMPULSE_ERROR kws_run_classifier_batch(kws_handle *h, const int16_t *pcm, size_t B, float *scores, float *features, int8_t *q_in)
{
    if (!h || !pcm || !scores) return fail(KWS_ERROR_BAD_ARGUMENT, "null argument");
    if (B == 0) return EI_IMPULSE_OK;
    HIP_TRY(hipSetDevice(h->device));
    if (h->is_float && q_in) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "int8 input tensor requested from a float32 model");
    const size_t n = h->model.raw_sample_count, F = h->model.nn_input_frame_size, C = h->model.labels.size();
    std::lock_guard<std::mutex> lk(h->pipe_mu);
    const size_t chunk = std::min(B, kHostChunk);
    EI_IMPULSE_ERROR e = ensure_pipe(h, chunk);
    if (e) return e;
    kws_handle::HostPipe &p = h->pipe;
    hipError_t he = hipSuccess;
    size_t i = 0;
    for (size_t off = 0; off < B && !e && he == hipSuccess; off += chunk, ++i) {
        const int k = (int)(i & 1);
        const size_t nb = std::min(chunk, B - off);
        he = hipMemcpyAsync(p.pcm[k], pcm + off * n, nb * n * sizeof(int16_t), hipMemcpyHostToDevice, p.st[k]);
        if (he != hipSuccess) break;
        e = classify_device(h, p.pcm[k], nb, p.s[k], (h->is_float || features || h->model.dsp.block == DSP_BLOCK_MFE) ? p.f[k] : nullptr,
                            h->is_float ? nullptr : p.q[k], p.st[k]);
        if (e) break;
        he = hipMemcpyAsync(scores + off * C, p.s[k], nb * C * sizeof(float), hipMemcpyDeviceToHost, p.st[k]);
        if (he == hipSuccess && features) he = hipMemcpyAsync(features + off * F, p.f[k], nb * F * sizeof(float), hipMemcpyDeviceToHost, p.st[k]);
        if (he == hipSuccess && q_in) he = hipMemcpyAsync(q_in + off * F, p.q[k], nb * F, hipMemcpyDeviceToHost, p.st[k]);
    }
    for (int k = 0; k < 2; ++k) { const hipError_t se = hipStreamSynchronize(p.st[k]); if (he == hipSuccess) he = se; }
    if (e) return e;
    if (he != hipSuccess) return fail(KWS_ERROR_HIP, "kws_run_classifier_batch: %s", hipGetErrorString(he));
    return EI_IMPULSE_OK;
}

#define TRY_OR_CLEAN(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return fail(KWS_ERROR_HIP, "%s: %s", #x, hipGetErrorString(e_)); } } while (0)
EI_IMPULSE_ERROR kws_nn_batch(kws_handle *h, const int8_t *q_in, size_t B, float *scores, int8_t *tap_pooled, int8_t *tap_fc, int8_t *tap_out)
{
    if (!h || !q_in || !scores) return fail(KWS_ERROR_BAD_ARGUMENT, "null argument");
    if (B == 0) return EI_IMPULSE_OK;
    HIP_TRY(hipSetDevice(h->device));
    const size_t F = h->model.nn_input_frame_size, C = h->model.labels.size(), PT = (size_t)h->pooled_tap_bytes;
    int8_t *d_q = nullptr, *d_p = nullptr, *d_fc = nullptr, *d_o = nullptr; float *d_s = nullptr;
    auto cleanup = [&]() { for (void *p : { (void *)d_q, (void *)d_p, (void *)d_fc, (void *)d_o, (void *)d_s }) if (p) (void)hipFree(p); };
    TRY_OR_CLEAN(hipMalloc((void **)&d_q, B * F));
    TRY_OR_CLEAN(hipMalloc((void **)&d_p, B * PT));
    TRY_OR_CLEAN(hipMalloc((void **)&d_fc, B * C));
    TRY_OR_CLEAN(hipMalloc((void **)&d_o, B * C));
    TRY_OR_CLEAN(hipMalloc((void **)&d_s, B * C * sizeof(float)));
    TRY_OR_CLEAN(hipMemcpy(d_q, q_in, B * F, hipMemcpyHostToDevice));
    EI_IMPULSE_ERROR e = kws_nn_batch_device(h, d_q, B, d_s, d_p, d_fc, d_o, nullptr);
    if (e == EI_IMPULSE_OK) {
        TRY_OR_CLEAN(hipMemcpy(scores, d_s, B * C * sizeof(float), hipMemcpyDeviceToHost));
        if (tap_pooled) TRY_OR_CLEAN(hipMemcpy(tap_pooled, d_p, B * PT, hipMemcpyDeviceToHost));
        if (tap_fc) TRY_OR_CLEAN(hipMemcpy(tap_fc, d_fc, B * C, hipMemcpyDeviceToHost));
        if (tap_out) TRY_OR_CLEAN(hipMemcpy(tap_out, d_o, B * C, hipMemcpyDeviceToHost));
    }
    cleanup();
    return e;
}

EI_IMPULSE_ERROR kws_mix_audio_device(const float *words, const int *word_len, size_t word_stride, const float *noise, size_t noise_len,
                                      const int *start, float word_vol, float bg_vol, size_t n_clips, size_t n, int16_t *out, void *stream)
{
    if (!out || (words && !word_len) || (noise && !start) || n == 0 || n > 0x7fffffff) return fail(KWS_ERROR_BAD_ARGUMENT, "kws_mix_audio_device: bad argument");
    if (noise && noise_len < n) return fail(KWS_ERROR_BAD_ARGUMENT, "background track shorter than a clip");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(KWS_ERROR_HIP, "no HIP device available (libkws_mi355x has no CPU fallback)");
    int rc = kws_launch_mix_audio(words, word_len, word_stride, noise, start, word_vol, bg_vol, (int)n, n_clips, out, (hipStream_t)stream);
    if (rc) return fail(KWS_ERROR_HIP, "mix kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    return EI_IMPULSE_OK;
}

EI_IMPULSE_ERROR kws_synth_clips_device(uint32_t seed, uint32_t first_clip, uint32_t n_clips, uint32_t clip_len, int16_t *out, void *stream)
{
    if (!out) return fail(KWS_ERROR_BAD_ARGUMENT, "null argument");
    int ndev = 0, dev = 0;      // no handle here: make sure the runtime is up before the first launch of the process
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || hipGetDevice(&dev) != hipSuccess)
        return fail(KWS_ERROR_HIP, "no HIP device available (libkws_mi355x has no CPU fallback)");
    int rc = kws_launch_synth(seed, first_clip, n_clips, clip_len, out, (hipStream_t)stream);
    if (rc) return fail(KWS_ERROR_HIP, "synth kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    return EI_IMPULSE_OK;
}

EI_IMPULSE_ERROR kws_device_malloc(void **ptr, size_t nbytes) { HIP_TRY(hipMalloc(ptr, nbytes)); return EI_IMPULSE_OK; }
EI_IMPULSE_ERROR kws_device_free(void *ptr) { HIP_TRY(hipFree(ptr)); return EI_IMPULSE_OK; }
EI_IMPULSE_ERROR kws_memcpy_h2d(void *dst, const void *src, size_t n) { HIP_TRY(hipMemcpy(dst, src, n, hipMemcpyHostToDevice)); return EI_IMPULSE_OK; }
EI_IMPULSE_ERROR kws_memcpy_d2h(void *dst, const void *src, size_t n) { HIP_TRY(hipMemcpy(dst, src, n, hipMemcpyDeviceToHost)); return EI_IMPULSE_OK; }
EI_IMPULSE_ERROR kws_device_synchronize(void) { HIP_TRY(hipDeviceSynchronize()); return EI_IMPULSE_OK; }

}  // extern "C"
#pragma GCC visibility pop
