// kws_sdk.cpp -- continuous mode for many streams (kws_streams_*) and the SDK-compatible single-clip entry points of
// include/kws/ei_compat.h (run_classifier, run_classifier_continuous, run_inference, ...).
#include "kws_internal.h"

#pragma GCC visibility push(default)     // the library is built with -fvisibility=hidden: only the C ABI is exported
extern "C" {
// ------------------------------------------------------------------------------------------------------------
//  continuous mode for S streams in lock step ("many concurrent streams, per-stream state in HBM")
// ------------------------------------------------------------------------------------------------------------
struct kws_stream_batch {
    kws_handle *h = nullptr;
    size_t S = 0;
    float *feat[2] = { nullptr, nullptr };   // rolling cepstra buffers [S][F] (ping-pong for the shift)
    float *running_sum = nullptr, *maf_buf = nullptr;   // [S][C], [S][C][taps]
    float *zeros = nullptr;                   // [S] end-of-signal samples when the caller gives none
    size_t slice_offset = 0;
    bool full = false, first_run = false;     // first_run: like the reference's function-static, never reset
    // Once the buffer is full the reference shifts it by one slice after every inference (ei_run_classifier.h:277-279).  Here the
    // first ring_rows rows (everything up to the end of the slice being written) form a ring whose head advances instead; the rows
    // behind them are the reference's never-written tail and stay in place.
    int ring_rows = 0, head = 0;
    uint32_t buf_idx = 0;
};
static const int kMafTaps = EI_CLASSIFIER_SLICES_PER_MODEL_WINDOW >> 1;

void kws_streams_destroy(kws_stream_batch *sb)
{
    if (!sb) return;
    for (void *p : { (void *)sb->feat[0], (void *)sb->feat[1], (void *)sb->running_sum, (void *)sb->maf_buf, (void *)sb->zeros })
        if (p) (void)hipFree(p);
    delete sb;
}

EI_IMPULSE_ERROR kws_streams_init(kws_stream_batch *sb)          // run_classifier_init for every stream
{
    if (!sb) return fail(KWS_ERROR_BAD_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(sb->h->device));
    const size_t C = sb->h->model.labels.size();
    if (sb->ring_rows && sb->head) {
        // run_classifier_init does not clear the feature buffer: put the rows back in plain order before slices are written
        // linearly again
        const KwsDspPlan &P = sb->h->dsp;
        const int rows = (int)(sb->h->model.nn_input_frame_size / (size_t)P.n_cepstral);
        // the last step may still be writing feat[0] on the caller's stream (a hipStreamNonBlocking stream is not ordered with the
        // default stream the copy goes to)
        HIP_TRY(hipDeviceSynchronize());
        int rc = kws_launch_unring(sb->feat[0], sb->feat[1], (int)sb->S, rows, P.n_cepstral, sb->ring_rows, sb->head, nullptr);
        if (rc) return fail(KWS_ERROR_HIP, "copy kernel launch failed");
        HIP_TRY(hipDeviceSynchronize());
        std::swap(sb->feat[0], sb->feat[1]);
    }
    sb->ring_rows = 0; sb->head = 0;
    sb->slice_offset = 0;
    sb->full = false;
    sb->buf_idx = 0;
    HIP_TRY(hipMemset(sb->running_sum, 0, sb->S * C * sizeof(float)));
    HIP_TRY(hipMemset(sb->maf_buf, 0, sb->S * C * kMafTaps * sizeof(float)));
    return EI_IMPULSE_OK;
}

EI_IMPULSE_ERROR kws_streams_create(kws_handle *h, size_t S, kws_stream_batch **out)
{
    if (!h || !out || S == 0 || S > 0x7fffffff) return fail(KWS_ERROR_BAD_ARGUMENT, "bad argument");
    *out = nullptr;
    HIP_TRY(hipSetDevice(h->device));
    kws_stream_batch *sb = new kws_stream_batch();
    sb->h = h; sb->S = S;
    const size_t F = h->model.nn_input_frame_size, C = h->model.labels.size();
    bool ok = hipMalloc((void **)&sb->feat[0], S * F * sizeof(float)) == hipSuccess &&
              hipMalloc((void **)&sb->feat[1], S * F * sizeof(float)) == hipSuccess &&
              hipMalloc((void **)&sb->running_sum, S * C * sizeof(float)) == hipSuccess &&
              hipMalloc((void **)&sb->maf_buf, S * C * kMafTaps * sizeof(float)) == hipSuccess &&
              hipMalloc((void **)&sb->zeros, S * sizeof(float)) == hipSuccess;
    if (ok) ok = hipMemset(sb->feat[0], 0, S * F * sizeof(float)) == hipSuccess && hipMemset(sb->feat[1], 0, S * F * sizeof(float)) == hipSuccess &&
                 hipMemset(sb->zeros, 0, S * sizeof(float)) == hipSuccess;
    if (!ok) { kws_streams_destroy(sb); return fail(EI_IMPULSE_ALLOC_FAILED, "device allocation failed"); }
    EI_IMPULSE_ERROR e = kws_streams_init(sb);
    if (e) { kws_streams_destroy(sb); return e; }
    *out = sb;
    return EI_IMPULSE_OK;
}

EI_IMPULSE_ERROR kws_streams_step_device(kws_stream_batch *sb, const int16_t *slices, size_t slice_samples, const float *end_of_signal,
                                         float *scores, int *produced, void *stream)
{
    if (!sb || !slices || !scores || !produced) return fail(KWS_ERROR_BAD_ARGUMENT, "null argument");
    kws_handle *h = sb->h;
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    const Model &m = h->model;
    const size_t F = m.nn_input_frame_size, C = m.labels.size(), S = sb->S;
    *produced = 0;
    // extract_mfcc_per_slice_features: every step but the first claims one more frame length (ei_run_dsp.h:319-325)
    size_t n_claimed = slice_samples;
    const bool grown = sb->first_run;           // committed below, once the slice has been accepted and enqueued
    if (grown) n_claimed += (size_t)(m.dsp.frame_length * (float)m.frequency);
    const int frame_len = h->dsp.frame_len, stride = h->dsp.frame_stride;
    const int nf = n_claimed >= (size_t)frame_len ? (int)floorf((float)(n_claimed - (size_t)frame_len) / (float)stride) : 0;
    const size_t feature_size = (size_t)(nf > 0 ? nf : 0) * (size_t)h->dsp.n_cepstral;
    if (nf < 1 || (!h->dsp.generic && nf > kws_mfcc_max_frames(h->dsp.n_filters)) || feature_size > F || sb->slice_offset + feature_size > F ||
        (size_t)(nf - 1) * stride + std::min(h->dsp.fft_len, frame_len) > slice_samples || (!h->dsp.generic && (slice_samples * 2) % 16 != 0))
        return fail(EI_IMPULSE_DSP_ERROR, "slice of %zu samples (claimed %zu) yields %d frames", slice_samples, n_claimed, nf);
    KwsDspPlan P = h->dsp;
    P.n_samples = (int)slice_samples;      // memory stride between the streams' slices
    P.n_frames = nf;
    // x[-1] of the slice: the reference takes the sample at total_length-1, which lies beyond the slice once it has grown
    const float *wrap = grown ? (end_of_signal ? end_of_signal : sb->zeros) : nullptr;
    float *feat = sb->feat[0];
    const int ncols = h->dsp.n_cepstral, row0 = (int)(sb->slice_offset / (size_t)ncols);
    EI_IMPULSE_ERROR e;
    if (sb->ring_rows && row0 + nf != sb->ring_rows)
        return fail(EI_IMPULSE_DSP_ERROR, "slice of %d frames in a window laid out for slices of %d", nf, sb->ring_rows - row0);
    if (sb->ring_rows) {                        // steady state: this slice's rows go behind the ring's head
        P.ring_rows = sb->ring_rows;
        P.ring_row0 = row0 + sb->head;
        e = spectral_device(h, P, slices, 0, S, feat, wrap, st, (int)F);
    } else
        e = spectral_device(h, P, slices, 0, S, feat + sb->slice_offset, wrap, st, (int)F);
    if (e) return e;
    sb->first_run = true;
    if (!sb->full) {
        sb->slice_offset += feature_size;
        if (sb->slice_offset > (F - feature_size)) {
            sb->full = true;
            sb->slice_offset -= feature_size;
            sb->ring_rows = (int)((sb->slice_offset + feature_size) / (size_t)ncols);      // rows written so far; head = 0: still plain
            sb->head = 0;
        }
    }
    if (!sb->full) return EI_IMPULSE_OK;
    {
        std::lock_guard<std::mutex> lk(h->mu);
        e = ensure_scratch(h, S);
        if (!e) {
            ScratchUse use(h, st);
            if (h->mode == KWS_MODE_FAST && h->fast_plain_ok)
                e = cmvn_nn_fast_device(h, feat, S, scores, st, sb->ring_rows, sb->head);
            else
                e = cmvn_nn_device(h, feat, S, nullptr, nullptr, scores, nullptr, nullptr, nullptr, st, sb->ring_rows, sb->head);
        }
    }
    if (e) return e;
    int rc = kws_launch_maf(scores, sb->running_sum, sb->maf_buf, (int)(S * C), (int)sb->buf_idx, kMafTaps, st);
    if (rc) return fail(KWS_ERROR_HIP, "moving-average kernel launch failed");
    if (++sb->buf_idx >= (uint32_t)kMafTaps) sb->buf_idx = 0;
    // "shift the feature buffer for new data": the ring's head moves on by one slice
    sb->head = (sb->head + nf) % sb->ring_rows;
    *produced = 1;
    return EI_IMPULSE_OK;
}

// ------------------------------------------------------------------------------------------------------------
//  SDK-compatible single-clip entry points
// ------------------------------------------------------------------------------------------------------------
// workspace for one window / slice of n_x float samples
static EI_IMPULSE_ERROR ensure_ws(kws_handle *h, size_t n_x)
{
    kws_handle::Ws &w = h->ws;
    const size_t F = h->model.nn_input_frame_size, C = h->model.labels.size();
    auto oom = [&]() { return fail(EI_IMPULSE_ALLOC_FAILED, "device allocation failed"); };
    if (!w.st) {
        if (hipStreamCreateWithFlags(&w.st, hipStreamNonBlocking) != hipSuccess) return fail(KWS_ERROR_HIP, "stream creation failed");
        for (hipEvent_t &ev : w.ev) if (hipEventCreate(&ev) != hipSuccess) return fail(KWS_ERROR_HIP, "event creation failed");
        if (hipMalloc((void **)&w.d_f, F * sizeof(float)) != hipSuccess || hipMalloc((void **)&w.d_s, C * sizeof(float)) != hipSuccess ||
            hipMalloc((void **)&w.d_w, 16) != hipSuccess || hipMalloc((void **)&w.d_q, F + 16) != hipSuccess ||
            hipHostMalloc((void **)&w.h_s, C * sizeof(float), hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc((void **)&w.h_f, F * sizeof(float), hipHostMallocDefault) != hipSuccess)
            return oom();
    }
    if (n_x > w.cap_x) {
        if (w.d_x) (void)hipFree(w.d_x);
        if (w.h_x) (void)hipHostFree(w.h_x);
        w.d_x = nullptr; w.h_x = nullptr; w.cap_x = 0;
        if (hipMalloc((void **)&w.d_x, n_x * sizeof(float)) != hipSuccess ||
            hipHostMalloc((void **)&w.h_x, n_x * sizeof(float), hipHostMallocDefault) != hipSuccess)
            return oom();
        // (the one-shot path fills the window frame by frame: samples no frame covers -- the tail behind the last frame, the slack -- are
        //  copied to the device with the rest and must not be whatever the allocator left there: ADVICE round 4)
        memset(w.h_x, 0, n_x * sizeof(float));
        w.cap_x = n_x;
    }
    return EI_IMPULSE_OK;
}

static kws_handle *g_default = nullptr;
static bool g_default_owned = false;
static std::mutex g_default_mu;

// kws_destroy(h) of the handle that is installed as the default model: the SDK entry points must not find it any more
void kws_sdk_forget_default(kws_handle *h)
{
    std::lock_guard<std::mutex> lk(g_default_mu);
    if (g_default == h) { g_default = nullptr; g_default_owned = false; }
}

// The application's ei_impulse_result_t is sized by ITS EI_CLASSIFIER_LABEL_COUNT (include/kws/ei_compat.h publishes it as
// kws_app_label_count; absent when the caller is not a C program built with that header, e.g. ctypes).  Writing a result of
// another label count would run past the caller's struct.
extern "C" { __attribute__((weak)) extern const int kws_app_label_count; }
static EI_IMPULSE_ERROR check_result_layout(const kws_handle *h)
{
    if (&kws_app_label_count && kws_app_label_count != (int)h->model.labels.size())
        return fail(EI_IMPULSE_ERROR_SHAPES_DONT_MATCH, "the application was compiled for EI_CLASSIFIER_LABEL_COUNT = %d, the loaded model has %zu labels",
                    kws_app_label_count, h->model.labels.size());
    return EI_IMPULSE_OK;
}

EI_IMPULSE_ERROR kws_set_default_model(kws_handle *h)
{
    kws_handle *old = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_default_mu);
        if (g_default && g_default_owned && g_default != h) old = g_default;
        g_default = h;
        g_default_owned = false;
    }
    if (old) kws_destroy(old);                 // outside the lock: kws_destroy() asks whether it destroys the default
    return EI_IMPULSE_OK;
}

kws_handle *kws_default_model(void)
{
    {
        std::lock_guard<std::mutex> lk(g_default_mu);
        if (g_default) return g_default;
    }
    // The model is created OUTSIDE the lock: when a plan builder refuses the file, kws_create() tears the handle down with
    // kws_destroy(), which asks kws_sdk_forget_default() -- under g_default_mu -- whether it is destroying the default.
    const char *path = getenv("KWS_MODEL");
    if (!path) { fail(KWS_ERROR_NO_MODEL, "no model: call kws_set_default_model() or set KWS_MODEL"); return nullptr; }
    const char *dev = getenv("KWS_DEVICE");
    kws_handle *h = nullptr;
    if (kws_create_from_file(path, dev ? atoi(dev) : 0, &h) != EI_IMPULSE_OK) return nullptr;
    kws_handle *loser = nullptr, *winner = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_default_mu);
        if (g_default) loser = h;                  // another thread installed a model in the meantime: keep that one
        else { g_default = h; g_default_owned = true; }
        winner = g_default;
    }
    if (loser) kws_destroy(loser);
    return winner;
}

// The label count of the loaded model decides where `anomaly` and `timing` sit behind the classification array
// (classifier/ei_classifier_types.h:41-45): the caller's ei_impulse_result_t must be compiled for that model.
static void fill_result(const kws_handle *h, ei_impulse_result_t *result, const float *scores, bool debug, int ms)
{
    const int C = (int)h->model.labels.size();
    ei_impulse_result_classification_t *cls = (ei_impulse_result_classification_t *)result;
    ei_impulse_result_timing_t *timing = (ei_impulse_result_timing_t *)((char *)result + (size_t)C * sizeof(*cls) + sizeof(float));
    timing->classification = ms;
    if (debug) ei_printf("Predictions (time: %d ms.):\n", ms);
    for (int ix = 0; ix < C; ix++) {
        if (debug) { ei_printf("%s:\t", h->model.labels[ix].c_str()); ei_printf_float(scores[ix]); ei_printf("\n"); }
        cls[ix].label = h->model.labels[ix].c_str();
        cls[ix].value = scores[ix];
    }
}
// run_inference polls the cancellation hook twice once the result is written: behind the classifier (ei_run_classifier.h:489-491) and at
// its end (:636-638, behind the -- absent -- anomaly block); the second poll is not made when the first one cancels.
static EI_IMPULSE_ERROR inference_polls(void)
{
    if (ei_run_impulse_check_canceled() == EI_IMPULSE_CANCELED) return EI_IMPULSE_CANCELED;
    if (ei_run_impulse_check_canceled() == EI_IMPULSE_CANCELED) return EI_IMPULSE_CANCELED;
    return EI_IMPULSE_OK;
}
static ei_impulse_result_timing_t *result_timing(const kws_handle *h, ei_impulse_result_t *result)
{
    const size_t C = h->model.labels.size();
    return (ei_impulse_result_timing_t *)((char *)result + C * sizeof(ei_impulse_result_classification_t) + sizeof(float));
}

EI_IMPULSE_ERROR run_inference(ei_matrix_t *fmatrix, ei_impulse_result_t *result, bool debug)
{
    kws_handle *h = kws_default_model();
    if (!h) return kws_last_error_code() != EI_IMPULSE_OK ? kws_last_error_code() : KWS_ERROR_NO_MODEL;   // why the default model is missing
    if (!fmatrix || !fmatrix->buffer || !result) return fail(KWS_ERROR_BAD_ARGUMENT, "null argument");
    if (EI_IMPULSE_ERROR le = check_result_layout(h)) return le;
    const size_t F = h->model.nn_input_frame_size, C = h->model.labels.size();
    if ((size_t)fmatrix->rows * fmatrix->cols != F) return fail(EI_IMPULSE_ERROR_SHAPES_DONT_MATCH, "feature matrix is %ux%u, model needs %zu", fmatrix->rows, fmatrix->cols, F);
    HIP_TRY(hipSetDevice(h->device));
    std::lock_guard<std::mutex> sdk_lk(h->sdk_mu);
    uint64_t t0 = ei_read_timer_ms();
    std::vector<float> scores(C);
    EI_IMPULSE_ERROR e = ensure_ws(h, 1);
    if (e) return e;
    kws_handle::Ws &w = h->ws;
    memcpy(w.h_f, fmatrix->buffer, F * sizeof(float));
    if (hipMemcpyAsync(w.d_f, w.h_f, F * sizeof(float), hipMemcpyHostToDevice, w.st) != hipSuccess) e = fail(KWS_ERROR_HIP, "h2d copy failed");
    if (!e) e = kws_run_inference_batch_device(h, w.d_f, 1, w.d_s, w.st);
    if (!e && (hipMemcpyAsync(w.h_s, w.d_s, C * sizeof(float), hipMemcpyDeviceToHost, w.st) != hipSuccess ||
               hipStreamSynchronize(w.st) != hipSuccess)) e = fail(KWS_ERROR_HIP, "d2h copy failed");
    if (e) return e;
    memcpy(scores.data(), w.h_s, C * sizeof(float));
    fill_result(h, result, scores.data(), debug, (int)(ei_read_timer_ms() - t0));
    return inference_polls();
}

// run_classifier's device work on the workspace stream: H2D of the window, extract_mfcc_features, the network, D2H of the
// scores.  debug: the stream is drained after the DSP block and the features are printed, as the SDK's sequential code does.
// other: the plan for a window of another length than the model's (kws_plan_for_length) -- its fewer rows fill the head of the feature
// matrix, the rest stays at the zeros the reference's calloc left there (ei_run_classifier.h: features_matrix), and the network's input
// is quantised from that whole matrix as run_inference does.
static EI_IMPULSE_ERROR oneshot_enqueue(kws_handle *h, kws_handle::Ws &w, size_t n, size_t C, bool debug, uint64_t t0, int *dsp_ms, uint64_t *t1,
                                        const KwsDspPlan *other = nullptr)
{
    const size_t F = h->model.nn_input_frame_size;
    EI_IMPULSE_ERROR e = EI_IMPULSE_OK;
    (void)hipEventRecord(w.ev[0], w.st);
    if (hipMemcpyAsync(w.d_x, w.h_x, n * sizeof(float), hipMemcpyHostToDevice, w.st) != hipSuccess) e = fail(KWS_ERROR_HIP, "h2d copy failed");
    if (!e && other) {
        if (hipMemsetAsync(w.d_f, 0, F * sizeof(float), w.st) != hipSuccess) e = fail(KWS_ERROR_HIP, "memset failed");
        if (!e) e = mfcc_fused_device_plan(h, *other, w.d_x, 1, 1, w.d_f, nullptr, w.st);
    } else if (!e) e = mfcc_fused_device(h, w.d_x, 1, 1, w.d_f, h->is_float ? nullptr : w.d_q, w.st);
    (void)hipEventRecord(w.ev[1], w.st);                     // the stage boundary of ei_run_classifier.h:669 / 696, on the device's clock
    if (debug) {
        if (!e && hipMemcpyAsync(w.h_f, w.d_f, F * sizeof(float), hipMemcpyDeviceToHost, w.st) != hipSuccess) e = fail(KWS_ERROR_HIP, "d2h copy failed");
        if (!e && hipStreamSynchronize(w.st) != hipSuccess) e = fail(KWS_ERROR_HIP, "MFCC kernel failed");
        if (e) return e;
        if (ei_run_impulse_check_canceled() == EI_IMPULSE_CANCELED) return EI_IMPULSE_CANCELED;   // ei_run_classifier.h:689-691
        *dsp_ms = (int)(ei_read_timer_ms() - t0);
        ei_printf("Features (%d ms.): ", *dsp_ms);
        for (size_t ix = 0; ix < F; ix++) { ei_printf_float(w.h_f[ix]); ei_printf(" "); }
        ei_printf("\n");
        ei_printf("Running neural network...\n");
        *t1 = ei_read_timer_ms();
    }
    if (!e && other) e = kws_run_inference_batch_device(h, w.d_f, 1, w.d_s, w.st);
    else if (!e) e = h->is_float ? nn_f32_device(h, w.d_f, 1, w.d_s, nullptr, w.st) : kws_nn_batch_device(h, w.d_q, 1, w.d_s, nullptr, nullptr, nullptr, w.st);
    if (!e && hipMemcpyAsync(w.h_s, w.d_s, C * sizeof(float), hipMemcpyDeviceToHost, w.st) != hipSuccess) e = fail(KWS_ERROR_HIP, "d2h copy failed");
    (void)hipEventRecord(w.ev[2], w.st);
    return e;
}

// milliseconds between two recorded (and completed) events of the workspace stream, rounded like the SDK's integer fields
static int stage_ms(hipEvent_t a, hipEvent_t b)
{
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, a, b) != hipSuccess) return 0;
    return (int)(ms + 0.5f);
}

// The caller's callback is asked for exactly what the reference's DSP block asks it, in the reference's order (a stateful callback --
// a file reader, a ring buffer that counts its reads -- sees the same sequence; 98 calls for the shipped 49-frame window):
//   processing.hpp:68       the pre-emphasis constructor: the window's last `shift` samples, return value unchecked (buffer calloc'd)
//   processing.hpp:86-94    per frame of speechpy::feature::mfe's loop (feature.hpp:263-281, through ei_run_dsp.h:252-253): the `shift`
//                           samples before the frame unless it starts inside the first `shift`, then the frame's frame_length samples
// An MFE block (L432 copy, ei_run_dsp.h:369-418, 420-470) hands the application's signal to feature::mfe itself -- no pre-emphasis object:
// the frames only (shift = 0 here), 49 calls for its window.
// Every answer lands at its own offset of the window x (what the kernels read: the frames, the sample before each, the wrap sample).
// The constructor runs before the reference sizes its output (ei_run_dsp.h:267, 279-286): its call is made on the error paths too.
// gather_frames_like_reference returns the callback's first non-zero return value (EIDSP_ERR(ret) in the reference), 0 otherwise.
static int gather_shift(const kws_handle *h) { return h->model.dsp.block == DSP_BLOCK_MFE ? 0 : h->dsp.pre_shift; }
static void gather_constructor_call(signal_t *signal, size_t total_length, int shift, float *eos)
{
    for (int i = 0; i < shift; ++i) eos[i] = 0.0f;
    if (shift > 0 && total_length >= (size_t)shift) (void)signal->get_data(total_length - (size_t)shift, (size_t)shift, eos);
}
static int gather_frames_like_reference(signal_t *signal, float *x, int n_frames, int frame_len, int stride, int shift)
{
    for (int f = 0; f < n_frames; ++f) {
        const size_t off = (size_t)f * (size_t)stride;
        // (the reference's shortening of a frame that would pass the end, feature.hpp:267-270, cannot fire: the frame count comes from the length)
        if (shift > 0 && off >= (size_t)shift) {
            const int r = signal->get_data(off - (size_t)shift, (size_t)shift, x + off - shift);
            if (r != 0) return r;
        }
        const int r = signal->get_data(off, (size_t)frame_len, x + off);
        if (r != 0) return r;
    }
    return 0;
}

EI_IMPULSE_ERROR run_classifier(signal_t *signal, ei_impulse_result_t *result, bool debug)
{
    kws_handle *h = kws_default_model();
    if (!h) return kws_last_error_code() != EI_IMPULSE_OK ? kws_last_error_code() : KWS_ERROR_NO_MODEL;   // why the default model is missing
    if (!signal || !signal->get_data || !result) return fail(KWS_ERROR_BAD_ARGUMENT, "null argument");
    if (EI_IMPULSE_ERROR le = check_result_layout(h)) return le;
    const size_t n = h->model.raw_sample_count, C = h->model.labels.size();
    // The reference sizes its frame count from signal->total_length (ei_run_dsp.h:277-286, processing.hpp:194-284).  A window with 1 ..
    // n_frames frames (640 .. 16 319 samples for the shipped impulse) it classifies from the frames that fit: normalised among themselves,
    // x[-1] = the last sample of THAT window, the rest of the network's input at its calloc'd zeros -- 15 999 samples give 48 frames, 16 001
    // give 49 with another wrap sample.  The same here, on the handle's plan re-targeted at that length; recorded from the compiled
    // reference in tests/golden/other_length_l476.npz.
    // Outside that range the reference has no result to match: its check for a matrix that does not fit (ei_run_dsp.h:279-284) and
    // feature.hpp's EIDSP_INPUT_MATRIX_EMPTY end in EIDSP_ERR = printf + assert(false) (dsp/config.hpp:65-67; the return-code form does not
    // compile for this SDK copy) -- an abort, or under NDEBUG a run into the overflow (observed: 16 320, 16 321, 16 640, 17 000 samples) or
    // a crash (639).  The library returns EI_IMPULSE_DSP_ERROR there with the message run_classifier prints for a failed DSP block.  An MFE
    // block follows the same rule (L432 ei_run_dsp.h:379-389; that copy's run_classifier cannot be compiled here, so its case is pinned by
    // the copy's compiled leaves: tests/golden/mfe_other_length_l432.npz).
    const size_t n2 = signal->total_length;
    const int shift = gather_shift(h);
    const int nfr2 = n2 == n ? h->dsp.n_frames : kws_frames_for_length(h, n2);
    if (n2 != n && (nfr2 > h->dsp.n_frames || nfr2 < 1)) {
        float eos_unused = 0.0f;                                                 // the pre-emphasis object exists by then: its call has been made
        gather_constructor_call(signal, n2, shift, &eos_unused);
        if (nfr2 > h->dsp.n_frames) {                                            // ei_run_dsp.h:280-282
            ei_printf("out_matrix = %hux%hu\n", (unsigned short)1, (unsigned short)h->model.nn_input_frame_size);
            ei_printf("calculated size = %hux%hu\n", (unsigned short)nfr2, (unsigned short)h->dsp.n_cepstral);
        }
        const int code = nfr2 < 1 ? -1006 : -1002;                               // EIDSP_INPUT_MATRIX_EMPTY / EIDSP_MATRIX_SIZE_MISMATCH
        ei_printf("ERR: Failed to run DSP process (%d)\n", code);
        return fail(EI_IMPULSE_DSP_ERROR, "signal length %zu (%d frames), model window %zu (%d frames)", n2, nfr2, n, h->dsp.n_frames);
    }
    HIP_TRY(hipSetDevice(h->device));
    std::lock_guard<std::mutex> sdk_lk(h->sdk_mu);
    uint64_t t0 = ei_read_timer_ms();
    EI_IMPULSE_ERROR e = ensure_ws(h, std::max(n, n2) + 16);
    if (e) return e;
    kws_handle::Ws &w = h->ws;
    KwsDspPlan other_plan;
    if (n2 != n && (e = kws_plan_for_length(h, n2, &other_plan))) return e;
    // gather the window through the caller's callback (float samples, as the SDK hands them to the DSP block) straight into pinned
    // memory, with the reference's own sequence of calls; samples no frame covers (between frames of a stride above the frame length,
    // behind the last frame) are never asked for -- nor read by the kernels
    {
        float eos = 0.0f;                                                        // (the plan admits pre_shift == 1 only)
        const int nfr = nfr2, frame_len = h->dsp.frame_len, stride = h->dsp.frame_stride;
        gather_constructor_call(signal, n2, shift, &eos);
        const int r = gather_frames_like_reference(signal, w.h_x, nfr, frame_len, stride, shift);
        if (r != 0) { ei_printf("ERR: Failed to run DSP process (%d)\n", r); return fail(EI_IMPULSE_DSP_ERROR, "signal->get_data returned %d", r); }
        // x[-1] of the window is its last sample (processing.hpp:104-106 reads _end_of_signal_buffer): where no frame reaches the
        // window's end (the shipped shape: the last frame ends at 15 680 of 16 000) the constructor's answer is the only copy.  A frame
        // that covers it has fetched the same sample (a callback that fails the first call only would differ: 0 in the reference)
        if (shift > 0 && (size_t)(nfr - 1) * (size_t)stride + (size_t)frame_len < n2) w.h_x[n2 - 1] = eos;
    }
    int dsp_ms = 0;
    uint64_t t1 = t0;
    // One window is launch-bound (one wave of work per kernel): without debug both stages are queued back to back and the host
    // waits once; the cancellation hook the SDK polls between the stages (ei_run_classifier.h:689-691) is polled after the
    // wait with the same return value.  (Replaying the four stream operations as a captured HIP graph was measured too:
    // 90.7 vs 90.3 us per call, no gain on this runtime -- most of the call is the single wave's MFCC.)
    e = oneshot_enqueue(h, w, n2, C, debug, t0, &dsp_ms, &t1, n2 != n ? &other_plan : nullptr);
    if (e == EI_IMPULSE_CANCELED) return e;
    if (!e && hipStreamSynchronize(w.st) != hipSuccess) e = fail(KWS_ERROR_HIP, "run_classifier: device work failed");
    if (e) return e;
    // (with debug the poll behind the DSP block has been made where the SDK makes it, before the features are printed: oneshot_enqueue)
    if (!debug && ei_run_impulse_check_canceled() == EI_IMPULSE_CANCELED) return EI_IMPULSE_CANCELED;
    // timing.dsp / timing.classification (ei_run_classifier.h:669, 696: the SDK reads its millisecond timer around each stage).  Both
    // stages are queued before the host waits once, so the split comes from events on the stream: the callback gather (host) + H2D +
    // DSP block kernels, and the network + D2H.  With debug the host drained the stream between the stages and its own timer is used.
    int nn_ms;
    if (debug) nn_ms = (int)(ei_read_timer_ms() - t1);
    else {
        const int wall = (int)(ei_read_timer_ms() - t0);
        nn_ms = stage_ms(w.ev[1], w.ev[2]);
        dsp_ms = std::max(wall - nn_ms, stage_ms(w.ev[0], w.ev[1]));
    }
    result_timing(h, result)->dsp = dsp_ms;
    std::vector<float> scores(C);
    memcpy(scores.data(), w.h_s, C * sizeof(float));
    fill_result(h, result, scores.data(), debug, nn_ms);
    return inference_polls();
}

// development / test aid (not in the public headers): the feature matrix the last run_classifier() call of the default model classified
// (the DSP block's output with the zero tail of a shorter window), copied from the device
EI_IMPULSE_ERROR kws_dev_oneshot_features(float *out, size_t n)
{
    kws_handle *h = kws_default_model();
    if (!h || !out) return fail(KWS_ERROR_BAD_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> sdk_lk(h->sdk_mu);
    if (!h->ws.d_f || n != h->model.nn_input_frame_size) return fail(KWS_ERROR_BAD_ARGUMENT, "no one-shot call yet, or %zu is not the model's feature count", n);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpy(out, h->ws.d_f, n * sizeof(float), hipMemcpyDeviceToHost));
    return EI_IMPULSE_OK;
}

// ei_run_classifier.h:134-145
float run_moving_average_filter(ei_impulse_maf *maf, float classification)
{
    maf->running_sum -= maf->maf_buffer[maf->buf_idx];
    maf->running_sum += classification;
    maf->maf_buffer[maf->buf_idx] = classification;
    if (++maf->buf_idx >= (EI_CLASSIFIER_SLICES_PER_MODEL_WINDOW >> 1)) maf->buf_idx = 0;
    return maf->running_sum / (float)(EI_CLASSIFIER_SLICES_PER_MODEL_WINDOW >> 1);
}

// ei_run_classifier.h:164-172
void run_classifier_init(void)
{
    kws_handle *h = kws_default_model();
    if (!h) return;
    std::lock_guard<std::mutex> sdk_lk(h->sdk_mu);
    h->slice_offset = 0;
    h->feature_buffer_full = false;
    h->cont_first_run = false;
    h->cont_features.assign(h->model.nn_input_frame_size, 0.0f);
    for (auto &m : h->maf) { m.buf_idx = 0; m.running_sum = 0; for (float &v : m.maf_buffer) v = 0.f; }
}

// ei_run_classifier.h:184-282 + ei_run_dsp.h:310-366 -- continuous (sliced) mode.  One call = one slice of audio:
// cepstra of the slice (speechpy::feature::mfcc, no CMVN) are appended to a rolling feature buffer; once it is full every
// call normalises a copy of the whole buffer (cmvnw), runs the network and a 2-tap moving average per class.
// The slice's MFCC and the window's cmvnw + network run on the GPU (kws_mfcc_kernel<WITH_CMVN=false>,
// kws_cmvn_nn_kernel); the rolling buffer and the filters are host state of the default model handle.
//
// Reference quirks that are kept: a function-static `first_run` (ei_run_dsp.h:313) that NOTHING resets makes every call
// but the process's first grow signal->total_length by one frame length IN THE CALLER'S STRUCT and take one more frame;
// the pre-emphasis constructor then asks get_data for the sample at total_length-1 (beyond the slice) and ignores the
// callback's return value (buffer pre-zeroed).
static bool g_cont_first_run = false;

EI_IMPULSE_ERROR run_classifier_continuous(signal_t *signal, ei_impulse_result_t *result, bool debug)
{
    kws_handle *h = kws_default_model();
    if (!h) return kws_last_error_code() != EI_IMPULSE_OK ? kws_last_error_code() : KWS_ERROR_NO_MODEL;   // why the default model is missing
    if (!signal || !signal->get_data || !result) return fail(KWS_ERROR_BAD_ARGUMENT, "null argument");
    EI_IMPULSE_ERROR le = check_result_layout(h);
    if (le) return le;
    std::lock_guard<std::mutex> sdk_lk(h->sdk_mu);                               // the reference is non-reentrant: callers are serialised
    const Model &m = h->model;
    const size_t F = m.nn_input_frame_size, C = m.labels.size();
    const int ncep = h->dsp.n_cepstral;         // columns of the feature matrix (mel filters for an MFE block)
    if (h->cont_features.size() != F) h->cont_features.assign(F, 0.0f);         // static_features_matrix (calloc'd)
    HIP_TRY(hipSetDevice(h->device));
    uint64_t dsp_start_ms = ei_read_timer_ms();

    // ---- extract_mfcc_per_slice_features ----------------------------------------------------------------------
    if (g_cont_first_run) signal->total_length += (size_t)(m.dsp.frame_length * (float)m.frequency);
    g_cont_first_run = true;
    const size_t n_claimed = signal->total_length;
    float eos = 0.0f;                                                           // _end_of_signal_buffer (calloc)
    gather_constructor_call(signal, n_claimed, gather_shift(h), &eos);          // before the size check below, as in the reference
    const int frame_len = h->dsp.frame_len, stride = h->dsp.frame_stride;
    const int nf = n_claimed >= (size_t)frame_len ? (int)floorf((float)(n_claimed - (size_t)frame_len) / (float)stride) : 0;
    const size_t feature_size = (size_t)(nf > 0 ? nf : 0) * (size_t)ncep;
    if (nf < 1 || feature_size > F || h->slice_offset + feature_size > F || (!h->dsp.generic && nf > kws_mfcc_max_frames(h->dsp.n_filters))) {
        ei_printf("ERR: MFCC failed (%d)\n", -1002);                           // EIDSP_MATRIX_SIZE_MISMATCH
        ei_printf("ERR: Failed to run DSP process (%d)\n", -1002);
        return fail(EI_IMPULSE_DSP_ERROR, "slice of %zu samples yields %d frames", n_claimed, nf);
    }
    const size_t needed = (size_t)(nf - 1) * stride + frame_len;                // last sample any frame reads
    const size_t n_x = std::max(n_claimed, needed) + 16;
    EI_IMPULSE_ERROR e = ensure_ws(h, n_x);
    if (e) return e;
    kws_handle::Ws &w = h->ws;
    memset(w.h_x, 0, n_x * sizeof(float));
    {
        // frame by frame, as the reference asks (gather_frames_like_reference)
        const int r = gather_frames_like_reference(signal, w.h_x, nf, frame_len, stride, gather_shift(h));
        if (r != 0) { ei_printf("ERR: Failed to run DSP process (%d)\n", r); return fail(EI_IMPULSE_DSP_ERROR, "signal->get_data returned %d", r); }
    }
    w.h_s[0] = eos;                                                              // staged through pinned memory
    if (hipMemcpyAsync(w.d_x, w.h_x, n_x * sizeof(float), hipMemcpyHostToDevice, w.st) != hipSuccess ||
        hipMemcpyAsync(w.d_w, w.h_s, sizeof(float), hipMemcpyHostToDevice, w.st) != hipSuccess) e = fail(KWS_ERROR_HIP, "h2d copy failed");
    if (!e) {
        KwsDspPlan P = h->dsp;                // same tables, this slice's framing
        P.n_samples = (int)n_claimed;
        P.n_frames = nf;
        e = spectral_device(h, P, w.d_x, 1, 1, w.d_f, w.d_w, w.st);
    }
    if (!e && (hipMemcpyAsync(w.h_f, w.d_f, feature_size * sizeof(float), hipMemcpyDeviceToHost, w.st) != hipSuccess ||
               hipStreamSynchronize(w.st) != hipSuccess))
        e = fail(KWS_ERROR_HIP, "d2h copy failed");
    if (e) return e;
    memcpy(h->cont_features.data() + h->slice_offset, w.h_f, feature_size * sizeof(float));
    if (ei_run_impulse_check_canceled() == EI_IMPULSE_CANCELED) return EI_IMPULSE_CANCELED;

    // ---- rolling buffer bookkeeping (ei_run_classifier.h:229-239) ------------------------------------------------
    if (!h->feature_buffer_full) {
        h->slice_offset += feature_size;
        if (h->slice_offset > (F - feature_size)) {
            h->feature_buffer_full = true;
            h->slice_offset -= feature_size;
        }
    }
    ei_impulse_result_timing_t *timing = result_timing(h, result);
    timing->dsp = (int)(ei_read_timer_ms() - dsp_start_ms);
    if (debug) {
        ei_printf("\r\nFeatures (%d ms.): ", timing->dsp);
        for (size_t ix = 0; ix < F; ix++) { ei_printf_float(h->cont_features[ix]); ei_printf(" "); }
        ei_printf("\n");
        ei_printf("Running neural network...\n");
    }
    if (h->feature_buffer_full) {
        dsp_start_ms = ei_read_timer_ms();
        // calc_cepstral_mean_and_var_normalization on a COPY of the buffer, then run_inference
        std::vector<float> scores(C);
        uint64_t t1 = 0;
        memcpy(w.h_f, h->cont_features.data(), F * sizeof(float));
        if (hipMemcpyAsync(w.d_f, w.h_f, F * sizeof(float), hipMemcpyHostToDevice, w.st) != hipSuccess) e = fail(KWS_ERROR_HIP, "h2d copy failed");
        if (!e) {
            std::lock_guard<std::mutex> lk(h->mu);
            e = ensure_scratch(h, 1);
            t1 = ei_read_timer_ms();
            if (!e) e = cmvn_nn_device(h, w.d_f, 1, nullptr, nullptr, w.d_s, nullptr, nullptr, nullptr, w.st);
        }
        if (!e && (hipMemcpyAsync(w.h_s, w.d_s, C * sizeof(float), hipMemcpyDeviceToHost, w.st) != hipSuccess ||
                   hipStreamSynchronize(w.st) != hipSuccess)) e = fail(KWS_ERROR_HIP, "d2h copy failed");
        if (e) return e;
        memcpy(scores.data(), w.h_s, C * sizeof(float));
        timing->dsp += (int)(t1 - dsp_start_ms);
        fill_result(h, result, scores.data(), debug, (int)(ei_read_timer_ms() - t1));
        // run_inference's return value is only handed on at the end: a cancelled call still filters the scores and shifts the buffer
        // (ei_run_classifier.h:268-281)
        const EI_IMPULSE_ERROR polled = inference_polls();
        ei_impulse_result_classification_t *cls = (ei_impulse_result_classification_t *)result;
        for (size_t ix = 0; ix < C; ix++) cls[ix].value = run_moving_average_filter(&h->maf[ix], cls[ix].value);
        // shift the feature buffer for new data (ei_run_classifier.h:277-279)
        for (size_t i = 0; i < F - feature_size; i++) h->cont_features[i] = h->cont_features[i + feature_size];
        return polled;
    }
    return EI_IMPULSE_OK;
}

}  // extern "C"
#pragma GCC visibility pop
