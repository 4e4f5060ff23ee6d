// kws_internal.h -- shared by the host translation units of libkws_mi355x.so (not installed).
#pragma once
#include <hip/hip_runtime.h>

#include <float.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <mutex>
#include <string>
#include <map>
#include <vector>

#include "../../include/kws/kws.h"
#include "kws_plan.h"
#include "kws_fast.h"

// launchers in kws_mfcc.hip, kws_nn_int8.hip, kws_nn_f32.hip, kws_misc.hip
int kws_launch_spectral(const KwsDspPlan &P, const void *pcm, int pcm_is_float, int n_clips, float *mfcc_out, const float *wrap,
                        int out_stride, int grid_cap, hipStream_t stream, const int *sel = nullptr);
int kws_launch_maf(float *scores, float *running_sum, float *maf_buf, int n, int buf_idx, int taps, hipStream_t stream);
int kws_launch_unring(const float *src, float *dst, int n_streams, int rows, int cols, int ring_rows, int head, hipStream_t stream);
int kws_launch_mfe(const KwsDspPlan &P, const void *pcm, int pcm_is_float, int n_clips, float *mel_out, float *energy_out, const float *wrap,
                   int out_stride, int grid_cap, hipStream_t stream);
int kws_launch_mfe_norm(float *feat, int n_clips, int rows, int cols, int win, const int *pad_map, int prow, int grid_cap, hipStream_t stream);
int kws_launch_mfcc_fused(const KwsDspPlan &P, const void *pcm, int pcm_is_float, int n_clips, float *features, int8_t *q_out,
                          float in_scale, int in_zp, int grid_cap, hipStream_t stream, const int *sel = nullptr);
int kws_launch_mfcc_fused_prof(const KwsDspPlan &P, const void *pcm, int n_clips, float *features, int8_t *q_out, float in_scale,
                               int in_zp, int grid_cap, long long *prof_out, hipStream_t stream);
int kws_launch_cmvn_nn(const KwsDspPlan &P, const KwsNnPlan &N, const float *mfcc, int n_clips, float *features, int8_t *q_out,
                       float *scores, int8_t *tap_pooled, int pooled_stride, int8_t *tap_fc, int8_t *tap_out_q, int grid_cap,
                       int *ran_nn, hipStream_t stream, const int *sel = nullptr);
int kws_launch_fast_from_cepstra(const KwsDspPlan &P, const KwsFastPlan &FP, const KwsFastPlan *d_plan, const float *cep, int n_clips, float *scores,
                                 float *features, int8_t *q_out, float in_scale, int in_zp, int *flag_count, int *flag_list, int n_cu, hipStream_t stream,
                                 const int *sel = nullptr, float *tap_logits = nullptr, int feat_in = 0);
int kws_launch_nn_f32(const KwsNnPlanF32 &N, const KwsNnPlanF32 *d_plan, const float *features, int n_clips, float *scores,
                      float *tap_logits, int n_cu, hipStream_t stream, const int *sel = nullptr);
size_t kws_nn_f32_smem_bytes(const KwsNnPlanF32 &N, int n_waves);
void kws_nn_f32_pick_blocking(KwsConvBlockF32 *k);
int kws_launch_nn(const KwsNnPlan &N, const int8_t *q_in, int n_clips, float *scores, int8_t *tap_pooled,
                  int pooled_stride, int8_t *tap_fc, int8_t *tap_out_q, int grid_cap, hipStream_t stream, const int *sel = nullptr);
int kws_launch_fast(const KwsDspPlan &P, const KwsFastPlan &FP, const KwsFastPlan *d_plan, const int16_t *pcm, int n_clips, float *scores, float *features,
                    int8_t *q_out, float in_scale, int in_zp, int *flag_count, int *flag_list, int n_cu, hipStream_t stream, const KwsNnPlan *d_nn = nullptr,
                    float *tap_logits = nullptr);
size_t kws_fast_qnet_bytes(int qcp);     // LDS bytes of the fused int8 network's shared tables (kws_fast.hip)
int kws_launch_fast_prof(const KwsDspPlan &P, const KwsFastPlan &FP, const KwsFastPlan *d_plan, const int16_t *pcm, int n_clips, float *scores,
                         int *flag_count, int *flag_list, int n_cu, long long *prof_out, hipStream_t stream);
int kws_launch_mix_audio(const float *words, const int *word_len, size_t word_stride, const float *noise, const int *start, float word_vol,
                         float bg_vol, int n, size_t n_clips, int16_t *out, hipStream_t stream);
int kws_launch_quantize(const float *f, int8_t *q, size_t n, float scale, int zp, hipStream_t stream);
int kws_launch_resample(const float *in, size_t n_in, float *out, size_t n_out, size_t n_valid, double ratio, const double *win, const double *delta, int nwin,
                        int precision, int exact, const double *treg, hipStream_t stream);
// kws_generic.hip: the exact MFCC block for configurations outside the tuned kernel (KwsDspPlan::generic)
size_t kws_generic_ws_bytes(const KwsDspPlan &P, int grid);
bool kws_generic_uses_lds(const KwsDspPlan &P);       // the LDS-resident cooperative kernel serves this configuration (else: the scratch-in-HBM kernel)
int kws_launch_spectral_generic(const KwsDspPlan &P, const void *pcm, int pcm_is_float, int n_clips, float *mfcc_out, const float *wrap,
                                int out_stride, float *ws, int grid, int lch /* frames per chunk of the LDS kernel: 4 or 8 */, hipStream_t stream);
int kws_launch_cmvn_generic(const KwsDspPlan &P, const float *mfcc, int n_clips, float *features, int8_t *q_out, float in_scale, int in_zp,
                            hipStream_t stream);
int kws_launch_synth(uint32_t seed, uint32_t first_clip, uint32_t n_clips, uint32_t clip_len, int16_t *out, hipStream_t stream);
size_t kws_nn_smem_bytes(const KwsNnPlan &N, int n_waves);
extern int kws_force_scalar_nn;
int kws_nn_uses_mfma(const KwsNnPlan &N);
int kws_mfcc_max_prow(void);
int kws_mfcc_max_win(int n_cepstral);
int kws_mfcc_max_frames_for(int n_filters, int n_cepstral);
int kws_mfcc_max_nz(void);
int kws_mfcc_cmvn_rows(void);
int kws_mfcc_max_frames(int n_filters);


// ---- errors: the last failure of the calling thread (message via kws_last_error()) ---------------------------------------
EI_IMPULSE_ERROR kws_fail(EI_IMPULSE_ERROR code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
EI_IMPULSE_ERROR kws_last_error_code(void);
#define fail kws_fail
#define HIP_TRY(expr)                                                                            \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) return fail(KWS_ERROR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// ---- model blob (layout: tools/eon_import.py) -----------------------------------------------------------------------------

enum { OP_RESHAPE = 0, OP_CONV_2D, OP_ADD, OP_MAX_POOL_2D, OP_FULLY_CONNECTED, OP_SOFTMAX, OP_DEPTHWISE_CONV_2D };
enum { TYPE_F32 = 1, TYPE_I32 = 2, TYPE_I8 = 9 };

struct Tensor {
    uint32_t type = 0;
    std::vector<int> dims;
    bool is_const = false;
    std::vector<float> scale;
    std::vector<int32_t> zero;
    int qdim = 0;
    uint32_t nbytes = 0;
    std::vector<uint8_t> data;
    int dim4(int i) const { int pad = 4 - (int)dims.size(); return i < pad ? 1 : dims[i - pad]; }
};
struct Node {
    uint32_t op = 0;
    std::vector<int> in, out;
    int p[8] = { 0 };
    float beta = 0;
};
// which extract_fn the model's ei_dsp_blocks[] entry names (MODEL/model-parameters/dsp_blocks.h:29-36): extract_mfcc_features, or
// extract_mfe_features of the newer SDK copy (nucleo-l432 .../classifier/ei_run_dsp.h:369-418)
enum { DSP_BLOCK_MFCC = 0, DSP_BLOCK_MFE = 1 };
struct DspCfg {
    int axes, num_cepstral, num_filters, fft_length, win_size, low_frequency, high_frequency, pre_shift;
    float frame_length, frame_stride, pre_cof;
    int block = DSP_BLOCK_MFCC;
    int quantize_fb = 0;           // EIDSP_QUANTIZE_FILTERBANK (SDK/dsp/config.hpp:75-77): mel weights snapped to numpy.hpp:52's table
};
struct Model {
    std::vector<Tensor> t;
    std::vector<Node> n;
    std::vector<std::string> labels;
    uint32_t t_in = 0, t_out = 0, raw_sample_count = 0, frequency = 0, nn_input_frame_size = 0;
    DspCfg dsp;
};

// kws_model.cpp: blob parser and the host arithmetic the table builders need (same formulas as the reference's setup code)
bool parse_model(const void *blob, size_t nbytes, Model &m);
float h_fast_log(float a);
float h_freq_to_mel(float f);
float h_mel_to_freq(float mel);
void h_linspace(float start, float stop, uint32_t number, float *out);
std::vector<float> h_filterbank(int num_filter, int coefficients, uint32_t fs, uint32_t low, uint32_t high, bool quantize = false);
float h_quantize_zero_one(float value);
void h_twiddles(int nfft, std::vector<float2> &tw);
void h_super_twiddles(int ncfft, std::vector<float2> &st);
void h_pad_map(int rows, int pad, std::vector<int> &map);
int32_t h_srdhm(int32_t a, int32_t b);
int32_t h_rdivpot(int32_t x, int e);
void h_quantize_multiplier(double m, int32_t *q, int *shift);
int32_t h_wadd(int32_t a, int32_t b);
int32_t h_wsub(int32_t a, int32_t b);
int32_t h_sat_shl(int32_t x, int e);
int32_t h_exp_interval(int32_t a);
int32_t h_exp_neg_q5_26(int32_t a);
void h_act_range(int activation, float scale, int32_t zp, int32_t *amin, int32_t *amax);
int h_out_size(int padding, int image, int filter, int stride, int dil);
int h_pad_amount(int stride, int dil, int in_size, int filter, int out);

// ------------------------------------------------------------------------------------------------------------
//  handle
// ------------------------------------------------------------------------------------------------------------
struct kws_handle {
    Model model;
    int device = 0;
    int n_cu = 256;
    KwsDspPlan dsp{};
    KwsNnPlan nn{};
    bool is_float = false;        // float32 model: nnf is the plan, nn only carries a neutral input quantisation
    KwsNnPlanF32 nnf{};
    const KwsNnPlanF32 *d_nnf = nullptr;   // the same plan in device memory (the float kernel reads it from there)
    int pooled_tap_bytes = 0;
    std::vector<void *> dev_allocs;
    // kws_spectral_lds_kernel's frames per chunk, measured per handle on its own first large calls (kws_api.cpp generic_chunk_begin / _end)
    struct GenericTune { int choice = 0, phase = 0, armed = 0; size_t clips = 0; hipEvent_t ev[2] = { nullptr, nullptr }; double ms_per_clip[2] = { 0.0, 0.0 };
                         hipStream_t owner = nullptr; bool owned = false; int warm = 0; } gen_tune;   // owner: the stream whose call armed the running sample; warm: bit per chunk length already launched once
    std::map<int, const int *> pad_maps_by_rows;     // kws_plan_for_length: cmvnw pad maps for other row counts than the model's (device, in dev_allocs)
    // scratch for the combined entry points (grown on demand)
    float *s_mfcc = nullptr;      // cepstra before CMVN, [B][n_features]
    int8_t *s_q = nullptr;
    size_t s_cap = 0;
    // the scratch (and the fast mode's clip list) is shared by every call on this handle: a call on another stream than the
    // previous one first waits for it (see ScratchUse)
    // general MFCC kernels (KwsDspPlan::generic): per-workgroup transform scratch, cepstra and feature buffers, grown on demand
    // One set PER STREAM: the transform scratch is indexed by workgroup, so two launches that run concurrently (the two streams of
    // kws_run_classifier_batch, or callers of the stage API on streams of their own) must not share it; launches on one stream are
    // ordered.  generic_for() hands out the set of a stream.
    struct GenericBuf {
        hipStream_t s = nullptr;
        bool used = false;
        float *ws = nullptr, *mfcc = nullptr, *feat = nullptr;
        size_t ws_bytes = 0, cap = 0;
    };
    static const int kGenericSets = 8;
    GenericBuf g_sets[kGenericSets];
    int g_next = 0;
    std::mutex g_mu;
    hipEvent_t scratch_ev = nullptr;
    hipStream_t scratch_stream = nullptr;
    bool scratch_used = false;
    // host-buffer entry point (kws_run_classifier_batch): two streams, each with its own chunk-sized device buffers, so that
    // the PCIe transfer of one chunk overlaps the kernels of the other; allocated once, grown on demand
    struct HostPipe {
        hipStream_t st[2] = { nullptr, nullptr };
        int16_t *pcm[2] = { nullptr, nullptr };
        float *s[2] = { nullptr, nullptr }, *f[2] = { nullptr, nullptr };
        int8_t *q[2] = { nullptr, nullptr };
        size_t cap = 0;
    } pipe;
    std::mutex pipe_mu;
    std::mutex mu;
    // single-clip workspace of the SDK entry points: allocated once, pinned host staging, own stream
    struct Ws {
        float *h_x = nullptr, *d_x = nullptr;     // samples / slice (host pinned, device)
        size_t cap_x = 0;
        float *d_f = nullptr, *d_s = nullptr, *d_w = nullptr;   // features (or cepstra), scores, wrap sample
        int8_t *d_q = nullptr;
        float *h_s = nullptr, *h_f = nullptr;     // pinned: scores, features
        hipStream_t st = nullptr;
        hipEvent_t ev[3] = { nullptr, nullptr, nullptr };   // before the DSP block / between the stages / after the network (timing fields)
    } ws;
    std::mutex sdk_mu;            // the SDK entry points are serialised (the reference is non-reentrant)
    // continuous-mode state (ei_run_classifier.h:115-121, 187)
    std::vector<float> cont_features;
    size_t slice_offset = 0;
    bool feature_buffer_full = false;
    bool cont_first_run = false;
    std::vector<ei_impulse_maf> maf;
    // KWS_MODE_FAST (kws_fast.h): host copies of a float graph's constants (the fused plan re-lays them out), the two plans,
    // and the list of clips the fast kernel hands back to the exact kernels
    struct HostF32 { std::vector<float> w[KWS_MAX_BLOCKS], bias[KWS_MAX_BLOCKS], addc[KWS_MAX_BLOCKS], fc_w, fc_b; } hostf;
    // kws_gain.cpp: logit gain per cepstral column of a float32 graph (calibrated at kws_create), the fused network's re-ordering noise
    struct Gain { std::vector<float> col; float sigma_net = 0.0f, total = 0.0f; int calibrated = 0, n_inputs = 0; } gain;
    KwsFastPlan fast_plain{}, fast_fused{}, fast_q{};   // features / int8 tensor to HBM; float32 graph fused; int8 two-block graph fused
    const KwsFastPlan *d_fast_plain = nullptr, *d_fast_fused = nullptr, *d_fast_q = nullptr;     // the same plans in device memory
    // the fused plan of the launches that start from cepstra or features (continuous mode, the second tier of a batch call, the features-in route): always
    // laid out for two waves per SIMD -- those forms are 15 - 47 % slower in the three-wave build (profiles/r06_occupancy.md); equal to fast_fused when that is
    KwsFastPlan fast_fused_cep{};
    const KwsFastPlan *d_fast_fused_cep = nullptr;
    const KwsNnPlan *d_nn = nullptr;                    // the int8 plan in device memory (the fused form reads it from there)
    bool fast_plain_ok = false, fast_fused_ok = false, fast_q_ok = false;
    std::string fast_why;
    // host copy of the guard's per-column coefficients, [tier - 1][kind][column]: kind 0 absolute, 1 per log-mel level, 2 per |window mean|,
    // 3 per |window mean| when column 0's window means were replayed in the reference's order (kws_fast_guard)
    std::vector<float> fast_guard_coef[2][4];
    int fast_dev_overrides = 0;                            // bit set: a KWS_DEV_FAST_* switch that changes results was read at kws_create
    int fast_entry_tier = 1;                               // 1: batch calls start in the fast kernel; 2: from exact cepstra; 3: exact kernels (build_guard)
    std::vector<float> fast_sil_row;                       // the reference's cepstral row of a digitally silent frame (record_silent_row); empty: not recorded
    std::vector<float> fast_gain_used;                     // the per-column gain those coefficients were built with (float32 graph: gain.col)
    int mode = KWS_MODE_EXACT;
    int *d_flags = nullptr;       // [0] = count, [1 + i] = clip index: the clips the fast kernel handed back (first tier)
    int *d_flags2 = nullptr;      // the same for the second tier: the clips that go to the exact cmvnw + network
    int *d_flags3 = nullptr;      // sink: when a call wants the feature matrix AND fused scores, the feature-emitting launch's list (P = 1/4) decides
                                  // for both -- the features are then the ones kws_extract_mfcc_batch_device returns -- and the fused launch's goes here
    float *tap_logits = nullptr;  // kws_set_logits_tap: FULLY_CONNECTED outputs of the batch calls' clips (float32 graphs), device [B][labels]
    float *s_cep = nullptr;       // [B][n_features] exact cepstra of the first tier's clips (indexed by clip)
    size_t flags_cap = 0, cep_cap = 0;

    template <typename T> EI_IMPULSE_ERROR upload(const std::vector<T> &v, const T **out)
    {
        void *d = nullptr;
        size_t nb = std::max<size_t>(v.size() * sizeof(T), 16);
        HIP_TRY(hipMalloc(&d, nb));
        dev_allocs.push_back(d);
        if (!v.empty()) HIP_TRY(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
        *out = (const T *)d;
        return EI_IMPULSE_OK;
    }
};

// Brackets a call that uses the handle's scratch on stream s (h->mu held): orders it behind the previous such call when that one
// was enqueued on a different stream, and records where this one ends.
struct ScratchUse {
    kws_handle *h;
    hipStream_t s;
    ScratchUse(kws_handle *h_, hipStream_t s_) : h(h_), s(s_)
    {
        if (!h->scratch_ev) (void)hipEventCreateWithFlags(&h->scratch_ev, hipEventDisableTiming);
        if (h->scratch_used && h->scratch_stream != s && h->scratch_ev) (void)hipStreamWaitEvent(s, h->scratch_ev, 0);
    }
    ~ScratchUse()
    {
        if (h->scratch_ev && hipEventRecord(h->scratch_ev, s) == hipSuccess) { h->scratch_stream = s; h->scratch_used = true; }
    }
};

// kws_plan.cpp: execution plans (tables computed once per model, uploaded to HBM)
EI_IMPULSE_ERROR build_dsp_plan(kws_handle *h);
EI_IMPULSE_ERROR build_nn_plan(kws_handle *h);
EI_IMPULSE_ERROR build_fast_plans(kws_handle *h);     // kws_fast_plan.cpp; never fatal: sets fast_*_ok
#define KWS_GAIN_INPUTS 48
#define KWS_GAIN_HEADROOM 1.25f        // the calibrated per-column gain is the largest value over the calibration matrices x this (kws_gain.cpp)
void kws_calibrate_gain(kws_handle *h);               // kws_gain.cpp
// Does the fused fast kernel run this block's contraction on split 22-bit operands (KwsFastBlock::hconv)?  ONE predicate for the plan builder
// (kws_fast_plan.cpp) and for the arithmetic kws_gain.cpp models when it measures sigma_net: CONV_2D blocks whose image the
// in-place split converts in one pass of at most 16 channel pairs per lane and whose outputs fit 4 x 2 tiles.
static inline bool kws_fast_block_splits(const KwsConvBlockF32 &s)
{
    const int in_cp = (s.in_c + 7) / 8 * 8;
    return !s.depthwise && s.in_w * (in_cp / 2) <= 16 * KWS_FAST_WAVE && in_cp <= 64 && (s.out_w + 15) / 16 <= 4 && (s.out_c + 15) / 16 <= 2;
}

// kws_api.cpp: stage launchers shared with the stream / SDK entry points (kws_sdk.cpp); internal, not exported
#define KWS_INTERNAL __attribute__((visibility("hidden")))
extern "C" {
KWS_INTERNAL EI_IMPULSE_ERROR ensure_scratch(kws_handle *h, size_t B);
KWS_INTERNAL void kws_sdk_forget_default(kws_handle *h);      // kws_sdk.cpp: kws_destroy() of the installed default model
KWS_INTERNAL int grid_cap_mfcc(const kws_handle *h);
KWS_INTERNAL EI_IMPULSE_ERROR generic_for(kws_handle *h, hipStream_t s, size_t B, kws_handle::GenericBuf **out);
KWS_INTERNAL EI_IMPULSE_ERROR cmvn_nn_fast_device(kws_handle *h, const float *mfcc, size_t B, float *scores, hipStream_t s, int ring_rows, int ring_head,
                                                  float *features = nullptr, int8_t *q_out = nullptr);
KWS_INTERNAL int grid_cap_nn(const kws_handle *h);
KWS_INTERNAL EI_IMPULSE_ERROR spectral_device(kws_handle *h, const KwsDspPlan &P, const void *pcm, int is_float, size_t B, float *mfcc,
                                 const float *wrap, hipStream_t s, int out_stride = 0);
KWS_INTERNAL EI_IMPULSE_ERROR mfcc_fused_device(kws_handle *h, const void *pcm, int is_float, size_t B, float *features, int8_t *q, hipStream_t s);
// the same with another plan than the handle's own (kws_plan_for_length: a window of another length, SDK one-shot path)
KWS_INTERNAL EI_IMPULSE_ERROR mfcc_fused_device_plan(kws_handle *h, const KwsDspPlan &P, const void *pcm, int is_float, size_t B, float *features, int8_t *q, hipStream_t s);
// frames the reference's framing yields for a window of n samples (processing.hpp:194-284; 0 when none fits), and the handle's DSP plan
// re-targeted at such a window: same tables, n_samples / n_frames / cmvnw's pad map for that row count (maps are cached on the handle)
KWS_INTERNAL int kws_frames_for_length(const kws_handle *h, size_t n);
KWS_INTERNAL EI_IMPULSE_ERROR kws_plan_for_length(kws_handle *h, size_t n, KwsDspPlan *out);
KWS_INTERNAL EI_IMPULSE_ERROR nn_f32_device(kws_handle *h, const float *features, size_t B, float *scores, float *tap_logits, hipStream_t s);
KWS_INTERNAL EI_IMPULSE_ERROR cmvn_nn_device(kws_handle *h, const float *mfcc, size_t B, float *features, int8_t *q, float *scores,
                                int8_t *tap_pooled, int8_t *tap_fc, int8_t *tap_out, hipStream_t s, int ring_rows = 0, int ring_head = 0);
}
