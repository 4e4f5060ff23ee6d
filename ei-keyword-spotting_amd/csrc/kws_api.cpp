// kws_api.cpp -- host side of libkws_mi355x.so: model ingestion, table building, the C-ABI of include/kws/kws.h
// and the SDK-compatible entry points of include/kws/ei_compat.h.
//
// Everything that the reference recomputes per clip on the CPU but that does not depend on the audio is computed
// HERE once per model, with the reference's own formulas and precisions (each builder cites its source), and
// uploaded to HBM; the per-clip arithmetic runs in kws_mfcc.hip / kws_nn_int8.hip / kws_nn_f32.hip.  There is no CPU fallback: if no HIP device or
// code object is available every entry point fails with KWS_ERROR_HIP.
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
#include <vector>

#include "../../include/kws/kws.h"
#include "kws_plan.h"

// launchers in kws_mfcc.hip, kws_nn_int8.hip, kws_nn_f32.hip, kws_misc.hip
int kws_launch_spectral(const KwsDspPlan &P, const void *pcm, int pcm_is_float, int n_clips, float *mfcc_out, const float *wrap,
                        int out_stride, int grid_cap, hipStream_t stream);
int kws_launch_maf(float *scores, float *running_sum, float *maf_buf, int n, int buf_idx, int taps, hipStream_t stream);
int kws_launch_shift(const float *src, float *dst, int n_streams, int F, int shift, hipStream_t stream);
int kws_launch_mfe(const KwsDspPlan &P, const void *pcm, int n_clips, float *mel_out, float *energy_out, int grid_cap, hipStream_t stream);
int kws_launch_mfcc_fused(const KwsDspPlan &P, const void *pcm, int pcm_is_float, int n_clips, float *features, int8_t *q_out,
                          float in_scale, int in_zp, int grid_cap, hipStream_t stream);
int kws_launch_mfcc_fused_prof(const KwsDspPlan &P, const void *pcm, int n_clips, float *features, int8_t *q_out, float in_scale,
                               int in_zp, int grid_cap, long long *prof_out, hipStream_t stream);
int kws_launch_cmvn_nn(const KwsDspPlan &P, const KwsNnPlan &N, const float *mfcc, int n_clips, float *features, int8_t *q_out,
                       float *scores, int8_t *tap_pooled, int pooled_stride, int8_t *tap_fc, int8_t *tap_out_q, int grid_cap,
                       int *ran_nn, hipStream_t stream);
int kws_launch_nn_f32(const KwsNnPlanF32 &N, const KwsNnPlanF32 *d_plan, const float *features, int n_clips, float *scores,
                      float *tap_logits, int n_cu, hipStream_t stream);
size_t kws_nn_f32_smem_bytes(const KwsNnPlanF32 &N, int n_waves);
void kws_nn_f32_pick_blocking(KwsConvBlockF32 *k);
int kws_launch_nn(const KwsNnPlan &N, const int8_t *q_in, int n_clips, float *scores, int8_t *tap_pooled,
                  int pooled_stride, int8_t *tap_fc, int8_t *tap_out_q, int grid_cap, hipStream_t stream);
int kws_launch_quantize(const float *f, int8_t *q, size_t n, float scale, int zp, hipStream_t stream);
int kws_launch_synth(uint32_t seed, uint32_t first_clip, uint32_t n_clips, uint32_t clip_len, int16_t *out, hipStream_t stream);
size_t kws_nn_smem_bytes(const KwsNnPlan &N);
extern int kws_force_scalar_nn;
int kws_nn_uses_mfma(const KwsNnPlan &N);
int kws_mfcc_max_prow(void);
int kws_mfcc_max_win(int n_cepstral);
int kws_mfcc_max_frames_for(int n_filters, int n_cepstral);
int kws_mfcc_max_nz(void);
int kws_mfcc_cmvn_rows(void);
int kws_mfcc_max_frames(int n_filters);

// ------------------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static thread_local EI_IMPULSE_ERROR g_err_code = EI_IMPULSE_OK;
static EI_IMPULSE_ERROR fail(EI_IMPULSE_ERROR code, const char *fmt, ...)
{
    g_err_code = code;
    char buf[512];
    va_list a;
    va_start(a, fmt);
    vsnprintf(buf, sizeof(buf), fmt, a);
    va_end(a);
    g_err = buf;
    return code;
}
#define HIP_TRY(expr)                                                                            \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) return fail(KWS_ERROR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

#pragma GCC visibility push(default)     // the library is built with -fvisibility=hidden: only the C ABI is exported
extern "C" const char *kws_last_error(void) { return g_err.c_str(); }

// ------------------------------------------------------------------------------------------------------------
//  weak platform hooks (porting/ei_classifier_porting.h:45-76)
// ------------------------------------------------------------------------------------------------------------
extern "C" {
__attribute__((weak)) EI_IMPULSE_ERROR ei_run_impulse_check_canceled(void) { return EI_IMPULSE_OK; }
__attribute__((weak)) EI_IMPULSE_ERROR ei_sleep(int32_t ms)
{
    struct timespec ts = { ms / 1000, (long)(ms % 1000) * 1000000L };
    nanosleep(&ts, NULL);
    return EI_IMPULSE_OK;
}
__attribute__((weak)) uint64_t ei_read_timer_us(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (uint64_t)ts.tv_sec * 1000000ull + (uint64_t)ts.tv_nsec / 1000;
}
__attribute__((weak)) uint64_t ei_read_timer_ms(void) { return ei_read_timer_us() / 1000; }
__attribute__((weak)) void ei_printf(const char *format, ...)
{
    va_list a;
    va_start(a, format);
    vprintf(format, a);
    va_end(a);
}
__attribute__((weak)) void ei_printf_float(float f) { ei_printf("%f", f); }
}

// ------------------------------------------------------------------------------------------------------------
//  model blob (layout: tools/eon_import.py)
// ------------------------------------------------------------------------------------------------------------
namespace {

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
struct DspCfg {
    int axes, num_cepstral, num_filters, fft_length, win_size, low_frequency, high_frequency, pre_shift;
    float frame_length, frame_stride, pre_cof;
};
struct Model {
    std::vector<Tensor> t;
    std::vector<Node> n;
    std::vector<std::string> labels;
    uint32_t t_in = 0, t_out = 0, raw_sample_count = 0, frequency = 0, nn_input_frame_size = 0;
    DspCfg dsp;
};

struct Reader {
    const uint8_t *p, *end;
    bool bad = false;
    uint32_t u32() { if (p + 4 > end) { bad = true; return 0; } uint32_t v; memcpy(&v, p, 4); p += 4; return v; }
    int32_t i32() { return (int32_t)u32(); }
    float f32() { uint32_t u = u32(); float f; memcpy(&f, &u, 4); return f; }
    const uint8_t *bytes(size_t n) { size_t pn = (n + 3) & ~(size_t)3; if (p + pn > end) { bad = true; return nullptr; } const uint8_t *q = p; p += pn; return q; }
};

bool parse_model(const void *blob, size_t nbytes, Model &m)
{
    if (nbytes < 8 || memcmp(blob, "KWSM", 4) != 0) return false;
    Reader r{ (const uint8_t *)blob + 4, (const uint8_t *)blob + nbytes };
    if (r.u32() != 1) return false;
    uint32_t nt = r.u32(), nn = r.u32(), nl = r.u32();
    m.t_in = r.u32(); m.t_out = r.u32();
    m.raw_sample_count = r.u32(); m.frequency = r.u32(); m.nn_input_frame_size = r.u32();
    DspCfg &d = m.dsp;
    d.axes = r.i32(); d.num_cepstral = r.i32(); d.num_filters = r.i32(); d.fft_length = r.i32(); d.win_size = r.i32();
    d.low_frequency = r.i32(); d.high_frequency = r.i32(); d.pre_shift = r.i32();
    d.frame_length = r.f32(); d.frame_stride = r.f32(); d.pre_cof = r.f32();
    if (r.bad || nt > 4096 || nn > 4096 || nl > 1024) return false;
    for (uint32_t i = 0; i < nl; i++) {
        uint32_t len = r.u32();
        const uint8_t *b = r.bytes(len);
        if (!b) return false;
        m.labels.emplace_back((const char *)b, len);
    }
    m.t.resize(nt);
    for (auto &t : m.t) {
        t.type = r.u32();
        uint32_t nd = r.u32();
        if (r.bad || nd > 8) return false;
        for (uint32_t k = 0; k < nd; k++) t.dims.push_back(r.i32());
        t.is_const = r.u32() != 0;
        uint32_t nq = r.u32();
        if (r.bad || nq > 65536) return false;
        for (uint32_t k = 0; k < nq; k++) t.scale.push_back(r.f32());
        for (uint32_t k = 0; k < nq; k++) t.zero.push_back(r.i32());
        t.qdim = r.i32();
        t.nbytes = r.u32();
        // self-consistency: known element type, positive dims, nbytes == element count x element size, int8 tensors
        // carry their quantisation, a per-channel scale list matches the quantised dimension
        if (r.bad || (t.type != TYPE_F32 && t.type != TYPE_I32 && t.type != TYPE_I8)) return false;
        uint64_t count = 1;
        for (int d : t.dims) { if (d <= 0 || d > (1 << 24)) return false; count *= (uint64_t)d; if (count > (1u << 28)) return false; }
        if ((uint64_t)t.nbytes != count * (t.type == TYPE_I8 ? 1u : 4u)) return false;
        if (t.type == TYPE_I8 && nq == 0) return false;
        if (nq > 1 && (t.qdim < 0 || t.qdim >= (int)t.dims.size() || (uint32_t)t.dims[t.qdim] != nq)) return false;
        if (t.is_const) {
            const uint8_t *b = r.bytes(t.nbytes);
            if (!b) return false;
            t.data.assign(b, b + t.nbytes);
        }
    }
    m.n.resize(nn);
    for (auto &n : m.n) {
        n.op = r.u32();
        uint32_t ni = r.u32();
        if (r.bad || ni > 8) return false;
        for (uint32_t k = 0; k < ni; k++) n.in.push_back(r.i32());
        uint32_t no = r.u32();
        if (r.bad || no > 8) return false;
        for (uint32_t k = 0; k < no; k++) n.out.push_back(r.i32());
        for (int k = 0; k < 8; k++) n.p[k] = r.i32();
        n.beta = r.f32();
    }
    if (r.bad || m.t_in >= nt || m.t_out >= nt) return false;
    for (auto &n : m.n) {
        for (int v : n.in) if (v >= (int)nt) return false;
        for (int v : n.out) if (v < 0 || v >= (int)nt) return false;
    }
    return true;
}

// ------------------------------------------------------------------------------------------------------------
//  host arithmetic the table builders need (same formulas as the reference's setup code)
// ------------------------------------------------------------------------------------------------------------
float h_fast_log(float a)                                   // numpy::log, SDK/dsp/numpy.hpp:1350-1371
{
    uint32_t gu; memcpy(&gu, &a, 4);
    int32_t g = (int32_t)gu;
    int32_t e = (int32_t)(((uint32_t)g - 0x3f2aaaabu) & 0xff800000u);
    g = (int32_t)((uint32_t)g - (uint32_t)e);
    float m; memcpy(&m, &g, 4);
    float i = (float)e * 1.19209290e-7f;
    float f = m - 1.0f, s = f * f;
    float r = fmaf(0.230836749f, f, -0.279208571f);
    float t = fmaf(0.331826031f, f, -0.498910338f);
    r = fmaf(r, s, t);
    r = fmaf(r, s, f);
    return fmaf(i, 0.693147182f, r);
}
float h_freq_to_mel(float f) { return (float)(1127.0 * (double)h_fast_log(1 + f / 700.0f)); }   // functions.hpp:42-44
float h_mel_to_freq(float mel) { return 700.0f * (expf(mel / 1127.0f) - 1.0f); }                // functions.hpp:52-54

void h_linspace(float start, float stop, uint32_t number, float *out)                            // numpy.hpp:1257-1280
{
    if (number == 1) { out[0] = start; return; }
    float step = (stop - start) / (number - 1);
    for (uint32_t ix = 0; ix < number - 1; ix++) out[ix] = start + ix * step;
    out[number - 1] = stop;
}

// feature::filterbanks (feature.hpp:54-171) + functions::triangle (functions.hpp:90-104), dense [coeff][M]
std::vector<float> h_filterbank(int num_filter, int coefficients, uint32_t fs, uint32_t low, uint32_t high)
{
    std::vector<float> fb((size_t)coefficients * num_filter, 0.0f);
    const int np = num_filter + 2;
    std::vector<float> mels(np), hertz(np);
    std::vector<int> idx(np);
    h_linspace(h_freq_to_mel((float)low), h_freq_to_mel((float)high), (uint32_t)np, mels.data());
    for (int ix = 0; ix < np; ix++) {
        hertz[ix] = h_mel_to_freq(mels[ix]);
        if (hertz[ix] < low) hertz[ix] = (float)low;
        if (hertz[ix] > high) hertz[ix] = (float)high;
        if (ix == np - 1) hertz[ix] = (float)((double)hertz[ix] - 0.001);
    }
    for (int ix = 0; ix < np; ix++) idx[ix] = (int)floorf((float)(coefficients + 1) * hertz[ix] / (float)fs);
    for (int i = 0; i < num_filter; i++) {
        const int left = idx[i], middle = idx[i + 1], right = idx[i + 2];
        const int zn = right - left + 1;
        if (zn < 1) continue;
        std::vector<float> z(zn), o(zn, 0.0f);
        h_linspace((float)left, (float)right, (uint32_t)zn, z.data());
        for (int k = 0; k < zn; k++) {
            const float x = z[k];
            if (x > left && x <= middle) o[k] = (x - left) / (middle - left);
            if (x < right && middle <= x) o[k] = (right - x) / (right - middle);
        }
        for (int zx = 0; zx < zn; zx++) {
            const int bin = left + zx;
            if (bin >= 0 && bin < coefficients) fb[(size_t)bin * num_filter + i] = o[zx];
        }
    }
    return fb;
}

void h_twiddles(int nfft, std::vector<float2> &tw)                                // kiss_fft.cpp:351-357
{
    tw.resize(nfft);
    for (int i = 0; i < nfft; ++i) {
        const double pi = 3.141592653589793238462643383279502884197169399375105820974944;
        double phase = -2 * pi * i / nfft;
        tw[i] = make_float2((float)cos(phase), (float)sin(phase));
    }
}
void h_super_twiddles(int ncfft, std::vector<float2> &st)                         // kiss_fftr.cpp:52-58
{
    st.resize(ncfft / 2);
    for (int i = 0; i < ncfft / 2; ++i) {
        double phase = -3.14159265358979323846264338327 * ((double)(i + 1) / ncfft + .5);
        st[i] = make_float2((float)cos(phase), (float)sin(phase));
    }
}
void h_pad_map(int rows, int pad, std::vector<int> &map)                          // numpy.hpp:479-541
{
    map.assign(rows + 2 * pad, 0);
    int idx = 0; bool up = true;
    for (int ix = pad - 1; ix >= 0; ix--) {
        map[ix] = idx;
        if (idx == 0 && !up) up = true;
        else if (idx == rows - 1 && up) up = false;
        else if (up) idx++;
        else idx--;
    }
    for (int r = 0; r < rows; r++) map[pad + r] = r;
    idx = rows - 1; up = false;
    for (int ix = 0; ix < pad; ix++) {
        map[ix + pad + rows] = idx;
        if (idx == 0 && !up) up = true;
        else if (idx == rows - 1 && up) up = false;
        else if (up) idx++;
        else idx--;
    }
}

// ---- fixed point (gemmlowp fixedpoint.h:329-368, TFL quantization_util.cc:53-91) ----------------------------
int32_t h_srdhm(int32_t a, int32_t b)
{
    bool overflow = (a == b) && (a == INT32_MIN);
    int64_t ab = (int64_t)a * (int64_t)b;
    int32_t nudge = ab >= 0 ? (1 << 30) : (1 - (1 << 30));
    int32_t hi = (int32_t)((ab + nudge) / (1ll << 31));
    return overflow ? INT32_MAX : hi;
}
int32_t h_rdivpot(int32_t x, int e)
{
    const int32_t mask = (int32_t)((1ll << e) - 1);
    const int32_t rem = x & mask;
    const int32_t thr = (mask >> 1) + (x < 0 ? 1 : 0);
    return (x >> e) + (rem > thr ? 1 : 0);
}
void h_quantize_multiplier(double m, int32_t *q, int *shift)
{
    if (m == 0.) { *q = 0; *shift = 0; return; }
    const double f = frexp(m, shift);
    int64_t qf = (int64_t)round(f * (double)(1ll << 31));
    if (qf == (1ll << 31)) { qf /= 2; ++*shift; }
    if (*shift < -31) { *shift = 0; qf = 0; }
    *q = (int32_t)qf;
}
int32_t h_wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
int32_t h_wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
int32_t h_sat_shl(int32_t x, int e)
{
    const int32_t thr = (int32_t)((1u << (31 - e)) - 1);
    if (x > thr) return INT32_MAX;
    if (x < -thr) return INT32_MIN;
    return (int32_t)((int64_t)x * (1 << e));
}
int32_t h_exp_interval(int32_t a)                                                 // fixedpoint.h:721-742
{
    const int32_t ct = 1895147668, third = 715827883;
    int32_t x = h_wadd(a, 1 << 28);
    int32_t x2 = h_srdhm(x, x), x3 = h_srdhm(x2, x), x4 = h_srdhm(x2, x2);
    int32_t x4_4 = h_rdivpot(x4, 2);
    int32_t t = h_rdivpot(h_wadd(h_srdhm(h_wadd(x4_4, x3), third), x2), 1);
    return h_wadd(ct, h_srdhm(ct, h_wadd(x, t)));
}
int32_t h_exp_neg_q5_26(int32_t a)                                                // fixedpoint.h:746-790, 5 integer bits
{
    const int32_t quarter = 1 << 24, mask = quarter - 1;
    int32_t amq = h_wsub(a & mask, quarter);
    int32_t result = h_exp_interval(h_sat_shl(amq, 5));
    int32_t rem = h_wsub(amq, a);
    static const int32_t mult[7] = { 1672461947, 1302514674, 790015084, 290630308, 39332535, 720401, 242 };
    for (int e = -2; e <= 4; e++)
        if (rem & (1 << (26 + e))) result = h_srdhm(result, mult[e + 2]);
    return a == 0 ? INT32_MAX : result;
}
void h_act_range(int activation, float scale, int32_t zp, int32_t *amin, int32_t *amax)   // kernel_util_lite.cc:174-226
{
    *amin = -128; *amax = 127;
    auto q = [&](float f) { return zp + (int32_t)roundf(f / scale); };
    if (activation == 1) { *amin = std::max(-128, q(0.0f)); }
    else if (activation == 3) { *amin = std::max(-128, q(0.0f)); *amax = std::min(127, q(6.0f)); }
    else if (activation == 2) { *amin = std::max(-128, q(-1.0f)); *amax = std::min(127, q(1.0f)); }
}
int h_out_size(int padding, int image, int filter, int stride, int dil)          // padding.h:44-55
{
    int eff = (filter - 1) * dil + 1;
    if (padding == 1) return (image + stride - 1) / stride;
    if (padding == 2) return (image + stride - eff) / stride;
    return 0;
}
int h_pad_amount(int stride, int dil, int in_size, int filter, int out)          // padding.h:32-41
{
    int eff = (filter - 1) * dil + 1;
    int total = (out - 1) * stride + eff - in_size;
    return (total > 0 ? total : 0) / 2;
}

}  // namespace

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
    // scratch for the combined entry points (grown on demand)
    float *s_mfcc = nullptr;      // cepstra before CMVN, [B][n_features]
    int8_t *s_q = nullptr;
    size_t s_cap = 0;
    std::mutex mu;
    // single-clip workspace of the SDK entry points: allocated once, pinned host staging, own stream
    struct Ws {
        float *h_x = nullptr, *d_x = nullptr;     // samples / slice (host pinned, device)
        size_t cap_x = 0;
        float *d_f = nullptr, *d_s = nullptr, *d_w = nullptr;   // features (or cepstra), scores, wrap sample
        int8_t *d_q = nullptr;
        float *h_s = nullptr, *h_f = nullptr;     // pinned: scores, features
        hipStream_t st = nullptr;
    } ws;
    std::mutex sdk_mu;            // the SDK entry points are serialised (the reference is non-reentrant)
    // continuous-mode state (ei_run_classifier.h:115-121, 187)
    std::vector<float> cont_features;
    size_t slice_offset = 0;
    bool feature_buffer_full = false;
    bool cont_first_run = false;
    std::vector<ei_impulse_maf> maf;

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

static EI_IMPULSE_ERROR build_dsp_plan(kws_handle *h)
{
    const Model &m = h->model;
    const DspCfg &c = m.dsp;
    KwsDspPlan &P = h->dsp;
    const uint32_t fs = m.frequency;
    // framing: processing.hpp:194-284
    const int frame_len = (int)roundf((float)fs * c.frame_length);
    const float stride_f = roundf((float)fs * c.frame_stride);
    const int stride = (int)stride_f;
    const size_t diff = (size_t)m.raw_sample_count - (size_t)frame_len;
    const int nfr = (int)floorf((float)diff / stride_f);
    P.n_samples = (int)m.raw_sample_count;
    P.n_frames = nfr;
    P.frame_stride = stride;
    P.frame_len = frame_len;
    P.fft_len = c.fft_length;
    P.n_bins = c.fft_length / 2 + 1;
    P.n_filters = c.num_filters;
    P.n_cepstral = c.num_cepstral;
    P.win_size = c.win_size;
    P.pad = (int)(uint16_t)((c.win_size - 1) / 2);
    P.pre_shift = c.pre_shift;
    P.pre_cof = c.pre_cof;
    P.inv_fft = (float)(1.0 / (double)(float)c.fft_length);
    const int N = c.num_filters;
    P.dct_s0 = sqrtf(1.0f / (float)(4 * N));
    P.dct_s1 = sqrtf(1.0f / (float)(2 * N));

    // what the gfx950 kernels implement (kws_device.h: KWS_FFT, KWS_NF_MAX, KWS_MAXF ...)
    if (c.axes != 1) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "MFCC block with %d axes", c.axes);
    if (c.fft_length != 256 || (c.num_filters != 32 && c.num_filters != 40))
        return fail(KWS_ERROR_UNSUPPORTED_MODEL, "MFCC kernel is built for fft_length 256 and 32 or 40 filters (got %d / %d)",
                    c.fft_length, c.num_filters);
    if (frame_len < c.fft_length || c.pre_shift != 1 || (c.win_size & 1) == 0 || nfr < 1 || nfr > kws_mfcc_max_frames(c.num_filters) ||
        c.num_cepstral < 1 || c.num_cepstral > c.num_filters || (stride * 2) % 16 != 0 || (P.n_samples * 2) % 16 != 0 ||
        (nfr - 1) * stride + c.fft_length > P.n_samples || nfr + 2 * P.pad > kws_mfcc_max_prow() || c.win_size > kws_mfcc_max_win(c.num_cepstral) || nfr > kws_mfcc_max_frames_for(c.num_filters, c.num_cepstral) || c.win_size < ((c.num_filters == 40 && c.num_cepstral > 16) ? 17 : 13) ||
        nfr > 4 * kws_mfcc_cmvn_rows() ||
        (size_t)nfr * c.num_cepstral != m.nn_input_frame_size)
        return fail(KWS_ERROR_UNSUPPORTED_MODEL, "MFCC framing outside the kernel's limits (frames %d, frame_len %d, stride %d, "
                    "cepstra %d, win %d, shift %d)", nfr, frame_len, stride, c.num_cepstral, c.win_size, c.pre_shift);

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
    const uint32_t high = c.high_frequency == 0 ? fs / 2 : (uint32_t)c.high_frequency;   // feature.hpp:203-205
    std::vector<float> fb = h_filterbank(N, P.n_bins, fs, (uint32_t)c.low_frequency, high);
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
    if (max_nz > kws_mfcc_max_nz())
        return fail(KWS_ERROR_UNSUPPORTED_MODEL, "mel filter with %d taps (kernel keeps at most %d in registers)", max_nz, kws_mfcc_max_nz());
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

static EI_IMPULSE_ERROR build_nn_plan_f32(kws_handle *h);

// Recognise the Edge Impulse 1-D CNN family and fold its per-model constants.
static EI_IMPULSE_ERROR build_nn_plan(kws_handle *h)
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
            if (a_id != cur || !cb.is_const || cb.type != TYPE_I8 || (int)cb.nbytes != out_c || cb.dims.back() != out_c)
                return fail(KWS_ERROR_UNSUPPORTED_MODEL, "ADD %zu is not a per-channel constant add", i);
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
            if (po != out_n || h_pad_amount(s, 1, cur_w, f, po) != 0 || (po - 1) * s + f > cur_w || f > 8 || pl.p[5] != 0)
                return fail(KWS_ERROR_UNSUPPORTED_MODEL, "pool %zu window/padding outside the kernel's limits", i);
            k.pool = f; k.pool_stride = s; k.pool_w = po;
            cur = pl.out[0]; cur_w = po;
            i++;
            if (!skip_reshapes()) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "reshape changes quantisation");
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
    if (i >= m.n.size() || m.n[i].op != OP_FULLY_CONNECTED || m.n[i].in[0] != cur)
        return fail(KWS_ERROR_UNSUPPORTED_MODEL, "expected FULLY_CONNECTED after the conv blocks");
    {
        const Node &fc = m.n[i];
        const Tensor &x = m.t[cur], &w = m.t[fc.in[1]], &y = m.t[fc.out[0]];
        const Tensor *bias = (fc.in.size() > 2 && fc.in[2] >= 0) ? &m.t[fc.in[2]] : nullptr;
        N.fc_in = w.dims.back(); N.fc_out = w.dims[0];
        const KwsConvBlock &lb = N.blk[N.n_blocks - 1];
        if (N.fc_in != lb.pool_w * lb.out_c || N.fc_in > 64 || N.fc_out > 48 || N.fc_out != N.n_labels || !w.is_const ||
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
            if (po != out_n || h_pad_amount(s, 1, cur_w, f, po) != 0 || (po - 1) * s + f > cur_w || f > 8)
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
        if (N.fc_in != lb.pool_w * lb.out_c || N.fc_in > 64 || N.fc_out > 48 || N.fc_out != N.n_labels || !w.is_const ||
            (int)w.nbytes != N.fc_in * N.fc_out * 4)
            return fail(KWS_ERROR_UNSUPPORTED_MODEL, "FULLY_CONNECTED shape %dx%d outside the kernel's limits", N.fc_out, N.fc_in);
        h_act_range_f32(fc.p[0], &N.fc_min, &N.fc_max);
        std::vector<float> wv = floats(w), bv(N.fc_out, 0.0f);
        if (bias) { if ((int)bias->nbytes != N.fc_out * 4) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "FULLY_CONNECTED bias size"); bv = floats(*bias); }
        EI_IMPULSE_ERROR e;
        if ((e = h->upload(wv, &N.fc_w))) return e;
        if ((e = h->upload(bv, &N.fc_bias))) return e;
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

// ------------------------------------------------------------------------------------------------------------
//  C ABI
// ------------------------------------------------------------------------------------------------------------
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
    (void)hipSetDevice(h->device);
    for (void *p : h->dev_allocs) (void)hipFree(p);
    if (h->s_mfcc) (void)hipFree(h->s_mfcc);
    if (h->s_q) (void)hipFree(h->s_q);
    for (void *p : { (void *)h->ws.d_x, (void *)h->ws.d_f, (void *)h->ws.d_s, (void *)h->ws.d_w, (void *)h->ws.d_q })
        if (p) (void)hipFree(p);
    for (void *p : { (void *)h->ws.h_x, (void *)h->ws.h_s, (void *)h->ws.h_f })
        if (p) (void)hipHostFree(p);
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
const char *kws_nn_kernel_name(const kws_handle *h)
{
    return h->is_float ? "kws_nn_f32_kernel" : kws_nn_uses_mfma(h->nn) ? "kws_nn_mfma_kernel" : "kws_nn_kernel";
}

static EI_IMPULSE_ERROR ensure_scratch(kws_handle *h, size_t B)
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

static int grid_cap_mfcc(const kws_handle *h) { return h->n_cu * 8; }
static int grid_cap_nn(const kws_handle *h) { return h->n_cu * 4; }

// speechpy::feature::mfcc for B windows (kernel 1)
static EI_IMPULSE_ERROR spectral_device(kws_handle *h, const KwsDspPlan &P, const void *pcm, int is_float, size_t B, float *mfcc,
                                        const float *wrap, hipStream_t s, int out_stride = 0)
{
    if (B > 0x7fffffff) return fail(KWS_ERROR_BAD_ARGUMENT, "batch too large");
    if (((uintptr_t)pcm & 15) != 0) return fail(KWS_ERROR_BAD_ARGUMENT, "pcm must be 16-byte aligned");
    int rc = kws_launch_spectral(P, pcm, is_float, (int)B, mfcc, wrap, out_stride, grid_cap_mfcc(h), s);
    if (rc) return fail(KWS_ERROR_HIP, "MFCC kernel launch failed: %s (is the gfx950 code object present?)", hipGetErrorString((hipError_t)rc));
    return EI_IMPULSE_OK;
}

// extract_mfcc_features + quantisation in one launch (fused kernel)
static EI_IMPULSE_ERROR mfcc_fused_device(kws_handle *h, const void *pcm, int is_float, size_t B, float *features, int8_t *q, hipStream_t s)
{
    if (B > 0x7fffffff) return fail(KWS_ERROR_BAD_ARGUMENT, "batch too large");
    if (((uintptr_t)pcm & 15) != 0) return fail(KWS_ERROR_BAD_ARGUMENT, "pcm must be 16-byte aligned");
    int rc = kws_launch_mfcc_fused(h->dsp, pcm, is_float, (int)B, features, q, h->nn.in_scale, h->nn.in_zp, h->n_cu * 8, s);
    if (rc) return fail(KWS_ERROR_HIP, "MFCC kernel launch failed: %s (is the gfx950 code object present?)", hipGetErrorString((hipError_t)rc));
    return EI_IMPULSE_OK;
}

// float32 models: the network reads the feature matrix itself (ei_run_classifier.h:447-452 copies it into the input tensor)
static EI_IMPULSE_ERROR nn_f32_device(kws_handle *h, const float *features, size_t B, float *scores, float *tap_logits, hipStream_t s)
{
    if (B > 0x7fffffff) return fail(KWS_ERROR_BAD_ARGUMENT, "batch too large");
    int rc = kws_launch_nn_f32(h->nnf, h->d_nnf, features, (int)B, scores, tap_logits, h->n_cu, s);
    if (rc) return fail(KWS_ERROR_HIP, "float NN kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    return EI_IMPULSE_OK;
}
#define KWS_INT8_ONLY(h) do { if ((h)->is_float) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "%s takes an int8 tensor; the loaded model is float32", __func__); } while (0)

// cmvnw + quantise + (optionally) the network (kernel 2; the generic NN kernel follows when the graph does not fit
// the matrix-core path)
static EI_IMPULSE_ERROR cmvn_nn_device(kws_handle *h, const float *mfcc, size_t B, float *features, int8_t *q, float *scores,
                                       int8_t *tap_pooled, int8_t *tap_fc, int8_t *tap_out, hipStream_t s)
{
    if (B > 0x7fffffff) return fail(KWS_ERROR_BAD_ARGUMENT, "batch too large");
    int ran_nn = 0;
    if (h->is_float) {
        if (q || tap_pooled || tap_fc || tap_out) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "int8 outputs requested from a float32 model");
        float *f = features ? features : h->s_mfcc;
        int rc = kws_launch_cmvn_nn(h->dsp, h->nn, mfcc, (int)B, f, nullptr, nullptr, nullptr, 0, nullptr, nullptr, grid_cap_nn(h), &ran_nn, s);
        if (rc) return fail(KWS_ERROR_HIP, "CMVN kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
        return scores ? nn_f32_device(h, f, B, scores, nullptr, s) : EI_IMPULSE_OK;
    }
    int8_t *qq = q;
    if (scores && !qq) qq = h->s_q;           // the generic NN kernel reads the quantised tensor from HBM
    int rc = kws_launch_cmvn_nn(h->dsp, h->nn, mfcc, (int)B, features, qq, scores, tap_pooled, h->pooled_tap_bytes, tap_fc, tap_out,
                                grid_cap_nn(h), &ran_nn, s);
    if (rc) return fail(KWS_ERROR_HIP, "CMVN kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    if (scores && !ran_nn) {
        rc = kws_launch_nn(h->nn, qq, (int)B, scores, tap_pooled, h->pooled_tap_bytes, tap_fc, tap_out, grid_cap_nn(h), s);
        if (rc) return fail(KWS_ERROR_HIP, "NN kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    }
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
    if (((uintptr_t)pcm & 15) != 0) return fail(KWS_ERROR_BAD_ARGUMENT, "pcm must be 16-byte aligned");
    HIP_TRY(hipSetDevice(h->device));
    int rc = kws_launch_mfe(h->dsp, pcm, (int)B, mel, energy, grid_cap_mfcc(h), (hipStream_t)stream);
    if (rc) return fail(KWS_ERROR_HIP, "MFE kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
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
    return mfcc_fused_device(h, pcm, 0, B, features, q_in, (hipStream_t)stream);
}

// development aid (not in the public headers): per-phase shader-clock totals of wave 0 of the float network kernel
extern long long *kws_dev_f32_prof;
void kws_dev_set_f32_prof(long long *dev_buf) { kws_dev_f32_prof = dev_buf; }

// development/test aid (not in the public headers): force the generic dot4 NN kernel
void kws_dev_force_scalar_nn(int on) { kws_force_scalar_nn = on; }

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
    int rc = kws_launch_quantize(features, h->s_q, B * h->model.nn_input_frame_size, h->nn.in_scale, h->nn.in_zp, (hipStream_t)stream);
    if (rc) return fail(KWS_ERROR_HIP, "quantise kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    return kws_nn_batch_device(h, h->s_q, B, scores, nullptr, nullptr, nullptr, stream);
}

EI_IMPULSE_ERROR kws_run_classifier_batch_device(kws_handle *h, const int16_t *pcm, size_t B, float *scores, float *features,
                                                 int8_t *q_in, void *stream)
{
    if (!h || !pcm || !scores) return fail(KWS_ERROR_BAD_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    std::lock_guard<std::mutex> lk(h->mu);
    EI_IMPULSE_ERROR e = ensure_scratch(h, B);
    if (e) return e;
    if (h->is_float) {                       // features -> HBM (caller's buffer or scratch) -> float network
        if (q_in) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "int8 input tensor requested from a float32 model");
        float *f = features ? features : h->s_mfcc;
        e = mfcc_fused_device(h, pcm, 0, B, f, nullptr, (hipStream_t)stream);
        return e ? e : nn_f32_device(h, f, B, scores, nullptr, (hipStream_t)stream);
    }
    // one fused launch for extract_mfcc_features + quantisation (the cepstra stay in LDS), then the network;
    // the float feature matrix only leaves the chip when the caller asks for it
    int8_t *q = q_in ? q_in : h->s_q;
    e = mfcc_fused_device(h, pcm, 0, B, features, q, (hipStream_t)stream);
    if (e) return e;
    return kws_nn_batch_device(h, q, B, scores, nullptr, nullptr, nullptr, stream);
}

EI_IMPULSE_ERROR kws_run_classifier_batch(kws_handle *h, const int16_t *pcm, size_t B, float *scores, float *features, int8_t *q_in)
{
    if (!h || !pcm || !scores) return fail(KWS_ERROR_BAD_ARGUMENT, "null argument");
    if (B == 0) return EI_IMPULSE_OK;
    HIP_TRY(hipSetDevice(h->device));
    const size_t n = h->model.raw_sample_count, F = h->model.nn_input_frame_size, C = h->model.labels.size();
    int16_t *d_pcm = nullptr; float *d_s = nullptr, *d_f = nullptr; int8_t *d_q = nullptr;
    EI_IMPULSE_ERROR e = EI_IMPULSE_OK;
    auto cleanup = [&]() { if (d_pcm) (void)hipFree(d_pcm); if (d_s) (void)hipFree(d_s); if (d_f) (void)hipFree(d_f); if (d_q) (void)hipFree(d_q); };
#define TRY_OR_CLEAN(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return fail(KWS_ERROR_HIP, "%s: %s", #x, hipGetErrorString(e_)); } } while (0)
    TRY_OR_CLEAN(hipMalloc((void **)&d_pcm, B * n * sizeof(int16_t)));
    TRY_OR_CLEAN(hipMalloc((void **)&d_s, B * C * sizeof(float)));
    TRY_OR_CLEAN(hipMalloc((void **)&d_f, B * F * sizeof(float)));
    if (h->is_float && q_in) { cleanup(); return fail(KWS_ERROR_UNSUPPORTED_MODEL, "int8 input tensor requested from a float32 model"); }
    if (!h->is_float) TRY_OR_CLEAN(hipMalloc((void **)&d_q, B * F));
    TRY_OR_CLEAN(hipMemcpy(d_pcm, pcm, B * n * sizeof(int16_t), hipMemcpyHostToDevice));
    e = kws_run_classifier_batch_device(h, d_pcm, B, d_s, d_f, d_q, nullptr);
    if (e == EI_IMPULSE_OK) {
        TRY_OR_CLEAN(hipMemcpy(scores, d_s, B * C * sizeof(float), hipMemcpyDeviceToHost));
        if (features) TRY_OR_CLEAN(hipMemcpy(features, d_f, B * F * sizeof(float), hipMemcpyDeviceToHost));
        if (q_in) TRY_OR_CLEAN(hipMemcpy(q_in, d_q, B * F, hipMemcpyDeviceToHost));
    }
    cleanup();
    return e;
}

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

// ------------------------------------------------------------------------------------------------------------
//  continuous mode for S streams in lock step ("many concurrent streams, per-stream state in HBM")
// ------------------------------------------------------------------------------------------------------------
struct kws_stream_batch {
    kws_handle *h = nullptr;
    size_t S = 0;
    float *feat[2] = { nullptr, nullptr };   // rolling cepstra buffers [S][F] (ping-pong for the shift)
    int cur = 0;
    float *running_sum = nullptr, *maf_buf = nullptr;   // [S][C], [S][C][taps]
    float *zeros = nullptr;                   // [S] end-of-signal samples when the caller gives none
    size_t slice_offset = 0;
    bool full = false, first_run = false;     // first_run: like the reference's function-static, never reset
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
    const bool grown = sb->first_run;
    if (grown) n_claimed += (size_t)(m.dsp.frame_length * (float)m.frequency);
    sb->first_run = true;
    const int frame_len = h->dsp.frame_len, stride = h->dsp.frame_stride;
    const int nf = n_claimed >= (size_t)frame_len ? (int)floorf((float)(n_claimed - (size_t)frame_len) / (float)stride) : 0;
    const size_t feature_size = (size_t)(nf > 0 ? nf : 0) * (size_t)m.dsp.num_cepstral;
    if (nf < 1 || nf > kws_mfcc_max_frames(h->dsp.n_filters) || feature_size > F || sb->slice_offset + feature_size > F ||
        (size_t)(nf - 1) * stride + h->dsp.fft_len > slice_samples || (slice_samples * 2) % 16 != 0)
        return fail(EI_IMPULSE_DSP_ERROR, "slice of %zu samples (claimed %zu) yields %d frames", slice_samples, n_claimed, nf);
    KwsDspPlan P = h->dsp;
    P.n_samples = (int)slice_samples;      // memory stride between the streams' slices
    P.n_frames = nf;
    // x[-1] of the slice: the reference takes the sample at total_length-1, which lies beyond the slice once it has grown
    const float *wrap = grown ? (end_of_signal ? end_of_signal : sb->zeros) : nullptr;
    float *feat = sb->feat[sb->cur];
    EI_IMPULSE_ERROR e = spectral_device(h, P, slices, 0, S, feat + sb->slice_offset, wrap, st, (int)F);
    if (e) return e;
    if (!sb->full) {
        sb->slice_offset += feature_size;
        if (sb->slice_offset > (F - feature_size)) { sb->full = true; sb->slice_offset -= feature_size; }
    }
    if (!sb->full) return EI_IMPULSE_OK;
    {
        std::lock_guard<std::mutex> lk(h->mu);
        e = ensure_scratch(h, S);
        if (!e) e = cmvn_nn_device(h, feat, S, nullptr, nullptr, scores, nullptr, nullptr, nullptr, st);
    }
    if (e) return e;
    int rc = kws_launch_maf(scores, sb->running_sum, sb->maf_buf, (int)(S * C), (int)sb->buf_idx, kMafTaps, st);
    if (rc) return fail(KWS_ERROR_HIP, "moving-average kernel launch failed");
    if (++sb->buf_idx >= (uint32_t)kMafTaps) sb->buf_idx = 0;
    rc = kws_launch_shift(feat, sb->feat[sb->cur ^ 1], (int)S, (int)F, (int)feature_size, st);
    if (rc) return fail(KWS_ERROR_HIP, "shift kernel launch failed");
    sb->cur ^= 1;
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
        w.cap_x = n_x;
    }
    return EI_IMPULSE_OK;
}

static kws_handle *g_default = nullptr;
static bool g_default_owned = false;
static std::mutex g_default_mu;

EI_IMPULSE_ERROR kws_set_default_model(kws_handle *h)
{
    std::lock_guard<std::mutex> lk(g_default_mu);
    if (g_default && g_default_owned && g_default != h) kws_destroy(g_default);
    g_default = h;
    g_default_owned = false;
    return EI_IMPULSE_OK;
}

kws_handle *kws_default_model(void)
{
    std::lock_guard<std::mutex> lk(g_default_mu);
    if (!g_default) {
        const char *path = getenv("KWS_MODEL");
        if (!path) { fail(KWS_ERROR_NO_MODEL, "no model: call kws_set_default_model() or set KWS_MODEL"); return nullptr; }
        const char *dev = getenv("KWS_DEVICE");
        kws_handle *h = nullptr;
        if (kws_create_from_file(path, dev ? atoi(dev) : 0, &h) != EI_IMPULSE_OK) return nullptr;
        g_default = h;
        g_default_owned = true;
    }
    return g_default;
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
static ei_impulse_result_timing_t *result_timing(const kws_handle *h, ei_impulse_result_t *result)
{
    const size_t C = h->model.labels.size();
    return (ei_impulse_result_timing_t *)((char *)result + C * sizeof(ei_impulse_result_classification_t) + sizeof(float));
}

EI_IMPULSE_ERROR run_inference(ei_matrix_t *fmatrix, ei_impulse_result_t *result, bool debug)
{
    kws_handle *h = kws_default_model();
    if (!h) return g_err_code != EI_IMPULSE_OK ? g_err_code : KWS_ERROR_NO_MODEL;   // why the default model is missing
    if (!fmatrix || !fmatrix->buffer || !result) return fail(KWS_ERROR_BAD_ARGUMENT, "null argument");
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
    if (ei_run_impulse_check_canceled() == EI_IMPULSE_CANCELED) return EI_IMPULSE_CANCELED;     // ei_run_classifier.h:489-491
    return EI_IMPULSE_OK;
}

EI_IMPULSE_ERROR run_classifier(signal_t *signal, ei_impulse_result_t *result, bool debug)
{
    kws_handle *h = kws_default_model();
    if (!h) return g_err_code != EI_IMPULSE_OK ? g_err_code : KWS_ERROR_NO_MODEL;   // why the default model is missing
    if (!signal || !signal->get_data || !result) return fail(KWS_ERROR_BAD_ARGUMENT, "null argument");
    const size_t n = h->model.raw_sample_count, F = h->model.nn_input_frame_size, C = h->model.labels.size();
    // The reference sizes its frame count from signal->total_length (ei_run_dsp.h:277-286); a length that yields another
    // feature count than the model's is EIDSP_MATRIX_SIZE_MISMATCH there (-> EI_IMPULSE_DSP_ERROR).
    if (signal->total_length != n) { ei_printf("ERR: Failed to run DSP process (%d)\n", -1002); return fail(EI_IMPULSE_DSP_ERROR, "signal length %zu, model window %zu", signal->total_length, n); }
    HIP_TRY(hipSetDevice(h->device));
    std::lock_guard<std::mutex> sdk_lk(h->sdk_mu);
    uint64_t t0 = ei_read_timer_ms();
    EI_IMPULSE_ERROR e = ensure_ws(h, n);
    if (e) return e;
    kws_handle::Ws &w = h->ws;
    // gather the window through the caller's callback (float samples, as the SDK hands them to the DSP block) straight
    // into pinned memory
    const size_t chunk = 4000;
    for (size_t off = 0; off < n; off += chunk) {
        const size_t len = std::min(chunk, n - off);
        int r = signal->get_data(off, len, w.h_x + off);
        if (r != 0) { ei_printf("ERR: Failed to run DSP process (%d)\n", r); return fail(EI_IMPULSE_DSP_ERROR, "signal->get_data returned %d", r); }
    }
    std::vector<float> scores(C);
    if (hipMemcpyAsync(w.d_x, w.h_x, n * sizeof(float), hipMemcpyHostToDevice, w.st) != hipSuccess) e = fail(KWS_ERROR_HIP, "h2d copy failed");
    if (!e) e = mfcc_fused_device(h, w.d_x, 1, 1, w.d_f, h->is_float ? nullptr : w.d_q, w.st);
    if (!e && debug && hipMemcpyAsync(w.h_f, w.d_f, F * sizeof(float), hipMemcpyDeviceToHost, w.st) != hipSuccess) e = fail(KWS_ERROR_HIP, "d2h copy failed");
    if (!e && hipStreamSynchronize(w.st) != hipSuccess) e = fail(KWS_ERROR_HIP, "MFCC kernel failed");
    if (e) return e;
    if (ei_run_impulse_check_canceled() == EI_IMPULSE_CANCELED) return EI_IMPULSE_CANCELED;   // ei_run_classifier.h:689-691
    const int dsp_ms = (int)(ei_read_timer_ms() - t0);
    result_timing(h, result)->dsp = dsp_ms;
    if (debug) {
        ei_printf("Features (%d ms.): ", dsp_ms);
        for (size_t ix = 0; ix < F; ix++) { ei_printf_float(w.h_f[ix]); ei_printf(" "); }
        ei_printf("\n");
        ei_printf("Running neural network...\n");
    }
    uint64_t t1 = ei_read_timer_ms();
    e = h->is_float ? nn_f32_device(h, w.d_f, 1, w.d_s, nullptr, w.st) : kws_nn_batch_device(h, w.d_q, 1, w.d_s, nullptr, nullptr, nullptr, w.st);
    if (!e && (hipMemcpyAsync(w.h_s, w.d_s, C * sizeof(float), hipMemcpyDeviceToHost, w.st) != hipSuccess ||
               hipStreamSynchronize(w.st) != hipSuccess)) e = fail(KWS_ERROR_HIP, "d2h copy failed");
    if (e) return e;
    memcpy(scores.data(), w.h_s, C * sizeof(float));
    fill_result(h, result, scores.data(), debug, (int)(ei_read_timer_ms() - t1));
    if (ei_run_impulse_check_canceled() == EI_IMPULSE_CANCELED) return EI_IMPULSE_CANCELED;
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
    if (!h) return g_err_code != EI_IMPULSE_OK ? g_err_code : KWS_ERROR_NO_MODEL;   // why the default model is missing
    if (!signal || !signal->get_data || !result) return fail(KWS_ERROR_BAD_ARGUMENT, "null argument");
    const Model &m = h->model;
    const size_t F = m.nn_input_frame_size, C = m.labels.size();
    const int ncep = m.dsp.num_cepstral;
    if (h->cont_features.size() != F) h->cont_features.assign(F, 0.0f);         // static_features_matrix (calloc'd)
    HIP_TRY(hipSetDevice(h->device));
    uint64_t dsp_start_ms = ei_read_timer_ms();

    // ---- extract_mfcc_per_slice_features ----------------------------------------------------------------------
    if (g_cont_first_run) signal->total_length += (size_t)(m.dsp.frame_length * (float)m.frequency);
    g_cont_first_run = true;
    const size_t n_claimed = signal->total_length;
    float eos = 0.0f;                                                           // _end_of_signal_buffer (calloc)
    if (n_claimed >= 1) (void)signal->get_data(n_claimed - (size_t)m.dsp.pre_shift, (size_t)m.dsp.pre_shift, &eos);
    const int frame_len = h->dsp.frame_len, stride = h->dsp.frame_stride;
    const int nf = n_claimed >= (size_t)frame_len ? (int)floorf((float)(n_claimed - (size_t)frame_len) / (float)stride) : 0;
    const size_t feature_size = (size_t)(nf > 0 ? nf : 0) * (size_t)ncep;
    if (nf < 1 || feature_size > F || h->slice_offset + feature_size > F || nf > kws_mfcc_max_frames(h->dsp.n_filters)) {
        ei_printf("ERR: MFCC failed (%d)\n", -1002);                           // EIDSP_MATRIX_SIZE_MISMATCH
        ei_printf("ERR: Failed to run DSP process (%d)\n", -1002);
        return fail(EI_IMPULSE_DSP_ERROR, "slice of %zu samples yields %d frames", n_claimed, nf);
    }
    const size_t needed = (size_t)(nf - 1) * stride + frame_len;                // last sample any frame reads
    const size_t n_x = std::max(n_claimed, needed) + 16;
    std::lock_guard<std::mutex> sdk_lk(h->sdk_mu);
    EI_IMPULSE_ERROR e = ensure_ws(h, n_x);
    if (e) return e;
    kws_handle::Ws &w = h->ws;
    memset(w.h_x, 0, n_x * sizeof(float));
    {
        int r = signal->get_data(0, needed, w.h_x);
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
        ei_impulse_result_classification_t *cls = (ei_impulse_result_classification_t *)result;
        for (size_t ix = 0; ix < C; ix++) cls[ix].value = run_moving_average_filter(&h->maf[ix], cls[ix].value);
        // shift the feature buffer for new data (ei_run_classifier.h:277-279)
        for (size_t i = 0; i < F - feature_size; i++) h->cont_features[i] = h->cont_features[i + feature_size];
        if (ei_run_impulse_check_canceled() == EI_IMPULSE_CANCELED) return EI_IMPULSE_CANCELED;
        return EI_IMPULSE_OK;
    }
    return EI_IMPULSE_OK;
}

}  // extern "C"
