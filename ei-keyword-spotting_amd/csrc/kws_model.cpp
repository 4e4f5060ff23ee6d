// kws_model.cpp -- error state, weak platform hooks, the .kwsm blob parser and the host arithmetic of the table builders.
#include "kws_internal.h"

static thread_local std::string g_err;
static thread_local EI_IMPULSE_ERROR g_err_code = EI_IMPULSE_OK;
EI_IMPULSE_ERROR kws_fail(EI_IMPULSE_ERROR code, const char *fmt, ...)
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
EI_IMPULSE_ERROR kws_last_error_code(void) { return g_err_code; }

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

#pragma GCC visibility pop

// ------------------------------------------------------------------------------------------------------------
//  model blob (layout: tools/eon_import.py)
// ------------------------------------------------------------------------------------------------------------
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
    const uint32_t version = r.u32();
    if (version != 1 && version != 2) return false;
    uint32_t nt = r.u32(), nn = r.u32(), nl = r.u32();
    m.t_in = r.u32(); m.t_out = r.u32();
    m.raw_sample_count = r.u32(); m.frequency = r.u32(); m.nn_input_frame_size = r.u32();
    DspCfg &d = m.dsp;
    d.axes = r.i32(); d.num_cepstral = r.i32(); d.num_filters = r.i32(); d.fft_length = r.i32(); d.win_size = r.i32();
    d.low_frequency = r.i32(); d.high_frequency = r.i32(); d.pre_shift = r.i32();
    d.frame_length = r.f32(); d.frame_stride = r.f32(); d.pre_cof = r.f32();
    {
        // version 2: one more i32 -- bits 0..7 the DSP block type, bit 8 EIDSP_QUANTIZE_FILTERBANK; anything else is not a model of ours
        const int v = version == 2 ? r.i32() : 0;
        if (v & ~0x1ff) return false;
        d.block = v & 0xff;
        d.quantize_fb = (v >> 8) & 1;
    }
    if (d.block != DSP_BLOCK_MFCC && d.block != DSP_BLOCK_MFE) return false;
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
        // a quantisation scale is a positive, finite, normal number (TFLite's converter never writes anything else); the plan
        // builders divide by it and feed the quotients to frexp / round (found by the UBSan run of tests/test_sanitizers.py: a NaN
        // scale reached the double -> int64 conversion of h_quantize_multiplier)
        for (float sc : t.scale) if (!(sc >= 1e-30f && sc <= 1e30f)) return false;
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
    // Per-op operand contract: the plan builders (kws_plan.cpp) index in[] / out[] / dims / scale without further checks, so a
    // blob must satisfy it here.  Mandatory operands are real tensor ids; only the bias slot (in[2]) of a convolution or of
    // FULLY_CONNECTED may be -1 (absent).  Every tensor an op touches has at least one dimension; activations of CONV_2D /
    // DEPTHWISE_CONV_2D / MAX_POOL_2D are 4-D; every int8 tensor an op touches carries a scale and a zero point (checked above
    // for all int8 tensors: nq >= 1).
    auto real = [&](int v) { return v >= 0 && v < (int)nt; };
    for (auto &n : m.n) {
        size_t need_in = 1, need_out = 1;
        switch (n.op) {
        case OP_RESHAPE: need_in = 1; break;                                 // in[1] (the shape operand) is optional
        case OP_CONV_2D: case OP_DEPTHWISE_CONV_2D: case OP_FULLY_CONNECTED: need_in = 2; break;
        case OP_ADD: need_in = 2; break;
        case OP_MAX_POOL_2D: case OP_SOFTMAX: need_in = 1; break;
        default: return false;                                                // unknown op code
        }
        if (n.in.size() < need_in || n.out.size() < need_out) return false;
        for (size_t k = 0; k < n.in.size(); k++) {
            const bool optional = (k >= need_in);
            if (!real(n.in[k]) && !(optional && n.in[k] == -1)) return false;
        }
        for (int v : n.out) if (!real(v)) return false;
        for (size_t k = 0; k < n.in.size(); k++)
            if (real(n.in[k]) && m.t[n.in[k]].dims.empty()) return false;
        for (int v : n.out) if (m.t[v].dims.empty()) return false;
        if (n.op == OP_CONV_2D || n.op == OP_DEPTHWISE_CONV_2D) {
            if (m.t[n.in[0]].dims.size() != 4 || m.t[n.in[1]].dims.size() != 4 || m.t[n.out[0]].dims.size() != 4) return false;
        }
        if (n.op == OP_MAX_POOL_2D && (m.t[n.in[0]].dims.size() != 4 || m.t[n.out[0]].dims.size() != 4)) return false;
        if (n.op == OP_FULLY_CONNECTED && m.t[n.in[1]].dims.size() != 2) return false;
    }
    if (m.t[m.t_in].dims.empty() || m.t[m.t_out].dims.empty() || m.labels.empty()) return false;
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

// EIDSP_QUANTIZE_FILTERBANK = 1 (the SDK's default, SDK/dsp/config.hpp:75-77): a mel weight is stored as an index into
// quantized_values_one_zero[] (numpy.hpp:52) and read back through it (numpy.hpp:423-468, 222-250).  That table holds, ascending, every
// fraction a / b with b <= 22 and every i / 100 -- 231 distinct values, each a correctly rounded float division, so it is generated here
// from the rule (tests/test_oracle_vs_reference.py pins the restatement's copy of the same rule entry by entry against the compiled
// reference).  quantize_zero_one's search keeps the reference's quirks: out-of-range values return the table VALUE cast to the index
// type (0 below; 1 above, i.e. 1/100), the final pick between the two neighbours is made in float arithmetic.
float h_quantize_zero_one(float value)
{
    static const std::vector<float> tab = [] {
        std::vector<std::pair<int, int>> fr;
        auto add = [&](int a, int b) {
            for (const auto &f : fr) if ((long)f.first * b == (long)a * f.second) return;
            fr.push_back({ a, b });
        };
        for (int b = 1; b <= 22; b++) for (int a = 0; a <= b; a++) add(a, b);
        for (int a = 0; a <= 100; a++) add(a, 100);
        std::sort(fr.begin(), fr.end(), [](const std::pair<int, int> &x, const std::pair<int, int> &y) { return (long)x.first * y.second < (long)y.first * x.second; });
        std::vector<float> t;
        for (const auto &f : fr) t.push_back((float)f.first / (float)f.second);
        return t;
    }();
    const int length = (int)tab.size();
    int ix = -1;
    for (int i = 0; i < length && ix < 0; i++) if (tab[(size_t)i] == value) ix = i;
    if (ix < 0) {
        if (value < tab[0]) ix = (uint8_t)tab[0];
        else if (value > tab[(size_t)length - 1]) ix = (uint8_t)tab[(size_t)length - 1];
        else {
            int lo = 0, hi = length - 1;
            bool hit = false;
            while (lo <= hi) {
                const int mid = (hi + lo) / 2;
                if (value < tab[(size_t)mid]) hi = mid - 1;
                else if (value > tab[(size_t)mid]) lo = mid + 1;
                else { ix = (uint8_t)tab[(size_t)mid]; hit = true; break; }     // unreachable for numbers; a NaN ends here as in the reference
            }
            if (!hit) ix = (tab[(size_t)lo] - value) < (value - tab[(size_t)hi]) ? lo : hi;
        }
    }
    return tab[(size_t)std::min(ix, std::min(247, length - 1))];
}

// feature::filterbanks (feature.hpp:54-171) + functions::triangle (functions.hpp:90-104), dense [coeff][M]
std::vector<float> h_filterbank(int num_filter, int coefficients, uint32_t fs, uint32_t low, uint32_t high, bool quantize)
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
            if (bin >= 0 && bin < coefficients) fb[(size_t)bin * num_filter + i] = quantize ? h_quantize_zero_one(o[zx]) : o[zx];
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
    if (m == 0. || !std::isfinite(m)) { *q = 0; *shift = 0; return; }
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

