/*
 * oracle/kws_oracle.c -- TEST INFRASTRUCTURE, not product code (see kws_oracle.h).
 *
 * Plain-C11 restatement of the reference arithmetic.  Must be compiled WITHOUT FP
 * contraction and for baseline x86-64 (oracle/Makefile: -O2 -ffp-contract=off) so that every
 * float operation rounds exactly as the reference's g++ -O2 build does (FLT_EVAL_METHOD 0).
 *
 * Conventions that matter for bit-exactness, as established against the compiled reference:
 *  - the reference is C++: unqualified sqrt()/cos()/sin()/round() on a float argument resolve to
 *    the float overloads (libstdc++ <math.h> wrapper), pow(float,int) promotes to double;
 *  - "float += double" is evaluated in double and rounded to float once per statement.
 */
#define _GNU_SOURCE 1   /* M_PI, clock_gettime */
#include "kws_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define KWSO_OK 0
#define KWSO_ERR_SIZE (-1002)      /* EIDSP_MATRIX_SIZE_MISMATCH  SDK/dsp/returntypes.hpp:30-43 */
#define KWSO_ERR_MEM (-1002 - 0)   /* placeholder, allocation failures are reported as -1005 */
#define KWSO_ERR_OOM (-1005)
#define KWSO_ERR_PARAM (-1008)
#define KWSO_ERR_UNSUPPORTED (-1099)

/* ====================================================================== */
/*  numpy::log -- fast natural log            SDK/dsp/numpy.hpp:1350-1371  */
/* ====================================================================== */
float kwso_log(float a)
{
    uint32_t gu;
    memcpy(&gu, &a, 4);
    int32_t g = (int32_t)gu;
    /* e = (g - 0x3f2aaaab) & 0xff800000 : integer exponent split around 2/3 */
    int32_t e = (int32_t)(((uint32_t)g - 0x3f2aaaabu) & 0xff800000u);
    g = (int32_t)((uint32_t)g - (uint32_t)e);
    float m;
    memcpy(&m, &g, 4);
    float i = (float)e * 1.19209290e-7f;
    float f = m - 1.0f;
    float s = f * f;
    float r = fmaf(0.230836749f, f, -0.279208571f);
    float t = fmaf(0.331826031f, f, -0.498910338f);
    r = fmaf(r, s, t);
    r = fmaf(r, s, f);
    r = fmaf(i, 0.693147182f, r);
    return r;
}

/* functions.hpp:42-44 : 1127.0 * numpy::log(1 + f / 700.0f)  (double multiply, float result) */
float kwso_frequency_to_mel(float f)
{
    return (float)(1127.0 * (double)kwso_log(1 + f / 700.0f));
}

/* functions.hpp:52-54 : 700.0f * (exp(mel / 1127.0f) - 1.0f) ; exp(float) is the float overload */
float kwso_mel_to_frequency(float mel)
{
    return 700.0f * (expf(mel / 1127.0f) - 1.0f);
}

/* ====================================================================== */
/*  framing                               processing.hpp:194-245, 260-284  */
/* ====================================================================== */
int kwso_frame_length_samples(const kwso_mfcc_config *c)
{
    return (int)roundf((float)c->sampling_frequency * c->frame_length);
}

int kwso_num_frames(size_t n, const kwso_mfcc_config *c)
{
    int frame_sample_length = kwso_frame_length_samples(c);
    float frame_stride = roundf((float)c->sampling_frequency * c->frame_stride);
    /* size_t arithmetic, as in the reference (wraps for n < frame length) */
    size_t diff = n - (size_t)frame_sample_length;
    return (int)floorf((float)diff / frame_stride);
}

/* ====================================================================== */
/*  numpy::linspace                                 numpy.hpp:1257-1280    */
/* ====================================================================== */
static void linspace(float start, float stop, uint32_t number, float *out)
{
    if (number == 1) { out[0] = start; return; }
    float step = (stop - start) / (number - 1);   /* uint32 -> float */
    for (uint32_t ix = 0; ix < number - 1; ix++) out[ix] = start + ix * step;
    out[number - 1] = stop;
}

/* ====================================================================== */
/*  EIDSP_QUANTIZE_FILTERBANK = 1             numpy.hpp:52, 423-468        */
/*  The table quantized_values_one_zero[] holds, in ascending order, every  */
/*  fraction a/b with b <= 22 and every i/100 (231 distinct values, each     */
/*  written as a float division): built here from that rule -- IEEE division */
/*  is correctly rounded, so the spelling of a fraction does not matter --    */
/*  and pinned entry by entry against the compiled reference                  */
/*  (tests/test_oracle_vs_reference.py).                                      */
/* ====================================================================== */
static float g_qtab[256];
static int g_qtab_n = 0;
static void qtab_build(void)
{
    /* (numerator, denominator) pairs in lowest terms, sorted by value */
    int num[256], den[256], n = 0;
    for (int pass = 0; pass < 2; pass++)
        for (int b = pass ? 100 : 1; b <= (pass ? 100 : 22); b++)
            for (int a = 0; a <= b; a++) {
                int dup = 0;
                for (int k = 0; k < n && !dup; k++) dup = (long)num[k] * b == (long)a * den[k];
                if (!dup && n < 256) { num[n] = a; den[n] = b; n++; }
            }
    for (int i = 1; i < n; i++)                           /* insertion sort by a/b (cross-multiplication: exact) */
        for (int j = i; j > 0 && (long)num[j] * den[j - 1] < (long)num[j - 1] * den[j]; j--) {
            int t = num[j]; num[j] = num[j - 1]; num[j - 1] = t;
            t = den[j]; den[j] = den[j - 1]; den[j - 1] = t;
        }
    for (int i = 0; i < n; i++) g_qtab[i] = (float)num[i] / (float)den[i];
    g_qtab_n = n;
}

/* quantize_zero_one followed by dequantize_zero_one, with the reference's quirks: the early returns for out-of-range values return the
 * TABLE VALUE converted to uint8_t (0 below the table; 1 -- i.e. entry 1 = 1/100 -- above it), the `return ...[mid]` inside the binary
 * search is unreachable (exact matches are caught by the linear scan), and the final pick is made in float arithmetic. */
float kwso_quantize_zero_one(float value)
{
    if (!g_qtab_n) qtab_build();
    const int length = g_qtab_n;
    int ix_out = -1;
    for (int ix = 0; ix < length && ix_out < 0; ix++) if (g_qtab[ix] == value) ix_out = ix;
    if (ix_out < 0) {
        if (value < g_qtab[0]) ix_out = (uint8_t)g_qtab[0];
        else if (value > g_qtab[length - 1]) ix_out = (uint8_t)g_qtab[length - 1];
        else {
            int lo = 0, hi = length - 1;
            while (lo <= hi) {
                int mid = (hi + lo) / 2;
                if (value < g_qtab[mid]) hi = mid - 1;
                else if (value > g_qtab[mid]) lo = mid + 1;
                else { lo = hi = mid; break; }               /* (NaN lands here in the reference too: neither < nor >) */
            }
            if (lo == hi) ix_out = (uint8_t)g_qtab[lo];       /* the reference returns the table VALUE as the index */
            else ix_out = (g_qtab[lo] - value) < (value - g_qtab[hi]) ? lo : hi;
        }
    }
    if (ix_out > 247) ix_out = 247;                           /* dequantize_zero_one's clamp (beyond the table's 231 entries: never reached) */
    return g_qtab[ix_out];
}

/* ====================================================================== */
/*  mel filterbank (transposed)        feature.hpp:54-171, functions.hpp:90-104 */
/* ====================================================================== */
int kwso_filterbanks(const kwso_mfcc_config *c, float *fb_t)
{
    const int num_filter = c->num_filters;
    const int coefficients = c->fft_length / 2 + 1;
    const uint32_t sampling_freq = (uint32_t)c->sampling_frequency;
    const uint32_t low_freq = (uint32_t)c->low_frequency;
    const uint32_t high_freq = c->high_frequency == 0 ? sampling_freq / 2 : (uint32_t)c->high_frequency;
    const int np = num_filter + 2;
    float *mels = (float *)malloc(sizeof(float) * np);
    float *hertz = (float *)malloc(sizeof(float) * np);
    int *freq_index = (int *)malloc(sizeof(int) * np);
    if (!mels || !hertz || !freq_index) { free(mels); free(hertz); free(freq_index); return KWSO_ERR_OOM; }
    memset(fb_t, 0, sizeof(float) * (size_t)coefficients * num_filter);

    linspace(kwso_frequency_to_mel((float)low_freq), kwso_frequency_to_mel((float)high_freq), (uint32_t)np, mels);
    for (int ix = 0; ix < np; ix++) {
        hertz[ix] = kwso_mel_to_frequency(mels[ix]);
        if (hertz[ix] < low_freq) hertz[ix] = (float)low_freq;      /* float vs uint32 compare -> float */
        if (hertz[ix] > high_freq) hertz[ix] = (float)high_freq;
        if (ix == np - 1) hertz[ix] = (float)((double)hertz[ix] - 0.001);  /* hertz[ix] -= 0.001 (double) */
    }
    for (int ix = 0; ix < np; ix++) {
        /* floor((coefficients + 1) * hertz / sampling_freq): int*float -> float, / uint32 -> float */
        float v = (float)(coefficients + 1) * hertz[ix] / (float)sampling_freq;
        freq_index[ix] = (int)floorf(v);
    }
    for (int i = 0; i < num_filter; i++) {
        int left = freq_index[i], middle = freq_index[i + 1], right = freq_index[i + 2];
        int zn = right - left + 1;
        if (zn < 1) continue;   /* the reference would ask linspace for <1 points (EIDSP_PARAMETER_INVALID) */
        float *z = (float *)calloc((size_t)zn, sizeof(float));
        float *o = (float *)calloc((size_t)zn, sizeof(float));
        linspace((float)left, (float)right, (uint32_t)zn, z);
        /* functions::triangle: the second condition overwrites the first at x == middle */
        for (int k = 0; k < zn; k++) {
            float x = z[k];
            if (x > left && x <= middle) o[k] = (x - left) / (middle - left);
            if (x < right && middle <= x) o[k] = (right - x) / (right - middle);
        }
        for (int zx = 0; zx < zn; zx++) {
            int bin = left + zx;
            /* EIDSP_QUANTIZE_FILTERBANK: the weight is stored as a table index and read back through the table (feature.hpp:154-158,
               numpy.hpp:222-250) */
            if (bin >= 0 && bin < coefficients) fb_t[(size_t)bin * num_filter + i] = c->quantize_filterbank ? kwso_quantize_zero_one(o[zx]) : o[zx];
        }
        free(z); free(o);
    }
    free(mels); free(hertz); free(freq_index);
    return KWSO_OK;
}

/* ====================================================================== */
/*  pre-emphasis                                    processing.hpp:52-138  */
/*  y[n] = x[n] - cof * x[n-shift]; for n < shift the "previous" sample is */
/*  x[N-shift+n] (captured at construction with N = total_length).         */
/*  x = int16 / 32768 (numpy::int16_to_float, numpy.hpp:1289-1298)         */
/* ====================================================================== */
/* end_of_signal (optional): the `shift` floats the reference's constructor fetched from
 * signal->get_data(total_length - shift, shift) (processing.hpp:68); NULL = take them from pcm[n-shift..n). */
static int preemphasis_ex(const int16_t *pcm, size_t n, float cof, int shift, size_t offset, size_t length, float *out,
                          const float *end_of_signal)
{
    if (shift < 1 || (size_t)shift > n) return KWSO_ERR_PARAM;
    if (offset + length > n) return -1004;   /* EIDSP_OUT_OF_BOUNDS */
    for (size_t ix = 0; ix < length; ix++) {
        size_t p = offset + ix;
        float now = (float)pcm[p] / 32768;
        float prev;
        if (p < (size_t)shift) prev = end_of_signal ? end_of_signal[p] : (float)pcm[n - (size_t)shift + p] / 32768;
        else prev = (float)pcm[p - (size_t)shift] / 32768;
        float prod = cof * prev;
        out[ix] = now - prod;
    }
    return KWSO_OK;
}

int kwso_preemphasis(const int16_t *pcm, size_t n, float cof, int shift,
                     size_t offset, size_t length, float *out)
{
    return preemphasis_ex(pcm, n, cof, shift, offset, length, out, NULL);
}

/* ====================================================================== */
/*  KissFFT restatement            kissfft/kiss_fft.cpp, kiss_fftr.cpp     */
/* ====================================================================== */
typedef struct { float r, i; } cpx;

typedef struct {
    int nfft;
    int factors[64];
    cpx *tw;
} fft_plan;

/* kf_factor (kiss_fft.cpp:303-324): powers of 4, then 2, then odd primes */
static void plan_factor(int n, int *facbuf)
{
    int p = 4;
    double floor_sqrt = floor(sqrt((double)n));
    do {
        while (n % p) {
            switch (p) {
            case 4: p = 2; break;
            case 2: p = 3; break;
            default: p += 2; break;
            }
            if (p > floor_sqrt) p = n;
        }
        n /= p;
        *facbuf++ = p;
        *facbuf++ = n;
    } while (n > 1);
}

/* kiss_fft_alloc (kiss_fft.cpp:333-370): twiddles cos/sin in double, stored as float */
static int plan_init(fft_plan *pl, int nfft)
{
    pl->nfft = nfft;
    pl->tw = (cpx *)malloc(sizeof(cpx) * (size_t)nfft);
    if (!pl->tw) return KWSO_ERR_OOM;
    for (int i = 0; i < nfft; ++i) {
        const double pi = 3.141592653589793238462643383279502884197169399375105820974944;
        double phase = -2 * pi * i / nfft;
        pl->tw[i].r = (float)cos(phase);
        pl->tw[i].i = (float)sin(phase);
    }
    plan_factor(nfft, pl->factors);
    return KWSO_OK;
}

static inline cpx cmul(cpx a, cpx b)
{
    cpx m;
    float rr = a.r * b.r, ii = a.i * b.i, ri = a.r * b.i, ir = a.i * b.r;
    m.r = rr - ii;
    m.i = ri + ir;
    return m;
}
static inline cpx cadd(cpx a, cpx b) { cpx c = { a.r + b.r, a.i + b.i }; return c; }
static inline cpx csub(cpx a, cpx b) { cpx c = { a.r - b.r, a.i - b.i }; return c; }

static void bfly2(cpx *F, size_t fstride, const fft_plan *pl, int m)          /* kiss_fft.cpp:15-36 */
{
    for (int k = 0; k < m; k++) {
        cpx t = cmul(F[k + m], pl->tw[(size_t)k * fstride]);
        F[k + m] = csub(F[k], t);
        F[k] = cadd(F[k], t);
    }
}

static void bfly4(cpx *F, size_t fstride, const fft_plan *pl, int m)          /* kiss_fft.cpp:38-84 */
{
    for (int k = 0; k < m; k++) {
        cpx s0 = cmul(F[k + m], pl->tw[(size_t)k * fstride]);
        cpx s1 = cmul(F[k + 2 * m], pl->tw[(size_t)k * fstride * 2]);
        cpx s2 = cmul(F[k + 3 * m], pl->tw[(size_t)k * fstride * 3]);
        cpx s5 = csub(F[k], s1);
        F[k] = cadd(F[k], s1);
        cpx s3 = cadd(s0, s2);
        cpx s4 = csub(s0, s2);
        F[k + 2 * m] = csub(F[k], s3);
        F[k] = cadd(F[k], s3);
        F[k + m].r = s5.r + s4.i;
        F[k + m].i = s5.i - s4.r;
        F[k + 3 * m].r = s5.r - s4.i;
        F[k + 3 * m].i = s5.i + s4.r;
    }
}

static void bfly3(cpx *F, size_t fstride, const fft_plan *pl, int m)          /* kiss_fft.cpp:86-129 */
{
    const cpx epi3 = pl->tw[fstride * (size_t)m];
    for (int k = 0; k < m; k++) {
        cpx s1 = cmul(F[k + m], pl->tw[(size_t)k * fstride]);
        cpx s2 = cmul(F[k + 2 * m], pl->tw[(size_t)k * fstride * 2]);
        cpx s3 = cadd(s1, s2);
        cpx s0 = csub(s1, s2);
        F[k + m].r = F[k].r - s3.r * 0.5f;     /* HALF_OF(x) = x*.5 : exact */
        F[k + m].i = F[k].i - s3.i * 0.5f;
        s0.r *= epi3.i;
        s0.i *= epi3.i;
        F[k] = cadd(F[k], s3);
        F[k + 2 * m].r = F[k + m].r + s0.i;
        F[k + 2 * m].i = F[k + m].i - s0.r;
        F[k + m].r -= s0.i;
        F[k + m].i += s0.r;
    }
}

static void bfly5(cpx *F, size_t fstride, const fft_plan *pl, int m)          /* kiss_fft.cpp:131-192 */
{
    const cpx ya = pl->tw[fstride * (size_t)m];
    const cpx yb = pl->tw[fstride * 2 * (size_t)m];
    for (int u = 0; u < m; u++) {
        cpx *F0 = F + u, *F1 = F0 + m, *F2 = F0 + 2 * m, *F3 = F0 + 3 * m, *F4 = F0 + 4 * m;
        cpx s0 = *F0;
        cpx s1 = cmul(*F1, pl->tw[(size_t)u * fstride]);
        cpx s2 = cmul(*F2, pl->tw[2 * (size_t)u * fstride]);
        cpx s3 = cmul(*F3, pl->tw[3 * (size_t)u * fstride]);
        cpx s4 = cmul(*F4, pl->tw[4 * (size_t)u * fstride]);
        cpx s7 = cadd(s1, s4), s10 = csub(s1, s4), s8 = cadd(s2, s3), s9 = csub(s2, s3);
        float t;
        t = s7.r + s8.r; F0->r = F0->r + t;
        t = s7.i + s8.i; F0->i = F0->i + t;
        cpx s5, s6, s11, s12;
        float a, b;
        a = s7.r * ya.r; b = s8.r * yb.r; s5.r = (s0.r + a) + b;
        a = s7.i * ya.r; b = s8.i * yb.r; s5.i = (s0.i + a) + b;
        a = s10.i * ya.i; b = s9.i * yb.i; s6.r = a + b;
        a = s10.r * ya.i; b = s9.r * yb.i; s6.i = (-a) - b;
        *F1 = csub(s5, s6);
        *F4 = cadd(s5, s6);
        a = s7.r * yb.r; b = s8.r * ya.r; s11.r = (s0.r + a) + b;
        a = s7.i * yb.r; b = s8.i * ya.r; s11.i = (s0.i + a) + b;
        a = s10.i * yb.i; b = s9.i * ya.i; s12.r = (-a) + b;
        a = s10.r * yb.i; b = s9.r * ya.i; s12.i = a - b;
        *F2 = cadd(s11, s12);
        *F3 = csub(s11, s12);
    }
}

/* kf_work (kiss_fft.cpp:232-296): recursive decimation in time */
static int fft_work(cpx *Fout, const cpx *f, size_t fstride, const int *factors, const fft_plan *pl)
{
    const int p = factors[0], m = factors[1];
    if (m == 1) {
        for (int k = 0; k < p; k++) Fout[k] = f[(size_t)k * fstride];
    } else {
        for (int k = 0; k < p; k++) {
            int rc = fft_work(Fout + (size_t)k * m, f + (size_t)k * fstride, fstride * (size_t)p, factors + 2, pl);
            if (rc) return rc;
        }
    }
    switch (p) {
    case 2: bfly2(Fout, fstride, pl, m); break;
    case 3: bfly3(Fout, fstride, pl, m); break;
    case 4: bfly4(Fout, fstride, pl, m); break;
    case 5: bfly5(Fout, fstride, pl, m); break;
    default: return KWSO_ERR_UNSUPPORTED;   /* kf_bfly_generic not restated */
    }
    return KWSO_OK;
}

/* kiss_fftr_alloc + kiss_fftr (kiss_fftr.cpp:21-120) */
int kwso_rfft_complex(const float *in, int nfft, float *out_ri)
{
    if (nfft & 1) return KWSO_ERR_PARAM;
    const int ncfft = nfft >> 1;
    fft_plan pl;
    int rc = plan_init(&pl, ncfft);
    if (rc) return rc;
    cpx *tmp = (cpx *)malloc(sizeof(cpx) * (size_t)ncfft);
    cpx *st = (cpx *)malloc(sizeof(cpx) * (size_t)(ncfft / 2 + 1));
    if (!tmp || !st) { free(tmp); free(st); free(pl.tw); return KWSO_ERR_OOM; }
    for (int i = 0; i < ncfft / 2; ++i) {
        double phase = -3.14159265358979323846264338327 * ((double)(i + 1) / ncfft + .5);
        st[i].r = (float)cos(phase);
        st[i].i = (float)sin(phase);
    }
    rc = fft_work(tmp, (const cpx *)in, 1, pl.factors, &pl);
    if (rc == KWSO_OK) {
        cpx *freq = (cpx *)out_ri;
        float tdc_r = tmp[0].r, tdc_i = tmp[0].i;
        freq[0].r = tdc_r + tdc_i;
        freq[ncfft].r = tdc_r - tdc_i;
        freq[ncfft].i = freq[0].i = 0;
        for (int k = 1; k <= ncfft / 2; ++k) {
            cpx fpk = tmp[k];
            cpx fpnk = { tmp[ncfft - k].r, -tmp[ncfft - k].i };
            cpx f1k = cadd(fpk, fpnk);
            cpx f2k = csub(fpk, fpnk);
            cpx tw = cmul(f2k, st[k - 1]);
            float a;
            a = f1k.r + tw.r; freq[k].r = a * 0.5f;
            a = f1k.i + tw.i; freq[k].i = a * 0.5f;
            a = f1k.r - tw.r; freq[ncfft - k].r = a * 0.5f;
            a = tw.i - f1k.i; freq[ncfft - k].i = a * 0.5f;
        }
    }
    free(tmp); free(st); free(pl.tw);
    return rc;
}

/* numpy::rfft float-out + software_rfft (numpy.hpp:1091-1156, 1388-1417) and
 * processing::power_spectrum (processing.hpp:295-312) */
int kwso_power_spectrum(const float *frame, size_t frame_size, float *out, int fft_length)
{
    const int nout = fft_length / 2 + 1;
    size_t src = frame_size > (size_t)fft_length ? (size_t)fft_length : frame_size;
    float *in = (float *)calloc((size_t)fft_length, sizeof(float));
    float *spec = (float *)malloc(sizeof(float) * 2 * (size_t)nout);
    if (!in || !spec) { free(in); free(spec); return KWSO_ERR_OOM; }
    memcpy(in, frame, src * sizeof(float));
    int rc = kwso_rfft_complex(in, fft_length, spec);
    if (rc == KWSO_OK) {
        for (int ix = 0; ix < nout; ix++) {
            double re = (double)spec[2 * ix], im = (double)spec[2 * ix + 1];
            /* sqrt(pow(r,2) + pow(i,2)) in double, rounded to float */
            float mag = (float)sqrt(re * re + im * im);
            float sq = mag * mag;
            /* (1.0 / static_cast<float>(fft_points)) * (x*x): double product, rounded to float */
            out[ix] = (float)((1.0 / (double)(float)fft_length) * (double)sq);
        }
    }
    free(in); free(spec);
    return rc;
}

/* ====================================================================== */
/*  mfe                                             feature.hpp:193-318    */
/* ====================================================================== */
static int mfe_ex(const int16_t *pcm, size_t n, size_t n_claimed, const kwso_mfcc_config *c, float *features, float *energies,
                  const float *end_of_signal);
int kwso_mfe(const int16_t *pcm, size_t n, const kwso_mfcc_config *c, float *features, float *energies)
{
    return mfe_ex(pcm, n, n, c, features, energies, NULL);
}

/* n = samples really available in pcm, n_claimed = signal->total_length the frame count is derived from
 * (continuous mode claims one extra frame length, ei_run_dsp.h:319-325) */
static int mfe_ex(const int16_t *pcm, size_t n, size_t n_claimed, const kwso_mfcc_config *c, float *features, float *energies,
                  const float *end_of_signal)
{
    const int nf = kwso_num_frames(n_claimed, c);
    const int flen = kwso_frame_length_samples(c);
    const int stride = (int)roundf((float)c->sampling_frequency * c->frame_stride);
    const int coeff = c->fft_length / 2 + 1;
    const int M = c->num_filters;
    if (nf < 1) return KWSO_ERR_SIZE;
    float *fb = (float *)malloc(sizeof(float) * (size_t)coeff * M);
    float *frame = (float *)malloc(sizeof(float) * (size_t)flen);
    float *ps = (float *)malloc(sizeof(float) * (size_t)coeff);
    if (!fb || !frame || !ps) { free(fb); free(frame); free(ps); return KWSO_ERR_OOM; }
    int rc = kwso_filterbanks(c, fb);
    memset(features, 0, sizeof(float) * (size_t)nf * M);
    for (int ix = 0; ix < nf && rc == KWSO_OK; ix++) {
        size_t off = (size_t)ix * (size_t)stride;
        rc = preemphasis_ex(pcm, n, c->pre_cof, c->pre_shift, off, (size_t)flen, frame, end_of_signal);
        if (rc) break;
        rc = kwso_power_spectrum(frame, (size_t)flen, ps, c->fft_length);
        if (rc) break;
        float energy = 0.0f;                               /* numpy::sum, numpy.hpp:88-94 */
        for (int k = 0; k < coeff; k++) energy += ps[k];
        if (energy == 0) energy = FLT_EPSILON;
        energies[ix] = energy;
        for (int j = 0; j < M; j++) {                      /* numpy::dot_by_row, numpy.hpp:183-211 */
            float acc = features[(size_t)ix * M + j];
            for (int k = 0; k < coeff; k++) {
                float prod = ps[k] * fb[(size_t)k * M + j];
                acc += prod;
            }
            features[(size_t)ix * M + j] = acc;
        }
    }
    if (rc == KWSO_OK)                                      /* functions::zero_handling */
        for (size_t i = 0; i < (size_t)nf * M; i++) if (features[i] == 0) features[i] = FLT_EPSILON;
    free(fb); free(frame); free(ps);
    return rc;
}

/* ====================================================================== */
/*  dct2 (ortho)             numpy.hpp:378-401, dct/fast-dct-fft.cpp:37-80  */
/*  Only outputs 0..n/2 come from the transform; the rest keep the input    */
/*  value (times the scale factors) -- reference behaviour, SURVEY section 0 */
/* ====================================================================== */
int kwso_dct2_ortho(float *v, int n)
{
    if (n == 0) return KWSO_OK;
    float *in = (float *)calloc((size_t)n, sizeof(float));
    float *spec = (float *)calloc(2 * (size_t)(n / 2 + 1), sizeof(float));
    if (!in || !spec) { free(in); free(spec); return KWSO_ERR_OOM; }
    int half = n / 2;
    for (int i = 0; i < half; i++) {
        in[i] = v[i * 2];
        in[n - 1 - i] = v[i * 2 + 1];
    }
    if (n % 2 == 1) in[half] = v[n - 1];
    int rc = kwso_rfft_complex(in, n, spec);
    if (rc == KWSO_OK) {
        for (int i = 0; i < n / 2 + 1; i++) {
            float temp = (float)((double)i * M_PI / (double)(n * 2));
            /* fft.r * cos(temp) + fft.i * sin(temp): float overloads of cos/sin */
            float a = spec[2 * i] * cosf(temp);
            float b = spec[2 * i + 1] * sinf(temp);
            v[i] = a + b;
        }
        for (int ix = 0; ix < n; ix++) v[ix] *= 2;
        v[0] = v[0] * sqrtf(1.0f / (float)(4 * n));
        for (int ix = 1; ix < n; ix++) v[ix] = v[ix] * sqrtf(1.0f / (float)(2 * n));
    }
    free(in); free(spec);
    return rc;
}

/* ====================================================================== */
/*  mfcc (no CMVN)                                  feature.hpp:370-439    */
/* ====================================================================== */
static int mfcc_ex(const int16_t *pcm, size_t n, size_t n_claimed, const kwso_mfcc_config *c, float *out, const float *end_of_signal);
int kwso_mfcc_nocmvn(const int16_t *pcm, size_t n, const kwso_mfcc_config *c, float *out)
{
    return mfcc_ex(pcm, n, n, c, out, NULL);
}

static int mfcc_ex(const int16_t *pcm, size_t n, size_t n_claimed, const kwso_mfcc_config *c, float *out, const float *end_of_signal)
{
    const int nf = kwso_num_frames(n_claimed, c);
    const int M = c->num_filters, K = c->num_cepstral;
    if (nf < 1) return KWSO_ERR_SIZE;
    float *mel = (float *)malloc(sizeof(float) * (size_t)nf * M);
    float *en = (float *)malloc(sizeof(float) * (size_t)nf);
    if (!mel || !en) { free(mel); free(en); return KWSO_ERR_OOM; }
    int rc = mfe_ex(pcm, n, n_claimed, c, mel, en, end_of_signal);
    if (rc == KWSO_OK) {
        for (size_t i = 0; i < (size_t)nf * M; i++) mel[i] = kwso_log(mel[i]);
        for (int r = 0; r < nf && rc == KWSO_OK; r++) rc = kwso_dct2_ortho(mel + (size_t)r * M, M);
        for (int r = 0; r < nf; r++) mel[(size_t)r * M] = kwso_log(en[r]);
        for (int r = 0; r < nf; r++)
            for (int i = 0; i < K; i++) out[(size_t)r * K + i] = mel[(size_t)r * M + i];
    }
    free(mel); free(en);
    return rc;
}

/* ====================================================================== */
/*  cmvnw      processing.hpp:326-389, numpy.hpp:479-541, 746-836          */
/* ====================================================================== */
/* source row of padded row p, replaying numpy::pad_1d_symmetric's walk */
static void pad_index_map(int rows, int pad_before, int pad_after, int *map)
{
    int idx = 0, up = 1;
    for (int ix = pad_before - 1; ix >= 0; ix--) {
        map[ix] = idx;
        if (idx == 0 && !up) up = 1;
        else if (idx == rows - 1 && up) up = 0;
        else if (up) idx++;
        else idx--;
    }
    for (int r = 0; r < rows; r++) map[pad_before + r] = r;
    idx = rows - 1; up = 0;
    for (int ix = 0; ix < pad_after; ix++) {
        map[ix + pad_before + rows] = idx;
        if (idx == 0 && !up) up = 1;
        else if (idx == rows - 1 && up) up = 0;
        else if (up) idx++;
        else idx--;
    }
}

int kwso_cmvnw(float *m, int rows, int cols, int win_size, int variance_normalization)
{
    if (rows == 0) return -1009;  /* EIDSP_INPUT_MATRIX_EMPTY */
    const int pad = (int)(uint16_t)((win_size - 1) / 2);
    const int prow = rows + 2 * pad;
    int *map = (int *)malloc(sizeof(int) * (size_t)prow);
    float *padm = (float *)malloc(sizeof(float) * (size_t)prow * cols);
    if (!map || !padm) { free(map); free(padm); return KWSO_ERR_OOM; }
    pad_index_map(rows, pad, pad, map);
    for (int p = 0; p < prow; p++) memcpy(padm + (size_t)p * cols, m + (size_t)map[p] * cols, sizeof(float) * cols);
    for (int r = 0; r < rows; r++) {
        for (int col = 0; col < cols; col++) {
            float sum = 0.0f;
            for (int j = 0; j < win_size; j++) sum += padm[(size_t)(r + j) * cols + col];
            float mean = sum / (float)(uint32_t)win_size;
            float x = m[(size_t)r * cols + col];
            if (variance_normalization) {
                float std = 0.0f;
                for (int j = 0; j < win_size; j++) {
                    float d = padm[(size_t)(r + j) * cols + col] - mean;
                    /* std += pow(d, 2): double square, float accumulator */
                    std = (float)((double)std + (double)d * (double)d);
                }
                float sd = sqrtf(std / (float)(uint32_t)win_size);
                m[(size_t)r * cols + col] = (x - mean) / (sd + FLT_EPSILON);
            } else {
                m[(size_t)r * cols + col] = x - mean;
            }
        }
    }
    free(map); free(padm);
    return KWSO_OK;
}

/* ====================================================================== */
/*  extract_mfcc_features                      ei_run_dsp.h:256-308        */
/* ====================================================================== */
int kwso_extract_mfcc(const int16_t *pcm, size_t n, const kwso_mfcc_config *c, float *features)
{
    int rc = kwso_mfcc_nocmvn(pcm, n, c, features);
    if (rc) return rc;
    return kwso_cmvnw(features, kwso_num_frames(n, c), c->num_cepstral, c->win_size, 1);
}

/* ====================================================================== */
/*  MFE block of the L432 SDK copy             (L432 = .../nucleo-l432-keyword-spotting/keyword-spotting-02-v3/edge-impulse-sdk) */
/*  extract_mfe_features                       L432 classifier/ei_run_dsp.h:369-418 */
/*     = speechpy::feature::mfe (feature.hpp:193-318, the same text in both SDK copies)                                    */
/*       -> processing::cmvnw(win_size, variance_normalization = false, scale = true)   L432 dsp/speechpy/processing.hpp:327-399 */
/*       -> numpy::normalize                   L432 dsp/numpy.hpp:1391-1429                                                 */
/*  Pinned against oracle/_ref/libei_ref_l432dsp.so (the two L432 headers compiled in place, ref_l432_dsp.cpp).              */
/* ====================================================================== */
int kwso_normalize(float *m, size_t n)
{
    float mn = FLT_MAX, mx = -FLT_MAX;                       /* numpy::min / max, numpy.hpp:842-905: strict comparisons */
    for (size_t i = 0; i < n; i++) if (m[i] < mn) mn = m[i];
    for (size_t i = 0; i < n; i++) if (m[i] > mx) mx = m[i];
    const float row_scale = 1.0f / (mx - mn);                /* numpy.hpp:1416 */
    for (size_t i = 0; i < n; i++) m[i] -= mn;               /* numpy::subtract, numpy.hpp:641-646 */
    if (row_scale != 1.0f)                                   /* numpy::scale returns early for 1.0f, numpy.hpp:548-549 */
        for (size_t i = 0; i < n; i++) m[i] *= row_scale;
    return KWSO_OK;
}

int kwso_cmvnw_scale(float *m, int rows, int cols, int win_size, int variance_normalization, int scale)
{
    int rc = kwso_cmvnw(m, rows, cols, win_size, variance_normalization);
    if (rc) return rc;
    return scale ? kwso_normalize(m, (size_t)rows * (size_t)cols) : KWSO_OK;
}

/* features: [frames][num_filters] (num_cepstral of the config is not used) */
int kwso_extract_mfe(const int16_t *pcm, size_t n, const kwso_mfcc_config *c, float *features)
{
    const int frames = kwso_num_frames(n, c);
    if (frames <= 0) return -1002;
    float *energies = (float *)malloc(sizeof(float) * (size_t)frames);
    if (!energies) return KWSO_ERR_OOM;
    int rc = kwso_mfe(pcm, n, c, features, energies);
    free(energies);
    if (rc) return rc;
    return kwso_cmvnw_scale(features, frames, c->num_filters, c->win_size, 0, 1);
}

/* ====================================================================== */
/*  fixed-point helpers                                                    */
/* ====================================================================== */
int32_t kwso_srdhm(int32_t a, int32_t b)                 /* gemmlowp fixedpoint.h:329-339 */
{
    int overflow = (a == b) && (a == INT32_MIN);
    int64_t ab = (int64_t)a * (int64_t)b;
    int32_t nudge = ab >= 0 ? (1 << 30) : (1 - (1 << 30));
    int32_t hi = (int32_t)((ab + nudge) / (1ll << 31));   /* truncating division, not a shift */
    return overflow ? INT32_MAX : hi;
}

int32_t kwso_rdivpot(int32_t x, int exponent)            /* fixedpoint.h:357-368 */
{
    const int32_t mask = (int32_t)((1ll << exponent) - 1);
    const int32_t remainder = x & mask;
    const int32_t threshold = (mask >> 1) + (x < 0 ? 1 : 0);
    return (x >> exponent) + (remainder > threshold ? 1 : 0);
}

int32_t kwso_mbqm(int32_t x, int32_t mult, int shift)    /* TFL/kernels/internal/common.h:153-162 */
{
    int left_shift = shift > 0 ? shift : 0;
    int right_shift = shift > 0 ? 0 : -shift;
    return kwso_rdivpot(kwso_srdhm((int32_t)((uint32_t)x * (1u << left_shift)), mult), right_shift);
}

static int32_t mbqm_smaller_than_one(int32_t x, int32_t mult, int left_shift)   /* common.h:138-144 */
{
    return kwso_rdivpot(kwso_srdhm(x, mult), -left_shift);
}

void kwso_quantize_multiplier(double m, int32_t *q, int *shift)   /* quantization_util.cc:53-91 */
{
    if (m == 0.) { *q = 0; *shift = 0; return; }
    const double f = frexp(m, shift);
    int64_t q_fixed = (int64_t)round(f * (double)(1ll << 31));
    if (q_fixed == (1ll << 31)) { q_fixed /= 2; ++*shift; }
    if (*shift < -31) { *shift = 0; q_fixed = 0; }
    *q = (int32_t)q_fixed;
}

static int32_t sat_shl(int32_t x, int e)   /* ImplSaturatingRoundingMultiplyByPOT<e>0>, fixedpoint.h:385-406 */
{
    const int32_t threshold = (int32_t)((1u << (31 - e)) - 1);
    if (x > threshold) return INT32_MAX;
    if (x < -threshold) return INT32_MIN;
    int64_t w = (int64_t)x * (1 << e);
    return w < INT32_MIN ? INT32_MIN : (w > INT32_MAX ? INT32_MAX : (int32_t)w);
}

static int32_t wrap_add(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static int32_t wrap_sub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }

/* exp on [-1/4, 0) in Q0.31                                   fixedpoint.h:721-742 */
static int32_t exp_interval(int32_t a)
{
    const int32_t constant_term = 1895147668, one_third = 715827883;
    int32_t x = wrap_add(a, 1 << 28);
    int32_t x2 = kwso_srdhm(x, x);
    int32_t x3 = kwso_srdhm(x2, x);
    int32_t x4 = kwso_srdhm(x2, x2);
    int32_t x4_4 = kwso_rdivpot(x4, 2);
    int32_t t = kwso_rdivpot(wrap_add(kwso_srdhm(wrap_add(x4_4, x3), one_third), x2), 1);
    return wrap_add(constant_term, kwso_srdhm(constant_term, wrap_add(x, t)));
}

/* exp_on_negative_values<int32, 5>                            fixedpoint.h:746-790 */
int32_t kwso_exp_on_negative_values_q5_26(int32_t a)
{
    const int kFractionalBits = 26;
    const int32_t one_quarter = 1 << (kFractionalBits - 2);
    const int32_t mask = one_quarter - 1;
    int32_t amq = wrap_sub(a & mask, one_quarter);
    int32_t result = exp_interval(sat_shl(amq, 5));
    int32_t remainder = wrap_sub(amq, a);
    static const int32_t mult[7] = { 1672461947, 1302514674, 790015084, 290630308, 39332535, 720401, 242 };
    for (int e = -2; e <= 4; e++) {
        int shift_amount = kFractionalBits + e;
        if (remainder & (1 << shift_amount)) result = kwso_srdhm(result, mult[e + 2]);
    }
    return a == 0 ? INT32_MAX : result;
}

/* one_over_one_plus_x_for_x_in_0_1                            fixedpoint.h:842-862 */
int32_t kwso_one_over_one_plus_x(int32_t a)
{
    int64_t sum = (int64_t)a + (int64_t)INT32_MAX;           /* RoundingHalfSum, fixedpoint.h:236-242 */
    int64_t sign = sum >= 0 ? 1 : -1;
    int32_t half_den = (int32_t)((sum + sign) / 2);
    const int32_t c48_17 = 1515870810, cneg32_17 = -1010580540;
    int32_t x = wrap_add(c48_17, kwso_srdhm(half_den, cneg32_17));
    for (int i = 0; i < 3; i++) {
        int32_t hdx = kwso_srdhm(half_den, x);
        int32_t one_minus = wrap_sub(1 << 29, hdx);
        x = wrap_add(x, sat_shl(kwso_srdhm(x, one_minus), 2));
    }
    return sat_shl(x, 1);
}

/* ====================================================================== */
/*  model blob                                                             */
/* ====================================================================== */
enum { OP_RESHAPE = 0, OP_CONV_2D, OP_ADD, OP_MAX_POOL_2D, OP_FULLY_CONNECTED, OP_SOFTMAX, OP_DEPTHWISE_CONV_2D };
enum { TYPE_F32 = 1, TYPE_I32 = 2, TYPE_I8 = 9 };

typedef struct {
    uint32_t type, ndims;
    int32_t dims[8];
    uint32_t is_const, nquant;
    float *scale;
    int32_t *zero;
    int32_t qdim;
    uint32_t nbytes;
    uint8_t *data;         /* constants */
    size_t tap_offset;     /* position in the flat tap buffer */
} o_tensor;

typedef struct {
    uint32_t op, n_in, n_out;
    int32_t in[4], out[2];
    int32_t p[8];
    float beta;
} o_node;

struct kwso_model {
    uint32_t n_tensors, n_nodes, n_labels, t_in, t_out;
    uint32_t raw_sample_count, frequency, nn_input_frame_size;
    kwso_mfcc_config dsp;
    int dsp_block;                    /* 0: extract_mfcc_features, 1: extract_mfe_features (L432 SDK copy) */
    char **labels;
    o_tensor *t;
    o_node *n;
    size_t tap_bytes;
};

typedef struct { const uint8_t *p, *end; int bad; } rd;
static uint32_t rd_u32(rd *r) { uint32_t v = 0; if (r->p + 4 > r->end) { r->bad = 1; return 0; } memcpy(&v, r->p, 4); r->p += 4; return v; }
static int32_t rd_i32(rd *r) { return (int32_t)rd_u32(r); }
static float rd_f32(rd *r) { uint32_t u = rd_u32(r); float f; memcpy(&f, &u, 4); return f; }
static const uint8_t *rd_bytes(rd *r, size_t n) { size_t pn = (n + 3) & ~(size_t)3; if (r->p + pn > r->end) { r->bad = 1; return NULL; } const uint8_t *q = r->p; r->p += pn; return q; }

void kwso_model_free(kwso_model *m)
{
    if (!m) return;
    if (m->labels) for (uint32_t i = 0; i < m->n_labels; i++) free(m->labels[i]);
    free(m->labels);
    if (m->t) for (uint32_t i = 0; i < m->n_tensors; i++) { free(m->t[i].scale); free(m->t[i].zero); free(m->t[i].data); }
    free(m->t); free(m->n); free(m);
}

kwso_model *kwso_model_load(const void *blob, size_t nbytes)
{
    rd r = { (const uint8_t *)blob, (const uint8_t *)blob + nbytes, 0 };
    if (nbytes < 8 || memcmp(blob, "KWSM", 4) != 0) return NULL;
    r.p += 4;
    const uint32_t version = rd_u32(&r);
    if (version != 1 && version != 2) return NULL;
    kwso_model *m = (kwso_model *)calloc(1, sizeof(*m));
    if (!m) return NULL;
    m->n_tensors = rd_u32(&r); m->n_nodes = rd_u32(&r); m->n_labels = rd_u32(&r);
    m->t_in = rd_u32(&r); m->t_out = rd_u32(&r);
    m->raw_sample_count = rd_u32(&r); m->frequency = rd_u32(&r); m->nn_input_frame_size = rd_u32(&r);
    (void)rd_i32(&r); /* axes */
    m->dsp.num_cepstral = rd_i32(&r); m->dsp.num_filters = rd_i32(&r); m->dsp.fft_length = rd_i32(&r);
    m->dsp.win_size = rd_i32(&r); m->dsp.low_frequency = rd_i32(&r); m->dsp.high_frequency = rd_i32(&r);
    m->dsp.pre_shift = rd_i32(&r);
    m->dsp.frame_length = rd_f32(&r); m->dsp.frame_stride = rd_f32(&r); m->dsp.pre_cof = rd_f32(&r);
    {
        /* version 2: one more i32 -- bits 0..7 the DSP block (0: extract_mfcc_features, 1: extract_mfe_features of the L432 SDK copy),
           bit 8 EIDSP_QUANTIZE_FILTERBANK */
        const int v = version == 2 ? rd_i32(&r) : 0;
        m->dsp_block = v & 0xff;
        m->dsp.quantize_filterbank = (v >> 8) & 1;
        if (v & ~0x1ff) r.bad = 1;
    }
    m->dsp.sampling_frequency = (int)m->frequency;
    if (r.bad || m->n_tensors > 4096 || m->n_nodes > 4096 || m->n_labels > 1024) { kwso_model_free(m); return NULL; }
    m->labels = (char **)calloc(m->n_labels ? m->n_labels : 1, sizeof(char *));
    for (uint32_t i = 0; i < m->n_labels; i++) {
        uint32_t len = rd_u32(&r);
        const uint8_t *b = rd_bytes(&r, len);
        if (!b) break;
        m->labels[i] = (char *)calloc(len + 1, 1);
        memcpy(m->labels[i], b, len);
    }
    m->t = (o_tensor *)calloc(m->n_tensors ? m->n_tensors : 1, sizeof(o_tensor));
    for (uint32_t i = 0; i < m->n_tensors && !r.bad; i++) {
        o_tensor *t = &m->t[i];
        t->type = rd_u32(&r); t->ndims = rd_u32(&r);
        if (t->ndims > 8) { r.bad = 1; break; }
        for (uint32_t d = 0; d < t->ndims; d++) t->dims[d] = rd_i32(&r);
        t->is_const = rd_u32(&r); t->nquant = rd_u32(&r);
        if (t->nquant > 65536) { r.bad = 1; break; }
        t->scale = (float *)calloc(t->nquant ? t->nquant : 1, sizeof(float));
        t->zero = (int32_t *)calloc(t->nquant ? t->nquant : 1, sizeof(int32_t));
        for (uint32_t q = 0; q < t->nquant; q++) t->scale[q] = rd_f32(&r);
        for (uint32_t q = 0; q < t->nquant; q++) t->zero[q] = rd_i32(&r);
        t->qdim = rd_i32(&r); t->nbytes = rd_u32(&r);
        if (t->is_const) {
            const uint8_t *b = rd_bytes(&r, t->nbytes);
            if (!b) break;
            t->data = (uint8_t *)malloc(t->nbytes ? t->nbytes : 1);
            memcpy(t->data, b, t->nbytes);
        }
        t->tap_offset = m->tap_bytes;
        m->tap_bytes += t->nbytes;
    }
    m->n = (o_node *)calloc(m->n_nodes ? m->n_nodes : 1, sizeof(o_node));
    for (uint32_t i = 0; i < m->n_nodes && !r.bad; i++) {
        o_node *nd = &m->n[i];
        nd->op = rd_u32(&r); nd->n_in = rd_u32(&r);
        if (nd->n_in > 4) { r.bad = 1; break; }
        for (uint32_t k = 0; k < nd->n_in; k++) nd->in[k] = rd_i32(&r);
        nd->n_out = rd_u32(&r);
        if (nd->n_out > 2) { r.bad = 1; break; }
        for (uint32_t k = 0; k < nd->n_out; k++) nd->out[k] = rd_i32(&r);
        for (int k = 0; k < 8; k++) nd->p[k] = rd_i32(&r);
        nd->beta = rd_f32(&r);
    }
    if (r.bad) { kwso_model_free(m); return NULL; }
    return m;
}

int kwso_model_label_count(const kwso_model *m) { return (int)m->n_labels; }
int kwso_model_dsp_block(const kwso_model *m) { return m->dsp_block; }
/* columns of the feature matrix: cepstra (MFCC block) or mel filters (MFE block) */
static int feature_cols(const kwso_model *m) { return m->dsp_block == 1 ? m->dsp.num_filters : m->dsp.num_cepstral; }
const char *kwso_model_label(const kwso_model *m, int i) { return m->labels[i]; }
int kwso_model_feature_count(const kwso_model *m) { return (int)m->nn_input_frame_size; }
int kwso_model_raw_sample_count(const kwso_model *m) { return (int)m->raw_sample_count; }
void kwso_model_mfcc_config(const kwso_model *m, kwso_mfcc_config *c) { *c = m->dsp; }
int kwso_model_tensor_count(const kwso_model *m) { return (int)m->n_tensors; }
int kwso_model_tensor_bytes(const kwso_model *m, int id) { return (int)m->t[id].nbytes; }

/* ====================================================================== */
/*  NN ops (TFLite-Micro reference kernels, int8)                          */
/* ====================================================================== */
static int dim4(const o_tensor *t, int i)  /* RuntimeShape::ExtendedShape(4, ...) : left-pad with 1 */
{
    int pad = 4 - (int)t->ndims;
    return i < pad ? 1 : t->dims[i - pad];
}

/* CalculateActivationRangeQuantized                        kernel_util_lite.cc:174-226 */
static void act_range(int activation, const o_tensor *out, int32_t *amin, int32_t *amax)
{
    const int32_t qmin = -128, qmax = 127;
    const float scale = out->scale[0];
    const int32_t zp = out->zero[0];
    *amin = qmin; *amax = qmax;
    if (activation == 1) {
        int32_t q = zp + (int32_t)roundf(0.0f / scale);
        *amin = q > qmin ? q : qmin;
    } else if (activation == 3) {
        int32_t q0 = zp + (int32_t)roundf(0.0f / scale), q6 = zp + (int32_t)roundf(6.0f / scale);
        *amin = q0 > qmin ? q0 : qmin; *amax = q6 < qmax ? q6 : qmax;
    } else if (activation == 2) {
        int32_t qa = zp + (int32_t)roundf(-1.0f / scale), qb = zp + (int32_t)roundf(1.0f / scale);
        *amin = qa > qmin ? qa : qmin; *amax = qb < qmax ? qb : qmax;
    }
}

static int out_size(int padding, int image, int filter, int stride, int dilation)   /* padding.h:44-55 */
{
    int eff = (filter - 1) * dilation + 1;
    if (padding == 1) return (image + stride - 1) / stride;
    if (padding == 2) return (image + stride - eff) / stride;
    return 0;
}
static int pad_amount(int stride, int dilation, int in_size, int filter, int out)     /* padding.h:32-41 */
{
    int eff = (filter - 1) * dilation + 1;
    int total = (out - 1) * stride + eff - in_size;
    total = total > 0 ? total : 0;
    return total / 2;
}

/* CONV_2D: conv.cc:540-637 (Prepare), kernel_util_lite.cc:47-120, integer_ops/conv.h:24-123 */
static int op_conv(const kwso_model *m, const o_node *nd, int8_t **buf, int depthwise)
{
    const o_tensor *in = &m->t[nd->in[0]], *flt = &m->t[nd->in[1]], *out = &m->t[nd->out[0]];
    const o_tensor *bias = nd->n_in > 2 && nd->in[2] >= 0 ? &m->t[nd->in[2]] : NULL;
    const int8_t *x = buf[nd->in[0]];
    const int8_t *w = (const int8_t *)flt->data;
    const int32_t *b = bias ? (const int32_t *)bias->data : NULL;
    int8_t *y = buf[nd->out[0]];
    const int padding = nd->p[0], stride_w = nd->p[1], stride_h = nd->p[2], activation = nd->p[3];
    const int dil_w = nd->p[4], dil_h = nd->p[5];
    const int batches = dim4(in, 0), in_h = dim4(in, 1), in_w = dim4(in, 2), in_d = dim4(in, 3);
    const int f_h = dim4(flt, 1), f_w = dim4(flt, 2);
    const int out_h = dim4(out, 1), out_w = dim4(out, 2), out_d = dim4(out, 3);
    int oh = out_size(padding, in_h, f_h, stride_h, dil_h), ow = out_size(padding, in_w, f_w, stride_w, dil_w);
    const int pad_h = pad_amount(stride_h, dil_h, in_h, f_h, oh), pad_w = pad_amount(stride_w, dil_w, in_w, f_w, ow);
    const int nch = depthwise ? dim4(flt, 3) : dim4(flt, 0);
    int32_t *mult = (int32_t *)malloc(sizeof(int32_t) * (size_t)nch);
    int *shift = (int *)malloc(sizeof(int) * (size_t)nch);
    if (!mult || !shift) { free(mult); free(shift); return -6; }
    const int per_channel = flt->nquant > 1;
    for (int i = 0; i < nch; i++) {
        const float scale = per_channel ? flt->scale[i] : flt->scale[0];
        const double eff = (double)in->scale[0] * (double)scale / (double)out->scale[0];
        kwso_quantize_multiplier(eff, &mult[i], &shift[i]);
    }
    int32_t amin, amax;
    act_range(activation, out, &amin, &amax);
    /* depthwise_conv.cc:618-620 (EvalQuantizedPerChannel): the int8 depthwise op clamps to the int8 range and IGNORES its
     * fused activation ("TODO(b/130439627): Use calculated value for clamping") -- kept, it is what the reference computes */
    if (depthwise) { amin = -128; amax = 127; }
    const int32_t input_offset = -in->zero[0], output_offset = out->zero[0];
    const int depth_mult = depthwise ? nd->p[6] : 1;
    for (int bt = 0; bt < batches; ++bt)
        for (int oy = 0; oy < out_h; ++oy)
            for (int ox = 0; ox < out_w; ++ox) {
                const int x0 = ox * stride_w - pad_w, y0 = oy * stride_h - pad_h;
                if (!depthwise) {
                    for (int oc = 0; oc < out_d; ++oc) {
                        int32_t acc = 0;
                        for (int fy = 0; fy < f_h; ++fy)
                            for (int fx = 0; fx < f_w; ++fx)
                                for (int ic = 0; ic < in_d; ++ic) {
                                    const int ix = x0 + dil_w * fx, iy = y0 + dil_h * fy;
                                    if (ix >= 0 && ix < in_w && iy >= 0 && iy < in_h) {
                                        int32_t iv = x[((bt * in_h + iy) * in_w + ix) * in_d + ic];
                                        int32_t fv = w[((oc * f_h + fy) * f_w + fx) * in_d + ic];
                                        acc += fv * (iv + input_offset);
                                    }
                                }
                        if (b) acc += b[oc];
                        acc = kwso_mbqm(acc, mult[oc], shift[oc]);
                        acc += output_offset;
                        acc = acc > amin ? acc : amin;
                        acc = acc < amax ? acc : amax;
                        y[((bt * out_h + oy) * out_w + ox) * out_d + oc] = (int8_t)acc;
                    }
                } else {   /* integer_ops/depthwise_conv.h:22-120 */
                    for (int ic = 0; ic < in_d; ++ic)
                        for (int mm = 0; mm < depth_mult; ++mm) {
                            const int oc = mm + ic * depth_mult;
                            int32_t acc = 0;
                            for (int fy = 0; fy < f_h; ++fy)
                                for (int fx = 0; fx < f_w; ++fx) {
                                    const int ix = x0 + dil_w * fx, iy = y0 + dil_h * fy;
                                    if (ix >= 0 && ix < in_w && iy >= 0 && iy < in_h) {
                                        int32_t iv = x[((bt * in_h + iy) * in_w + ix) * in_d + ic];
                                        int32_t fv = w[(fy * f_w + fx) * out_d + oc];
                                        acc += fv * (iv + input_offset);
                                    }
                                }
                            if (b) acc += b[oc];
                            acc = kwso_mbqm(acc, mult[oc], shift[oc]);
                            acc += output_offset;
                            acc = acc > amin ? acc : amin;
                            acc = acc < amax ? acc : amax;
                            y[((bt * out_h + oy) * out_w + ox) * out_d + oc] = (int8_t)acc;
                        }
                }
            }
    free(mult); free(shift);
    return 0;
}

/* ADD: add.cc:271-309 (CalculateOpData), integer_ops/add.h:30-138 */
static int op_add(const kwso_model *m, const o_node *nd, int8_t **buf)
{
    const o_tensor *t1 = &m->t[nd->in[0]], *t2 = &m->t[nd->in[1]], *out = &m->t[nd->out[0]];
    const int8_t *a = buf[nd->in[0]], *bq = buf[nd->in[1]];
    int8_t *y = buf[nd->out[0]];
    const int32_t off1 = -t1->zero[0], off2 = -t2->zero[0], off_out = out->zero[0];
    const int left_shift = 20;
    const float max_scale = t1->scale[0] > t2->scale[0] ? t1->scale[0] : t2->scale[0];
    const double twice_max = 2 * (double)max_scale;
    const double rm1 = (double)t1->scale[0] / twice_max, rm2 = (double)t2->scale[0] / twice_max;
    const double rmo = twice_max / ((double)(1 << left_shift) * (double)out->scale[0]);
    int32_t m1, m2, mo; int s1, s2, so;
    kwso_quantize_multiplier(rm1, &m1, &s1);
    kwso_quantize_multiplier(rm2, &m2, &s2);
    kwso_quantize_multiplier(rmo, &mo, &so);
    int32_t amin, amax;
    act_range(nd->p[0], out, &amin, &amax);
    /* broadcast over 4-D extended shapes (NdArrayDescsForElementwiseBroadcast) */
    int od[4], d1[4], d2[4];
    for (int i = 0; i < 4; i++) { od[i] = dim4(out, i); d1[i] = dim4(t1, i); d2[i] = dim4(t2, i); }
    for (int b0 = 0; b0 < od[0]; ++b0)
        for (int yy = 0; yy < od[1]; ++yy)
            for (int xx = 0; xx < od[2]; ++xx)
                for (int c = 0; c < od[3]; ++c) {
                    int i1 = (((d1[0] == 1 ? 0 : b0) * d1[1] + (d1[1] == 1 ? 0 : yy)) * d1[2] + (d1[2] == 1 ? 0 : xx)) * d1[3] + (d1[3] == 1 ? 0 : c);
                    int i2 = (((d2[0] == 1 ? 0 : b0) * d2[1] + (d2[1] == 1 ? 0 : yy)) * d2[2] + (d2[2] == 1 ? 0 : xx)) * d2[3] + (d2[3] == 1 ? 0 : c);
                    const int32_t v1 = off1 + a[i1], v2 = off2 + bq[i2];
                    const int32_t sv1 = v1 * (1 << left_shift), sv2 = v2 * (1 << left_shift);
                    const int32_t q1 = mbqm_smaller_than_one(sv1, m1, s1);
                    const int32_t q2 = mbqm_smaller_than_one(sv2, m2, s2);
                    const int32_t raw = q1 + q2;
                    int32_t o = mbqm_smaller_than_one(raw, mo, so) + off_out;
                    o = o > amin ? o : amin;
                    o = o < amax ? o : amax;
                    y[((b0 * od[1] + yy) * od[2] + xx) * od[3] + c] = (int8_t)o;
                }
    return 0;
}

/* MAX_POOL_2D: pooling.cc:440-548, integer_ops/pooling.h:82-137 */
static int op_maxpool(const kwso_model *m, const o_node *nd, int8_t **buf)
{
    const o_tensor *in = &m->t[nd->in[0]], *out = &m->t[nd->out[0]];
    const int8_t *x = buf[nd->in[0]];
    int8_t *y = buf[nd->out[0]];
    const int padding = nd->p[0], stride_w = nd->p[1], stride_h = nd->p[2], f_w = nd->p[3], f_h = nd->p[4];
    const int batches = dim4(in, 0), in_h = dim4(in, 1), in_w = dim4(in, 2), depth = dim4(in, 3);
    const int out_h = dim4(out, 1), out_w = dim4(out, 2);
    int oh = out_size(padding, in_h, f_h, stride_h, 1), ow = out_size(padding, in_w, f_w, stride_w, 1);
    const int pad_h = pad_amount(stride_h, 1, in_h, f_h, oh), pad_w = pad_amount(stride_w, 1, in_w, f_w, ow);
    int32_t amin, amax;
    act_range(nd->p[5], out, &amin, &amax);
    for (int bt = 0; bt < batches; ++bt)
        for (int oy = 0; oy < out_h; ++oy)
            for (int ox = 0; ox < out_w; ++ox)
                for (int c = 0; c < depth; ++c) {
                    const int x0 = ox * stride_w - pad_w, y0 = oy * stride_h - pad_h;
                    const int fxs = 0 > -x0 ? 0 : -x0, fxe = f_w < in_w - x0 ? f_w : in_w - x0;
                    const int fys = 0 > -y0 ? 0 : -y0, fye = f_h < in_h - y0 ? f_h : in_h - y0;
                    int8_t mx = INT8_MIN;
                    for (int fy = fys; fy < fye; ++fy)
                        for (int fx = fxs; fx < fxe; ++fx) {
                            int8_t v = x[((bt * in_h + (y0 + fy)) * in_w + (x0 + fx)) * depth + c];
                            mx = v > mx ? v : mx;
                        }
                    mx = mx > (int8_t)amin ? mx : (int8_t)amin;
                    mx = mx < (int8_t)amax ? mx : (int8_t)amax;
                    y[((bt * out_h + oy) * out_w + ox) * depth + c] = mx;
                }
    return 0;
}

/* FULLY_CONNECTED: fully_connected.cc:322-396, integer_ops/fully_connected.h:23-63 */
static int op_fc(const kwso_model *m, const o_node *nd, int8_t **buf)
{
    const o_tensor *in = &m->t[nd->in[0]], *flt = &m->t[nd->in[1]], *out = &m->t[nd->out[0]];
    const o_tensor *bias = nd->n_in > 2 && nd->in[2] >= 0 ? &m->t[nd->in[2]] : NULL;
    const int8_t *x = buf[nd->in[0]];
    const int8_t *w = (const int8_t *)flt->data;
    const int32_t *b = bias ? (const int32_t *)bias->data : NULL;
    int8_t *y = buf[nd->out[0]];
    /* GetQuantizedConvolutionMultipler: (double)(float)(in*filter) / out   kernel_util_lite.cc:160-172 */
    const double in_prod = (double)(in->scale[0] * flt->scale[0]);
    const double real_mult = in_prod / (double)out->scale[0];
    int32_t mult; int exponent;
    kwso_quantize_multiplier(real_mult, &mult, &exponent);
    /* data->output_shift = -exponent; op_params.output_shift = -data->output_shift */
    const int shift = exponent;
    int32_t amin, amax;
    act_range(nd->p[0], out, &amin, &amax);
    const int32_t in_off = -in->zero[0], f_off = -flt->zero[0], out_off = out->zero[0];
    const int batches = out->dims[0], out_d = out->dims[1];
    const int accum = flt->dims[flt->ndims - 1];
    for (int bt = 0; bt < batches; ++bt)
        for (int oc = 0; oc < out_d; ++oc) {
            int32_t acc = 0;
            for (int d = 0; d < accum; ++d) {
                int32_t iv = x[bt * accum + d], fv = w[oc * accum + d];
                acc += (fv + f_off) * (iv + in_off);
            }
            if (b) acc += b[oc];
            acc = kwso_mbqm(acc, mult, shift);
            acc += out_off;
            acc = acc > amin ? acc : amin;
            acc = acc < amax ? acc : amax;
            y[oc + out_d * bt] = (int8_t)acc;
        }
    return 0;
}

/* SOFTMAX int8->int8: softmax.cc:187-226, quantization_util.cc:269-335, reference/softmax.h:66-144 */
static int op_softmax(const kwso_model *m, const o_node *nd, int8_t **buf)
{
    const o_tensor *in = &m->t[nd->in[0]], *out = &m->t[nd->out[0]];
    const int8_t *x = buf[nd->in[0]];
    int8_t *y = buf[nd->out[0]];
    const int kScaledDiffIntegerBits = 5, kAccumulationIntegerBits = 12;
    double rm = (double)nd->beta * (double)in->scale[0] * (double)(1 << (31 - kScaledDiffIntegerBits));
    const double cap = (double)((1ll << 31) - 1.0);
    if (rm > cap) rm = cap;
    int32_t mult; int left_shift;
    kwso_quantize_multiplier(rm, &mult, &left_shift);
    const double max_in = 1.0 * ((1 << kScaledDiffIntegerBits) - 1) *
                          (double)(1ll << (31 - kScaledDiffIntegerBits)) / (double)(1ll << left_shift);
    const int diff_min = (int)(-1.0 * (double)(int)floor(max_in));
    const int depth = in->dims[in->ndims - 1];
    int outer = 1;
    for (uint32_t i = 0; i + 1 < in->ndims; i++) outer *= in->dims[i];
    for (int i = 0; i < outer; ++i) {
        int8_t mx = INT8_MIN;
        for (int c = 0; c < depth; ++c) mx = x[i * depth + c] > mx ? x[i * depth + c] : mx;
        int32_t sum = 0;
        for (int c = 0; c < depth; ++c) {
            int32_t diff = (int32_t)x[i * depth + c] - mx;
            if (diff >= diff_min) {
                int32_t resc = kwso_srdhm((int32_t)((uint32_t)diff * (1u << left_shift)), mult);
                sum = wrap_add(sum, kwso_rdivpot(kwso_exp_on_negative_values_q5_26(resc), kAccumulationIntegerBits));
            }
        }
        /* GetReciprocal                                    common.h:530-546 */
        int headroom_plus_one = sum ? __builtin_clz((uint32_t)sum) : 32;
        int num_bits_over_unit = kAccumulationIntegerBits - headroom_plus_one;
        int32_t shifted_sum_minus_one = (int32_t)(((uint32_t)sum << headroom_plus_one) - (1u << 31));
        int32_t shifted_scale = kwso_one_over_one_plus_x(shifted_sum_minus_one);
        for (int c = 0; c < depth; ++c) {
            int32_t diff = (int32_t)x[i * depth + c] - mx;
            if (diff >= diff_min) {
                int32_t resc = kwso_srdhm((int32_t)((uint32_t)diff * (1u << left_shift)), mult);
                int32_t e = kwso_exp_on_negative_values_q5_26(resc);
                int32_t unsat = kwso_rdivpot(kwso_srdhm(shifted_scale, e), num_bits_over_unit + 31 - 8);
                int32_t so = unsat + (int32_t)INT8_MIN;
                so = so < 127 ? so : 127;
                so = so > -128 ? so : -128;
                y[i * depth + c] = (int8_t)so;
            } else {
                y[i * depth + c] = INT8_MIN;
            }
        }
    }
    (void)out;
    return 0;
}

/* ====================================================================== */
/*  NN ops, float32 variants (TFLite-Micro float reference kernels)        */
/* ====================================================================== */
/* CalculateActivationRange (TFL/kernels/kernel_util.h): none -> [lowest, max], relu -> [0, max], ... */
static void act_range_f32(int activation, float *amin, float *amax)
{
    *amin = -FLT_MAX; *amax = FLT_MAX;
    if (activation == 1) *amin = 0.f;
    else if (activation == 3) { *amin = 0.f; *amax = 6.f; }
    else if (activation == 2) { *amin = -1.f; *amax = 1.f; }
}
static float clampf(float x, float lo, float hi)   /* ActivationFunctionWithMinMax: std::min(std::max(x, lo), hi) */
{
    float a = x < lo ? lo : x;        /* std::max(x, lo): returns x unless x < lo */
    return hi < a ? hi : a;           /* std::min(a, hi): returns a unless hi < a */
}

/* CONV_2D float: TFL/kernels/internal/reference/conv.h:28-99 (sequential total += in * filter, then + bias, clamp) */
static int op_conv_f32(const kwso_model *m, const o_node *nd, float **buf, int depthwise)
{
    const o_tensor *in = &m->t[nd->in[0]], *flt = &m->t[nd->in[1]], *out = &m->t[nd->out[0]];
    const o_tensor *bias = nd->n_in > 2 && nd->in[2] >= 0 ? &m->t[nd->in[2]] : NULL;
    const float *x = buf[nd->in[0]], *w = (const float *)flt->data, *b = bias ? (const float *)bias->data : NULL;
    float *y = buf[nd->out[0]];
    const int padding = nd->p[0], stride_w = nd->p[1], stride_h = nd->p[2], activation = nd->p[3];
    const int dil_w = nd->p[4], dil_h = nd->p[5];
    const int batches = dim4(in, 0), in_h = dim4(in, 1), in_w = dim4(in, 2), in_d = dim4(in, 3);
    const int f_h = dim4(flt, 1), f_w = dim4(flt, 2);
    const int out_h = dim4(out, 1), out_w = dim4(out, 2), out_d = dim4(out, 3);
    int oh = out_size(padding, in_h, f_h, stride_h, dil_h), ow = out_size(padding, in_w, f_w, stride_w, dil_w);
    const int pad_h = pad_amount(stride_h, dil_h, in_h, f_h, oh), pad_w = pad_amount(stride_w, dil_w, in_w, f_w, ow);
    float amin, amax;
    act_range_f32(activation, &amin, &amax);
    if (depthwise) {   /* reference/depthwiseconv_float.h:25-97, depthwise_conv.cc:582-603 (EvalFloat honours the activation) */
        const int depth_mult = nd->p[6];
        for (int bt = 0; bt < batches; ++bt)
            for (int oy = 0; oy < out_h; ++oy)
                for (int ox = 0; ox < out_w; ++ox)
                    for (int ic = 0; ic < in_d; ++ic)
                        for (int mm = 0; mm < depth_mult; ++mm) {
                            const int oc = mm + ic * depth_mult;
                            const int x0 = ox * stride_w - pad_w, y0 = oy * stride_h - pad_h;
                            float total = 0.f;
                            for (int fy = 0; fy < f_h; ++fy)
                                for (int fx = 0; fx < f_w; ++fx) {
                                    const int ix = x0 + dil_w * fx, iy = y0 + dil_h * fy;
                                    if (ix >= 0 && ix < in_w && iy >= 0 && iy < in_h) {
                                        float prod = x[((bt * in_h + iy) * in_w + ix) * in_d + ic] * w[(fy * f_w + fx) * out_d + oc];
                                        total += prod;
                                    }
                                }
                            float bv = b ? b[oc] : 0.0f;
                            y[((bt * out_h + oy) * out_w + ox) * out_d + oc] = clampf(total + bv, amin, amax);
                        }
        return 0;
    }
    for (int bt = 0; bt < batches; ++bt)
        for (int oy = 0; oy < out_h; ++oy)
            for (int ox = 0; ox < out_w; ++ox)
                for (int oc = 0; oc < out_d; ++oc) {
                    const int x0 = ox * stride_w - pad_w, y0 = oy * stride_h - pad_h;
                    float total = 0.f;
                    for (int fy = 0; fy < f_h; ++fy)
                        for (int fx = 0; fx < f_w; ++fx)
                            for (int ic = 0; ic < in_d; ++ic) {
                                const int ix = x0 + dil_w * fx, iy = y0 + dil_h * fy;
                                if (ix >= 0 && ix < in_w && iy >= 0 && iy < in_h) {
                                    float prod = x[((bt * in_h + iy) * in_w + ix) * in_d + ic] * w[((oc * f_h + fy) * f_w + fx) * in_d + ic];
                                    total += prod;
                                }
                            }
                    float bv = b ? b[oc] : 0.0f;
                    y[((bt * out_h + oy) * out_w + ox) * out_d + oc] = clampf(total + bv, amin, amax);
                }
    return 0;
}

/* ADD float: add.cc:101-120, reference/add.h:179-215 */
static int op_add_f32(const kwso_model *m, const o_node *nd, float **buf)
{
    const o_tensor *t1 = &m->t[nd->in[0]], *t2 = &m->t[nd->in[1]], *out = &m->t[nd->out[0]];
    const float *a = buf[nd->in[0]], *bq = buf[nd->in[1]];
    float *y = buf[nd->out[0]];
    float amin, amax;
    act_range_f32(nd->p[0], &amin, &amax);
    int od[4], d1[4], d2[4];
    for (int i = 0; i < 4; i++) { od[i] = dim4(out, i); d1[i] = dim4(t1, i); d2[i] = dim4(t2, i); }
    for (int b0 = 0; b0 < od[0]; ++b0)
        for (int yy = 0; yy < od[1]; ++yy)
            for (int xx = 0; xx < od[2]; ++xx)
                for (int c = 0; c < od[3]; ++c) {
                    int i1 = (((d1[0] == 1 ? 0 : b0) * d1[1] + (d1[1] == 1 ? 0 : yy)) * d1[2] + (d1[2] == 1 ? 0 : xx)) * d1[3] + (d1[3] == 1 ? 0 : c);
                    int i2 = (((d2[0] == 1 ? 0 : b0) * d2[1] + (d2[1] == 1 ? 0 : yy)) * d2[2] + (d2[2] == 1 ? 0 : xx)) * d2[3] + (d2[3] == 1 ? 0 : c);
                    y[((b0 * od[1] + yy) * od[2] + xx) * od[3] + c] = clampf(a[i1] + bq[i2], amin, amax);
                }
    return 0;
}

/* MAX_POOL_2D float: reference/pooling.h:189-237 */
static int op_maxpool_f32(const kwso_model *m, const o_node *nd, float **buf)
{
    const o_tensor *in = &m->t[nd->in[0]], *out = &m->t[nd->out[0]];
    const float *x = buf[nd->in[0]];
    float *y = buf[nd->out[0]];
    const int padding = nd->p[0], stride_w = nd->p[1], stride_h = nd->p[2], f_w = nd->p[3], f_h = nd->p[4];
    const int batches = dim4(in, 0), in_h = dim4(in, 1), in_w = dim4(in, 2), depth = dim4(in, 3);
    const int out_h = dim4(out, 1), out_w = dim4(out, 2);
    int oh = out_size(padding, in_h, f_h, stride_h, 1), ow = out_size(padding, in_w, f_w, stride_w, 1);
    const int pad_h = pad_amount(stride_h, 1, in_h, f_h, oh), pad_w = pad_amount(stride_w, 1, in_w, f_w, ow);
    float amin, amax;
    act_range_f32(nd->p[5], &amin, &amax);
    for (int bt = 0; bt < batches; ++bt)
        for (int oy = 0; oy < out_h; ++oy)
            for (int ox = 0; ox < out_w; ++ox)
                for (int c = 0; c < depth; ++c) {
                    const int x0 = ox * stride_w - pad_w, y0 = oy * stride_h - pad_h;
                    const int fxs = 0 > -x0 ? 0 : -x0, fxe = f_w < in_w - x0 ? f_w : in_w - x0;
                    const int fys = 0 > -y0 ? 0 : -y0, fye = f_h < in_h - y0 ? f_h : in_h - y0;
                    float mx = -FLT_MAX;
                    for (int fy = fys; fy < fye; ++fy)
                        for (int fx = fxs; fx < fxe; ++fx) {
                            float v = x[((bt * in_h + (y0 + fy)) * in_w + (x0 + fx)) * depth + c];
                            mx = mx < v ? v : mx;                     /* std::max(mx, v) */
                        }
                    y[((bt * out_h + oy) * out_w + ox) * depth + c] = clampf(mx, amin, amax);
                }
    return 0;
}

/* FULLY_CONNECTED float: reference/fully_connected.h:26-60 */
static int op_fc_f32(const kwso_model *m, const o_node *nd, float **buf)
{
    const o_tensor *flt = &m->t[nd->in[1]], *out = &m->t[nd->out[0]];
    const o_tensor *bias = nd->n_in > 2 && nd->in[2] >= 0 ? &m->t[nd->in[2]] : NULL;
    const float *x = buf[nd->in[0]], *w = (const float *)flt->data, *b = bias ? (const float *)bias->data : NULL;
    float *y = buf[nd->out[0]];
    float amin, amax;
    act_range_f32(nd->p[0], &amin, &amax);
    const int out_d = out->dims[out->ndims - 1];
    int batches = 1;
    for (uint32_t i = 0; i + 1 < out->ndims; i++) batches *= out->dims[i];
    const int accum = flt->dims[flt->ndims - 1];
    for (int bt = 0; bt < batches; ++bt)
        for (int oc = 0; oc < out_d; ++oc) {
            float total = 0.f;
            for (int d = 0; d < accum; ++d) {
                float prod = x[bt * accum + d] * w[oc * accum + d];
                total += prod;
            }
            float bv = b ? b[oc] : 0.0f;
            y[oc + out_d * bt] = clampf(total + bv, amin, amax);
        }
    return 0;
}

/* SOFTMAX float: reference/softmax.h:31-63 (std::exp on float = expf) */
static int op_softmax_f32(const kwso_model *m, const o_node *nd, float **buf)
{
    const o_tensor *in = &m->t[nd->in[0]];
    const float *x = buf[nd->in[0]];
    float *y = buf[nd->out[0]];
    const int depth = in->dims[in->ndims - 1];
    int outer = 1;
    for (uint32_t i = 0; i + 1 < in->ndims; i++) outer *= in->dims[i];
    const float beta = nd->beta;
    for (int i = 0; i < outer; ++i) {
        float mx = -FLT_MAX;
        for (int c = 0; c < depth; ++c) mx = mx < x[i * depth + c] ? x[i * depth + c] : mx;
        float sum = 0.f;
        for (int c = 0; c < depth; ++c) sum += expf((x[i * depth + c] - mx) * beta);
        for (int c = 0; c < depth; ++c) y[i * depth + c] = expf((x[i * depth + c] - mx) * beta) / sum;
    }
    return 0;
}

int kwso_model_is_float(const kwso_model *m) { return m->t[m->t_in].type == TYPE_F32; }

/* float graph: input[nn_input_frame_size] -> out[label_count]; taps (optional) = every tensor, tensor-id order */
int kwso_nn_invoke_f32(const kwso_model *m, const float *input, float *out, float *taps)
{
    if (!kwso_model_is_float(m)) return -3;      /* an int8 graph's tensors are not sized for float values */
    float **buf = (float **)calloc(m->n_tensors, sizeof(float *));
    float *own = (float *)calloc(m->tap_bytes / 4 + 1, sizeof(float));
    if (!buf || !own) { free(buf); free(own); return -6; }
    for (uint32_t i = 0; i < m->n_tensors; i++) {
        buf[i] = own + m->t[i].tap_offset / 4;
        if (m->t[i].is_const) memcpy(buf[i], m->t[i].data, m->t[i].nbytes);
    }
    memcpy(buf[m->t_in], input, m->t[m->t_in].nbytes);
    int rc = 0;
    for (uint32_t i = 0; i < m->n_nodes && rc == 0; i++) {
        const o_node *nd = &m->n[i];
        switch (nd->op) {
        case OP_RESHAPE: memcpy(buf[nd->out[0]], buf[nd->in[0]], m->t[nd->out[0]].nbytes); break;
        case OP_CONV_2D: rc = op_conv_f32(m, nd, buf, 0); break;
        case OP_DEPTHWISE_CONV_2D: rc = op_conv_f32(m, nd, buf, 1); break;
        case OP_ADD: rc = op_add_f32(m, nd, buf); break;
        case OP_MAX_POOL_2D: rc = op_maxpool_f32(m, nd, buf); break;
        case OP_FULLY_CONNECTED: rc = op_fc_f32(m, nd, buf); break;
        case OP_SOFTMAX: rc = op_softmax_f32(m, nd, buf); break;
        default: rc = -3;
        }
    }
    if (rc == 0) {
        memcpy(out, buf[m->t_out], m->t[m->t_out].nbytes);
        if (taps) memcpy(taps, own, m->tap_bytes);
    }
    free(buf); free(own);
    return rc;
}

/* ei_run_classifier.h:436-444 : static_cast<int8_t>(round(f / scale) + zero_point), no clamp.
 * x86 semantics of the out-of-range cast: cvttss2si to int32 (INT_MIN when unrepresentable),
 * then the low 8 bits. */
void kwso_quantize_input(const kwso_model *m, const float *features, int8_t *q)
{
    const o_tensor *in = &m->t[m->t_in];
    const float scale = in->scale[0];
    const int32_t zp = in->zero[0];
    for (uint32_t ix = 0; ix < m->nn_input_frame_size; ix++) {
        float v = roundf(features[ix] / scale) + (float)zp;
        int32_t iv;
        if (!(v >= -2147483648.0f && v < 2147483648.0f)) iv = INT32_MIN;   /* also NaN */
        else iv = (int32_t)v;
        q[ix] = (int8_t)(uint8_t)((uint32_t)iv & 0xffu);
    }
}

void kwso_dequantize_output(const kwso_model *m, const int8_t *out_q, float *scores)
{
    const o_tensor *out = &m->t[m->t_out];
    for (uint32_t ix = 0; ix < m->n_labels; ix++)
        scores[ix] = (float)((int32_t)out_q[ix] - out->zero[0]) * out->scale[0];
}

int kwso_nn_invoke(const kwso_model *m, const int8_t *input_q, int8_t *out_q, int8_t *taps)
{
    if (kwso_model_is_float(m)) return -3;
    /* every tensor gets its own buffer (the reference overlays them in a 3392-byte arena;
     * the values are the same) */
    int8_t **buf = (int8_t **)calloc(m->n_tensors, sizeof(int8_t *));
    int8_t *own = (int8_t *)calloc(m->tap_bytes ? m->tap_bytes : 1, 1);
    if (!buf || !own) { free(buf); free(own); return -6; }
    for (uint32_t i = 0; i < m->n_tensors; i++) {
        buf[i] = own + m->t[i].tap_offset;
        if (m->t[i].is_const) memcpy(buf[i], m->t[i].data, m->t[i].nbytes);
    }
    memcpy(buf[m->t_in], input_q, m->t[m->t_in].nbytes);
    int rc = 0;
    for (uint32_t i = 0; i < m->n_nodes && rc == 0; i++) {
        const o_node *nd = &m->n[i];
        switch (nd->op) {
        case OP_RESHAPE:   /* reshape.cc:76-89 : byte copy */
            memcpy(buf[nd->out[0]], buf[nd->in[0]], m->t[nd->out[0]].nbytes);
            break;
        case OP_CONV_2D: rc = op_conv(m, nd, buf, 0); break;
        case OP_DEPTHWISE_CONV_2D: rc = op_conv(m, nd, buf, 1); break;
        case OP_ADD: rc = op_add(m, nd, buf); break;
        case OP_MAX_POOL_2D: rc = op_maxpool(m, nd, buf); break;
        case OP_FULLY_CONNECTED: rc = op_fc(m, nd, buf); break;
        case OP_SOFTMAX: rc = op_softmax(m, nd, buf); break;
        default: rc = -3;
        }
    }
    if (rc == 0) {
        memcpy(out_q, buf[m->t_out], m->t[m->t_out].nbytes);
        if (taps) memcpy(taps, own, m->tap_bytes);
    }
    free(buf); free(own);
    return rc;
}

int kwso_run_inference(const kwso_model *m, const float *features, float *scores)
{
    if (kwso_model_is_float(m)) return kwso_nn_invoke_f32(m, features, scores, NULL);   /* ei_run_classifier.h:441-443, 475 */
    int8_t *q = (int8_t *)malloc(m->nn_input_frame_size);
    int8_t oq[1024];
    if (!q) return -8;
    kwso_quantize_input(m, features, q);
    int rc = kwso_nn_invoke(m, q, oq, NULL);
    if (rc == 0) kwso_dequantize_output(m, oq, scores);
    free(q);
    return rc;
}

int kwso_run_classifier_batch(const kwso_model *m, const int16_t *pcm, size_t n, size_t B,
                              float *scores, float *features_out, int8_t *q_out)
{
    const size_t F = m->nn_input_frame_size, C = m->n_labels;
    float *feat = (float *)calloc(F, sizeof(float));
    int8_t *q = (int8_t *)malloc(F);
    int8_t oq[1024];
    if (!feat || !q) { free(feat); free(q); return -8; }
    int rc = 0;
    for (size_t b = 0; b < B && rc == 0; b++) {
        const int nf = kwso_num_frames(n, &m->dsp);
        if (nf < 1 || (size_t)nf * (size_t)feature_cols(m) > F) { rc = -5; break; }
        memset(feat, 0, sizeof(float) * F);
        if (m->dsp_block == 1) {
            kwso_mfcc_config c = m->dsp;
            c.pre_cof = 0.0f;                  /* extract_mfe_features hands the raw signal to feature::mfe (L432 ei_run_dsp.h:398-400) */
            rc = kwso_extract_mfe(pcm + b * n, n, &c, feat);
        } else
            rc = kwso_extract_mfcc(pcm + b * n, n, &m->dsp, feat);
        if (rc) { rc = -5; break; }   /* EI_IMPULSE_DSP_ERROR */
        if (kwso_model_is_float(m)) {
            memset(q, 0, F);
            rc = kwso_nn_invoke_f32(m, feat, scores + b * C, NULL);
            if (rc) break;
        } else {
            kwso_quantize_input(m, feat, q);
            rc = kwso_nn_invoke(m, q, oq, NULL);
            if (rc) break;
            kwso_dequantize_output(m, oq, scores + b * C);
        }
        if (features_out) memcpy(features_out + b * F, feat, sizeof(float) * F);
        if (q_out) memcpy(q_out + b * F, q, F);
    }
    free(feat); free(q);
    return rc;
}

int kwso_run_classifier(const kwso_model *m, const int16_t *pcm, size_t n, float *scores)
{
    return kwso_run_classifier_batch(m, pcm, n, 1, scores, NULL, NULL);
}

double kwso_time_run_classifier(const kwso_model *m, const int16_t *pcm, size_t n_clips, size_t n,
                                int iters, float *checksum)
{
    struct timespec t0, t1;
    float acc = 0.f;
    float sc[1024];
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int it = 0; it < iters; it++)
        for (size_t c = 0; c < n_clips; c++) {
            kwso_run_classifier(m, pcm + c * n, n, sc);
            acc += sc[0];
        }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (checksum) *checksum = acc;
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ====================================================================== */
/*  continuous (sliced) mode      ei_run_classifier.h:134-145, 164-282,    */
/*                                ei_run_dsp.h:310-366                     */
/* ====================================================================== */
#define KWSO_SLICES_PER_WINDOW 4     /* EI_CLASSIFIER_SLICES_PER_MODEL_WINDOW, model_metadata.h:66-68 */
struct kwso_continuous {
    const kwso_model *m;
    float *features;                 /* static_features_matrix (ei_run_classifier.h:187) */
    size_t slice_offset;
    int feature_buffer_full;
    int first_run;                   /* function-static in extract_mfcc_per_slice_features: NOT reset by init */
    struct { uint32_t buf_idx; float running_sum; float buf[KWSO_SLICES_PER_WINDOW >> 1]; } maf[64];
};

kwso_continuous *kwso_continuous_create(const kwso_model *m)
{
    kwso_continuous *s = (kwso_continuous *)calloc(1, sizeof(*s));
    if (!s) return NULL;
    s->m = m;
    s->features = (float *)calloc(m->nn_input_frame_size, sizeof(float));
    if (!s->features) { free(s); return NULL; }
    return s;
}
void kwso_continuous_free(kwso_continuous *s) { if (s) { free(s->features); free(s); } }

void kwso_continuous_init(kwso_continuous *s)          /* run_classifier_init, ei_run_classifier.h:164-172 */
{
    s->slice_offset = 0;
    s->feature_buffer_full = 0;
    memset(s->maf, 0, sizeof(s->maf));
}

/* One slice.  end_of_signal: what signal->get_data(total_length - shift, shift) delivered to the pre-emphasis
 * constructor (the reference asks for it AFTER growing total_length by one frame, i.e. beyond the slice; callers
 * pass what their callback returns there, 0 when it refuses).  *produced = inference ran. */
int kwso_continuous_step(kwso_continuous *s, const int16_t *slice, size_t n, const float *end_of_signal, float *scores,
                         int *produced)
{
    const kwso_model *m = s->m;
    const size_t F = m->nn_input_frame_size;
    const kwso_mfcc_config *c = &m->dsp;
    *produced = 0;
    size_t n_claimed = n;
    if (s->first_run) n_claimed += (size_t)(c->frame_length * (float)c->sampling_frequency);
    s->first_run = 1;
    const int nf = kwso_num_frames(n_claimed, c);
    if (nf < 1) return -5;
    const size_t feature_size = (size_t)nf * (size_t)feature_cols(m);
    if (feature_size > F) return -5;                 /* EIDSP_MATRIX_SIZE_MISMATCH */
    if (s->slice_offset + feature_size > F) return -5;
    float eos0 = 0.0f;
    const float *eos = end_of_signal;
    if (!eos && n_claimed != n) eos = &eos0;         /* the reference's calloc'd buffer when get_data refused */
    int rc;
    if (m->dsp_block == 1) {
        /* extract_mfe_per_slice_features (L432 classifier/ei_run_dsp.h:420-470): feature::mfe of the slice, nothing else */
        kwso_mfcc_config cm = *c;
        cm.pre_cof = 0.0f;
        float *en = (float *)malloc(sizeof(float) * (size_t)nf);
        if (!en) return -8;
        rc = mfe_ex(slice, n, n_claimed, &cm, s->features + s->slice_offset, en, eos);
        free(en);
    } else
        rc = mfcc_ex(slice, n, n_claimed, c, s->features + s->slice_offset, eos);
    if (rc) return -5;
    if (!s->feature_buffer_full) {
        s->slice_offset += feature_size;
        if (s->slice_offset > (F - feature_size)) {
            s->feature_buffer_full = 1;
            s->slice_offset -= feature_size;
        }
    }
    if (s->feature_buffer_full) {
        float *cm = (float *)malloc(sizeof(float) * F);
        if (!cm) return -8;
        memcpy(cm, s->features, sizeof(float) * F);
        if (m->dsp_block == 1)      /* calc_cepstral_mean_and_var_normalization_mfe (L432 classifier/ei_run_classifier.h:745-775) */
            rc = kwso_cmvnw_scale(cm, (int)(F / (size_t)c->num_filters), c->num_filters, c->win_size, 0, 1);
        else
            rc = kwso_cmvnw(cm, (int)(F / (size_t)c->num_cepstral), c->num_cepstral, c->win_size, 1);
        if (rc == 0) rc = kwso_run_inference(m, cm, scores);
        free(cm);
        if (rc) return rc;
        for (uint32_t ix = 0; ix < m->n_labels; ix++) {          /* run_moving_average_filter */
            float *buf = s->maf[ix].buf;
            s->maf[ix].running_sum -= buf[s->maf[ix].buf_idx];
            s->maf[ix].running_sum += scores[ix];
            buf[s->maf[ix].buf_idx] = scores[ix];
            if (++s->maf[ix].buf_idx >= (KWSO_SLICES_PER_WINDOW >> 1)) s->maf[ix].buf_idx = 0;
            scores[ix] = s->maf[ix].running_sum / (float)(KWSO_SLICES_PER_WINDOW >> 1);
        }
        for (size_t i = 0; i < F - feature_size; i++) s->features[i] = s->features[i + feature_size];
        *produced = 1;
    }
    return 0;
}

/* ====================================================================== */
/*  mix_audio            /root/reference/dataset-curation.py:93-137        */
/*  PARITY UNPINNED: the script needs librosa + soundfile, neither of      */
/*  which exists in the build container; restated line by line from its    */
/*  text, minus librosa.load's resampling (inputs are 16 kHz mono float32). */
/* ====================================================================== */
void kwso_mix_audio(const float *word, int word_len, const float *noise_window, float word_vol, float bg_vol, int n, int16_t *out)
{
    for (int i = 0; i < n; i++) {
        double w = 0.0;                                     /* lines 107-109: no word -> zeros; 114-120: pad with zeros, truncate */
        if (word && i < word_len) w = (double)word[i];
        double x;
        if (!noise_window) x = w;                           /* lines 123-124 */
        else {
            double a = 0.5 * (double)word_vol * w;          /* [0.5 * word_vol * i for i in waveform]: Python floats */
            float b = (float)(0.5 * (double)bg_vol) * noise_window[i];   /* 0.5 * bg_vol * ndarray(float32): the scalar takes the array's dtype */
            x = a + (double)b;                              /* list + ndarray -> float64 */
        }
        /* sf.write(subtype = 'PCM_16') of float64 data: python-soundfile switches libsndfile's clipping ON when it opens a file
           (SoundFile._open: sf_command(SFC_SET_CLIPPING, SF_TRUE)), so pcm.c converts with d2s_clip_array, not d2s_array:
               scaled = x * 2^31;  >= 0x7FFFFFFF -> 0x7FFF;  <= -2^31 -> 0x8000;  else lrint(scaled) >> 16
           -- saturating, and a TRUNCATING (floor) reduction of the rounded 32-bit value, which sits one LSB below lrint(x * 32767) for
           about half of the negative samples.  Round 3 had the no-clipping rule (lrint(x * 0x7FFF), wrapped); ADVICE round 3 pointed to
           python-soundfile's source.  Restated from libsndfile's / python-soundfile's published sources, not run against them here
           (PARITY UNPINNED).  NaN: what x86's cvtsd2si returns (INT_MIN) >> 16. */
        const double scaled = x * 2147483648.0;
        int32_t v;
        if (scaled >= 2147483647.0) { out[i] = 0x7FFF; continue; }
        if (scaled <= -2147483648.0 || scaled != scaled) { out[i] = (int16_t)-32768; continue; }
        v = (int32_t)lrint(scaled);
        out[i] = (int16_t)(v >> 16);
    }
}

/* ====================================================================== */
/*  synthetic clips (shared integer generator, include/kws/kws_synth.h)    */
/* ====================================================================== */
#include "../include/kws/kws_synth.h"
void kwso_synth_fill(uint32_t seed, uint32_t first_clip, uint32_t n_clips, uint32_t clip_len, int16_t *out)
{
    kws_synth_fill(seed, first_clip, n_clips, clip_len, out);
}
