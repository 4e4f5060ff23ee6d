// kws_misc.hip -- small kernels: feature quantisation, the synthetic clip generator, the per-stream moving average and
// feature-buffer shift of continuous mode.
#include "kws_device.h"
#include "../../include/kws/kws_synth.h"

// ---------------------------------------------------------------------------------------------------------
//  float features -> int8 input tensor (the quantise loop of run_inference, ei_run_classifier.h:436-444)
// ---------------------------------------------------------------------------------------------------------
__global__ void kws_quantize_kernel(const float *__restrict__ f, int8_t *__restrict__ q, size_t n, float scale, int zp)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float qv = roundf(f[i] / scale) + (float)zp;
        int iv = (qv >= -2147483648.0f && qv < 2147483648.0f) ? (int)qv : (int)0x80000000;
        q[i] = (int8_t)(iv & 0xff);
    }
}

// ---------------------------------------------------------------------------------------------------------
//  synthetic clips, generated in HBM (include/kws/kws_synth.h; bit-identical to the host generator)
// ---------------------------------------------------------------------------------------------------------
__global__ void kws_synth_kernel(uint32_t seed, uint32_t first_clip, uint32_t n_clips, uint32_t clip_len, int16_t *out)
{
    for (uint32_t c = blockIdx.y; c < n_clips; c += gridDim.y) {
        const kws_synth_params p = kws_synth_clip_params(seed, first_clip + c);
        for (uint32_t n = blockIdx.x * blockDim.x + threadIdx.x; n < clip_len; n += gridDim.x * blockDim.x)
            out[(size_t)c * clip_len + n] = kws_synth_sample(&p, n);
    }
}

// ---------------------------------------------------------------------------------------------------------
//  launchers (called from kws_api.cpp)
// ---------------------------------------------------------------------------------------------------------
// ---- continuous mode, many streams: per-class 2-tap moving average (ei_run_classifier.h:134-145) and the feature-buffer
//      shift (ei_run_classifier.h:277-279) for S streams advancing in lock step
__global__ void kws_maf_kernel(float *__restrict__ scores, float *__restrict__ running_sum, float *__restrict__ maf_buf, int n,
                               int buf_idx, int taps)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float rs = running_sum[i];
        const float v = scores[i];
        rs -= maf_buf[(size_t)i * taps + buf_idx];
        rs += v;
        maf_buf[(size_t)i * taps + buf_idx] = v;
        running_sum[i] = rs;
        scores[i] = rs / (float)taps;
    }
}

// linear copy of ring-indexed rolling buffers (KwsDspPlan::ring_*): dst[stream][row][col] = src[stream][physical row][col].  Used where a
// consumer wants plain rows (the general cmvnw kernel, the MFE normalisation, kws_streams_init) -- the tuned cmvnw + network kernel
// reads the ring directly.
__global__ void kws_unring_kernel(const float *__restrict__ src, float *__restrict__ dst, int n_streams, int rows, int cols, int ring_rows, int head)
{
    const size_t F = (size_t)rows * cols, total = (size_t)n_streams * F;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t st = i / F;
        const int k = (int)(i - st * F), r = k / cols, c = k - r * cols;
        const int pr = (ring_rows && r < ring_rows) ? (r + head) % ring_rows : r;
        dst[i] = src[st * F + (size_t)pr * cols + c];
    }
}

int kws_launch_maf(float *scores, float *running_sum, float *maf_buf, int n, int buf_idx, int taps, hipStream_t stream)
{
    (void)hipGetLastError();      // the status returned below is this launch's, not a stale error of an earlier call
    if (n <= 0) return 0;
    hipLaunchKernelGGL(kws_maf_kernel, dim3((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024), dim3(256), 0, stream, scores, running_sum,
                       maf_buf, n, buf_idx, taps);
    return (int)hipGetLastError();
}

int kws_launch_unring(const float *src, float *dst, int n_streams, int rows, int cols, int ring_rows, int head, hipStream_t stream)
{
    (void)hipGetLastError();      // the status returned below is this launch's, not a stale error of an earlier call
    if (n_streams <= 0) return 0;
    size_t blocks = ((size_t)n_streams * rows * cols + 255) / 256;
    hipLaunchKernelGGL(kws_unring_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, stream, src, dst, n_streams, rows, cols,
                       ring_rows, head);
    return (int)hipGetLastError();
}

int kws_launch_quantize(const float *f, int8_t *q, size_t n, float scale, int zp, hipStream_t stream)
{
    (void)hipGetLastError();      // the status returned below is this launch's, not a stale error of an earlier call
    if (n == 0) return 0;
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(kws_quantize_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, f, q, n, scale, zp);
    return (int)hipGetLastError();
}

int kws_launch_synth(uint32_t seed, uint32_t first_clip, uint32_t n_clips, uint32_t clip_len, int16_t *out, hipStream_t stream)
{
    (void)hipGetLastError();      // the status returned below is this launch's, not a stale error of an earlier call
    if (n_clips == 0) return 0;
    dim3 grid((clip_len + 255) / 256, n_clips < 65535u ? n_clips : 65535u);
    hipLaunchKernelGGL(kws_synth_kernel, grid, dim3(256), 0, stream, seed, first_clip, n_clips, clip_len, out);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
//  MFE block of the newer SDK copy (L432 .../edge-impulse-sdk): extract_mfe_features (classifier/ei_run_dsp.h:369-418) =
//  speechpy::feature::mfe -> processing::cmvnw(win_size, variance_normalization = false, scale = true)
//  (dsp/speechpy/processing.hpp:327-399) -> numpy::normalize (dsp/numpy.hpp:1391-1429).  This kernel is the part after mfe:
//  one wave per clip, the [frames][filters] matrix in LDS, windowed mean subtraction with cmvn_columns<.., VARIANCE = false>,
//  then (x - min) * (1 / (max - min)) over the whole matrix, in place in HBM.
// ---------------------------------------------------------------------------------------------------------
constexpr int KWS_MFE_NORM_WAVES = 4;
__global__ __launch_bounds__(KWS_WAVE * KWS_MFE_NORM_WAVES) void kws_mfe_norm_kernel(float *__restrict__ feat, int n_clips, int rows, int cols, int win,
                                                                                   const int *__restrict__ pad_map, int prow)
{
    constexpr int MAXR = KWS_MAXF, MAXC = KWS_NF_MAX, MELS = MAXC + 1;
    __shared__ int s_map[KWS_MAXPROW];
    __shared__ float s_in[KWS_MFE_NORM_WAVES][MAXR * MELS];
    __shared__ float s_out[KWS_MFE_NORM_WAVES][MAXR * MAXC];
    __shared__ __attribute__((aligned(16))) int s_off[KWS_MFE_NORM_WAVES][2 * KWS_ZF];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < prow; i += blockDim.x) s_map[i] = pad_map[i];
    __syncthreads();
    float *in = s_in[wave], *out = s_out[wave];
    const int n = rows * cols;
    for (int clip = blockIdx.x * KWS_MFE_NORM_WAVES + wave; clip < n_clips; clip += gridDim.x * KWS_MFE_NORM_WAVES) {
        float *g = feat + (size_t)clip * n;
        for (int i = lane; i < n; i += KWS_WAVE) {
            const int r = i / cols, c = i - r * cols;
            in[r * MELS + c] = g[i];
        }
        WAVE_SYNC();
        float mn = FLT_MAX, mx = -FLT_MAX;                      // numpy::min / max (numpy.hpp:842-905): strict comparisons
        auto emit = [&](int row, int c, float o) {
            out[row * cols + c] = o;
            if (o < mn) mn = o;
            if (o > mx) mx = o;
        };
        if (cols > 16) cmvn_columns<17, 20, decltype(emit), false>(in, MELS, s_map, s_off[wave], lane, rows, cols, prow, win, emit);
        else cmvn_columns<13, 16, decltype(emit), false>(in, MELS, s_map, s_off[wave], lane, rows, cols, prow, win, emit);
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) {               // min / max are exact: any reduction order gives the same value
            const float a = __shfl_xor(mn, sft), b = __shfl_xor(mx, sft);
            mn = a < mn ? a : mn;
            mx = b > mx ? b : mx;
        }
        WAVE_SYNC();
        const float row_scale = 1.0f / (mx - mn);               // numpy.hpp:1416
        for (int i = lane; i < n; i += KWS_WAVE) {
            float o = out[i] - mn;                              // numpy::subtract, then numpy::scale (skipped for 1.0f)
            if (row_scale != 1.0f) o = o * row_scale;
            g[i] = o;
        }
        WAVE_SYNC();
    }
}

int kws_launch_mfe_norm(float *feat, int n_clips, int rows, int cols, int win, const int *pad_map, int prow, int grid_cap, hipStream_t stream)
{
    (void)hipGetLastError();
    if (n_clips <= 0) return 0;
    int grid = (n_clips + KWS_MFE_NORM_WAVES - 1) / KWS_MFE_NORM_WAVES;
    if (grid > grid_cap) grid = grid_cap;
    hipLaunchKernelGGL(kws_mfe_norm_kernel, dim3(grid), dim3(KWS_WAVE * KWS_MFE_NORM_WAVES), 0, stream, feat, n_clips, rows, cols, win, pad_map, prow);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
//  mix_audio (/root/reference/dataset-curation.py:93-137), batched: one thread per output sample.
//    word  : float32 waveform as librosa.load(sr = 16 kHz, mono) would return it, word_len[b] samples at words + b * word_stride;
//            padded with zeros / truncated to n samples (lines 114-120); words == NULL: the "just background noise" case (107-109)
//    noise : float32 background track; the clip takes noise[start[b] .. start[b] + n) (the reference draws start with Python's
//            random.randint(0, len - n): the caller supplies it); noise == NULL: the word alone (123-124)
//    out   = 0.5 * word_vol * word (a Python float product: double)  +  0.5 * bg_vol * noise (NumPy scalar * float32 array: float32)
//            summed in double (list + ndarray), lines 134-135; then PCM16 as sf.write(..., subtype = "PCM_16") stores float64 data:
//            python-soundfile switches libsndfile's clipping on when it opens a file (SFC_SET_CLIPPING), so pcm.c converts with
//            d2s_clip_array: scaled = x * 2^31, saturated at 0x7FFF / 0x8000, else lrint(scaled) >> 16 -- a floor of the rounded 32-bit
//            value (one LSB below lrint(x * 32767) for about half of the negative samples).  Round 3 had the no-clipping rule (wrapping);
//            ADVICE round 3 pointed to python-soundfile's source.
//  PARITY UNPINNED: librosa / soundfile are not available where this was written, so neither the resampling (excluded: inputs are
//  already 16 kHz mono) nor the PCM16 conversion rule could be checked against the reference; tests hold the kernel to the
//  restatement in oracle/kws_oracle.c only.
// ---------------------------------------------------------------------------------------------------------
__global__ void kws_mix_audio_kernel(const float *__restrict__ words, const int *__restrict__ word_len, size_t word_stride,
                                     const float *__restrict__ noise, const int *__restrict__ start, float word_vol, float bg_vol, int n,
                                     size_t total, int16_t *__restrict__ out)
{
    const float bgs = (float)(0.5 * (double)bg_vol);            // the scalar 0.5 * bg_vol, cast to the array's dtype
    const double ws = 0.5 * (double)word_vol;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t b = e / (size_t)n;
        const int i = (int)(e - b * (size_t)n);
        double w = 0.0;
        if (words && i < word_len[b]) w = (double)words[b * word_stride + i];
        double x = noise ? ws * w + (double)(bgs * noise[(size_t)start[b] + i]) : w;        // no noise: the waveform itself is returned
        const double scaled = x * 2147483648.0;                                             // libsndfile's d2s_clip_array (see above)
        int16_t smp;
        if (scaled >= 2147483647.0) smp = 0x7FFF;
        else if (!(scaled > -2147483648.0)) smp = (int16_t)-32768;                          // full-scale negative, and NaN (x86's cvtsd2si: INT_MIN)
        else smp = (int16_t)((int)rint(scaled) >> 16);
        out[e] = smp;
    }
}

int kws_launch_mix_audio(const float *words, const int *word_len, size_t word_stride, const float *noise, const int *start, float word_vol,
                         float bg_vol, int n, size_t n_clips, int16_t *out, hipStream_t stream)
{
    (void)hipGetLastError();
    if (n_clips == 0) return 0;
    const size_t total = n_clips * (size_t)n;
    const int grid = (int)std::min<size_t>((total + 255) / 256, 65536);
    hipLaunchKernelGGL(kws_mix_audio_kernel, dim3(grid), dim3(256), 0, stream, words, word_len, word_stride, noise, start, word_vol, bg_vol, n,
                       total, out);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
//  Resampling (the `sr = 16000` of librosa.load, dataset-curation.py:111,126): band-limited interpolation with a tabulated
//  Kaiser-windowed sinc (kws_audio.cpp builds the table), one thread per output sample.  Output sample t sits at input time t / ratio;
//  the filter's left wing runs over x[n], x[n - 1], ... and its right wing over x[n + 1], ...; when down-sampling (ratio < 1) the filter is
//  stretched by 1 / ratio and scaled by ratio (anti-aliasing).
//  EXACT = false (the default of kws_resample_device: the REFERENCE's behaviour): the published loop of resampy's
//  resample_f as librosa.load ran it in the reference's day -- a wing walks the table in steps of the truncated integer
//  int(scale x precision) from offset int(frac x precision), with ONE interpolation factor eta per wing; at most (nwin - offset) / step
//  taps with nwin = the table's 32 769 entries (len(interp_win)); the output is a float32 array, so every `y[t] += weight * x` rounds to
//  float32; resampy produces floor(n ratio) samples and
//  librosa's fix_length pads with zeros up to ceil(n ratio): n_valid.  resampy advances its time register by repeated addition
//  (time_register += time_increment): at the output samples whose time is an integer number of input samples -- every 160th for 44.1 -> 16
//  kHz -- the accumulated value can sit one ulp below it and the loop then starts one input sample earlier with a fraction just under 1, so
//  the register's values come from the host, which builds them by that same repeated addition (treg).
//  EXACT = true (KWS_RESAMPLE_EXACT_POSITIONS): every tap's table position computed exactly, products summed in double, every sample
//  computed: 5e-8 .. 7e-7 from the analytic signal, where the integer stepping costs 6e-4 .. 2e-3 at non-integer ratios.
// ---------------------------------------------------------------------------------------------------------
template <bool EXACT>
__global__ void kws_resample_kernel(const float *__restrict__ x, int n_in, float *__restrict__ y, size_t n_out, size_t n_valid, double ratio,
                                    const double *__restrict__ win, const double *__restrict__ delta, int nwin, int precision, const double *__restrict__ treg)
{
    const double scale = ratio < 1.0 ? ratio : 1.0, time_inc = 1.0 / ratio;
    const int index_step = (int)(scale * (double)precision);
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_out; t += (size_t)gridDim.x * blockDim.x) {
        if (t >= n_valid) { y[t] = 0.0f; continue; }               // librosa.util.fix_length's zero padding
        const double time_register = (!EXACT && treg) ? treg[t] : (double)t * time_inc;
        const int n = (int)time_register;
        const double frac = scale * (time_register - (double)n);
        if constexpr (EXACT) {
            double acc = 0.0;
            auto wing = [&](double fr, int count, int first, int dir) {
                for (int i = 0; i < count; ++i) {
                    const double pos = (fr + (double)i * scale) * (double)precision;
                    const int k = (int)pos;
                    if (k >= nwin) break;
                    acc += (win[k] + (pos - (double)k) * delta[k]) * (double)x[first + dir * i];
                }
            };
            wing(frac, n + 1, n, -1);                              // left wing: x[n], x[n - 1], ...
            wing(scale - frac, n_in - n - 1, n + 1, 1);            // right wing: x[n + 1], ...
            y[t] = (float)(scale * acc);
        } else {
            float acc = 0.0f;                                      // y[t, j]: an element of a float32 array
            auto wing = [&](double fr, int avail, int first, int dir) {
                const double index_frac = fr * (double)precision;
                const int offset = (int)index_frac;
                const double eta = index_frac - (double)offset;
                const int count = min(avail, (nwin - offset) / index_step);
                for (int i = 0; i < count; ++i) {
                    const int k = offset + i * index_step;
                    // resampy scales the table itself (interp_win *= ratio) and takes the differences of the SCALED table: the same two
                    // roundings here; the table's last entry has no successor (interp_delta's last entry is 0)
                    const double wk = scale * win[k];
                    const double dk = k + 1 < nwin ? scale * win[k + 1] - wk : 0.0;
                    const double weight = wk + eta * dk;
                    acc = (float)((double)acc + weight * (double)x[first + dir * i]);
                }
            };
            wing(frac, n + 1, n, -1);
            wing(scale - frac, n_in - n - 1, n + 1, 1);
            y[t] = acc;
        }
    }
}

int kws_launch_resample(const float *in, size_t n_in, float *out, size_t n_out, size_t n_valid, double ratio, const double *win, const double *delta, int nwin,
                        int precision, int exact, const double *treg, hipStream_t stream)
{
    (void)hipGetLastError();
    if (n_out == 0) return 0;
    const int grid = (int)std::min<size_t>((n_out + 255) / 256, 65536);
    if (exact) hipLaunchKernelGGL(kws_resample_kernel<true>, dim3(grid), dim3(256), 0, stream, in, (int)n_in, out, n_out, n_valid, ratio, win, delta, nwin, precision, nullptr);
    else hipLaunchKernelGGL(kws_resample_kernel<false>, dim3(grid), dim3(256), 0, stream, in, (int)n_in, out, n_out, n_valid, ratio, win, delta, nwin + 1, precision, treg);   // len(interp_win) = zeros x precision + 1
    return (int)hipGetLastError();
}
