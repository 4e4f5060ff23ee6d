// kws_audio.cpp -- the step before the path, SURVEY 8(f)4: what `librosa.load(path, sr = 16000, mono = True)` does for
// /root/reference/dataset-curation.py:111,126 -- decode a WAV file (libsndfile's conversion rules), mix it down to mono, resample it to
// the model's rate -- so that a harness can feed real recordings to kws_mix_audio_device / the classifier.  Host code for the container
// format (a few hundred bytes of header per file; nothing to accelerate) and the filter table, a GPU kernel for the resampling.
// PARITY UNPINNED: librosa / soundfile / resampy cannot be installed here, so the decoder is held to Python's own
// `wave` / scipy.io.wavfile writers and readers (independent implementations of the same container), the mono mix-down to NumPy, and the
// resampler to scipy.signal.resample_poly within a stated tolerance (tests/test_audio_ingest.py) -- not to the reference's output.
#include "kws_internal.h"

#include <map>

#include <cmath>

static uint32_t rd32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint16_t rd16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

#pragma GCC visibility push(default)
extern "C" {

// RIFF / WAVE: "fmt " (PCM = 1, IEEE float = 3, or WAVE_FORMAT_EXTENSIBLE = 0xFFFE whose sub-format's first two bytes say which) and
// "data"; every other chunk is skipped (chunks are word-aligned).  A data chunk that claims more bytes than the file holds -- what a
// streaming writer leaves behind -- is cut to what is there, as libsndfile does.
EI_IMPULSE_ERROR kws_wav_info_from_memory(const void *bytes, size_t nbytes, kws_wav_info *info)
{
    if (!bytes || !info) return fail(KWS_ERROR_BAD_ARGUMENT, "null argument");
    memset(info, 0, sizeof(*info));
    const uint8_t *b = (const uint8_t *)bytes;
    if (nbytes < 12 || memcmp(b, "RIFF", 4) != 0 || memcmp(b + 8, "WAVE", 4) != 0) return fail(KWS_ERROR_BAD_ARGUMENT, "not a RIFF/WAVE file");
    size_t pos = 12;
    bool have_fmt = false;
    int block_align = 0;
    while (pos + 8 <= nbytes) {
        const uint32_t sz = rd32(b + pos + 4);
        const size_t body = pos + 8;
        if (memcmp(b + pos, "fmt ", 4) == 0) {
            if (sz < 16 || body + 16 > nbytes) return fail(KWS_ERROR_BAD_ARGUMENT, "truncated fmt chunk");
            int tag = rd16(b + body);
            info->channels = rd16(b + body + 2);
            info->sample_rate = (int)rd32(b + body + 4);
            block_align = rd16(b + body + 12);
            info->bits_per_sample = rd16(b + body + 14);
            if (tag == 0xFFFE) {                                   // WAVE_FORMAT_EXTENSIBLE: the real tag opens the sub-format GUID
                if (sz < 40 || body + 26 > nbytes) return fail(KWS_ERROR_BAD_ARGUMENT, "truncated extensible fmt chunk");
                tag = rd16(b + body + 24);
            }
            if (tag != 1 && tag != 3) return fail(KWS_ERROR_UNSUPPORTED_MODEL, "WAV format tag %d (PCM and IEEE float are decoded)", tag);
            info->is_float = tag == 3;
            const int bps = info->bits_per_sample;
            if (info->channels < 1 || info->channels > 64 || info->sample_rate < 1 ||
                !(info->is_float ? bps == 32 : (bps == 8 || bps == 16 || bps == 24 || bps == 32)) || block_align != info->channels * bps / 8)
                return fail(KWS_ERROR_UNSUPPORTED_MODEL, "WAV: %d channels, %d bits, block align %d", info->channels, bps, block_align);
            have_fmt = true;
        } else if (memcmp(b + pos, "data", 4) == 0) {
            if (!have_fmt) return fail(KWS_ERROR_BAD_ARGUMENT, "data chunk before fmt chunk");
            size_t avail = nbytes - body;
            if ((size_t)sz < avail) avail = sz;
            info->data_offset = body;
            info->frames = avail / (size_t)block_align;
            return EI_IMPULSE_OK;
        }
        if ((size_t)sz > nbytes - body) break;
        pos = body + sz + (sz & 1);
    }
    return fail(KWS_ERROR_BAD_ARGUMENT, "no data chunk");
}

// libsndfile's integer -> float conversion (what soundfile.read(dtype = 'float32') returns and librosa.load passes on): value / 2^(bits - 1),
// 8-bit WAV samples are unsigned with a bias of 128; then librosa.to_mono = np.mean over the channels (float32 sum in channel order,
// one division).
EI_IMPULSE_ERROR kws_wav_decode_mono(const void *bytes, size_t nbytes, float *out, size_t out_cap, size_t *frames, int *sample_rate)
{
    kws_wav_info w;
    EI_IMPULSE_ERROR e = kws_wav_info_from_memory(bytes, nbytes, &w);
    if (e) return e;
    if (frames) *frames = w.frames;
    if (sample_rate) *sample_rate = w.sample_rate;
    if (!out) return EI_IMPULSE_OK;                              // a query for the sizes
    if (out_cap < w.frames) return fail(KWS_ERROR_BAD_ARGUMENT, "output holds %zu samples, the file has %zu frames", out_cap, w.frames);
    const uint8_t *p = (const uint8_t *)bytes + w.data_offset;
    const int bytes_ps = w.bits_per_sample / 8;
    for (size_t f = 0; f < w.frames; f++) {
        float sum = 0.0f;
        for (int c = 0; c < w.channels; c++, p += bytes_ps) {
            float v;
            if (w.is_float) { uint32_t u = rd32(p); memcpy(&v, &u, 4); }
            else if (bytes_ps == 1) v = (float)((int)p[0] - 128) / 128.0f;
            else if (bytes_ps == 2) v = (float)(int16_t)rd16(p) / 32768.0f;
            else if (bytes_ps == 3) v = (float)(((int32_t)((uint32_t)p[0] << 8 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 24)) >> 8) / 8388608.0f;
            else v = (float)((double)(int32_t)rd32(p) / 2147483648.0);
            sum += v;
        }
        out[f] = w.channels == 1 ? sum : sum / (float)w.channels;
    }
    return EI_IMPULSE_OK;
}

// librosa.resample's output length
size_t kws_resample_length(size_t n_in, int sr_in, int sr_out)
{
    if (sr_in <= 0 || sr_out <= 0) return 0;
    return (size_t)ceil((double)n_in * ((double)sr_out / (double)sr_in));
}

// Band-limited interpolation after the published design of resampy's "kaiser_best" filter, the resampler behind librosa.load in the
// reference's day (J. O. Smith's algorithm: a Kaiser-windowed sinc with 64 zero crossings, roll-off 0.9476, beta 14.77, tabulated at
// 512 points per zero crossing and interpolated linearly; the filter is stretched by the rate ratio when down-sampling).  The table is
// built here from those parameters -- resampy ships a pre-computed copy, which cannot be compared from this container.
static double bessel_i0(double x)
{
    double sum = 1.0, term = 1.0;
    for (int k = 1; k < 200; k++) { term *= (x / (2.0 * k)) * (x / (2.0 * k)); sum += term; if (term < 1e-18 * sum) break; }
    return sum;
}
static const int kResZeros = 64, kResPrecision = 512;
static void resample_table(std::vector<double> &win, std::vector<double> &delta)
{
    const double rolloff = 0.9475937167399596, beta = 14.769656459379492;
    const int n = kResZeros * kResPrecision;
    win.resize((size_t)n + 1); delta.resize((size_t)n + 1);
    const double i0b = bessel_i0(beta);
    for (int i = 0; i <= n; i++) {
        const double t = (double)kResZeros * (double)i / (double)n, a = rolloff * t;
        const double sinc = a == 0.0 ? 1.0 : sin(M_PI * a) / (M_PI * a);
        const double r = (double)i / (double)n;                   // position in the right half of the symmetric window
        const double taper = bessel_i0(beta * sqrt(std::max(0.0, 1.0 - r * r))) / i0b;
        win[(size_t)i] = rolloff * sinc * taper;                  // float64, as resampy's table
    }
    for (int i = 0; i < n; i++) delta[(size_t)i] = win[(size_t)i + 1] - win[(size_t)i];
    delta[(size_t)n] = 0.0;
}

EI_IMPULSE_ERROR kws_resample_device_ex(const float *in, size_t n_in, int sr_in, float *out, size_t n_out, int sr_out, int flags, void *stream)
{
    if (!in || !out || sr_in <= 0 || sr_out <= 0 || n_in == 0 || n_in > 0x7fffffff) return fail(KWS_ERROR_BAD_ARGUMENT, "kws_resample_device: bad argument");
    if (n_out != kws_resample_length(n_in, sr_in, sr_out)) return fail(KWS_ERROR_BAD_ARGUMENT, "n_out must be kws_resample_length(n_in, sr_in, sr_out) = %zu",
                                                                       kws_resample_length(n_in, sr_in, sr_out));
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(KWS_ERROR_HIP, "no HIP device available (libkws_mi355x has no CPU fallback)");
    if (sr_in == sr_out) {                                       // librosa.load does not touch a file that already has the target rate
        HIP_TRY(hipMemcpyAsync(out, in, n_in * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
        return EI_IMPULSE_OK;
    }
    // one table pair per device, built on first use; the pointers are copied while the lock is held (the function statics
    // of round 3 were re-allocated -- and leaked -- when another device called, under a concurrent caller on the first one)
    struct Tables { double *win = nullptr, *delta = nullptr; };
    static std::mutex mu;
    static std::map<int, Tables> tables;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    Tables tb;
    {
        std::lock_guard<std::mutex> lk(mu);
        Tables &slot = tables[dev];
        if (!slot.win) {
            std::vector<double> win, delta;
            resample_table(win, delta);
            double *dw = nullptr, *dd = nullptr;
            HIP_TRY(hipMalloc((void **)&dw, win.size() * sizeof(double)));
            if (hipMalloc((void **)&dd, delta.size() * sizeof(double)) != hipSuccess) { (void)hipFree(dw); return fail(KWS_ERROR_HIP, "hipMalloc failed"); }
            if (hipMemcpy(dw, win.data(), win.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(dd, delta.data(), delta.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
                (void)hipFree(dw); (void)hipFree(dd);
                return fail(KWS_ERROR_HIP, "uploading the resampler's table failed");
            }
            slot.win = dw; slot.delta = dd;
        }
        tb = slot;
    }
    const double ratio = (double)sr_out / (double)sr_in;
    // resampy's index_step = int(min(1, ratio) x precision): below 1 / precision it is 0 and the kernel would divide by it
    if ((int)(std::min(1.0, ratio) * (double)kResPrecision) < 1) return fail(KWS_ERROR_BAD_ARGUMENT, "kws_resample_device: ratio %g is below 1/%d", ratio, kResPrecision);
    const bool exact = (flags & KWS_RESAMPLE_EXACT_POSITIONS) != 0;
    // the reference's length: resampy writes int(n ratio) samples, librosa's fix_length pads with zeros up to ceil(n ratio)
    const size_t n_valid = exact ? n_out : std::min(n_out, (size_t)((double)n_in * ratio));
    // the reference's time register: resampy adds the increment once per output sample (interpn.py: time_register += time_increment); the values
    // it passes through are built the same way here and handed to the kernel (this step is data-set preparation, not the hot path: the
    // table is allocated, copied and freed around the launch)
    double *d_treg = nullptr;
    if (!exact && n_valid > 0) {
        std::vector<double> treg(n_valid);
        const double inc = 1.0 / ratio;
        double tr = 0.0;
        for (size_t t = 0; t < n_valid; t++) { treg[t] = tr; tr += inc; }
        HIP_TRY(hipMalloc((void **)&d_treg, n_valid * sizeof(double)));
        if (hipMemcpyAsync(d_treg, treg.data(), n_valid * sizeof(double), hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess ||
            hipStreamSynchronize((hipStream_t)stream) != hipSuccess) { (void)hipFree(d_treg); return fail(KWS_ERROR_HIP, "uploading the resampler's time register failed"); }
    }
    int rc = kws_launch_resample(in, n_in, out, n_out, n_valid, ratio, tb.win, tb.delta, kResZeros * kResPrecision, kResPrecision, exact ? 1 : 0, d_treg, (hipStream_t)stream);
    if (d_treg) { (void)hipStreamSynchronize((hipStream_t)stream); (void)hipFree(d_treg); }
    if (rc) return fail(KWS_ERROR_HIP, "resample kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    return EI_IMPULSE_OK;
}

EI_IMPULSE_ERROR kws_resample_device(const float *in, size_t n_in, int sr_in, float *out, size_t n_out, int sr_out, void *stream)
{
    return kws_resample_device_ex(in, n_in, sr_in, out, n_out, sr_out, 0, stream);
}

}  // extern "C"
#pragma GCC visibility pop
