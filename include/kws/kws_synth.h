/*
 * kws_synth.h -- deterministic synthetic 16 kHz int16 clips, pure 32/64-bit integer arithmetic so
 * that host C, numpy-free Python (through ctypes) and HIP device code produce the SAME samples.
 *
 * Recipe: per clip three "tones" (parabolic sine approximation,
 * f in [100, 7000] Hz, amplitude in [500, 12000]) under a smooth amplitude envelope placed at a
 * random offset (so the windowed CMVN sees non-stationary energy), plus uniform noise in
 * [-512, 511].  The reference has no input generator of its own (its inputs are microphone data,
 * L476/Core/Src/main.cpp:507-531); this one only exists for tests and benchmarks.
 */
#ifndef KWS_SYNTH_H
#define KWS_SYNTH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define KWS_HD __host__ __device__ static inline
#else
#define KWS_HD static inline
#endif

KWS_HD uint32_t kws_hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

KWS_HD uint32_t kws_hash3(uint32_t seed, uint32_t clip, uint32_t k)
{
    return kws_hash32(kws_hash32(kws_hash32(seed * 0x9e3779b9u + 0x85ebca6bu) ^ clip) + k * 0xc2b2ae35u);
}

/* parabolic "sine": phase in [0,2^32) -> [-32768, 32768] */
KWS_HD int32_t kws_psin(uint32_t phase)
{
    int32_t x = (int32_t)(phase >> 16) - 32768;          /* [-32768, 32767] */
    int32_t ax = x < 0 ? -x : x;
    return -((x * (32768 - ax)) >> 13);                   /* 0 at phase 0, +peak at quarter period */
}

typedef struct {
    uint32_t step[3], phase0[3];
    int32_t amp[3];
    int32_t env_center, env_width;
    uint32_t noise_key;
} kws_synth_params;

KWS_HD kws_synth_params kws_synth_clip_params(uint32_t seed, uint32_t clip)
{
    kws_synth_params p;
    for (uint32_t k = 0; k < 3; k++) {
        uint32_t h = kws_hash3(seed, clip, k);
        uint32_t f_hz = 100u + (h % 6901u);                                /* [100, 7000] */
        p.step[k] = (uint32_t)(((uint64_t)f_hz << 32) / 16000u);
        p.phase0[k] = kws_hash3(seed, clip, 8u + k);
        p.amp[k] = 500 + (int32_t)((h >> 13) % 11501u);                    /* [500, 12000] */
    }
    uint32_t he = kws_hash3(seed, clip, 16u);
    p.env_center = 2000 + (int32_t)(he % 12001u);                          /* [2000, 14000] */
    p.env_width = 3000 + (int32_t)((he >> 14) % 5001u);                    /* [3000, 8000] */
    p.noise_key = kws_hash3(seed, clip, 24u);
    return p;
}

KWS_HD int16_t kws_synth_sample(const kws_synth_params *p, uint32_t n)
{
    int32_t t = 0;
    for (int k = 0; k < 3; k++) {
        uint32_t ph = p->phase0[k] + n * p->step[k];
        t += (p->amp[k] * kws_psin(ph)) >> 15;
    }
    int32_t d = (int32_t)n - p->env_center;
    if (d < 0) d = -d;
    int32_t e = 0;
    if (d < p->env_width) {
        int32_t tri = 32768 - (int32_t)(((int64_t)d << 15) / p->env_width);   /* (0, 32768] */
        e = (int32_t)(((int64_t)tri * tri) >> 15);
    }
    int32_t env = 3277 + (int32_t)(((int64_t)29491 * e) >> 15);              /* [0.1, 1.0] in Q15 */
    int32_t noise = (int32_t)(kws_hash32(p->noise_key ^ (n * 0x9e3779b9u)) & 1023u) - 512;
    int32_t s = (int32_t)(((int64_t)t * env) >> 15) + noise;
    if (s > 32767) s = 32767;
    if (s < -32768) s = -32768;
    return (int16_t)s;
}

/* host helper: fill out[n_clips][clip_len] with clips first_clip .. first_clip+n_clips-1 */
KWS_HD void kws_synth_fill(uint32_t seed, uint32_t first_clip, uint32_t n_clips, uint32_t clip_len, int16_t *out)
{
    for (uint32_t c = 0; c < n_clips; c++) {
        kws_synth_params p = kws_synth_clip_params(seed, first_clip + c);
        for (uint32_t n = 0; n < clip_len; n++) out[(uint64_t)c * clip_len + n] = kws_synth_sample(&p, n);
    }
}

#endif
