"""Input families that probe KWS_MODE_FAST away from the noise-floored synthetic clips of include/kws/kws_synth.h (VERDICT round 2,
item 1).  Every family is a deterministic function of (name, n, seed) -> int16 [n][16000]; nothing here reads the reference.

What they are for: the fast kernel's cmvnw divides every cepstral column by its windowed deviation, so whatever the fp32
re-ordering moved in the cepstra is amplified by 1 / deviation.  The synthetic bench clips never have a quiet column; these do:

  amp_sweep      speech-like tone groups under an envelope, NO noise floor, peak amplitude swept 1 .. 32767 LSB
  word_silence   a word shorter than the window followed by digital silence -- what /root/reference/dataset-curation.py:114-116
                 produces for every Speech-Commands file shorter than 1 s (the test makes these with kws_mix_audio_device)
  dc_tone        DC offset (either sign, up to 20000 LSB) plus one tone, with and without a little noise
  clipped        tones + noise driven 2 .. 16x past full scale and hard-clipped to int16
  pure_tone      one sinusoid, no noise: most mel columns sit at log(FLT_EPSILON) in some frames and not in others
  bursts         digital silence with 1 .. 6 bursts shorter than one frame (<= 320 samples)
  quiet_noise    uniform noise of 1 .. 50 LSB peak and nothing else
  near_constant  a frame-periodic signal (every frame the same samples: constant cepstral columns) plus a perturbation whose
                 size is swept over five decades: walks every cmvnw column's deviation THROUGH the guard of the fast kernel
  detuned_tone   a tone a few millihertz .. hertz away from a multiple of the frame rate (50 Hz): nearly identical frames
  word_background  the reference's OWN data shape, for scale: dataset-curation.py mixes every word with a window of a background
                 recording at its default volumes (word 1.0, background 0.1: 0.5 word + 0.05 background, lines 134-135, 167-181) --
                 a word shorter than the window over white / pink / amplitude-modulated noise (the test mixes it with
                 kws_mix_audio_device)
  word_noise_gain  the same mix with BOTH volumes drawn per clip (word 0.05 .. 1, background 0.01 .. 1: a quiet word over a loud floor up to a
                 loud word over a barely audible one): 0.5 word_vol word + 0.5 bg_vol background (dataset-curation.py:133-135) -- bench.py's
                 mix_audio-shaped input family
"""
import numpy as np

CLIP_LEN = 16000
FS = 16000.0
FAMILIES = ("amp_sweep", "word_silence", "dc_tone", "clipped", "pure_tone", "bursts", "quiet_noise", "near_constant", "detuned_tone", "word_background",
            "word_noise_gain")


def _to_pcm(x):
    return np.clip(np.rint(x), -32768, 32767).astype(np.int16)


def _tones(rng, n, k, t, fmin=100.0, fmax=7000.0):
    """[n][len(t)] sum of k unit sinusoids with random frequency / phase and random relative weights"""
    out = np.zeros((n, t.size))
    for _ in range(k):
        f = rng.uniform(fmin, fmax, (n, 1))
        ph = rng.uniform(0, 2 * np.pi, (n, 1))
        w = rng.uniform(0.3, 1.0, (n, 1))
        out += w * np.sin(2 * np.pi * f * t[None, :] / FS + ph)
    return out / np.abs(out).max(axis=1, keepdims=True)


def _envelope(rng, n, length):
    """Hann-shaped bump of random width at a random position, [n][length] in 0 .. 1"""
    pos = rng.uniform(0.15, 0.85, (n, 1)) * length
    wid = rng.uniform(0.1, 0.5, (n, 1)) * length
    t = np.arange(length)[None, :]
    u = np.clip((t - pos) / wid, -0.5, 0.5)
    return 0.5 * (1.0 + np.cos(2 * np.pi * u))


def word_waveforms(n, seed):
    """float32 words in -1 .. 1 of 0.2 .. 0.9 s (what librosa.load hands mix_audio): [n][16000] zero beyond word_len[n]"""
    rng = np.random.default_rng([seed, 77])
    t = np.arange(CLIP_LEN, dtype=np.float64)
    length = rng.integers(int(0.2 * FS), int(0.9 * FS), n)
    x = _tones(rng, n, 3, t) * rng.uniform(0.02, 0.9, (n, 1))
    x += rng.uniform(-1, 1, (n, CLIP_LEN)) * rng.choice([0.0, 1e-4, 3e-3], (n, 1))
    for i in range(n):
        x[i, :length[i]] *= np.hanning(length[i]) ** 0.5
        x[i, length[i]:] = 0.0
    return x.astype(np.float32), length.astype(np.int32)


def background_track(seed, seconds=30):
    """float32 background recording in -1 .. 1: thirds of white noise, pink noise and slowly amplitude-modulated noise"""
    rng = np.random.default_rng([seed, 99])
    m = int(seconds * FS) // 3
    white = rng.standard_normal(m) * 0.15
    spec = np.fft.rfft(rng.standard_normal(m))
    spec[1:] /= np.sqrt(np.arange(1, spec.size))
    spec[0] = 0
    pink = np.fft.irfft(spec, m)
    pink *= 0.2 / pink.std()
    mod = rng.standard_normal(m) * 0.2 * (0.55 + 0.45 * np.sin(2 * np.pi * 0.7 * np.arange(m) / FS + 1.0))
    return np.clip(np.concatenate([white, pink, mod]), -1, 1).astype(np.float32)


def family(name, n, seed=0):
    rng = np.random.default_rng([seed, FAMILIES.index(name)])
    t = np.arange(CLIP_LEN, dtype=np.float64)
    if name == "amp_sweep":
        amp = np.exp(rng.uniform(np.log(1.0), np.log(32767.0), (n, 1)))
        return _to_pcm(amp * _tones(rng, n, 3, t) * (0.05 + 0.95 * _envelope(rng, n, CLIP_LEN)))
    if name == "word_silence":
        # host twin of what the GPU test makes with kws_mix_audio_device (word_vol 1, no background): 0.5 * word, then PCM16
        w, _ = word_waveforms(n, seed)
        return _to_pcm(0.5 * w.astype(np.float64) * 32767.0)
    if name == "word_background":
        w, _ = word_waveforms(n, seed)
        track = background_track(seed)
        start = rng.integers(0, track.size - CLIP_LEN + 1, n)
        noise = np.stack([track[st:st + CLIP_LEN] for st in start])
        return _to_pcm((0.5 * w.astype(np.float64) + (np.float32(0.05) * noise).astype(np.float64)) * 32767.0)
    if name == "word_noise_gain":
        w, _ = word_waveforms(n, seed)
        track = background_track(seed)
        start = rng.integers(0, track.size - CLIP_LEN + 1, n)
        noise = np.stack([track[st:st + CLIP_LEN] for st in start])
        wv = np.exp(rng.uniform(np.log(0.05), np.log(1.0), (n, 1)))
        bv = np.exp(rng.uniform(np.log(0.01), np.log(1.0), (n, 1)))
        return _to_pcm((0.5 * wv * w.astype(np.float64) + 0.5 * bv * noise.astype(np.float64)) * 32767.0)
    if name == "dc_tone":
        dc = rng.choice([-1.0, 1.0], (n, 1)) * np.exp(rng.uniform(np.log(50.0), np.log(20000.0), (n, 1)))
        amp = np.exp(rng.uniform(np.log(20.0), np.log(10000.0), (n, 1)))
        noise = rng.uniform(-1, 1, (n, CLIP_LEN)) * rng.choice([0.0, 2.0, 100.0], (n, 1))
        return _to_pcm(dc + amp * _tones(rng, n, 1, t) + noise)
    if name == "clipped":
        gain = rng.uniform(2.0, 16.0, (n, 1)) * 32767.0
        x = gain * _tones(rng, n, 3, t) * (0.2 + 0.8 * _envelope(rng, n, CLIP_LEN)) + rng.uniform(-400, 400, (n, CLIP_LEN))
        return _to_pcm(x)
    if name == "pure_tone":
        amp = np.exp(rng.uniform(np.log(30.0), np.log(32000.0), (n, 1)))
        return _to_pcm(amp * _tones(rng, n, 1, t, 60.0, 7900.0))
    if name == "bursts":
        x = np.zeros((n, CLIP_LEN))
        for i in range(n):
            for _ in range(rng.integers(1, 7)):
                ln = int(rng.integers(1, 321))
                at = int(rng.integers(0, CLIP_LEN - ln))
                a = np.exp(rng.uniform(np.log(5.0), np.log(30000.0)))
                x[i, at:at + ln] += a * rng.uniform(-1, 1, ln) if rng.random() < 0.5 else a * np.sin(2 * np.pi * rng.uniform(200, 6000) * np.arange(ln) / FS)
        return _to_pcm(x)
    if name == "quiet_noise":
        amp = np.exp(rng.uniform(np.log(1.0), np.log(50.0), (n, 1)))
        return _to_pcm(amp * rng.uniform(-1, 1, (n, CLIP_LEN)))
    if name == "near_constant":
        # every frame starts a new period of the same 320 samples (frame stride 320): all cepstral columns are constant; the
        # perturbation (a tone under a slow envelope, or sparse noise) is swept from far below one LSB of effect to dominant
        base = rng.integers(-3000, 3000, (n, 320)).astype(np.float64) * rng.choice([0.05, 0.3, 1.0, 4.0], (n, 1))
        x = np.tile(base, (1, CLIP_LEN // 320))
        eps = np.exp(rng.uniform(np.log(0.3), np.log(3000.0), (n, 1)))
        kind = rng.integers(0, 3, (n, 1))
        pert_tone = _tones(rng, n, 1, t) * _envelope(rng, n, CLIP_LEN)
        pert_noise = rng.uniform(-1, 1, (n, CLIP_LEN)) * (rng.random((n, CLIP_LEN)) < 0.02)
        pert_ramp = (t[None, :] / CLIP_LEN - 0.5) * 2.0 * _tones(rng, n, 1, t, 300.0, 4000.0)
        pert = np.where(kind == 0, pert_tone, np.where(kind == 1, pert_noise, pert_ramp))
        return _to_pcm(x + eps * pert)
    if name == "detuned_tone":
        k = rng.integers(4, 150, (n, 1)).astype(np.float64)
        df = rng.choice([-1.0, 1.0], (n, 1)) * np.exp(rng.uniform(np.log(1e-3), np.log(5.0), (n, 1)))
        amp = np.exp(rng.uniform(np.log(100.0), np.log(30000.0), (n, 1)))
        ph = rng.uniform(0, 2 * np.pi, (n, 1))
        x = amp * np.sin(2 * np.pi * (50.0 * k + df) * t[None, :] / FS + ph)
        x += rng.choice([0.0, 0.0, 1.0], (n, 1)) * rng.uniform(-1, 1, (n, CLIP_LEN))
        return _to_pcm(x)
    raise KeyError(name)


def column_conditioning(cep, win_size, full=False):
    """cmvnw's windows over cepstra [n][rows][cols] (processing.hpp:326-389: symmetric padding of (win - 1) / 2 rows, every
    window's population deviation per column): returns per clip the smallest deviation / max(1, |mean|) over all (row, column)
    -- the quantity the fast kernel's guard is stated in -- and the smallest plain deviation."""
    n, rows, cols = cep.shape
    pad = (win_size - 1) // 2
    idx = np.arange(-pad, rows + pad)
    m = np.mod(idx, 2 * rows)
    pm = np.where(m < rows, m, 2 * rows - 1 - m)
    x = cep.astype(np.float64)[:, pm, :]                                   # [n][rows + 2 pad][cols]
    c1 = np.concatenate([np.zeros((n, 1, cols)), np.cumsum(x, axis=1)], axis=1)
    c2 = np.concatenate([np.zeros((n, 1, cols)), np.cumsum(x * x, axis=1)], axis=1)
    s1 = (c1[:, win_size:win_size + rows] - c1[:, :rows]) / win_size
    s2 = (c2[:, win_size:win_size + rows] - c2[:, :rows]) / win_size
    sd = np.sqrt(np.maximum(s2 - s1 * s1, 0.0))
    if full:
        return sd, s1                                                     # [n][rows][cols] deviation and mean of every window
    rel = sd / np.maximum(1.0, np.abs(s1))
    return rel.reshape(n, -1).min(axis=1), sd.reshape(n, -1).min(axis=1)
