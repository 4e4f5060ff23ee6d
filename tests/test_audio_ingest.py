"""The step before the path: what `librosa.load(path, sr = 16000, mono = True)` (/root/reference/dataset-curation.py:111,126)
does with a WAV file -- decode, mono mix-down, resampling -- in the library (include/kws/kws.h: kws_wav_*, kws_resample_*).
PARITY UNPINNED: librosa / soundfile / resampy cannot be installed in this container (pip: "no matching distribution", no network), so
these tests hold the pieces to INDEPENDENT implementations instead of the reference's own output:
  * the container parser and the sample conversion to Python's `wave` module and scipy.io.wavfile (writers of the same format) and to
    libsndfile's documented rule  value / 2^(bits - 1)  (8-bit WAV: unsigned, bias 128);
  * the mono mix-down to numpy.mean over the channels in float32 (librosa.to_mono);
  * the resampler to the ANALYTIC value of band-limited test signals at the new sample times and to scipy.signal.resample_poly."""
import io
import os
import struct
import wave

import numpy as np
import pytest

from kws_testlib import MODELS, ROOT


@pytest.fixture(scope="module")
def pkg():
    import sys
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_package
    return load_package()


def wave_bytes(samples, rate, width):
    """samples [frames][channels] of integers -> WAV image written by Python's `wave` module"""
    buf = io.BytesIO()
    with wave.open(buf, "wb") as w:
        w.setnchannels(samples.shape[1])
        w.setsampwidth(width)
        w.setframerate(rate)
        if width == 1:
            raw = (samples + 128).astype(np.uint8).tobytes()
        elif width == 2:
            raw = samples.astype("<i2").tobytes()
        elif width == 3:
            raw = b"".join(int(v).to_bytes(3, "little", signed=True) for v in samples.reshape(-1))
        else:
            raw = samples.astype("<i4").tobytes()
        w.writeframes(raw)
    return buf.getvalue()


def test_wav_decoder_against_independent_writers(pkg):
    rng = np.random.default_rng(1)
    for width, ch, rate in ((2, 1, 16000), (2, 2, 44100), (1, 1, 8000), (3, 2, 22050), (4, 3, 48000)):
        hi = 1 << (8 * width - 1)
        x = rng.integers(-hi, hi, (1234, ch))
        x[0, 0], x[1, 0] = -hi, hi - 1                                       # the extremes
        img = wave_bytes(x, rate, width)
        w = pkg.wav_info(img)
        assert (w.channels, w.sample_rate, w.bits_per_sample, w.is_float, w.frames) == (ch, rate, 8 * width, 0, 1234)
        got, sr = pkg.wav_decode_mono(img)
        f = (x.astype(np.float64) / hi).astype(np.float32)                   # libsndfile: value / 2^(bits - 1)
        want = f[:, 0] if ch == 1 else np.mean(f, axis=1, dtype=np.float32)  # librosa.to_mono
        assert sr == rate and got.dtype == np.float32 and (got.view(np.uint32) == want.view(np.uint32)).all(), (width, ch)
    # scipy's writer: float32 and int16, with the header layouts it chooses (WAVE_FORMAT_IEEE_FLOAT with a fact chunk)
    from scipy.io import wavfile
    xf = rng.uniform(-1, 1, (777, 2)).astype(np.float32)
    buf = io.BytesIO()
    wavfile.write(buf, 32000, xf)
    got, sr = pkg.wav_decode_mono(buf.getvalue())
    assert sr == 32000 and (got.view(np.uint32) == np.mean(xf, axis=1, dtype=np.float32).view(np.uint32)).all()
    assert pkg.wav_info(buf.getvalue()).is_float == 1


def test_wav_decoder_chunks_truncation_and_errors(pkg):
    x = (np.arange(500)[:, None] * 17 % 2000 - 1000).astype(np.int16)
    img = wave_bytes(x, 16000, 2)
    # an unknown (odd-sized, hence padded) chunk between fmt and data, and one after the data
    fmt_end = img.index(b"data")
    extra = b"LIST" + struct.pack("<I", 5) + b"abcde" + b"\0"
    img2 = img[:fmt_end] + extra + img[fmt_end:] + b"junk" + struct.pack("<I", 2) + b"zz"
    img2 = img2[:4] + struct.pack("<I", len(img2) - 8) + img2[8:]
    got, sr = pkg.wav_decode_mono(img2)
    assert sr == 16000 and (got == x[:, 0].astype(np.float32) / np.float32(32768)).all()
    # WAVE_FORMAT_EXTENSIBLE around the same samples
    fmt = struct.pack("<HHIIHHHHIH14s", 0xFFFE, 1, 16000, 32000, 2, 16, 22, 16, 4, 1, b"\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71")
    data = x.tobytes()
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", len(data)) + data
    got, _ = pkg.wav_decode_mono(b"RIFF" + struct.pack("<I", len(body)) + body)
    assert (got == x[:, 0].astype(np.float32) / np.float32(32768)).all()
    # a data chunk that claims more than the file holds (a recorder that was cut off): what is there is decoded
    cut = img[:len(img) - 301]
    got, _ = pkg.wav_decode_mono(cut)
    assert len(got) == (500 * 2 - 301) // 2 and (got == x[:len(got), 0].astype(np.float32) / np.float32(32768)).all()
    for bad in (b"", b"RIFF", img[:20], b"RIFX" + img[4:], img[:12] + b"data" + struct.pack("<I", 4) + b"\0\0\0\0",
                img.replace(struct.pack("<HH", 1, 1), struct.pack("<HH", 85, 1), 1)):                     # format tag 85 = MP3
        with pytest.raises(pkg.KwsError) as e:
            pkg.wav_decode_mono(bad)
        assert e.value.code in (-20, -18)
    assert pkg.resample_length(22050, 22050, 16000) == 16000 and pkg.resample_length(1000, 44100, 16000) == 363 and pkg.resample_length(5, 16000, 16000) == 5


def band_limited(t_sec, freqs, amps, phases):
    return sum(a * np.sin(2 * np.pi * f * t_sec + p) for f, a, p in zip(freqs, amps, phases))


@pytest.mark.gpu
@pytest.mark.parametrize("sr_in", [8000, 11025, 22050, 32000, 44100, 48000])
def test_resampler_against_the_analytic_signal_and_scipy(sr_in, pkg):
    """Band-limited test signals (sines below 0.4 x the lower Nyquist rate): the resampled values must equal the signal evaluated at the
    new sample times, away from the edges where the filter sees the signal's start / end; scipy.signal.resample_poly (another filter
    design) must agree to its own accuracy."""
    import torch
    from scipy.signal import resample_poly
    rng = np.random.default_rng(sr_in)
    sr_out, n_in = 16000, int(1.3 * sr_in)
    top = 0.4 * min(sr_in, sr_out)
    freqs, amps, phases = rng.uniform(50, top, 6), rng.uniform(0.05, 0.15, 6), rng.uniform(0, 6.28, 6)
    x = band_limited(np.arange(n_in) / sr_in, freqs, amps, phases).astype(np.float32)
    n_out = pkg.resample_length(n_in, sr_in, sr_out)
    d_in = torch.from_numpy(x).to("cuda:0")
    d_out = torch.zeros(n_out, dtype=torch.float32, device="cuda:0")
    pkg.resample_device(d_in.data_ptr(), n_in, sr_in, d_out.data_ptr(), n_out, sr_out, flags=pkg.RESAMPLE_EXACT_POSITIONS)
    torch.cuda.synchronize()
    y = d_out.cpu().numpy()
    want = band_limited(np.arange(n_out) / sr_out, freqs, amps, phases)
    edge = 200
    err = np.abs(y - want)[edge:-edge].max()
    # the default (the reference's behaviour: resampy's truncated integer table step) is the coarser one at non-integer ratios
    d_ref = torch.zeros(n_out, dtype=torch.float32, device="cuda:0")
    pkg.resample_device(d_in.data_ptr(), n_in, sr_in, d_ref.data_ptr(), n_out, sr_out)
    torch.cuda.synchronize()
    err_ref = np.abs(d_ref.cpu().numpy() - want)[edge:-edge].max()
    print("\n%d -> %d Hz: reference-behaviour resampler: max |resampled - analytic| %.3g" % (sr_in, sr_out, err_ref))
    assert err_ref <= 5e-3
    g = np.gcd(sr_in, sr_out)
    ys = resample_poly(x.astype(np.float64), sr_out // g, sr_in // g)[:n_out]
    err_scipy = np.abs(ys - want[:len(ys)])[edge:-edge].max()
    print("\n%d -> %d Hz: max |resampled - analytic| %.3g (scipy.signal.resample_poly: %.3g)" % (sr_in, sr_out, err, err_scipy))
    assert err <= 5e-6                                   # measured 5e-8 .. 7e-7: float32 output of a sum carried in double
    assert np.abs(y[edge:-edge] - ys[edge:len(y) - edge]).max() <= err_scipy + 5e-6      # scipy's default polyphase filter is the coarser of the two
    with pytest.raises(pkg.KwsError):
        pkg.resample_device(d_in.data_ptr(), n_in, sr_in, d_out.data_ptr(), n_out + 1, sr_out)


def resampy_loop(x, sr_in, sr_out):
    """The published loop of resampy 0.2's resample_f (interpn.py) with the "kaiser_best" filter, as librosa.load(sr = ...) ran it in the
    reference's day, followed by librosa's fix_length -- restated from the published sources for this test (PARITY UNPINNED: neither package is
    installable here).  float64 table built from the filter's published parameters; a float32 output array."""
    from scipy.special import i0
    num_zeros, precision, rolloff, beta = 64, 512, 0.9475937167399596, 14.769656459379492
    n = num_zeros * precision
    sinc_win = rolloff * np.sinc(rolloff * np.linspace(0, num_zeros, num=n + 1, endpoint=True))
    r = np.arange(n + 1) / n
    taper = i0(beta * np.sqrt(np.maximum(0.0, 1.0 - r * r))) / i0(beta)          # scipy.signal.kaiser(2 n + 1, beta)[n:]
    interp_win = taper * sinc_win
    ratio = float(sr_out) / sr_in
    if ratio < 1:
        interp_win = interp_win * ratio
    interp_delta = np.zeros_like(interp_win)
    interp_delta[:-1] = np.diff(interp_win)
    y = np.zeros(int(len(x) * ratio), np.float32)
    scale = min(1.0, ratio)
    time_increment = 1.0 / ratio
    index_step = int(scale * precision)
    time_register = 0.0
    nwin, n_orig = len(interp_win), len(x)
    for t in range(len(y)):
        nn = int(time_register)
        frac = scale * (time_register - nn)
        index_frac = frac * precision
        offset = int(index_frac)
        eta = index_frac - offset
        for i in range(min(nn + 1, (nwin - offset) // index_step)):
            weight = interp_win[offset + i * index_step] + eta * interp_delta[offset + i * index_step]
            y[t] += weight * x[nn - i]
        frac = scale - frac
        index_frac = frac * precision
        offset = int(index_frac)
        eta = index_frac - offset
        for k in range(min(n_orig - nn - 1, (nwin - offset) // index_step)):
            weight = interp_win[offset + k * index_step] + eta * interp_delta[offset + k * index_step]
            y[t] += weight * x[nn + k + 1]
        time_register += time_increment
    n_fix = int(np.ceil(len(x) * ratio))                                          # librosa.resample: util.fix_length(y_hat, n_samples)
    return np.concatenate([y, np.zeros(n_fix - len(y), np.float32)])


@pytest.mark.gpu
@pytest.mark.parametrize("sr_in", [8000, 22050, 44100])
def test_default_resampler_follows_resampys_published_loop(sr_in, pkg):
    """kws_resample_device (no flags) = the reference's behaviour: resampy's integer table stepping with one
    interpolation factor per wing, float32 accumulation, floor(n ratio) samples + fix_length's zero padding -- held to the loop restated
    above on a short signal, sample for sample within float32 rounding of the sum."""
    import torch
    rng = np.random.default_rng(sr_in)
    n_in = 1501
    x = (rng.standard_normal(n_in) * 0.2).astype(np.float32)
    want = resampy_loop(x, sr_in, 16000)
    n_out = pkg.resample_length(n_in, sr_in, 16000)
    assert n_out == len(want)
    d_in = torch.from_numpy(x).to("cuda:0")
    d_out = torch.full((n_out,), 7.0, dtype=torch.float32, device="cuda:0")
    pkg.resample_device(d_in.data_ptr(), n_in, sr_in, d_out.data_ptr(), n_out, 16000)
    torch.cuda.synchronize()
    y = d_out.cpu().numpy()
    n_valid = int(n_in * 16000.0 / sr_in)
    assert (y[n_valid:] == 0).all()                                               # fix_length's padding, not a computed sample
    d = np.abs(y - want).max()
    print("\n%d -> 16000 Hz, %d samples: max |kernel - resampy's loop| = %.3g (%d of %d identical)" % (sr_in, n_out, d, int((y == want).sum()), n_out))
    assert d <= 2e-6                          # same taps, same order, float32 accumulation on both sides; the time register differs by ~1e-12 relative


@pytest.mark.gpu
def test_wav_file_to_scores_end_to_end(pkg, oracle):
    """A 22.05 kHz stereo PCM16 WAV of a 0.7 s word, as a data-set tool would feed it: decode + mono (host), resample to 16 kHz (GPU), pad
    to one second and mix with a background window (kws_mix_audio_device), classify.  The same PCM through the oracle gives the same
    scores bit for bit (the ingestion steps only produce the PCM; parity of the path itself is test_gpu_parity.py's)."""
    import torch
    from kws_testlib import OracleModel
    rng = np.random.default_rng(5)
    sr_in, n_in = 22050, int(0.7 * 22050)
    t = np.arange(n_in) / sr_in
    word = 0.3 * np.sin(2 * np.pi * 440 * t) * np.hanning(n_in) + 0.1 * np.sin(2 * np.pi * 1800 * t)
    st = np.stack([word * 0.9, word * 1.1], axis=1)
    img = wave_bytes(np.clip(np.rint(st * 32767), -32768, 32767).astype(np.int64), sr_in, 2)
    mono, sr = pkg.wav_decode_mono(img)
    assert sr == sr_in and len(mono) == n_in
    n_out = pkg.resample_length(n_in, sr_in, 16000)
    d_in = torch.from_numpy(mono).to("cuda:0")
    words = torch.zeros((1, 16000), dtype=torch.float32, device="cuda:0")
    pkg.resample_device(d_in.data_ptr(), n_in, sr_in, words.data_ptr(), n_out, 16000)
    same = torch.zeros(n_in, dtype=torch.float32, device="cuda:0")
    pkg.resample_device(d_in.data_ptr(), n_in, sr_in, same.data_ptr(), n_in, sr_in)          # equal rates: a copy, as librosa.load leaves it
    noise = torch.from_numpy((rng.standard_normal(3 * 16000) * 0.02).astype(np.float32)).to("cuda:0")
    lens = torch.tensor([n_out], dtype=torch.int32, device="cuda:0")
    start = torch.tensor([1234], dtype=torch.int32, device="cuda:0")
    pcm = torch.zeros((1, 16000), dtype=torch.int16, device="cuda:0")
    pkg.mix_audio_device(words.data_ptr(), lens.data_ptr(), 16000, noise.data_ptr(), noise.numel(), start.data_ptr(), 1.0, 0.5, 1, 16000, pcm.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(same, d_in)
    host = pcm.cpu().numpy()
    assert np.abs(host[0, :n_out]).max() > 2000 and np.abs(host[0, n_out + 100:]).max() < 1500     # the word, then background only
    path = os.path.join(MODELS, "l476_no_yes.kwsm")
    gm = pkg.Model(path, device=0)
    s = torch.zeros((1, gm.n_labels), dtype=torch.float32, device="cuda:0")
    gm.run_classifier_batch_device(pcm.data_ptr(), 1, s.data_ptr())
    torch.cuda.synchronize()
    so = OracleModel(oracle, path).run_batch(host)
    assert (s.cpu().numpy().view(np.uint32) == so.view(np.uint32)).all()
    gm.close()
