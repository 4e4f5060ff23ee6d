#!/usr/bin/env python3
"""Golden vectors for the step BEFORE the hot path: mix_audio of /root/reference/dataset-curation.py:93-137 followed by
the script's own sf.write(..., subtype = "PCM_16") (:345-348), run from the REFERENCE ITSELF -> tests/golden/mix_audio.npz.

The script needs librosa (+ resampy) and soundfile.  Neither is installable in the build container (no package index), so this tool
says so and exits with status 3 there; the row stays PARITY UNPINNED until it has run somewhere that has them --
nothing of the Python reference travels: only the .npz (inputs as PCM16 arrays, the start offsets the script drew, its output samples).

The script runs argparse and its whole curation loop at import time (:146-240 and below), so it is not imported: the text above its
"# Main" banner (the imports and mix_audio) is compiled and executed as it stands -- nothing is copied into this repository.

    python tools/make_golden_mix.py [/root/reference/dataset-curation.py]
"""
import os
import random
import sys
import tempfile
import wave

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/dataset-curation.py"


def write_wav(path, pcm, sr):
    with wave.open(path, "wb") as w:
        w.setnchannels(1 if pcm.ndim == 1 else pcm.shape[1])
        w.setsampwidth(2)
        w.setframerate(sr)
        w.writeframes(np.ascontiguousarray(pcm, "<i2").tobytes())


def read_wav(path):
    with wave.open(path, "rb") as w:
        assert w.getsampwidth() == 2
        return np.frombuffer(w.readframes(w.getnframes()), "<i2").reshape(-1, w.getnchannels())[:, 0].copy(), w.getframerate()


def main():
    missing = []
    for mod in ("librosa", "soundfile"):
        try:
            __import__(mod)
        except Exception as e:                       # ImportError, or a broken native dependency
            missing.append("%s (%s)" % (mod, e.__class__.__name__))
    if missing:
        print("make_golden_mix.py: cannot run the reference's mix_audio here -- missing: %s.\n"
              "tests/golden/mix_audio.npz was NOT written; kws_mix_audio_device / kws_resample_device / kws_wav_decode_mono stay PARITY UNPINNED."
              % ", ".join(missing), file=sys.stderr)
        return 3
    if not os.path.exists(SCRIPT):
        print("make_golden_mix.py: %s not found" % SCRIPT, file=sys.stderr)
        return 3
    src = open(SCRIPT).read()
    cut = src.index("# Main")
    cut = src.rfind("#####", 0, cut)                 # the banner line above "# Main"
    ns = {"__name__": "dataset_curation_functions"}
    exec(compile(src[:cut], SCRIPT, "exec"), ns)     # the script's imports and function definitions, as they stand
    mix_audio, sf = ns["mix_audio"], ns["sf"]
    rng = np.random.default_rng(2024)
    out = {}
    cases = []
    with tempfile.TemporaryDirectory() as td:
        k = 0
        for sr_in in (16000, 22050, 44100, 8000):
            for word_len_s, word_vol, bg_vol in ((0.7, 1.0, 0.1), (1.3, 0.5, 1.0), (1.0, 1.0, 1.0), (0.6, 2.5, 2.5)):
                t = np.arange(int(word_len_s * sr_in)) / sr_in
                word = (0.5 * np.sin(2 * np.pi * rng.uniform(200, 1500) * t) * np.hanning(len(t)) + 0.02 * rng.standard_normal(len(t)))
                bg = 0.3 * rng.standard_normal(int(3.0 * sr_in))
                wp, bp, op = (os.path.join(td, "%s%d.wav" % (n, k)) for n in ("word", "bg", "out"))
                wpcm, bpcm = np.clip(np.rint(word * 32767), -32768, 32767).astype(np.int16), np.clip(np.rint(bg * 32767), -32768, 32767).astype(np.int16)
                write_wav(wp, wpcm, sr_in)
                write_wav(bp, bpcm, sr_in)
                drawn = []
                real_randint = random.randint
                random.randint = lambda a, b: drawn.append(real_randint(a, b)) or drawn[-1]      # record the script's draw
                random.seed(1000 + k)
                try:
                    waveform = mix_audio(word_path=wp, bg_path=bp, word_vol=word_vol, bg_vol=bg_vol, sample_time=1.0, sample_rate=16000)
                finally:
                    random.randint = real_randint
                sf.write(op, waveform, 16000, subtype="PCM_16")                                  # the script's own call (:345-348)
                pcm, sr = read_wav(op)
                assert sr == 16000
                out["word_%d" % k], out["bg_%d" % k], out["out_%d" % k] = wpcm, bpcm, pcm
                out["mixed_float_%d" % k] = np.asarray(waveform, np.float64)
                cases.append((sr_in, word_vol, bg_vol, drawn[0] if drawn else -1))
                k += 1
    out["cases"] = np.array(cases, np.float64)
    import librosa
    import soundfile
    out["versions"] = np.array(["librosa %s" % librosa.__version__, "soundfile %s" % soundfile.__version__, "numpy %s" % np.__version__])
    dst = os.path.join(ROOT, "tests", "golden", "mix_audio.npz")
    np.savez_compressed(dst, **out)
    print("wrote %s (%d cases)" % (dst, len(cases)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
