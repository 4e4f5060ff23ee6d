"""ctypes bindings used by the tests: the C oracle (oracle/libkws_oracle.so), the compiled
reference (oracle/_ref/libei_ref_l476.so, only where it has been built) and small helpers.

Test infrastructure only -- nothing here is imported by the product package.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# KWS_ORACLE_SO: another build of oracle/kws_oracle.c (tests/test_sanitizers.py points it at the ASan + UBSan build)
ORACLE_SO = os.environ.get("KWS_ORACLE_SO") or os.path.join(ROOT, "oracle", "libkws_oracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libei_ref_l476.so")
REF_QFB_SO = os.path.join(ROOT, "oracle", "_ref", "libei_ref_l476_qfb.so")      # the same sources with EIDSP_QUANTIZE_FILTERBANK at the SDK's default (1)
MODELS = os.path.join(ROOT, "models")
GOLDEN = os.path.join(ROOT, "tests", "golden")
CLIP_LEN = 16000


class MfccConfig(C.Structure):
    _fields_ = [("num_cepstral", C.c_int), ("frame_length", C.c_float), ("frame_stride", C.c_float),
                ("num_filters", C.c_int), ("fft_length", C.c_int), ("win_size", C.c_int),
                ("low_frequency", C.c_int), ("high_frequency", C.c_int), ("pre_cof", C.c_float),
                ("pre_shift", C.c_int), ("sampling_frequency", C.c_int), ("quantize_filterbank", C.c_int)]

    def copy(self, **kw):
        c = MfccConfig()
        for f, _ in self._fields_:
            setattr(c, f, kw.get(f, getattr(self, f)))
        return c


def L476_CONFIG():
    return MfccConfig(13, 0.02, 0.02, 32, 256, 101, 300, 4000, 0.98, 1, 16000)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def build_oracle():
    if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(
            os.path.join(ROOT, "oracle", "kws_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"])


class Oracle:
    """The plain-C restatement (oracle/kws_oracle.c)."""

    def __init__(self):
        build_oracle()
        L = self.L = C.CDLL(ORACLE_SO)
        L.kwso_log.restype = C.c_float
        L.kwso_log.argtypes = [C.c_float]
        L.kwso_frequency_to_mel.restype = C.c_float
        L.kwso_frequency_to_mel.argtypes = [C.c_float]
        L.kwso_mel_to_frequency.restype = C.c_float
        L.kwso_mel_to_frequency.argtypes = [C.c_float]
        L.kwso_num_frames.argtypes = [C.c_size_t, C.POINTER(MfccConfig)]
        L.kwso_filterbanks.argtypes = [C.POINTER(MfccConfig), C.c_void_p]
        L.kwso_preemphasis.argtypes = [C.c_void_p, C.c_size_t, C.c_float, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p]
        L.kwso_rfft_complex.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.kwso_power_spectrum.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
        L.kwso_mfe.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(MfccConfig), C.c_void_p, C.c_void_p]
        L.kwso_dct2_ortho.argtypes = [C.c_void_p, C.c_int]
        L.kwso_mfcc_nocmvn.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(MfccConfig), C.c_void_p]
        L.kwso_cmvnw.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.kwso_cmvnw_scale.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.kwso_extract_mfe.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(MfccConfig), C.c_void_p]
        L.kwso_normalize.argtypes = [C.c_void_p, C.c_size_t]
        L.kwso_extract_mfcc.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(MfccConfig), C.c_void_p]
        L.kwso_srdhm.restype = C.c_int32
        L.kwso_srdhm.argtypes = [C.c_int32, C.c_int32]
        L.kwso_rdivpot.restype = C.c_int32
        L.kwso_rdivpot.argtypes = [C.c_int32, C.c_int]
        L.kwso_mbqm.restype = C.c_int32
        L.kwso_mbqm.argtypes = [C.c_int32, C.c_int32, C.c_int]
        L.kwso_quantize_multiplier.argtypes = [C.c_double, C.POINTER(C.c_int32), C.POINTER(C.c_int)]
        L.kwso_exp_on_negative_values_q5_26.restype = C.c_int32
        L.kwso_exp_on_negative_values_q5_26.argtypes = [C.c_int32]
        L.kwso_one_over_one_plus_x.restype = C.c_int32
        L.kwso_one_over_one_plus_x.argtypes = [C.c_int32]
        L.kwso_model_load.restype = C.c_void_p
        L.kwso_model_load.argtypes = [C.c_void_p, C.c_size_t]
        L.kwso_model_free.argtypes = [C.c_void_p]
        for f in ("label_count", "feature_count", "raw_sample_count", "tensor_count", "dsp_block"):
            getattr(L, "kwso_model_" + f).argtypes = [C.c_void_p]
        L.kwso_model_label.restype = C.c_char_p
        L.kwso_model_label.argtypes = [C.c_void_p, C.c_int]
        L.kwso_model_tensor_bytes.argtypes = [C.c_void_p, C.c_int]
        L.kwso_model_mfcc_config.argtypes = [C.c_void_p, C.POINTER(MfccConfig)]
        L.kwso_quantize_input.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.kwso_nn_invoke.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.kwso_dequantize_output.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.kwso_run_inference.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.kwso_run_classifier.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.kwso_run_classifier_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p,
                                                C.c_void_p, C.c_void_p]
        L.kwso_time_run_classifier.restype = C.c_double
        L.kwso_time_run_classifier.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p]
        L.kwso_synth_fill.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.kwso_mix_audio.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_void_p]
        L.kwso_model_is_float.argtypes = [C.c_void_p]
        L.kwso_nn_invoke_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.kwso_continuous_create.restype = C.c_void_p
        L.kwso_continuous_create.argtypes = [C.c_void_p]
        L.kwso_continuous_free.argtypes = [C.c_void_p]
        L.kwso_continuous_init.argtypes = [C.c_void_p]
        L.kwso_continuous_step.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]

    def mix_audio(self, word, noise_window, word_vol, bg_vol, n):
        """mix_audio of dataset-curation.py:93-137 (PARITY UNPINNED, see oracle/kws_oracle.h); word / noise_window may be None"""
        out = np.zeros(n, np.int16)
        w = None if word is None else np.ascontiguousarray(word, np.float32)
        b = None if noise_window is None else np.ascontiguousarray(noise_window, np.float32)
        self.L.kwso_mix_audio(None if w is None else _ptr(w), 0 if w is None else w.size, None if b is None else _ptr(b), word_vol, bg_vol, n, _ptr(out))
        return out

    # ---- clips
    def synth(self, seed, first, n, clip_len=CLIP_LEN):
        out = np.empty((n, clip_len), np.int16)
        self.L.kwso_synth_fill(seed, first, n, clip_len, _ptr(out))
        return out

    # ---- DSP
    def num_frames(self, n, cfg):
        return self.L.kwso_num_frames(n, C.byref(cfg))

    def quantize_zero_one(self, v):
        self.L.kwso_quantize_zero_one.restype = C.c_float
        self.L.kwso_quantize_zero_one.argtypes = [C.c_float]
        return self.L.kwso_quantize_zero_one(float(v))

    def filterbanks(self, cfg):
        out = np.zeros((cfg.fft_length // 2 + 1, cfg.num_filters), np.float32)
        rc = self.L.kwso_filterbanks(C.byref(cfg), _ptr(out))
        assert rc == 0, rc
        return out

    def preemphasis(self, pcm, cof, shift, offset, length):
        pcm = np.ascontiguousarray(pcm, np.int16)
        out = np.zeros(length, np.float32)
        rc = self.L.kwso_preemphasis(_ptr(pcm), pcm.size, cof, shift, offset, length, _ptr(out))
        assert rc == 0, rc
        return out

    def rfft_complex(self, x):
        x = np.ascontiguousarray(x, np.float32)
        out = np.zeros((x.size // 2 + 1, 2), np.float32)
        rc = self.L.kwso_rfft_complex(_ptr(x), x.size, _ptr(out))
        assert rc == 0, rc
        return out

    def power_spectrum(self, frame, fft_length):
        frame = np.ascontiguousarray(frame, np.float32)
        out = np.zeros(fft_length // 2 + 1, np.float32)
        rc = self.L.kwso_power_spectrum(_ptr(frame), frame.size, _ptr(out), fft_length)
        assert rc == 0, rc
        return out

    def mfe(self, pcm, cfg):
        pcm = np.ascontiguousarray(pcm, np.int16)
        nf = self.num_frames(pcm.size, cfg)
        feat = np.zeros((nf, cfg.num_filters), np.float32)
        en = np.zeros(nf, np.float32)
        rc = self.L.kwso_mfe(_ptr(pcm), pcm.size, C.byref(cfg), _ptr(feat), _ptr(en))
        assert rc == 0, rc
        return feat, en

    def dct2_ortho(self, v):
        v = np.array(v, np.float32)
        rc = self.L.kwso_dct2_ortho(_ptr(v), v.size)
        assert rc == 0, rc
        return v

    def mfcc_nocmvn(self, pcm, cfg):
        pcm = np.ascontiguousarray(pcm, np.int16)
        nf = self.num_frames(pcm.size, cfg)
        out = np.zeros((nf, cfg.num_cepstral), np.float32)
        rc = self.L.kwso_mfcc_nocmvn(_ptr(pcm), pcm.size, C.byref(cfg), _ptr(out))
        assert rc == 0, rc
        return out

    def cmvnw(self, m, win_size, var_norm=True):
        m = np.array(m, np.float32)
        rc = self.L.kwso_cmvnw(_ptr(m), m.shape[0], m.shape[1], win_size, int(var_norm))
        assert rc == 0, rc
        return m

    def cmvnw_scale(self, m, win_size, var_norm=False, scale=True):
        """L432 SDK copy: processing::cmvnw(matrix, win_size, variance_normalization, scale)."""
        m = np.array(m, np.float32)
        rc = self.L.kwso_cmvnw_scale(_ptr(m), m.shape[0], m.shape[1], win_size, int(var_norm), int(scale))
        assert rc == 0, rc
        return m

    def extract_mfe(self, pcm, cfg):
        """L432 SDK copy: extract_mfe_features (no pre-emphasis whatever cfg.pre_cof says)."""
        pcm = np.ascontiguousarray(pcm, np.int16)
        c = MfccConfig.from_buffer_copy(cfg)
        c.pre_cof = 0.0
        nf = self.num_frames(pcm.size, c)
        out = np.zeros(nf * c.num_filters, np.float32)
        rc = self.L.kwso_extract_mfe(_ptr(pcm), pcm.size, C.byref(c), _ptr(out))
        assert rc == 0, rc
        return out

    def extract_mfcc(self, pcm, cfg):
        pcm = np.ascontiguousarray(pcm, np.int16)
        nf = self.num_frames(pcm.size, cfg)
        out = np.zeros(nf * cfg.num_cepstral, np.float32)
        rc = self.L.kwso_extract_mfcc(_ptr(pcm), pcm.size, C.byref(cfg), _ptr(out))
        assert rc == 0, rc
        return out

    def quantize_multiplier(self, m):
        q, s = C.c_int32(), C.c_int()
        self.L.kwso_quantize_multiplier(m, C.byref(q), C.byref(s))
        return q.value, s.value


class OracleModel:
    def __init__(self, oracle, path):
        self.o = oracle
        self.blob = open(path, "rb").read()
        self.h = oracle.L.kwso_model_load(self.blob, len(self.blob))
        assert self.h, "kwso_model_load failed for %s" % path
        L = oracle.L
        self.n_labels = L.kwso_model_label_count(self.h)
        self.labels = [L.kwso_model_label(self.h, i).decode() for i in range(self.n_labels)]
        self.n_features = L.kwso_model_feature_count(self.h)
        self.raw_sample_count = L.kwso_model_raw_sample_count(self.h)
        self.cfg = MfccConfig()
        L.kwso_model_mfcc_config(self.h, C.byref(self.cfg))
        self.tensor_bytes = [L.kwso_model_tensor_bytes(self.h, i) for i in range(L.kwso_model_tensor_count(self.h))]

    def quantize_input(self, feat):
        feat = np.ascontiguousarray(feat, np.float32)
        q = np.zeros(self.n_features, np.int8)
        self.o.L.kwso_quantize_input(self.h, _ptr(feat), _ptr(q))
        return q

    def nn_invoke(self, q, taps=False):
        q = np.ascontiguousarray(q, np.int8)
        out = np.zeros(self.n_labels, np.int8)
        tp = np.zeros(sum(self.tensor_bytes), np.int8) if taps else None
        rc = self.o.L.kwso_nn_invoke(self.h, _ptr(q), _ptr(out), _ptr(tp) if taps else None)
        assert rc == 0, rc
        if not taps:
            return out
        offs = np.cumsum([0] + self.tensor_bytes)
        return out, [tp[offs[i]:offs[i + 1]] for i in range(len(self.tensor_bytes))]

    def nn_invoke_f32(self, feat, taps=False):
        """float graph; taps -> list of float32 arrays, one per tensor (tensor-id order)"""
        feat = np.ascontiguousarray(feat, np.float32)
        out = np.zeros(self.n_labels, np.float32)
        tp = np.zeros(sum(self.tensor_bytes) // 4 + 1, np.float32) if taps else None
        rc = self.o.L.kwso_nn_invoke_f32(self.h, _ptr(feat), _ptr(out), _ptr(tp) if taps else None)
        assert rc == 0, rc
        if not taps:
            return out
        offs = np.cumsum([0] + [b // 4 for b in self.tensor_bytes])
        return out, [tp[offs[i]:offs[i + 1]] for i in range(len(self.tensor_bytes))]

    def dequantize(self, out_q):
        out_q = np.ascontiguousarray(out_q, np.int8)
        s = np.zeros(self.n_labels, np.float32)
        self.o.L.kwso_dequantize_output(self.h, _ptr(out_q), _ptr(s))
        return s

    def run_inference(self, feat):
        feat = np.ascontiguousarray(feat, np.float32)
        s = np.zeros(self.n_labels, np.float32)
        rc = self.o.L.kwso_run_inference(self.h, _ptr(feat), _ptr(s))
        assert rc == 0, rc
        return s

    def run_batch(self, pcm, want_features=False):
        pcm = np.ascontiguousarray(pcm, np.int16)
        if pcm.ndim == 1:
            pcm = pcm[None]
        B, n = pcm.shape
        s = np.zeros((B, self.n_labels), np.float32)
        f = np.zeros((B, self.n_features), np.float32)
        q = np.zeros((B, self.n_features), np.int8)
        rc = self.o.L.kwso_run_classifier_batch(self.h, _ptr(pcm), n, B, _ptr(s), _ptr(f), _ptr(q))
        if rc != 0:
            return rc
        return (s, f, q) if want_features else s

    def time_run(self, pcm, iters=1):
        pcm = np.ascontiguousarray(pcm, np.int16)
        chk = C.c_float()
        return self.o.L.kwso_time_run_classifier(self.h, _ptr(pcm), pcm.shape[0], pcm.shape[1], iters, C.byref(chk))


class OracleContinuous:
    """run_classifier_init / run_classifier_continuous restated (oracle/kws_oracle.c)."""

    def __init__(self, model):
        self.m = model
        self.L = model.o.L
        self.h = self.L.kwso_continuous_create(model.h)
        assert self.h

    def init(self):
        self.L.kwso_continuous_init(self.h)

    def step(self, slice_pcm):
        slice_pcm = np.ascontiguousarray(slice_pcm, np.int16)
        s = np.zeros(self.m.n_labels, np.float32)
        produced = C.c_int()
        rc = self.L.kwso_continuous_step(self.h, _ptr(slice_pcm), slice_pcm.size, None, _ptr(s), C.byref(produced))
        return rc, bool(produced.value), s


def have_reference():
    return os.path.exists(REF_SO)


REF432_SO = os.path.join(ROOT, "oracle", "_ref", "libei_ref_l432dsp.so")


class ReferenceL432Dsp:
    """processing.hpp + numpy.hpp of the L432 SDK copy, compiled in place (oracle/ref_l432_dsp.cpp)."""

    def __init__(self):
        L = self.L = C.CDLL(REF432_SO)
        L.eiref432_cmvnw.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.eiref432_normalize.argtypes = [C.c_void_p, C.c_int, C.c_int]

    def cmvnw(self, m, win_size, var_norm, scale):
        m = np.array(m, np.float32)
        rc = self.L.eiref432_cmvnw(_ptr(m), m.shape[0], m.shape[1], win_size, int(var_norm), int(scale))
        assert rc == 0, rc
        return m

    def normalize(self, m):
        m = np.array(m, np.float32)
        rc = self.L.eiref432_normalize(_ptr(m), m.shape[0], m.shape[1])
        assert rc == 0, rc
        return m


class Reference:
    """The unmodified reference SDK compiled by oracle/Makefile (target ref).  qfb: the build with EIDSP_QUANTIZE_FILTERBANK = 1."""

    def __init__(self, qfb=False):
        L = self.L = C.CDLL(REF_QFB_SO if qfb else REF_SO)
        assert L.eiref_quantize_filterbank() == (1 if qfb else 0)
        L.eiref_quantize_zero_one.restype = C.c_float
        L.eiref_quantize_zero_one.argtypes = [C.c_float]
        L.eiref_quantized_table.argtypes = [C.c_void_p, C.c_int]
        L.eiref_label.restype = C.c_char_p
        L.eiref_run_classifier.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        L.eiref_extract_mfcc.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_size_t]
        L.eiref_mfcc_nocmvn.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int,
                                        C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]
        L.eiref_mfe.argtypes = [C.c_void_p, C.c_size_t, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_float, C.c_int, C.c_void_p, C.c_void_p]
        L.eiref_preemphasis.argtypes = [C.c_void_p, C.c_size_t, C.c_float, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p]
        L.eiref_power_spectrum.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
        L.eiref_filterbanks.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.eiref_log.restype = C.c_float
        L.eiref_log.argtypes = [C.c_float]
        L.eiref_frequency_to_mel.restype = C.c_float
        L.eiref_frequency_to_mel.argtypes = [C.c_float]
        L.eiref_mel_to_frequency.restype = C.c_float
        L.eiref_mel_to_frequency.argtypes = [C.c_float]
        L.eiref_dct2_ortho.argtypes = [C.c_void_p, C.c_size_t]
        L.eiref_cmvnw.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.eiref_num_frames.argtypes = [C.c_size_t, C.c_float, C.c_float]
        L.eiref_rfft_complex.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.eiref_run_inference.argtypes = [C.c_void_p, C.c_void_p]
        L.eiref_nn_taps.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.eiref_continuous.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_int), C.c_void_p]
        L.eiref_time_run_classifier.restype = C.c_double
        L.eiref_time_run_classifier.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p]
        L.eiref_graph_run.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        L.eiref_time_graph_classifier.restype = C.c_double
        L.eiref_time_graph_classifier.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p]
        self.n_labels = L.eiref_label_count()
        self.labels = [L.eiref_label(i).decode() for i in range(self.n_labels)]
        self.n_features = L.eiref_feature_count()
        self.tensor_bytes = [L.eiref_tensor_bytes(i) for i in range(L.eiref_tensor_count())]

    def graph_run(self, blob, x):
        """A .kwsm graph through the reference's own TFLite-Micro op registrations (init/prepare/invoke).
        x: the input tensor (int8 or float32).  Returns (output, [every tensor as raw bytes, tensor-id order])."""
        import eon_import
        tens, _, t_in, t_out, _ = eon_import.parse_blob(blob)
        x = np.ascontiguousarray(x)
        np_t = {1: np.float32, 2: np.int32, 9: np.int8}
        out = np.zeros(tens[t_out]["nbytes"] // np.dtype(np_t[tens[t_out]["type"]]).itemsize, np_t[tens[t_out]["type"]])
        taps = np.zeros(sum(t["nbytes"] for t in tens), np.uint8)
        rc = self.L.eiref_graph_run(blob, len(blob), _ptr(x), x.nbytes, _ptr(out), out.nbytes, _ptr(taps))
        assert rc == 0, rc
        offs = np.cumsum([0] + [t["nbytes"] for t in tens])
        return out, [taps[offs[i]:offs[i + 1]].view(np_t[t["type"]]) for i, t in enumerate(tens)]

    def time_graph(self, blob, pcm, iters=1):
        """seconds for iters passes of extract_mfcc_features + the blob's graph (reference op code) over pcm [n][len]"""
        pcm = np.ascontiguousarray(pcm, np.int16)
        chk = C.c_float()
        t = self.L.eiref_time_graph_classifier(blob, len(blob), _ptr(pcm), pcm.shape[0], pcm.shape[1], iters, C.byref(chk))
        assert t >= 0, t
        return t

    # ---- boundary behaviour (debug text, the cancellation hook): oracle/ref_driver.cpp eiref_*_full -------------------------
    def boundary_call(self, kind, data, debug=False, cancel_at=0):
        """kind: 'oneshot' (int16 window) | 'inference' (float features) | 'continuous' (int16 slice).  Returns (rc, polls of the
        cancellation hook, the caller's ei_impulse_result_t byte for byte -- pre-filled with 0xA5 --, label flags, printed text)."""
        L = self.L
        L.eiref_capture.argtypes = [C.c_void_p, C.c_size_t]
        L.eiref_capture_len.restype = C.c_size_t
        fn = {"oneshot": L.eiref_run_classifier_full, "inference": L.eiref_run_inference_full, "continuous": L.eiref_continuous_full}[kind]
        buf = C.create_string_buffer(1 << 16)
        res = np.zeros(L.eiref_result_size(), np.uint8)
        lab = np.zeros(self.n_labels, np.int32)
        L.eiref_capture(buf, len(buf))
        L.eiref_cancel_at(int(cancel_at))
        if kind == "inference":
            x = np.ascontiguousarray(data, np.float32)
            fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
            rc = fn(_ptr(x), int(debug), _ptr(res), _ptr(lab))
        else:
            x = np.ascontiguousarray(data, np.int16)
            fn.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
            rc = fn(_ptr(x), x.size, int(debug), _ptr(res), _ptr(lab))
        n = L.eiref_capture_len()
        polls = L.eiref_cancel_polls()
        L.eiref_capture(None, 0)
        L.eiref_cancel_at(0)
        return rc, polls, res, lab, buf.raw[:n]

    def run_classifier(self, pcm):
        pcm = np.ascontiguousarray(pcm, np.int16)
        s = np.zeros(self.n_labels, np.float32)
        tl, gc = C.c_size_t(), C.c_size_t()
        rc = self.L.eiref_run_classifier(_ptr(pcm), pcm.size, _ptr(s), C.byref(tl), C.byref(gc))
        return rc, s, tl.value, gc.value

    def _cfg_args(self, cfg, with_win):
        a = [cfg.num_cepstral, cfg.frame_length, cfg.frame_stride, cfg.num_filters, cfg.fft_length]
        if with_win:
            a.append(cfg.win_size)
        return a + [cfg.low_frequency, cfg.high_frequency, cfg.pre_cof, cfg.pre_shift]

    def num_frames(self, n, cfg):
        return self.L.eiref_num_frames(n, cfg.frame_length, cfg.frame_stride)

    def extract_mfcc(self, pcm, cfg):
        pcm = np.ascontiguousarray(pcm, np.int16)
        nf = self.num_frames(pcm.size, cfg)
        out = np.zeros(nf * cfg.num_cepstral, np.float32)
        rc = self.L.eiref_extract_mfcc(_ptr(pcm), pcm.size, *self._cfg_args(cfg, True), _ptr(out), out.size)
        assert rc == 0, rc
        return out

    def mfcc_nocmvn(self, pcm, cfg):
        pcm = np.ascontiguousarray(pcm, np.int16)
        nf = self.num_frames(pcm.size, cfg)
        out = np.zeros((nf, cfg.num_cepstral), np.float32)
        rc = self.L.eiref_mfcc_nocmvn(_ptr(pcm), pcm.size, *self._cfg_args(cfg, False), _ptr(out))
        assert rc == 0, rc
        return out

    def mfe(self, pcm, cfg):
        pcm = np.ascontiguousarray(pcm, np.int16)
        nf = self.num_frames(pcm.size, cfg)
        feat = np.zeros((nf, cfg.num_filters), np.float32)
        en = np.zeros(nf, np.float32)
        rc = self.L.eiref_mfe(_ptr(pcm), pcm.size, cfg.frame_length, cfg.frame_stride, cfg.num_filters,
                              cfg.fft_length, cfg.low_frequency, cfg.high_frequency, cfg.pre_cof, cfg.pre_shift,
                              _ptr(feat), _ptr(en))
        assert rc == 0, rc
        return feat, en

    def preemphasis(self, pcm, cof, shift, offset, length):
        pcm = np.ascontiguousarray(pcm, np.int16)
        out = np.zeros(length, np.float32)
        rc = self.L.eiref_preemphasis(_ptr(pcm), pcm.size, cof, shift, offset, length, _ptr(out))
        assert rc == 0, rc
        return out

    def power_spectrum(self, frame, fft_length):
        frame = np.array(frame, np.float32)
        out = np.zeros(fft_length // 2 + 1, np.float32)
        rc = self.L.eiref_power_spectrum(_ptr(frame), frame.size, _ptr(out), fft_length)
        assert rc == 0, rc
        return out

    def filterbanks(self, cfg):
        out = np.zeros((cfg.fft_length // 2 + 1, cfg.num_filters), np.float32)
        hf = cfg.high_frequency if cfg.high_frequency else cfg.sampling_frequency // 2
        rc = self.L.eiref_filterbanks(cfg.num_filters, cfg.fft_length, cfg.low_frequency, hf, _ptr(out))
        assert rc == 0, rc
        return out

    def rfft_complex(self, x):
        x = np.ascontiguousarray(x, np.float32)
        out = np.zeros((x.size // 2 + 1, 2), np.float32)
        rc = self.L.eiref_rfft_complex(_ptr(x), x.size, _ptr(out))
        assert rc == 0, rc
        return out

    def dct2_ortho(self, v):
        v = np.array(v, np.float32)
        rc = self.L.eiref_dct2_ortho(_ptr(v), v.size)
        assert rc == 0, rc
        return v

    def cmvnw(self, m, win_size, var_norm=True):
        m = np.array(m, np.float32)
        rc = self.L.eiref_cmvnw(_ptr(m), m.shape[0], m.shape[1], win_size, int(var_norm))
        assert rc == 0, rc
        return m

    def run_inference(self, feat):
        feat = np.ascontiguousarray(feat, np.float32)
        s = np.zeros(self.n_labels, np.float32)
        rc = self.L.eiref_run_inference(_ptr(feat), _ptr(s))
        assert rc == 0, rc
        return s

    def nn_taps(self, q):
        """all op outputs (tensor id -> bytes) of the int8 graph for input tensor q"""
        q = np.ascontiguousarray(q, np.int8)
        ids = np.arange(len(self.tensor_bytes), dtype=np.int32)
        out = np.zeros(sum(self.tensor_bytes), np.int8)
        sizes = np.zeros(ids.size, np.int32)
        rc = self.L.eiref_nn_taps(_ptr(q), ids.size, _ptr(ids), _ptr(out), _ptr(sizes))
        assert rc == 0, rc
        offs = np.cumsum([0] + self.tensor_bytes)
        return {int(i): out[offs[i]:offs[i + 1]].copy() for i in ids if sizes[i] >= 0}

    def traced(self, fn, *args, cap=4096):
        """fn(*args) with every get_data call the reference makes recorded: (result of fn, int64 [n_calls, 3] = offset, length, return value)."""
        buf = np.zeros((cap, 3), np.int64)
        self.L.eiref_trace_get_data.argtypes = [C.c_void_p, C.c_int]
        self.L.eiref_trace_get_data(_ptr(buf), cap)
        try:
            out = fn(*args)
            n = self.L.eiref_trace_count()
        finally:
            self.L.eiref_trace_get_data(None, 0)
        assert n <= cap, n
        return out, buf[:n].copy()

    def continuous_init(self):
        self.L.eiref_continuous_init()

    def continuous(self, slice_pcm):
        slice_pcm = np.ascontiguousarray(slice_pcm, np.int16)
        s = np.zeros(self.n_labels, np.float32)
        produced = C.c_int()
        tl = C.c_size_t()
        rc = self.L.eiref_continuous(_ptr(slice_pcm), slice_pcm.size, _ptr(s), C.byref(produced), C.byref(tl))
        return rc, bool(produced.value), s, tl.value

    def time_run(self, pcm, iters=1):
        pcm = np.ascontiguousarray(pcm, np.int16)
        chk = C.c_float()
        return self.L.eiref_time_run_classifier(_ptr(pcm), pcm.shape[0], pcm.shape[1], iters, C.byref(chk))


def reference_float_twin(reference, tensors, feat):
    """Run the fp32 twin of the shipped graph through the REFERENCE's float TFLite-Micro kernels,
    called leaf by leaf (oracle/ref_driver.cpp eiref_f32_*).  tensors: constant tensors of the .kwsm (id -> float array).
    Returns (logits[4], scores[4])."""
    L = reference.L
    fp = C.POINTER(C.c_float)
    FLT_MAX = float(np.finfo(np.float32).max)

    def P(a):
        return a.ctypes.data_as(fp)

    def F(x):
        return C.c_float(x)

    x = np.ascontiguousarray(feat, np.float32)
    y1 = np.zeros(49 * 30, np.float32)
    L.eiref_f32_conv(P(x), 1, 49, 13, P(tensors[7]), 30, 1, 7, P(tensors[6]), 3, 0, F(-FLT_MAX), F(FLT_MAX), P(y1), 1, 49)
    y2 = np.zeros(49 * 30, np.float32)
    L.eiref_f32_add_bcast(P(y1), (C.c_int * 4)(1, 1, 49, 30), P(tensors[2]), (C.c_int * 4)(1, 1, 1, 30),
                          (C.c_int * 4)(1, 1, 49, 30), F(0.0), F(FLT_MAX), P(y2))
    y3 = np.zeros(7 * 30, np.float32)
    L.eiref_f32_maxpool(P(y2), 49, 1, 30, 7, 1, 7, 1, F(-FLT_MAX), F(FLT_MAX), P(y3), 7, 1)
    y4 = np.zeros(70, np.float32)
    L.eiref_f32_conv(P(y3), 1, 7, 30, P(tensors[9]), 10, 1, 7, P(tensors[8]), 3, 0, F(-FLT_MAX), F(FLT_MAX), P(y4), 1, 7)
    y5 = np.zeros(70, np.float32)
    L.eiref_f32_add_bcast(P(y4), (C.c_int * 4)(1, 1, 7, 10), P(tensors[3]), (C.c_int * 4)(1, 1, 1, 10),
                          (C.c_int * 4)(1, 1, 7, 10), F(0.0), F(FLT_MAX), P(y5))
    y6 = np.zeros(10, np.float32)
    L.eiref_f32_maxpool(P(y5), 7, 1, 10, 7, 1, 7, 1, F(-FLT_MAX), F(FLT_MAX), P(y6), 1, 1)
    lg = np.zeros(4, np.float32)
    L.eiref_f32_fc(P(y6), 10, P(tensors[5]), 4, P(tensors[4]), F(-FLT_MAX), F(FLT_MAX), P(lg))
    sc = np.zeros(4, np.float32)
    L.eiref_f32_softmax(P(lg), 4, F(1.0), P(sc))
    return lg, sc


def bits(a):
    """view float32 array as uint32 for exact comparison"""
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def special_clips():
    """known-answer / edge-case clips"""
    z = np.zeros(CLIP_LEN, np.int16)
    alt = np.empty(CLIP_LEN, np.int16)
    alt[0::2] = 32767
    alt[1::2] = -32767
    step = np.zeros(CLIP_LEN, np.int16)
    step[8000:] = 20000
    imp = np.zeros(CLIP_LEN, np.int16)
    imp[5000] = 32767
    imp[15999] = -32768
    mn = np.full(CLIP_LEN, -32768, np.int16)
    ramp = (np.arange(CLIP_LEN) * 4 - 32000).astype(np.int16)
    return {"zeros": z, "alternating_fullscale": alt, "step": step, "impulses": imp, "min": mn, "ramp": ramp}


# ---------------------------------------------------------------------------------------------------------------
#  synthetic models of the Edge Impulse 1-D CNN family (random weights / quantisation): tools/synth_model.py
# ---------------------------------------------------------------------------------------------------------------
sys.path.insert(0, os.path.join(ROOT, "tools"))
from synth_model import synth_model_blob  # noqa: E402,F401

# Named synthetic graphs shared by the CPU pins (oracle vs the reference's op registrations), tools/make_golden.py and
# the GPU parity tests.  `dw` / `pw` blocks: DEPTHWISE_CONV_2D / pointwise CONV_2D (BASELINE config 5).
SYNTH_SPECS = {
    "seed1": dict(seed=1),                                                                # shipped shape, random weights
    "seed2": dict(seed=2, ncep=10, win_size=51, high=0, blocks=((16, 5, 7), (8, 3, 7)), n_labels=3),
    "seed4": dict(seed=4, ncep=16, blocks=((32, 8, 7), (16, 8, 7)), n_labels=5, add_bias=False),   # matrix-core limits, even taps
    "seed5": dict(seed=5, ncep=13, blocks=((30, 7, 7), (10, 7, 7)), n_labels=4, conv_bias=True),
    "seed6": dict(seed=6, ncep=12, win_size=13, low=0, high=8000, blocks=((20, 3, 7), (12, 5, 1), (6, 3, 7)), n_labels=2),  # 3 blocks
    "seed7": dict(seed=7, ncep=13, blocks=((40, 7, 7), (10, 7, 7)), n_labels=4),           # 40 channels
    "f40c40": dict(seed=11, num_filters=40, ncep=40, low=300, high=0, blocks=((16, 5, 7), (8, 3, 7)), n_labels=3),
    "dscnn_a": dict(seed=21, blocks=((16, 5, 1), ("dw", 1, 3, 1, 1), ("pw", 24, 1), ("dw", 1, 5, 7, 0), ("pw", 8, 3), (8, 3, 7)), n_labels=5),
    # un-pooled CONV_2D blocks of every row width: 13 -> 16-byte rows with an even tap count and two 32-row tiles, 24 -> 32-byte
    # rows but 40 outputs (stays on v_dot4), 40 -> 64-byte rows; then a pooled tail
    "mfma_mix": dict(seed=23, ncep=13, blocks=((24, 4, 1), (40, 2, 1), (32, 3, 1), (8, 3, 7)), n_labels=4),
    # one 32-row tile with 64-byte rows (40 channels after a pooled block), then an even-tap pooled tail
    "mfma_mix2": dict(seed=24, ncep=13, blocks=((40, 3, 7), (16, 3, 1), (8, 2, 7)), n_labels=3),
    # the shape of Edge Impulse's default 1-D conv export: Conv(8, k3) - MaxPool(2) - Conv(16, k3) - MaxPool(2) - Dense over
    # 12 x 16 = 192 inputs (a FULLY_CONNECTED input far longer than the shipped models' 10)
    "ei_default": dict(seed=30, ncep=13, blocks=((8, 3, -2), (16, 3, -2)), n_labels=4),           # negative pool: VALID (49 -> 24 -> 12)
    "ei_default_same": dict(seed=32, ncep=13, blocks=((8, 3, 2), (16, 3, 2)), n_labels=4),           # SAME pooling, ragged last windows: 49 -> 25 -> 13
    "ei_default40": dict(seed=31, num_filters=40, ncep=40, low=300, high=0, blocks=((16, 3, -2), (32, 3, -2), (32, 3, 1)), n_labels=12),
    "dscnn_b": dict(seed=22, ncep=10, blocks=(("dw", 2, 7, 7, 3), ("pw", 12, 1), ("dw", 1, 3, 7, 1)), n_labels=3),
    # BASELINE config 5 as worded: 49x40 MFCC, deeper depthwise-separable CNN, 10 keywords (+ noise/unknown); synthetic weights
    # (logit_std: calibrated activation ranges and head, tools/synth_model.py calibrate_head -- a model whose softmax is not saturated)
    "cfg5_dscnn": dict(seed=50, num_filters=40, ncep=40, low=300, high=0, n_labels=12, logit_std=3.5,
                       blocks=((32, 5, 1), ("dw", 1, 5, 1, 1), ("pw", 32, 1), ("dw", 1, 5, 7, 1), ("pw", 32, 1), ("dw", 1, 3, 7, 1), ("pw", 12, 0))),
}


def random_graph_spec(seed):
    """A random member of the 1-D conv graph family within the kernels' documented limits (for the fuzz tests): 1-4 blocks of
    CONV_2D / depthwise / pointwise, taps 1-8, channels 4-48, pooling 1 / 2 / 3 / 7 with SAME (positive) or VALID (negative)
    padding, 2-12 labels.  Returns synth_model_blob keyword arguments, or None when the draw leaves the limits."""
    rng = np.random.default_rng(1000 + seed)
    ncep = int(rng.choice([10, 13, 16, 20]))
    w, c = 49, ncep
    blocks = []
    for b in range(int(rng.integers(1, 5))):
        kind = rng.choice(["conv", "conv", "conv", "dw", "pw"]) if b else "conv"
        pool = int(rng.choice([1, 1, 2, -2, 3, -3, 7]))
        if kind == "conv":
            oc = int(rng.choice([4, 8, 12, 16, 24, 30, 32, 40, 48]))
            blocks.append((oc, int(rng.integers(1, 9)), pool))
            c = oc
        elif kind == "dw":
            mult = int(rng.choice([1, 1, 2]))
            if c * mult > 64:
                return None
            blocks.append(("dw", mult, int(rng.choice([3, 5, 7])), pool, int(rng.choice([0, 1, 3]))))
            c = c * mult
        else:
            oc = int(rng.choice([8, 12, 16, 32]))
            blocks.append(("pw", oc, int(rng.choice([0, 1]))))
            c = oc
            pool = 1
        if pool not in (0, 1):
            p = abs(pool)
            w = w // p if pool < 0 else (w + p - 1) // p
        if w < 1:
            return None
    n_labels = int(rng.integers(2, 13))
    if w * c > 1024 or w * c * n_labels * 4 > 32768:
        return None
    return dict(seed=500 + seed, ncep=ncep, blocks=tuple(blocks), n_labels=n_labels, conv_bias=bool(rng.integers(0, 2)))


def random_dsp_spec(seed):
    """A random MFCC configuration the GPU kernel is built for (fft 256, 32 or 40 filters, 49 frames): cepstra, CMVN window and
    the filterbank's frequency range vary.  Returns (MfccConfig keyword overrides, synth_model_blob keyword arguments)."""
    rng = np.random.default_rng(7000 + seed)
    nf = int(rng.choice([32, 40]))
    ncep = int(rng.integers(2, nf + 1))
    win = int(rng.choice([17, 21, 33, 51, 75, 101, 121, 137])) if ncep > 16 and nf == 40 else int(rng.choice([13, 15, 33, 51, 101, 137]))
    low = int(rng.choice([0, 50, 300, 600]))
    high = int(rng.choice([0, 3500, 4000, 6000]))
    cfg_kw = dict(num_filters=nf, num_cepstral=ncep, win_size=win, low_frequency=low, high_frequency=high)
    blob_kw = dict(seed=900 + seed, num_filters=nf, ncep=ncep, win_size=win, low=low, high=high, blocks=((8, 3, 7), (4, 3, 7)), n_labels=3)
    return cfg_kw, blob_kw
