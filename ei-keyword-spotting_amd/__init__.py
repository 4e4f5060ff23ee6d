"""ei-keyword-spotting_amd -- MI355X-native drop-in for the Edge Impulse `run_classifier()` hot path.

The product is the C-ABI shared library `libkws_mi355x.so` built from `csrc/` (hand-written HIP kernels for
gfx950 + the C host side; headers in `include/kws/`).  This module is only a thin ctypes binding over that C ABI
for Python callers, the tests and `bench.py`; it contains no arithmetic of its own and has NO CPU fallback: if the
library (or a GPU) is missing, loading / creating a model raises.

Because the directory name carries a hyphen, import it through `__graft_entry__.load_package()` or
`importlib` (see `__graft_entry__.py`).
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
# KWS_LIB: another build of the same library (development aid: same-box A/B runs of tools/ab_rate.py)
LIB_PATH = os.environ.get("KWS_LIB") or os.path.join(_HERE, "libkws_mi355x.so")
MODELS_DIR = os.path.join(ROOT, "models")
DEFAULT_MODEL = os.path.join(MODELS_DIR, "l476_no_yes.kwsm")

EI_IMPULSE_OK = 0
MODE_EXACT, MODE_FAST = 0, 1
ERROR_NAMES = {0: "EI_IMPULSE_OK", -1: "EI_IMPULSE_ERROR_SHAPES_DONT_MATCH", -2: "EI_IMPULSE_CANCELED",
               -3: "EI_IMPULSE_TFLITE_ERROR", -5: "EI_IMPULSE_DSP_ERROR", -6: "EI_IMPULSE_TFLITE_ARENA_ALLOC_FAILED",
               -7: "EI_IMPULSE_CUBEAI_ERROR", -8: "EI_IMPULSE_ALLOC_FAILED", -17: "KWS_ERROR_NO_MODEL",
               -18: "KWS_ERROR_UNSUPPORTED_MODEL", -19: "KWS_ERROR_HIP", -20: "KWS_ERROR_BAD_ARGUMENT"}

# every symbol include/kws/kws.h and include/kws/ei_compat.h declare
EXPORTED_SYMBOLS = [
    "run_classifier", "run_inference", "run_classifier_init", "run_classifier_continuous",
    "run_moving_average_filter", "ei_run_impulse_check_canceled", "ei_sleep", "ei_read_timer_ms",
    "ei_read_timer_us", "ei_printf", "ei_printf_float",
    "kws_create", "kws_create_from_file", "kws_destroy", "kws_last_error", "kws_label_count", "kws_label",
    "kws_feature_count", "kws_clip_samples", "kws_frame_count", "kws_pooled_tap_bytes", "kws_model_is_float",
    "kws_nn_f32_batch_device", "kws_nn_kernel_name", "kws_mfcc_kernel_name", "kws_mfe_batch_device", "kws_filter_count", "kws_set_default_model",
    "kws_default_model", "kws_run_classifier_batch_device", "kws_run_classifier_batch",
    "kws_extract_mfcc_batch_device", "kws_run_inference_batch_device", "kws_mfcc_batch_device",
    "kws_cmvn_inference_batch_device", "kws_nn_batch_device", "kws_nn_batch",
    "kws_streams_create", "kws_streams_destroy", "kws_streams_init", "kws_streams_step_device",
    "kws_extract_mfe_batch_device", "kws_set_mode", "kws_get_mode", "kws_fast_is_fused", "kws_fast_fallback_count", "kws_fast_exact_count", "kws_fast_guard",
    "kws_set_logits_tap", "kws_fast_gain", "kws_fast_tolerance_info",
    "kws_comm_unique_id", "kws_comm_create", "kws_comm_world_size", "kws_comm_rank", "kws_comm_ranks_seen", "kws_comm_rccl_version", "kws_comm_wait", "kws_allgather_scores", "kws_comm_destroy",
    "kws_wav_info_from_memory", "kws_wav_decode_mono", "kws_resample_length", "kws_resample_device", "kws_resample_device_ex",
    "kws_synth_clips_device", "kws_mix_audio_device", "kws_device_malloc", "kws_device_free", "kws_memcpy_h2d", "kws_memcpy_d2h",
    "kws_device_synchronize",
]


class FastTolerance(C.Structure):
    _fields_ = [("score_tol", C.c_float), ("k_sigma", C.c_float), ("lin_margin", C.c_float), ("logit_cap", C.c_float), ("g_c1", C.c_float),
                ("g_c2", C.c_float), ("sigma_net", C.c_float), ("total_gain", C.c_float), ("uniform_feature_tol", C.c_float),
                ("calibrated", C.c_int), ("n_columns", C.c_int), ("n_frames", C.c_int), ("entry_tier", C.c_int), ("dev_overrides", C.c_int),
                ("k_sigma_worst_column", C.c_float), ("silent_rows_exact", C.c_int), ("systematic_ratio", C.c_float),
                ("fused_waves_per_simd", C.c_int), ("fused_waves", C.c_int)]


class KwsError(RuntimeError):
    def __init__(self, code, detail):
        super().__init__("%s (%d): %s" % (ERROR_NAMES.get(code, "?"), code, detail))
        self.code = code


def build(force=False):
    """Compile csrc/ for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "csrc")])
    # the development build (KWS_DEV_* switches compiled in): what the tests that force a tier / a layout and the A/B tools load
    # (best effort: a failure of the development build must not keep the product library from loading)
    if subprocess.call(["make", "-s", "-C", os.path.join(_HERE, "csrc"), "dev"]) != 0:
        print("ei-keyword-spotting_amd.build(): the development build (libkws_mi355x_dev.so) failed; the product library is built", file=sys.stderr)
    return LIB_PATH


_lib = None


def lib():
    """Load libkws_mi355x.so (fails loudly if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError("%s is missing: run __graft_entry__.build() (hipcc, gfx950) first" % LIB_PATH)
        # One HIP runtime per process: when PyTorch is going to be used in this process (bench.py, tests) its bundled
        # libamdhip64 / libhsa-runtime64 must be the copies that get loaded, BEFORE this library pulls in /opt/rocm's --
        # otherwise the runtime that comes second finds no device.  (A plain C application links /opt/rocm's only.)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        vp, sz, i32, u32 = C.c_void_p, C.c_size_t, C.c_int, C.c_uint32
        L.kws_last_error.restype = C.c_char_p
        L.kws_create.argtypes = [vp, sz, i32, C.POINTER(vp)]
        L.kws_create_from_file.argtypes = [C.c_char_p, i32, C.POINTER(vp)]
        L.kws_destroy.argtypes = [vp]
        for f in ("kws_label_count", "kws_feature_count", "kws_clip_samples", "kws_frame_count", "kws_pooled_tap_bytes",
                  "kws_model_is_float", "kws_filter_count"):
            getattr(L, f).argtypes = [vp]
        L.kws_label.restype = C.c_char_p
        L.kws_comm_unique_id.argtypes = [vp, sz]
        L.kws_comm_create.argtypes = [vp, sz, i32, i32, i32, C.POINTER(vp)]
        L.kws_comm_world_size.argtypes = [vp]
        L.kws_comm_rank.argtypes = [vp]
        if hasattr(L, "kws_comm_ranks_seen"):
            L.kws_comm_ranks_seen.argtypes = [vp]
            L.kws_comm_wait.argtypes = [vp, vp]
        L.kws_allgather_scores.argtypes = [vp, vp, vp, sz, i32, vp]
        L.kws_comm_destroy.argtypes = [vp]
        L.kws_set_mode.argtypes = [vp, i32]
        L.kws_get_mode.argtypes = [vp]
        L.kws_fast_is_fused.argtypes = [vp]
        L.kws_fast_fallback_count.argtypes = [vp, C.POINTER(sz)]
        if hasattr(L, "kws_fast_guard"):                 # absent from older builds compared in tools/ab_rate.py
            L.kws_fast_guard.argtypes = [vp, i32, vp]
        if hasattr(L, "kws_fast_gain"):
            L.kws_fast_gain.argtypes = [vp, vp]
            L.kws_fast_tolerance_info.argtypes = [vp, vp]
        if hasattr(L, "kws_fast_exact_count"):
            L.kws_fast_exact_count.argtypes = [vp, C.POINTER(sz)]
        if hasattr(L, "kws_set_logits_tap"):
            L.kws_set_logits_tap.argtypes = [vp, vp]
        L.kws_nn_kernel_name.restype = C.c_char_p
        L.kws_nn_kernel_name.argtypes = [vp]
        L.kws_mfcc_kernel_name.restype = C.c_char_p
        L.kws_mfcc_kernel_name.argtypes = [vp]
        L.kws_label.argtypes = [vp, i32]
        L.kws_set_default_model.argtypes = [vp]
        L.kws_default_model.restype = vp
        L.kws_run_classifier_batch_device.argtypes = [vp, vp, sz, vp, vp, vp, vp]
        L.kws_run_classifier_batch.argtypes = [vp, vp, sz, vp, vp, vp]
        L.kws_extract_mfcc_batch_device.argtypes = [vp, vp, sz, vp, vp, vp]
        L.kws_run_inference_batch_device.argtypes = [vp, vp, sz, vp, vp]
        L.kws_mfcc_batch_device.argtypes = [vp, vp, sz, vp, vp]
        L.kws_cmvn_inference_batch_device.argtypes = [vp, vp, sz, vp, vp, vp, vp]
        L.kws_nn_batch_device.argtypes = [vp, vp, sz, vp, vp, vp, vp, vp]
        L.kws_nn_batch.argtypes = [vp, vp, sz, vp, vp, vp, vp]
        L.kws_nn_f32_batch_device.argtypes = [vp, vp, sz, vp, vp, vp]
        L.kws_mfe_batch_device.argtypes = [vp, vp, sz, vp, vp, vp]
        L.kws_extract_mfe_batch_device.argtypes = [vp, vp, sz, vp, vp]
        L.kws_synth_clips_device.argtypes = [u32, u32, u32, u32, vp, vp]
        L.kws_mix_audio_device.argtypes = [vp, vp, sz, vp, sz, vp, C.c_float, C.c_float, sz, sz, vp, vp]
        if hasattr(L, "kws_wav_decode_mono"):
            L.kws_wav_info_from_memory.argtypes = [vp, sz, vp]
            L.kws_wav_decode_mono.argtypes = [vp, sz, vp, sz, C.POINTER(sz), C.POINTER(i32)]
            L.kws_resample_length.restype = sz
            L.kws_resample_length.argtypes = [sz, i32, i32]
            L.kws_resample_device.argtypes = [vp, sz, i32, vp, sz, i32, vp]
        if hasattr(L, "kws_resample_device_ex"):
            L.kws_resample_device_ex.argtypes = [vp, sz, i32, vp, sz, i32, i32, vp]
        L.kws_streams_create.argtypes = [vp, sz, C.POINTER(vp)]
        L.kws_streams_destroy.argtypes = [vp]
        L.kws_streams_init.argtypes = [vp]
        L.kws_streams_step_device.argtypes = [vp, vp, sz, vp, vp, C.POINTER(C.c_int), vp]
        L.kws_device_malloc.argtypes = [C.POINTER(vp), sz]
        L.kws_device_free.argtypes = [vp]
        L.kws_memcpy_h2d.argtypes = [vp, vp, sz]
        L.kws_memcpy_d2h.argtypes = [vp, vp, sz]
        L.run_classifier.argtypes = [vp, vp, C.c_bool]
        L.run_inference.argtypes = [vp, vp, C.c_bool]
        L.run_moving_average_filter.restype = C.c_float
        L.run_moving_average_filter.argtypes = [vp, C.c_float]
        _lib = L
    return _lib


def _check(rc):
    if rc != EI_IMPULSE_OK:
        raise KwsError(rc, lib().kws_last_error().decode())


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# ---- SDK structs (include/kws/ei_compat.h) for callers that want the reference's own call shape --------------
GET_DATA_FN = C.CFUNCTYPE(C.c_int, C.c_size_t, C.c_size_t, C.POINTER(C.c_float))


class Signal(C.Structure):                       # signal_t, EIDSP_SIGNAL_C_FN_POINTER=1 layout
    _fields_ = [("get_data", GET_DATA_FN), ("total_length", C.c_size_t)]


class Matrix(C.Structure):                       # ei::matrix_t data members
    _fields_ = [("buffer", C.POINTER(C.c_float)), ("rows", C.c_uint32), ("cols", C.c_uint32),
                ("buffer_managed_by_me", C.c_bool)]


def result_struct(n_labels):
    class Classification(C.Structure):
        _fields_ = [("label", C.c_char_p), ("value", C.c_float)]

    class Timing(C.Structure):
        _fields_ = [("sampling", C.c_int), ("dsp", C.c_int), ("classification", C.c_int), ("anomaly", C.c_int)]

    class Result(C.Structure):                   # ei_impulse_result_t for EI_CLASSIFIER_LABEL_COUNT = n_labels
        _fields_ = [("classification", Classification * n_labels), ("anomaly", C.c_float), ("timing", Timing)]

    return Result


class Model:
    """A loaded .kwsm model on one MI355X (kws_handle)."""

    def __init__(self, path=DEFAULT_MODEL, device=0, blob=None):
        self.L = lib()
        h = C.c_void_p()
        if blob is not None:
            _check(self.L.kws_create(blob, len(blob), device, C.byref(h)))
        else:
            _check(self.L.kws_create_from_file(path.encode(), device, C.byref(h)))
        self.h = h
        self.device = device
        self.n_labels = self.L.kws_label_count(h)
        self.labels = [self.L.kws_label(h, i).decode(errors="replace") for i in range(self.n_labels)]
        self.n_features = self.L.kws_feature_count(h)
        self.clip_samples = self.L.kws_clip_samples(h)
        self.n_frames = self.L.kws_frame_count(h)
        self.n_filters = self.L.kws_filter_count(h)
        self.pooled_tap_bytes = self.L.kws_pooled_tap_bytes(h)
        self.is_float = bool(self.L.kws_model_is_float(h))
        self.nn_kernel = self.L.kws_nn_kernel_name(h).decode()
        self.mfcc_kernel = self.L.kws_mfcc_kernel_name(h).decode()

    def close(self):
        if getattr(self, "h", None):
            self.L.kws_destroy(self.h)
            self.h = None

    def set_mode(self, mode):
        """MODE_EXACT (default) or MODE_FAST (include/kws/kws.h)"""
        _check(self.L.kws_set_mode(self.h, mode))

    @property
    def fast_is_fused(self):
        return bool(self.L.kws_fast_is_fused(self.h))

    def fast_fallback_count(self):
        n = C.c_size_t()
        _check(self.L.kws_fast_fallback_count(self.h, C.byref(n)))
        return n.value

    def fast_exact_count(self):
        """clips of the last fast-mode batch call that ended in the exact kernels (the second tier handed them back)"""
        n = C.c_size_t()
        _check(self.L.kws_fast_exact_count(self.h, C.byref(n)))
        return n.value

    def set_logits_tap(self, dev_ptr):
        """float32 graphs: the batch calls also write every clip's FULLY_CONNECTED outputs to dev_ptr [B][labels] (None: off)"""
        _check(self.L.kws_set_logits_tap(self.h, dev_ptr))

    def fast_guard(self, tier=1):
        """coefficients [4][n_columns] of that tier's guard: absolute, per log-mel level, per |window mean|, per |window mean| when column 0's
        window means were replayed (tier 1: the fast kernel; tier 2: fast cmvnw + network on exact cepstra).  See kws.h for the rule."""
        import numpy as np
        n = self.n_features // self.n_frames
        coef = np.zeros((4, n), np.float32)
        _check(self.L.kws_fast_guard(self.h, tier, coef.ctypes.data_as(C.c_void_p)))
        return coef

    def fast_gain(self):
        """float32 graphs: logit-difference error per unit of feature error, per cepstral column (kws_gain.cpp)"""
        import numpy as np
        g = np.zeros(self.n_features // self.n_frames, np.float32)
        _check(self.L.kws_fast_gain(self.h, g.ctypes.data_as(C.c_void_p)))
        return g

    def fast_tolerance(self):
        """kws_fast_tolerance as a dict"""
        t = FastTolerance()
        _check(self.L.kws_fast_tolerance_info(self.h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in FastTolerance._fields_}

    def set_default(self):
        _check(self.L.kws_set_default_model(self.h))

    # ---- host (numpy) convenience paths: copy in, run on the GPU, copy out --------------------------------
    def run_classifier_batch(self, pcm, want_features=False):
        pcm = np.ascontiguousarray(pcm, np.int16)
        if pcm.ndim == 1:
            pcm = pcm[None]
        B = pcm.shape[0]
        assert pcm.shape[1] == self.clip_samples
        s = np.zeros((B, self.n_labels), np.float32)
        if not want_features:                                                      # scores only: nothing else crosses PCIe
            _check(self.L.kws_run_classifier_batch(self.h, _p(pcm), B, _p(s), None, None))
            return s
        f = np.zeros((B, self.n_features), np.float32)
        q = None if self.is_float else np.zeros((B, self.n_features), np.int8)     # float models have no int8 tensor
        _check(self.L.kws_run_classifier_batch(self.h, _p(pcm), B, _p(s), _p(f), None if q is None else _p(q)))
        return s, f, q

    def nn_batch(self, q):
        q = np.ascontiguousarray(q, np.int8).reshape(-1, self.n_features)
        B = q.shape[0]
        s = np.zeros((B, self.n_labels), np.float32)
        tp = np.zeros((B, self.pooled_tap_bytes), np.int8)
        tf = np.zeros((B, self.n_labels), np.int8)
        to = np.zeros((B, self.n_labels), np.int8)
        _check(self.L.kws_nn_batch(self.h, _p(q), B, _p(s), _p(tp), _p(tf), _p(to)))
        return s, tp, tf, to

    # ---- device-pointer paths (torch tensors or raw pointers); asynchronous on `stream` -------------------
    def run_classifier_batch_device(self, pcm_ptr, B, scores_ptr, features_ptr=None, q_ptr=None, stream=None):
        _check(self.L.kws_run_classifier_batch_device(self.h, pcm_ptr, B, scores_ptr, features_ptr, q_ptr, stream))

    def extract_mfcc_batch_device(self, pcm_ptr, B, features_ptr, q_ptr=None, stream=None):
        _check(self.L.kws_extract_mfcc_batch_device(self.h, pcm_ptr, B, features_ptr, q_ptr, stream))

    def mfe_batch_device(self, pcm_ptr, B, mel_ptr, energy_ptr=None, stream=None):
        _check(self.L.kws_mfe_batch_device(self.h, pcm_ptr, B, mel_ptr, energy_ptr, stream))

    def extract_mfe_batch_device(self, pcm_ptr, B, features_ptr, stream=None):
        _check(self.L.kws_extract_mfe_batch_device(self.h, pcm_ptr, B, features_ptr, stream))

    def mfcc_batch_device(self, pcm_ptr, B, mfcc_ptr, stream=None):
        _check(self.L.kws_mfcc_batch_device(self.h, pcm_ptr, B, mfcc_ptr, stream))

    def cmvn_inference_batch_device(self, mfcc_ptr, B, scores_ptr, features_ptr=None, q_ptr=None, stream=None):
        _check(self.L.kws_cmvn_inference_batch_device(self.h, mfcc_ptr, B, scores_ptr, features_ptr, q_ptr, stream))

    def run_inference_batch_device(self, features_ptr, B, scores_ptr, stream=None):
        _check(self.L.kws_run_inference_batch_device(self.h, features_ptr, B, scores_ptr, stream))

    def nn_f32_batch_device(self, features_ptr, B, scores_ptr, logits_ptr=None, stream=None):
        _check(self.L.kws_nn_f32_batch_device(self.h, features_ptr, B, scores_ptr, logits_ptr, stream))

    def nn_batch_device(self, q_ptr, B, scores_ptr, stream=None):
        _check(self.L.kws_nn_batch_device(self.h, q_ptr, B, scores_ptr, None, None, None, stream))


class StreamBatch:
    """S audio streams in continuous mode, state in HBM (kws_stream_batch)."""

    def __init__(self, model, n_streams):
        self.model = model
        self.L = model.L
        sb = C.c_void_p()
        _check(self.L.kws_streams_create(model.h, n_streams, C.byref(sb)))
        self.sb = sb
        self.n_streams = n_streams

    def init(self):
        _check(self.L.kws_streams_init(self.sb))

    def step_device(self, slices_ptr, slice_samples, scores_ptr, end_of_signal_ptr=None, stream=None):
        produced = C.c_int()
        _check(self.L.kws_streams_step_device(self.sb, slices_ptr, slice_samples, end_of_signal_ptr, scores_ptr,
                                              C.byref(produced), stream))
        return bool(produced.value)

    def close(self):
        if getattr(self, "sb", None):
            self.L.kws_streams_destroy(self.sb)
            self.sb = None


class Comm:
    """RCCL communicator of one rank (kws_comm): the all-gather of the per-clip scores over xGMI, through the C ABI."""
    ID_BYTES = 128

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(Comm.ID_BYTES)
        _check(lib().kws_comm_unique_id(buf, Comm.ID_BYTES))
        return bytes(buf.raw)

    def __init__(self, unique_id, world_size, rank, device):
        self.L = lib()
        c = C.c_void_p()
        _check(self.L.kws_comm_create(unique_id, len(unique_id), world_size, rank, device, C.byref(c)))
        self.c, self.world_size, self.rank = c, world_size, rank

    def allgather_scores(self, local_ptr, all_ptr, clips_per_rank, label_count, stream=None):
        _check(self.L.kws_allgather_scores(self.c, local_ptr, all_ptr, clips_per_rank, label_count, stream))

    @property
    def ranks_seen_by_rccl(self):
        """ncclCommCount of this communicator"""
        return int(self.L.kws_comm_ranks_seen(self.c))

    @property
    def rccl_version(self):
        return int(self.L.kws_comm_rccl_version())

    def wait(self, stream=None):
        """everything enqueued on `stream` has completed -- or KwsError after KWS_COMM_TIMEOUT_MS / when a peer failed (communicator aborted)"""
        _check(self.L.kws_comm_wait(self.c, stream))

    def close(self):
        if getattr(self, "c", None):
            self.L.kws_comm_destroy(self.c)
            self.c = None


def mix_audio_device(words_ptr, word_len_ptr, word_stride, noise_ptr, noise_len, start_ptr, word_vol, bg_vol, n_clips, n, out_ptr, stream=None):
    _check(lib().kws_mix_audio_device(words_ptr, word_len_ptr, word_stride, noise_ptr, noise_len, start_ptr, word_vol, bg_vol, n_clips, n,
                                      out_ptr, stream))


class WavInfo(C.Structure):                      # kws_wav_info
    _fields_ = [("channels", C.c_int), ("sample_rate", C.c_int), ("bits_per_sample", C.c_int), ("is_float", C.c_int),
                ("frames", C.c_size_t), ("data_offset", C.c_size_t)]


def wav_info(data):
    """container facts of a RIFF/WAVE image (bytes)"""
    w = WavInfo()
    _check(lib().kws_wav_info_from_memory(data, len(data), C.byref(w)))
    return w


def wav_decode_mono(data):
    """(float32 mono samples, sample rate) of a WAV image, as librosa.load(sr = None, mono = True) would return them"""
    import numpy as np
    n, sr = C.c_size_t(), C.c_int()
    _check(lib().kws_wav_decode_mono(data, len(data), None, 0, C.byref(n), C.byref(sr)))
    out = np.zeros(n.value, np.float32)
    _check(lib().kws_wav_decode_mono(data, len(data), out.ctypes.data_as(C.c_void_p), out.size, C.byref(n), C.byref(sr)))
    return out, sr.value


def resample_length(n_in, sr_in, sr_out):
    return lib().kws_resample_length(n_in, sr_in, sr_out)


RESAMPLE_EXACT_POSITIONS = 1


def resample_device(in_ptr, n_in, sr_in, out_ptr, n_out, sr_out, stream=None, flags=0):
    """flags = 0: the reference's behaviour (resampy's integer table stepping + librosa's fix_length); RESAMPLE_EXACT_POSITIONS: exact tap positions"""
    _check(lib().kws_resample_device_ex(in_ptr, n_in, sr_in, out_ptr, n_out, sr_out, flags, stream))


def synth_clips_device(seed, first_clip, n_clips, clip_len, out_ptr, stream=None):
    _check(lib().kws_synth_clips_device(seed, first_clip, n_clips, clip_len, out_ptr, stream))


# ---- multi-GPU: embarrassingly parallel sharding + the one collective of the path ---------------------------
def shard_first_clip(rank, clips_per_rank):
    """Rank r owns the contiguous block of clips [r*B, (r+1)*B)."""
    return rank * clips_per_rank


def all_gather_scores(local_scores, world_size):
    """All-gather the per-clip scores [B][C] of every rank into [world*B][C] (rank-major = global clip order).
    Works with any torch.distributed backend: "nccl" (= RCCL over xGMI) on the GPUs, "gloo" in the CPU tests."""
    import torch
    import torch.distributed as dist
    if world_size == 1:
        return local_scores
    out = torch.empty((world_size * local_scores.shape[0], local_scores.shape[1]), dtype=local_scores.dtype,
                      device=local_scores.device)
    dist.all_gather_into_tensor(out, local_scores.contiguous())
    return out
