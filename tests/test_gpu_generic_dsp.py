"""-m gpu: MFCC configurations outside the tuned kernel's instantiations (fft_length 128 / 512, overlapping frames -> 98 frames,
clips of other lengths incl. unaligned ones, other mel filter counts, wide mel filters) run on the general kernels
(csrc/kws_generic.hip) and are bit-identical to the oracle -- the cases of tests/test_oracle_vs_reference.py::test_mfcc_configs
and ::test_short_clips (which hold the oracle to the compiled reference), through the HIP path instead of being refused."""
import os

import numpy as np
import pytest

from kws_testlib import L476_CONFIG, ROOT, OracleModel, bits, special_clips, synth_model_blob

pytestmark = pytest.mark.gpu

BLOCKS = dict(blocks=((8, 3, 7), (4, 3, 7)), n_labels=3)
# what kws_mfcc_kernel_name says for a plan the tuned kernels do not serve as a whole.  Round 6: where only the frame count or the cmvnw window
# makes the plan general (fft 256, 32 / 40 filters), int16 batches take the tuned spectral kernel over chunks of frames and the general cmvnw
GENERAL_KERNELS = ("kws_spectral_lds_kernel", "kws_spectral_generic_kernel", "kws_mfcc8_kernel (chunked)")
CASES = {
    "fft512": dict(fft_length=512),                                  # zero-padded frames
    "fft128_win51": dict(fft_length=128, win_size=51),               # truncated frames
    "stride10ms_win31": dict(frame_stride=0.01, win_size=31),        # overlapping frames: 98 of them (two chunks of 49 on the tuned spectral kernel)
    "fft256_2s_40filters": dict(raw_samples=32000, num_filters=40, ncep=20, blocks=((8, 3, 1), (4, 3, 1))),   # 99 frames: three chunks of 33
    "fft256_win201": dict(win_size=201),                             # 49 frames, a cmvnw window the tuned kernel does not hold: one chunk
    "pre_cof0": dict(pre_cof=0.0),
    "clip4000": dict(raw_samples=4000, blocks=((8, 3, 1), (4, 3, 1))),
    "clip640_one_frame": dict(raw_samples=640, blocks=((8, 3, 1), (4, 3, 1))),      # 1 frame (no pooling: SAME pooling of 1-2 rows
    "clip1000": dict(raw_samples=1000, blocks=((8, 3, 1), (4, 3, 1))),               # would need padding, which the network plan refuses)
    "clip15999_unaligned": dict(raw_samples=15999),                  # (odd clip length: the int16 samples are fetched one by one, not as dword pairs)
    "odd_stride_fft512": dict(fft_length=512, frame_length=0.0200625, frame_stride=0.0100625, win_size=31),   # 321-sample frames every 161: likewise
    "filters24": dict(num_filters=24, ncep=10),                      # DCT of 24 points: radix 4, 3
    "filters64_wide": dict(num_filters=64, ncep=20, high=0),         # 64 filters
    "filters20_0_8000": dict(num_filters=20, ncep=12, low=0, high=0),  # few, wide filters: more than 12 taps each
    "fft1024_filters36": dict(fft_length=1024, num_filters=36, ncep=17, frame_length=0.05, frame_stride=0.025, win_size=21,
                              blocks=((8, 3, 1), (4, 3, 1))),
}


@pytest.fixture(scope="module")
def pkg():
    import sys
    sys.path.insert(0, ROOT)
    import torch  # noqa: F401
    from __graft_entry__ import load_package
    return load_package()


@pytest.mark.parametrize("name", sorted(CASES))
def test_general_mfcc_kernels_bit_exact(name, pkg, oracle, tmp_path):
    kw = dict(BLOCKS, **CASES[name])
    blob = synth_model_blob(seed=3, **kw)
    path = str(tmp_path / "m.kwsm")
    open(path, "wb").write(blob)
    om = OracleModel(oracle, path)
    gm = pkg.Model(blob=blob)
    n = om.raw_sample_count
    sp = special_clips()
    clips = np.concatenate([oracle.synth(5, 0, 70, n), np.stack([np.resize(sp[k], n) for k in ("impulses", "zeros", "alternating_fullscale")])])
    s, f, q = gm.run_classifier_batch(clips, want_features=True)
    so, fo, qo = om.run_batch(clips, want_features=True)
    assert (bits(f) == bits(fo)).all(), name
    assert (q == qo).all() and (bits(s) == bits(so)).all(), name
    # the stage API: cepstra before cmvnw, then cmvnw + inference
    import torch
    d = torch.from_numpy(np.ascontiguousarray(clips)).to("cuda:0")
    mf = torch.zeros((len(clips), gm.n_features), dtype=torch.float32, device="cuda:0")
    gm.mfcc_batch_device(d.data_ptr(), len(clips), mf.data_ptr())
    torch.cuda.synchronize()
    cfg = om.cfg
    want = np.stack([oracle.mfcc_nocmvn(c, cfg).reshape(-1) for c in clips[:8]])
    assert (bits(mf[:8].cpu().numpy()) == bits(want)).all(), name
    s2 = torch.zeros((len(clips), gm.n_labels), dtype=torch.float32, device="cuda:0")
    gm.cmvn_inference_batch_device(mf.data_ptr(), len(clips), s2.data_ptr())
    torch.cuda.synchronize()
    assert (bits(s2.cpu().numpy()) == bits(so)).all(), name
    if gm.mfcc_kernel in GENERAL_KERNELS:
        with pytest.raises(pkg.KwsError):
            gm.set_mode(pkg.MODE_FAST)                  # the fast kernel is built for the tuned configurations only
    else:
        # short aligned clips and pre_cof = 0 are tuned configurations (listed because test_mfcc_configs / test_short_clips list
        # them): the fast kernel serves them too, when it has at least two frames to normalise over
        assert name in ("pre_cof0", "clip4000", "clip640_one_frame", "clip1000")
        if gm.n_frames >= 2:
            gm.set_mode(pkg.MODE_FAST)
            _, f2, _ = gm.run_classifier_batch(clips[:70], want_features=True)
            assert np.abs(f2 - fo[:70]).max() <= 2e-3, name
    gm.close()


def test_general_kernels_float_model_and_drop_in_entry_points(pkg, oracle, tmp_path):
    """A float32 twin on a general configuration, and run_classifier() / run_classifier_continuous() through it."""
    import ctypes
    import subprocess
    import sys
    blob = synth_model_blob(seed=4, **BLOCKS, fft_length=512, frame_stride=0.01, win_size=31)
    p8, pf = str(tmp_path / "i8.kwsm"), str(tmp_path / "f32.kwsm")
    open(p8, "wb").write(blob)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "dequantize_model.py"), p8, pf])
    gm = pkg.Model(pf)
    om = OracleModel(oracle, pf)
    clips = oracle.synth(8, 0, 40)
    s, f, _ = gm.run_classifier_batch(clips, want_features=True)
    so, fo, _ = om.run_batch(clips, want_features=True)
    assert (bits(f) == bits(fo)).all() and np.abs(s - so).max() <= 1e-6
    gm.set_default()
    res = pkg.result_struct(gm.n_labels)()
    for ci in range(3):
        buf = clips[ci].astype(np.float32) / np.float32(32768)

        @pkg.GET_DATA_FN
        def get_data(offset, length, out):
            ctypes.memmove(out, buf[offset:offset + length].ctypes.data, 4 * length)
            return 0
        sig = pkg.Signal(get_data=get_data, total_length=16000)
        assert pkg.lib().run_classifier(ctypes.byref(sig), ctypes.byref(res), False) == 0
        got = np.float32([res.classification[i].value for i in range(gm.n_labels)])
        assert np.abs(got - so[ci]).max() <= 1e-6
    gm.close()


def test_general_kernels_host_batch_over_two_streams(pkg, oracle, tmp_path):
    """kws_run_classifier_batch cuts a host batch into chunks of 8 192 clips that alternate between two streams.  The general
    kernels' transform scratch is indexed by workgroup: each stream has to own its set (one shared set was a race
    that corrupted features for B > 8 192 on a general-configuration model).  Three chunks here, so both streams carry kernels at
    the same time; every clip must equal the single-stream device path, and a strided sample the oracle."""
    import torch
    blob = synth_model_blob(seed=3, **dict(BLOCKS, fft_length=512))
    path = str(tmp_path / "m.kwsm")
    open(path, "wb").write(blob)
    gm = pkg.Model(blob=blob)
    assert gm.mfcc_kernel in GENERAL_KERNELS
    om = OracleModel(oracle, path)
    n = 2 * 8192 + 700
    clips = oracle.synth(21, 0, n)
    for _ in range(2):                                                  # twice: the second pass finds every buffer allocated
        s, f, q = gm.run_classifier_batch(clips, want_features=True)
    d = torch.from_numpy(clips).to("cuda:0")
    s1 = torch.zeros((n, gm.n_labels), dtype=torch.float32, device="cuda:0")
    f1 = torch.zeros((n, gm.n_features), dtype=torch.float32, device="cuda:0")
    q1 = torch.zeros((n, gm.n_features), dtype=torch.int8, device="cuda:0")
    gm.run_classifier_batch_device(d.data_ptr(), n, s1.data_ptr(), f1.data_ptr(), q1.data_ptr())
    torch.cuda.synchronize()
    assert (bits(f) == bits(f1.cpu().numpy())).all() and (q == q1.cpu().numpy()).all() and (bits(s) == bits(s1.cpu().numpy())).all()
    idx = np.arange(0, n, 61)
    so, fo, qo = om.run_batch(clips[idx], want_features=True)
    assert (bits(f[idx]) == bits(fo)).all() and (q[idx] == qo).all() and (bits(s[idx]) == bits(so)).all()
    # two caller-owned streams on the stage API at the same time
    st = [torch.cuda.Stream(), torch.cuda.Stream()]
    out = [torch.zeros((4096, gm.n_features), dtype=torch.float32, device="cuda:0") for _ in range(2)]
    for rep in range(3):
        for k in range(2):
            gm.mfcc_batch_device(d[k * 4096:].data_ptr(), 4096, out[k].data_ptr(), st[k].cuda_stream)
    torch.cuda.synchronize()
    want = np.stack([oracle.mfcc_nocmvn(clips[i], om.cfg).reshape(-1) for i in (0, 4095, 4096, 8191)])
    got = np.stack([out[0][0].cpu().numpy(), out[0][4095].cpu().numpy(), out[1][0].cpu().numpy(), out[1][4095].cpu().numpy()])
    assert (bits(got) == bits(want)).all()
    gm.close()


@pytest.mark.parametrize("name", ["l476", "fft512", "fft1024_f20"])
def test_quantized_filterbank_models_on_gpu(name, pkg, oracle, tmp_path):
    """A model built for an application that compiles the SDK with its default EIDSP_QUANTIZE_FILTERBANK = 1 (.kwsm: bit 8 of the DSP word,
    tools/eon_import.py --quantize-filterbank): the mel weights are snapped to numpy.hpp:52's table.  The exact kernels must reproduce the
    reference's features (golden from the reference built with the option at its default) and the oracle's scores, bit for bit."""
    from test_oracle_golden import QFB_CASES
    kw = QFB_CASES[name]
    g = np.load(os.path.join(ROOT, "tests", "golden", "qfb_l476.npz"))
    blob_kw = dict(BLOCKS, seed=3, quantize_filterbank=True, fft_length=kw.get("fft_length", 256), num_filters=kw.get("num_filters", 32),
                   ncep=kw.get("num_cepstral", 13), low=kw.get("low_frequency", 300), high=kw.get("high_frequency", 4000))
    blob = synth_model_blob(**blob_kw)
    path = str(tmp_path / "q.kwsm")
    open(path, "wb").write(blob)
    om = OracleModel(oracle, path)
    assert om.cfg.quantize_filterbank == 1
    gm = pkg.Model(blob=blob)
    clips = oracle.synth(int(g["seed"]), 0, int(g["n"]))
    s, f, q = gm.run_classifier_batch(clips, want_features=True)
    assert (bits(f) == bits(g[name])).all(), name                      # the reference's own features
    so, fo, qo = om.run_batch(clips, want_features=True)
    assert (bits(f) == bits(fo)).all() and (q == qo).all() and (bits(s) == bits(so)).all(), name
    gm.close()

@pytest.mark.parametrize("name", ["fft1024_filters36", "stride10ms_win31", "fft128_win51"])
def test_chunk_length_is_measured_per_handle_and_never_changes_a_bit(name, dev_pkg, oracle, tmp_path, monkeypatch):
    """kws_spectral_lds_kernel runs with eight or four frames per chunk (csrc/kws_generic.hip); which is faster depends on the shape, so a handle
    measures it on its own first large calls (kws_api.cpp generic_chunk_begin).  Both pinned values and the measured path give the oracle's bits."""
    pkg = dev_pkg            # KWS_DEV_GENERIC_LCH is a development switch: the development build of the library (conftest.py)
    monkeypatch.setenv("KWS_DEV_GENERIC_NO_TUNED_SPECTRAL", "1")     # (this test is about the cooperative kernel: a shape whose spectral stage the tuned kernel could take stays on it)
    import ctypes
    import torch
    kw = dict(BLOCKS, **CASES[name])
    blob = synth_model_blob(seed=3, **kw)
    path = str(tmp_path / "m.kwsm")
    open(path, "wb").write(blob)
    om = OracleModel(oracle, path)
    n = om.raw_sample_count
    B = 4096
    clips = np.ascontiguousarray(np.tile(oracle.synth(6, 0, 64, n), (B // 64, 1)))
    _, fo, _ = om.run_batch(clips[:64], want_features=True)
    d = torch.from_numpy(clips).to("cuda:0")
    L = pkg.lib()
    L.kws_dev_generic_chunk.argtypes = [ctypes.c_void_p]

    def features(gm):
        ft = torch.zeros((B, gm.n_features), dtype=torch.float32, device="cuda:0")
        gm.extract_mfcc_batch_device(d.data_ptr(), B, ft.data_ptr())
        torch.cuda.synchronize()
        f = ft.cpu().numpy()
        assert (bits(f[:64]) == bits(fo)).all() and (bits(f[64:128]) == bits(fo)).all() and (bits(f[-64:]) == bits(fo)).all()

    # ... and so do both builds of the kernel (registers for two waves per SIMD with deep batches, for four with shallow ones), pinned with
    # KWS_DEV_GENERIC_WPS; left alone the launch picks the build by how many waves the LDS lets a CU hold
    # ... and the two ways the int16 samples are fetched (dword pairs requested one sub-batch ahead; sample by sample: KWS_DEV_GENERIC_NOPAIRS)
    for pinned in ("8", "4"):
        for build in ("2", "4"):
            for nopairs in (False, True):
                monkeypatch.setenv("KWS_DEV_GENERIC_LCH", pinned)
                monkeypatch.setenv("KWS_DEV_GENERIC_WPS", build)
                if nopairs:
                    monkeypatch.setenv("KWS_DEV_GENERIC_NOPAIRS", "1")
                gm = pkg.Model(blob=blob)
                assert gm.mfcc_kernel == "kws_spectral_lds_kernel"
                features(gm)
                assert L.kws_dev_generic_chunk(gm.h) == 0              # pinned from outside: the handle has not measured anything
                gm.close()
                monkeypatch.delenv("KWS_DEV_GENERIC_NOPAIRS", raising=False)
    monkeypatch.delenv("KWS_DEV_GENERIC_LCH")
    monkeypatch.delenv("KWS_DEV_GENERIC_WPS")
    gm = pkg.Model(blob=blob)
    for k in range(8):                                              # two un-timed first launches (one per chunk length) + four timed samples, collected by the next call
        assert L.kws_dev_generic_chunk(gm.h) == 0 or k >= 7
        features(gm)
    assert L.kws_dev_generic_chunk(gm.h) in (4, 8)
    features(gm)
    # a small batch is never timed and runs with the default
    gm2 = pkg.Model(blob=blob)
    s, f, _ = gm2.run_classifier_batch(clips[:64], want_features=True)
    assert (bits(f) == bits(fo)).all() and L.kws_dev_generic_chunk(gm2.h) == 0
    gm.close()
    gm2.close()



@pytest.mark.parametrize("name", ["stride10ms_win31", "fft256_2s_40filters", "fft256_win201"])
def test_tuned_spectral_chunks_equal_the_general_kernel(name, dev_pkg, oracle, tmp_path, monkeypatch):
    """Round 6: a general plan whose spectral stage fits the tuned kernel runs kws_mfcc8_kernel over chunks of at most 49
    frames (kws_api.cpp: launch_spectral_tuned_chunks; the predecessor of a later chunk's first sample is the sample before it, not the window's last).
    Cepstra and features must be the cooperative kernel's bits (development switch KWS_DEV_GENERIC_NO_TUNED_SPECTRAL) and the oracle's, on clips whose
    chunk boundaries matter (full-scale alternation, impulses) as well."""
    import torch
    pkg = dev_pkg
    kw = dict(BLOCKS, **CASES[name])
    blob = synth_model_blob(seed=3, **kw)
    path = str(tmp_path / "m.kwsm")
    open(path, "wb").write(blob)
    om = OracleModel(oracle, path)
    n = om.raw_sample_count
    sp = special_clips()
    clips = np.concatenate([oracle.synth(8, 0, 125, n), np.stack([np.resize(sp[k], n) for k in ("impulses", "zeros", "alternating_fullscale")])])
    d = torch.from_numpy(np.ascontiguousarray(clips)).to("cuda:0")

    def stage(gm):
        mf = torch.zeros((len(clips), gm.n_features), dtype=torch.float32, device="cuda:0")
        ft = torch.zeros((len(clips), gm.n_features), dtype=torch.float32, device="cuda:0")
        gm.mfcc_batch_device(d.data_ptr(), len(clips), mf.data_ptr())
        gm.extract_mfcc_batch_device(d.data_ptr(), len(clips), ft.data_ptr())
        torch.cuda.synchronize()
        return mf.cpu().numpy(), ft.cpu().numpy()
    gm = pkg.Model(blob=blob)
    assert gm.mfcc_kernel == "kws_mfcc8_kernel (chunked)"
    mf1, ft1 = stage(gm)
    gm.close()
    monkeypatch.setenv("KWS_DEV_GENERIC_NO_TUNED_SPECTRAL", "1")
    gm = pkg.Model(blob=blob)
    assert gm.mfcc_kernel in ("kws_spectral_lds_kernel", "kws_spectral_generic_kernel")
    mf0, ft0 = stage(gm)
    gm.close()
    assert (bits(mf1) == bits(mf0)).all() and (bits(ft1) == bits(ft0)).all(), name
    want = np.stack([oracle.mfcc_nocmvn(c, om.cfg).reshape(-1) for c in clips[-8:]])
    assert (bits(mf1[-8:]) == bits(want)).all(), name
    _, fo, _ = om.run_batch(clips, want_features=True)
    assert (bits(ft1) == bits(fo)).all(), name
