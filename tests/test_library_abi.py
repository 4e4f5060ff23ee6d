"""CPU-only: the C-ABI library loads and exports every symbol include/kws/*.h declares; no compute calls."""
import ctypes
import os
import re

import pytest

from kws_testlib import ROOT


@pytest.fixture(scope="module")
def pkg():
    import sys
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_package
    p = load_package()
    if not os.path.exists(p.LIB_PATH):
        p.build()
    return p


def _declared(header):
    src = open(os.path.join(ROOT, "include", "kws", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b([a-z_0-9]+)\s*\([^;{]*\)\s*;", src)) - {"defined", "int"}


def test_exports_every_declared_symbol(pkg):
    lib = ctypes.CDLL(pkg.LIB_PATH)
    declared = _declared("kws.h") | _declared("ei_compat.h")
    assert {"run_classifier", "run_inference", "kws_create", "kws_run_classifier_batch_device"} <= declared
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert set(pkg.EXPORTED_SYMBOLS) == declared


def test_struct_layouts_match_reference_abi(pkg):
    # signal_t 16 B {fn @0, total_length @8}; ei_impulse_result_t (4 labels) 88 B,
    # classification[i] @16i (label @0, value @8), anomaly @64, timing @68
    assert ctypes.sizeof(pkg.Signal) == 16 and pkg.Signal.total_length.offset == 8
    R = pkg.result_struct(4)
    assert ctypes.sizeof(R) == 88 and R.anomaly.offset == 64 and R.timing.offset == 68
    R3 = pkg.result_struct(3)
    assert R3.anomaly.offset == 48 and R3.timing.offset == 52
    assert ctypes.sizeof(pkg.Matrix) == 24


def test_fails_loudly_without_gpu(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg.KwsError) as e:
        pkg.Model(pkg.DEFAULT_MODEL)
    assert e.value.code == -19          # KWS_ERROR_HIP: no CPU fallback


def test_moving_average_filter(pkg):
    # ei_run_classifier.h:134-145 with EI_CLASSIFIER_SLICES_PER_MODEL_WINDOW = 4 (2-tap)
    class Maf(ctypes.Structure):
        _fields_ = [("buf_idx", ctypes.c_uint32), ("running_sum", ctypes.c_float), ("maf_buffer", ctypes.c_float * 2)]
    m = Maf()
    L = pkg.lib()
    outs = [L.run_moving_average_filter(ctypes.byref(m), v) for v in (1.0, 0.5, 0.0, 0.25)]
    assert outs == [0.5, 0.75, 0.25, 0.125]


def test_malformed_model_blobs_are_rejected(pkg):
    """The blob parser runs before any device is touched: truncations, bad magic / version, out-of-range tensor indices
    and absurd counts are KWS_ERROR_BAD_ARGUMENT (-20) everywhere -- never a crash."""
    good = open(pkg.DEFAULT_MODEL, "rb").read()
    import struct
    bad = [b"", b"KWS", b"KWSM", b"XXXX" + good[4:], good[:4] + struct.pack("<I", 2) + good[8:]]
    bad += [good[:n] for n in (8, 40, 100, 200, len(good) // 2, len(good) - 1)]
    for off in (8, 12, 16, 20, 24):                       # n_tensors, n_nodes, n_labels, input / output tensor index
        bad.append(good[:off] + struct.pack("<I", 0x7fffffff) + good[off + 4:])
    rng = __import__("numpy").random.default_rng(0)
    for _ in range(200):                                  # random single-word corruptions of the header / tables
        off = int(rng.integers(8, 400)) & ~3
        bad.append(good[:off] + struct.pack("<I", int(rng.integers(0, 2 ** 32))) + good[off + 4:])
    import torch
    n_bad_arg = 0
    for b in bad:
        try:
            m = pkg.Model(blob=b)
            m.close()                                      # a corruption may still be a loadable model (GPU box)
        except pkg.KwsError as e:
            assert e.code in (-20, -19, -18), e            # bad blob / no device (CPU box) / unsupported model
            n_bad_arg += e.code == -20
    assert n_bad_arg >= 16


def test_generated_dct_tables_are_current(tmp_path):
    """kws_dct_tables.h (the DCT constants the MFCC kernel carries as literals) is what tools/gen_dct_tables.cpp emits on
    this host; the library re-checks them against its run-time tables whenever a plan is built on a GPU."""
    import subprocess
    exe = tmp_path / "gen_dct"
    subprocess.run(["g++", "-O0", "-o", str(exe), os.path.join(ROOT, "tools", "gen_dct_tables.cpp")], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    with open(os.path.join(ROOT, "ei-keyword-spotting_amd", "csrc", "kws_dct_tables.h")) as f:
        assert f.read() == out


def test_product_library_carries_no_development_switches(pkg):
    """the KWS_DEV_* environment switches (guard off / scaled, no re-run, forced tiers and layouts) exist only in the
    development build (csrc/Makefile: `make dev`, -DKWS_DEV_SWITCHES -> libkws_mi355x_dev.so).  The product library does not contain
    their names -- so no environment can put KWS_MODE_FAST outside its tolerance -- and the development build, when present, does."""
    blob = open(pkg.LIB_PATH, "rb").read()
    assert b"KWS_DEV_" not in blob
    dev = os.path.join(os.path.dirname(pkg.LIB_PATH), "libkws_mi355x_dev.so")
    if os.path.exists(dev):
        names = set(re.findall(rb"KWS_DEV_[A-Z0-9_]+", open(dev, "rb").read()))
        assert {b"KWS_DEV_FAST_ENTRY", b"KWS_DEV_FAST_GUARD_SCALE", b"KWS_DEV_FAST_NO_RERUN", b"KWS_DEV_GENERIC_LCH", b"KWS_DEV_MFCC_OLD_LAYOUT"} <= names, names
