"""CPU: the sequence of calls run_classifier / run_classifier_continuous make to the APPLICATION's signal_t::get_data (SURVEY 8(b);
VERDICT round 3, weak item 11: "a caller whose callback is stateful sees a different sequence").

tests/golden/get_data_trace_l476.npz holds what the compiled reference asks its callback -- (offset, length, return value) per call -- in
five scenarios (tools/make_golden.py --only-trace): 98 calls for the shipped one-shot window (processing.hpp:68, 86-94 under
feature.hpp:263-281), the process's first continuous slice, a later one (total_length grown by a frame length in the caller's struct,
ei_run_dsp.h:318-326; the constructor's look beyond the slice is refused by the callback and ignored), a half slice, and a one-shot
window one sample short.  The library's gather is host code that runs before any device work, so the stub build
(tests/sanitize/host_driver.cpp --trace) shows it without a GPU; tests/test_gpu_parity.py repeats the one-shot case on the real path.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from kws_testlib import GOLDEN, MODELS, ROOT, have_reference

SCENARIOS = ("oneshot", "oneshot_short", "continuous_first", "continuous_second", "continuous_short")


def product_traces(host_exe):
    out = subprocess.run([host_exe, "--trace", os.path.join(MODELS, "l476_no_yes.kwsm")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert out.returncode == 0 and "Sanitizer" not in out.stderr, out.stderr[-2000:]
    res = {}
    text = out.stdout
    for ln in text.splitlines():
        if ln.startswith("TRACE "):
            head, tail = ln.split("|")
            w = head.split()
            t = tail.split()
            res[w[1]] = (np.int64(w[2:]).reshape(-1, 3), int(t[1]), int(t[3]))
    return res


def test_library_asks_the_callback_what_the_reference_asks(host_exe):
    g = np.load(os.path.join(GOLDEN, "get_data_trace_l476.npz"))
    got = product_traces(host_exe)
    assert set(got) == set(SCENARIOS)
    assert len(g["oneshot"]) == 98 and len(g["continuous_first"]) == 22 and len(g["continuous_second"]) == 24
    for k in SCENARIOS:
        trace, total_after, err = got[k]
        assert trace.shape == g[k].shape and (trace == g[k]).all(), (k, trace[:6].tolist(), g[k][:6].tolist())
        assert [total_after, err] == g[k + "_meta"].tolist(), k
    # (a one-shot window one sample short: 48 frames, 96 calls, no error -- tests/test_other_window_length.py holds the results)
    assert len(g["oneshot_short"]) == 96 and g["oneshot_short_meta"].tolist() == [15999, 0]


@pytest.mark.skipif(not have_reference(), reason="oracle/_ref not built (no /root/reference here)")
def test_golden_trace_is_what_the_compiled_reference_does(tmp_path):
    # a fresh process: the reference's first_run (ei_run_dsp.h:313) is function-static
    code = ("import sys, os, numpy as np\n"
            "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import make_golden\n"
            "make_golden.GOLDEN = %r\n"
            "from kws_testlib import Reference\n"
            "make_golden.get_data_trace(Reference())\n") % (os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools"), str(tmp_path))
    subprocess.check_call([sys.executable, "-c", code], stdout=subprocess.DEVNULL, timeout=300)
    a, b = np.load(os.path.join(str(tmp_path), "get_data_trace_l476.npz")), np.load(os.path.join(GOLDEN, "get_data_trace_l476.npz"))
    assert sorted(a.files) == sorted(b.files)
    for k in a.files:
        assert a[k].shape == b[k].shape and (a[k] == b[k]).all(), k
