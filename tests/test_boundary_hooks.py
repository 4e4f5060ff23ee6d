"""The two boundary behaviours an application sees besides the scores:

  * debug = true -- the text run_classifier / run_inference / run_classifier_continuous print through the application's ei_printf
    (classifier/ei_run_classifier.h:698-705, 463-479; continuous :242-253);
  * the cancellation hook -- ei_run_impulse_check_canceled is polled behind the DSP block (:689; continuous :221) and twice by
    run_inference once the result is written (:489, :636): a call cancelled at its n-th poll returns EI_IMPULSE_CANCELED (-2) with the
    caller's ei_impulse_result_t touched exactly where the reference has touched it by then, and continuous mode's state moves on as the
    reference's does (a slice cancelled behind the DSP block is not committed; one cancelled inside run_inference still filters and shifts).

tests/golden/debug_cancel_l476.npz holds what the compiled reference does in the scenarios of tools/make_golden.py BOUNDARY_SCENARIOS
(tools/make_golden.py --only-debug-cancel).  tests/boundary/boundary_driver.c is a C11 application with its own hooks that walks the same
scenarios through the library:
  * CPU (stub HIP runtime, tests/sanitize): kernels do not run, so numbers mean nothing -- return codes, poll counts, WHICH bytes of the
    result were written, the labels and the text with every number masked must equal the reference's;
  * GPU (-m gpu): everything must be equal, the text character for character except the `%d ms` fields.
"""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from kws_testlib import GOLDEN, MODELS, ROOT, have_reference

sys.path.insert(0, os.path.join(ROOT, "tools"))


def scenarios():
    from make_golden import BOUNDARY_SCENARIOS
    return BOUNDARY_SCENARIOS


def write_inputs(tmp_path, g):
    inp, scn = str(tmp_path / "input.bin"), str(tmp_path / "scenarios.txt")
    with open(inp, "wb") as f:
        f.write(np.ascontiguousarray(g["clips"], np.int16).tobytes())
        f.write(np.ascontiguousarray(g["features_a"], np.float32).tobytes())
    with open(scn, "w") as f:
        for name, kind, what, debug, cancel_at in scenarios():
            f.write("%s %s %s %d %d %d\n" % (name, kind, what[0] if what else "-", what[1] if what else 0, debug, cancel_at))
    return inp, scn


def parse(stdout):
    """{name: (rc, polls, result bytes, labels, text)} from boundary_driver's output (bytes)"""
    res, pos = {}, 0
    while True:
        m = re.compile(rb"SCEN (\S+) (-?\d+) (\d+) (\d+)\nRES ([0-9a-f]+)\nLAB([^\n]*)\n").search(stdout, pos)
        if not m:
            break
        n = int(m.group(4))
        text = stdout[m.end():m.end() + n]
        assert stdout[m.end() + n:m.end() + n + 5] == b"\nEND\n", m.group(1)
        res[m.group(1).decode()] = (int(m.group(2)), int(m.group(3)), np.frombuffer(bytes.fromhex(m.group(5).decode()), np.uint8),
                                    m.group(6).decode().split(), text)
        pos = m.end() + n + 5
    return res


NUM = re.compile(rb"-?\d+\.\d+|-?nan|-?inf")
MS = re.compile(rb"(\d+)( ms\.)")


def mask_ms(t):
    return MS.sub(rb"#\2", t)


def skeleton(t):
    return NUM.sub(b"#", mask_ms(t))


def fields(res, n_labels=4):
    """the struct's fields as (touched?) flags: label pointer and value per class, anomaly, the four timing ints (classifier/ei_classifier_types.h:30-52)"""
    t = lambda a, b: bool((res[a:b] != 0xA5).any())   # noqa: E731
    out = []
    for i in range(n_labels):
        out += [t(16 * i, 16 * i + 8), t(16 * i + 8, 16 * i + 12)]
    base = 16 * n_labels
    return out + [t(base, base + 4)] + [t(base + 4 + 4 * k, base + 8 + 4 * k) for k in range(4)]


def check(got, g, exact):
    labels = [str(x) for x in g["labels"]]
    for name, kind, _what, _debug, _cancel in scenarios():
        if kind == "init":
            continue
        rc, polls, res, lab, text = got[name]
        want_rc, want_polls = g[name + "_meta"].tolist()
        assert (rc, polls) == (want_rc, want_polls), (name, rc, polls, want_rc, want_polls)
        ref_res = g[name + "_result"]
        assert res.size == ref_res.size
        assert fields(res) == fields(ref_res), (name, fields(res), fields(ref_res))
        assert lab == [labels[i] if g[name + "_labels"][i] else "-" for i in range(len(labels))], (name, lab)
        ref_text = g[name + "_text"].tobytes()
        if exact:
            assert mask_ms(text) == mask_ms(ref_text), (name, text[:300], ref_text[:300])
            # the scores the application reads: the same bits (the timing ints differ)
            for i in range(len(labels)):
                assert (res[16 * i + 8:16 * i + 12] == ref_res[16 * i + 8:16 * i + 12]).all(), (name, i)
        else:
            assert skeleton(text) == skeleton(ref_text), (name, text[:300], ref_text[:300])


def test_fixture_shapes_the_reference_itself():
    """the facts the library is held to, read off the fixture: three polls per classified window, one behind the DSP block; a call cancelled behind
    the DSP block leaves the result alone; one cancelled inside run_inference has the scores"""
    g = np.load(os.path.join(GOLDEN, "debug_cancel_l476.npz"))
    assert g["oneshot_dbg_a_meta"].tolist() == [0, 3] and g["inference_dbg_a_meta"].tolist() == [0, 2]
    assert g["oneshot_cancel1_meta"].tolist() == [-2, 1] and not (g["oneshot_cancel1_result"] != 0xA5).any()
    assert g["oneshot_cancel2_meta"].tolist() == [-2, 2] and fields(g["oneshot_cancel2_result"]) == fields(g["oneshot_dbg_a_result"])
    assert g["cont_dbg_0_meta"].tolist() == [0, 1] and g["cont_dbg_3_meta"].tolist() == [0, 3]
    assert g["cont_cancel1_first_meta"].tolist() == [-2, 1] and g["cont_cancel2_full_meta"].tolist() == [-2, 2]
    t = g["oneshot_dbg_a_text"].tobytes()
    assert t.startswith(b"Features (") and b"\nRunning neural network...\nPredictions (time: " in t and t.count(b" ") >= 637
    assert g["cont_dbg_0_text"].tobytes().startswith(b"\r\nFeatures (") and g["oneshot_cancel1_dbg_text"].size == 0


def test_hooks_on_the_stub_runtime(host_exe, tmp_path):
    g = np.load(os.path.join(GOLDEN, "debug_cancel_l476.npz"))
    inp, scn = write_inputs(tmp_path, g)
    exe = os.path.join(os.path.dirname(host_exe), "kws_boundary_san")
    env = dict(os.environ, KWS_MODEL=os.path.join(MODELS, "l476_no_yes.kwsm"))
    out = subprocess.run([exe, inp, scn], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=env)
    assert out.returncode == 0 and b"Sanitizer" not in out.stderr, (out.stdout[-500:], out.stderr[-2000:])
    check(parse(out.stdout), g, exact=False)


@pytest.mark.gpu
def test_hooks_on_the_gpu(tmp_path):
    from __graft_entry__ import load_package
    pkg = load_package()
    g = np.load(os.path.join(GOLDEN, "debug_cancel_l476.npz"))
    inp, scn = write_inputs(tmp_path, g)
    exe = str(tmp_path / "boundary_driver")
    libdir = os.path.dirname(pkg.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "boundary", "boundary_driver.c"),
                           "-L" + libdir, "-lkws_mi355x", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    env = dict(os.environ, KWS_MODEL=os.path.join(MODELS, "l476_no_yes.kwsm"))
    out = subprocess.run([exe, inp, scn], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=env)
    assert out.returncode == 0, (out.stdout[-500:], out.stderr[-2000:])
    check(parse(out.stdout), g, exact=True)


@pytest.mark.skipif(not have_reference(), reason="oracle/_ref not built (no /root/reference here)")
def test_fixture_is_what_the_compiled_reference_does(tmp_path):
    # a fresh process: the reference's first_run (ei_run_dsp.h:313) is function-static
    code = ("import sys, os, numpy as np\n"
            "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import make_golden\n"
            "make_golden.GOLDEN = %r\n"
            "from kws_testlib import Oracle, Reference, L476_CONFIG\n"
            "make_golden.debug_cancel(Reference(), Oracle(), L476_CONFIG())\n") % (os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools"), str(tmp_path))
    subprocess.check_call([sys.executable, "-c", code], stdout=subprocess.DEVNULL, timeout=300)
    a, b = np.load(os.path.join(str(tmp_path), "debug_cancel_l476.npz")), np.load(os.path.join(GOLDEN, "debug_cancel_l476.npz"))
    assert sorted(a.files) == sorted(b.files)
    for k in a.files:
        if k.endswith("_text"):
            assert mask_ms(a[k].tobytes()) == mask_ms(b[k].tobytes()), k
        elif k.endswith("_result"):
            assert fields(a[k]) == fields(b[k]), k
        else:
            assert a[k].shape == b[k].shape and (a[k] == b[k]).all(), k
