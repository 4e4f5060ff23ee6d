"""Mutation / truncation fuzz over the shipped .kwsm blobs: kws_create must answer a malformed blob with an error code --
KWS_ERROR_BAD_ARGUMENT (-20) from the parser, KWS_ERROR_UNSUPPORTED_MODEL (-18) from the plan builders -- never with a crash.
Without a GPU the call ends after the parser (KWS_ERROR_HIP, -19, for a blob that parses); on the GPU the plan builders run too."""
import glob
import os
import subprocess
import sys

import pytest

from kws_testlib import MODELS, ROOT

WORKER = os.path.join(ROOT, "tests", "fuzz_worker.py")


def run_worker(path, seed, n):
    out = subprocess.run([sys.executable, WORKER, path, str(seed), str(n)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    lines = [ln for ln in out.stdout.splitlines() if " rc " in ln or " begin " in ln]
    last = lines[-1] if lines else "(nothing)"
    assert out.returncode == 0, "kws_create crashed (exit %d) at: %s\n%s" % (out.returncode, last, out.stderr[-1500:])
    return [int(ln.split()[2]) for ln in lines if " rc " in ln]


@pytest.mark.parametrize("name", sorted(os.path.basename(p) for p in glob.glob(os.path.join(MODELS, "*.kwsm"))))
def test_malformed_blobs_are_refused_by_the_parser(name):
    codes = run_worker(os.path.join(MODELS, name), 11, 160)
    assert len(codes) == 160
    assert set(codes) <= {-20, -19, -18, 0}, sorted(set(codes))
    assert codes.count(-20) >= 30                                  # the truncations at least


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["l476_no_yes.kwsm", "cfg2_mfcc40_f32.kwsm", "cfg5_dscnn_mfcc40_int8.kwsm", "cfg5_dscnn_mfcc40_f32.kwsm"])
def test_malformed_blobs_do_not_crash_the_plan_builders(name):
    codes = run_worker(os.path.join(MODELS, name), 12, 400)
    assert len(codes) == 400
    assert set(codes) <= {-20, -18, 0, -8}, sorted(set(codes))
    assert codes.count(-18) >= 5 and codes.count(-20) >= 30
