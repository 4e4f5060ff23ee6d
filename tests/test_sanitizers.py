"""AddressSanitizer + UndefinedBehaviorSanitizer over the host code (SURVEY.md section 5 "race detection / sanitizers"; VERDICT
round 2 item 8).  No GPU needed:
  * the library's HOST side -- .kwsm parser, plan builders (DSP / int8 / float / fast-mode tables), table uploads, the C ABI's argument
    checks, scratch management, continuous-mode bookkeeping, SDK entry points -- is compiled from the same sources
    (--cuda-host-only) with -fsanitize=address,undefined and linked against tests/sanitize/hip_stub.cpp, where "device" memory is
    host heap and kernel launches do nothing: an upload that over-reads, a table indexed out of range or a signed overflow in a plan
    builder aborts the run.  It is fed every shipped model and the mutation fuzzer's blobs (tests/fuzz_worker.py), which the
    un-sanitised fuzz test can only catch when they crash;
  * the C oracle (oracle/kws_oracle.c) is built the same way and the golden-vector tests are run against that build.
Everything is built into a scratch directory; nothing here is part of, or linked into, the product."""
import glob
import os
import shutil
import subprocess
import sys

import pytest

from kws_testlib import MODELS, ROOT

CLANG = "/opt/rocm/lib/llvm/bin/clang++"
pytestmark = pytest.mark.skipif(not (os.path.exists(CLANG) and shutil.which("gcc")), reason="needs ROCm's clang++ and gcc")


@pytest.fixture(scope="module")
def san_dir(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("kws_sanitize"))
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "sanitize"), "OUT=" + out])
    return out


def check_clean(proc, what):
    assert proc.returncode == 0 and "Sanitizer" not in proc.stderr and "runtime error" not in proc.stderr, \
        "%s: exit %d\n%s\n%s" % (what, proc.returncode, proc.stdout[-1500:], proc.stderr[-4000:])


def test_host_code_under_asan_ubsan_with_shipped_and_mutated_models(san_dir, tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from fuzz_worker import mutations
    exe = os.path.join(san_dir, "kws_host_san")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    models = sorted(glob.glob(os.path.join(MODELS, "*.kwsm")))
    p = subprocess.run([exe] + models, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=env)
    check_clean(p, "shipped models")
    assert p.stdout.count(" rc 0") == len(models)                 # every shipped model is accepted (and walked through the C ABI)
    served = refused = 0
    for m in models:
        blob = open(m, "rb").read()
        files = []
        for i, (what, b) in enumerate(mutations(blob, 21, 120)):
            fn = str(tmp_path / ("%s.%03d" % (os.path.basename(m), i)))
            open(fn, "wb").write(b)
            files.append(fn)
        p = subprocess.run([exe] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env)
        check_clean(p, "mutations of " + os.path.basename(m))
        codes = [int(ln.split()[-1]) for ln in p.stdout.splitlines() if " rc " in ln]
        assert len(codes) == len(files) and set(codes) <= {0, -8, -18, -20}, sorted(set(codes))
        served += codes.count(0)
        refused += len(codes) - codes.count(0)
        for fn in files:
            os.remove(fn)
    assert served > 50 and refused > 200                           # both the accepting and the refusing paths were exercised


def test_oracle_under_asan_ubsan(san_dir):
    lib = os.path.join(san_dir, "libkws_oracle_san.so")
    asan_rt = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    env = dict(os.environ, KWS_ORACLE_SO=lib, LD_PRELOAD=asan_rt, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_oracle_golden.py"),
                        os.path.join(ROOT, "tests", "test_mix_audio.py")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env, cwd=ROOT)
    check_clean(p, "oracle golden tests")
    assert " passed" in p.stdout and "failed" not in p.stdout, p.stdout[-2000:]
    maps = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, 'tests'); from kws_testlib import Oracle; Oracle(); print(open('/proc/self/maps').read())"],
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT).stdout
    assert "libkws_oracle_san.so" in maps and "libkws_oracle.so" not in maps      # the sanitised build was the one under test
