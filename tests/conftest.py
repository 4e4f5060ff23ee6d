import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def host_exe(tmp_path_factory):
    """The library's HOST code linked against the stub HIP runtime of tests/sanitize (device memory = host heap, launches do nothing):
    what kws_create builds and what the SDK entry points do before any device work can be looked at without a GPU."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not (os.path.exists("/opt/rocm/lib/llvm/bin/clang++") and shutil.which("gcc")):
        pytest.skip("needs ROCm's clang++ and gcc")
    out = str(tmp_path_factory.mktemp("kws_host_stub"))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "tests", "sanitize"), "OUT=" + out, os.path.join(out, "kws_host_san"),
                           os.path.join(out, "kws_boundary_san")])
    return os.path.join(out, "kws_host_san")


@pytest.fixture(scope="session")
def dev_pkg():
    """The binding over libkws_mi355x_dev.so (built with -DKWS_DEV_SWITCHES): the product library does not read the KWS_DEV_* environment
    switches, so tests that force a tier or a kernel layout run on this build of the same sources."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch  # noqa: F401
    from __graft_entry__ import load_package
    return load_package(dev=True)


@pytest.fixture(scope="session")
def oracle():
    from kws_testlib import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    from kws_testlib import Reference, have_reference
    if not have_reference():
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    return Reference()


@pytest.fixture(scope="session")
def l476(oracle):
    from kws_testlib import MODELS, OracleModel
    return OracleModel(oracle, os.path.join(MODELS, "l476_no_yes.kwsm"))


@pytest.fixture(scope="session")
def l432(oracle):
    from kws_testlib import MODELS, OracleModel
    return OracleModel(oracle, os.path.join(MODELS, "l432_trick_or_treat.kwsm"))
