import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
