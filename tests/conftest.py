import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def port():
    from oracle import oracle_lib
    return oracle_lib.load("port")


@pytest.fixture(scope="session")
def reference():
    from oracle import oracle_lib
    if not oracle_lib.available("reference"):
        pytest.skip("oracle/_ref/libmdc_ref.so not built (reference tree absent)")
    return oracle_lib.load("reference")
