import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc32():
    from oracle.orc import Oracle
    o = Oracle("f32")
    o.set_threads(min(8, os.cpu_count() or 1))
    return o


@pytest.fixture(scope="session")
def orc64():
    from oracle.orc import Oracle
    o = Oracle("f64")
    o.set_threads(min(8, os.cpu_count() or 1))
    return o
