import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import gomel_oracle
    return gomel_oracle


@pytest.fixture(scope="session")
def coracle():
    """The C restatement (oracle/libgomel_oracle.so), built on demand."""
    import oracle_c
    return oracle_c.load()


@pytest.fixture
def fp32_start_mode(monkeypatch):
    """Since round 4 a handle's default arithmetic is the calibrated fp64-then-fp32 schedule (gomel_set_precision(h, 0)).  The test
    modules written for the fp32 kernels -- now the opt-in fast mode -- start their handles in that mode through the library's
    GOMEL_PRECISION knob (read at gomel_create); a test that selects a mode itself overrides it.  tests/test_gpu_round4.py and the
    full-size tests of test_gpu_round2.py check what a caller gets WITHOUT any knob."""
    monkeypatch.setenv("GOMEL_PRECISION", "32")
