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
