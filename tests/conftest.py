import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oraclelib
    return oraclelib.Oracle()


@pytest.fixture(scope="session")
def ref():
    import oraclelib
    if not oraclelib.Ref.available():
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    return oraclelib.Ref()


@pytest.fixture(scope="session")
def decoder():
    import oraclelib
    return oraclelib.Decoder()
