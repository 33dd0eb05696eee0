import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session", autouse=True)
def _oracle_threads():
    """The C oracle's OpenMP loops use the CPUs this container is actually granted (hosts show every core)."""
    try:
        from oracle import c_oracle
        c_oracle.Oracle.set_threads(c_oracle.usable_cpus())
    except Exception:       # the oracle library is built on first use by the tests that need it
        pass
    yield
