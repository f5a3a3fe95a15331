import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "minbpe-cc_amd", "python")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(GOLDEN, "data")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    import __graft_entry__ as ge
    ge.build()


def read_data(name):
    with open(os.path.join(DATA, name), "rb") as f:
        return f.read()


def read_golden(name):
    with open(os.path.join(GOLDEN, name), "rb") as f:
        return f.read()
