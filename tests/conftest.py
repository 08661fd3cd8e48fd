import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(HERE, "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gps_log_bytes():
    """The one data fixture the reference ships (data/original_gps_data.txt), copied verbatim."""
    with open(os.path.join(GOLDEN, "original_gps_data.txt"), "rb") as f:
        return f.read()
