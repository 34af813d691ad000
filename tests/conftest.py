import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def fe():
    """One opusfe context on cuda:0 for the whole GPU session."""
    import opus_amd
    ctx = opus_amd.Context(0)
    yield ctx
    ctx.close()
