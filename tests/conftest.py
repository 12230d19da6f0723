import os
import sys

import pytest

# exercise the Winograd path on the small-channel operator cases too (product default: >= 256 channels)
os.environ.setdefault("SWN_WINO_MINC", "32")

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(REPO, "tests", "golden")
