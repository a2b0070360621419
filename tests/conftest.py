import os
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

GOLDEN = REPO / "tests" / "golden"
CALIB = REPO / "tacex_amd" / "assets" / "calib" / "gsmini_640x480"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def calib_dir():
    return CALIB


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
