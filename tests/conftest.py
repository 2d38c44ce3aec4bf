import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: CPU test that takes more than ~20 s")


@pytest.fixture(scope="session")
def param_npz():
    return np.load(GOLDEN / "param_weights.npz")


@pytest.fixture(scope="session")
def param_words(param_npz):
    """[(words uint64 [PE][TILES], bias int8 [Cout])] * 8 — the reference's PARAM:: tables."""
    return [(param_npz[f"w{n}_words"], param_npz[f"b{n}"]) for n in range(8)]


@pytest.fixture(scope="session")
def param_closed_form():
    from oracle import sicn_ref
    return sicn_ref.load_param_fixture(GOLDEN / "param_weights.npz")
