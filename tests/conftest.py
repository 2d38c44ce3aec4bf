import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


ALT_LIB = ROOT / "simple_image_compression_network_amd" / "libsicn_alt.so"


def running_on_alt_library() -> bool:
    """True in the child processes that the two `test_alt_build_*` driver tests start with SICN_LIB = libsicn_alt.so."""
    import os
    return os.environ.get("SICN_LIB", "") != "" and Path(os.environ["SICN_LIB"]).name == ALT_LIB.name


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: CPU test that takes more than ~20 s")
    config.addinivalue_line("markers", "alt: exercises a kernel form that lives in the ALT build only (libsicn_alt.so: k_l0p, k_l7s, k_l7g, the K "
                                       "split — forms that measured a loss against the product's defaults).  Skipped in an ordinary session; the "
                                       "driver tests test_alt_build_* run them in a child pytest whose SICN_LIB is the ALT library")


def pytest_collection_modifyitems(config, items):
    """`alt` tests only run where the library is the ALT build; everything else only where it is NOT (the child session is there for the alt
    tests alone)."""
    on_alt = running_on_alt_library()
    keep, drop = [], []
    for it in items:
        (keep if (it.get_closest_marker("alt") is not None) == on_alt else drop).append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


def run_alt_session(marker_expr: str, timeout: int = 1500):
    """Start `pytest -m "<marker_expr>"` over tests/ in a child process that loads libsicn_alt.so; returns the CompletedProcess."""
    import os
    import subprocess
    assert ALT_LIB.exists(), "build() makes libsicn_alt.so"
    return subprocess.run([sys.executable, "-m", "pytest", str(ROOT / "tests"), "-x", "-q", "-m", marker_expr, "-p", "no:cacheprovider"],
                          capture_output=True, text=True, env=dict(os.environ, SICN_LIB=str(ALT_LIB)), timeout=timeout, cwd=str(ROOT))


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
