import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "swift-homomorphic-encryption_amd")
for p in (ROOT, PKG, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def kats():
    with open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    import oracle as orc

    orc.build()
    return orc


def host_threads():
    """Host threads the multi-threaded oracle may use (the affinity mask, not the machine's core count)."""
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        return os.cpu_count() or 1


# Full-size parity tests compare EVERY unit with the oracle when the host can do that in seconds (the GPU boxes have 256
# threads; 8 are enough for every BASELINE size), and fall back to a sample spread over the slab on a smaller host.
# HEAMD_EXHAUSTIVE=0/1 forces either branch.
def exhaustive_parity():
    forced = os.environ.get("HEAMD_EXHAUSTIVE")
    if forced is not None:
        return forced not in ("0", "", "false")
    return host_threads() >= 8
