"""pytest configuration: the `gpu` marker, import paths and shared builders.

CPU suite  (-m "not gpu"): oracle vs golden vectors / independent numpy-scipy formulations, the host
           C++ logic (IESKF, map insert rule), C-ABI symbol export, the 2-rank gloo bench plumbing.
GPU suite  (-m gpu): parity of the HIP path against the oracle through the C ABI.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device")


@pytest.fixture(scope="session")
def built():
    """Make sure the native libraries exist (hipcc cross-compiles without a GPU)."""
    from fast_limo_amd import build as b
    from fast_limo_amd import _lib
    # never a stale binary: the libraries carry the hash of the sources they were built from (fast_limo_amd/.build_stamp);
    # after a source edit they are rebuilt here, otherwise this is a no-op
    if b.is_stale():
        b.build_native()
    import oracle_py
    oracle_py.build()
    return True


@pytest.fixture(scope="session")
def oracle(built):
    import oracle_py
    return oracle_py
