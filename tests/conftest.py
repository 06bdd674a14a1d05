import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.pyoracle import Oracle

    return Oracle("port")


@pytest.fixture(scope="session")
def reference():
    """The reference's own device code compiled for the host (oracle/_ref); skipped when absent."""
    from oracle import build_ref
    from oracle.pyoracle import Oracle, have_reference

    if build_ref.available():
        build_ref.build()
    if not have_reference():
        pytest.skip("oracle/_ref/libevogp_ref.so not available (needs /root/reference at build time)")
    return Oracle("reference")


@pytest.fixture
def rng(request):
    """a generator of the test's OWN: seeded by the test's node id, so that its draws depend neither on which other tests ran nor on
    their order (rounds 1-4 shared one session-wide generator: adding or removing a test changed the data of every later one)"""
    import zlib

    return np.random.default_rng([20260925, zlib.crc32(request.node.nodeid.encode())])
