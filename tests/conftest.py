import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "dev_variants: runs with the package pointed at libtomo_mi355x_dev.so, the build that "
                                       "also carries the independent kernel implementations / A-B variants")


@pytest.fixture(autouse=True)
def _library_flavour(request):
    """GPU tests run the SHIPPED library (tomobar_amd/libtomo_mi355x.so) with every kernel class at its default -- since
    round 4 the defaults reproduce the oracle's roundings, so bit-for-bit comparisons need no variant switch.  A test (or
    parameter) marked ``dev_variants`` compares another implementation of a kernel: for its duration the package is
    pointed at libtomo_mi355x_dev.so (same sources + -DTOMO_DEV_VARIANTS).  Variant switches are per library and are
    reset after every test."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    from tomobar_amd import _lib, ops
    dev = request.node.get_closest_marker("dev_variants") is not None
    ctx = _lib.use_flavour("dev") if dev else None
    try:
        yield
    finally:
        for k in ("bp", "fp", "pdtv", "roftv"):
            ops.set_variant(k, 0)
        if ctx is not None:
            ctx.__exit__(None, None, None)


@pytest.fixture(scope="session")
def oracle():
    from oracle import tomo_oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
