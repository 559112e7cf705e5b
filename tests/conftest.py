import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

REFERENCE_MODELS = "/root/reference/resources/models/testing/"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pbr():
    import pbr_loader
    return pbr_loader.load()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.lib()
    return orc


@pytest.fixture()
def cfg_defaults(pbr):
    """Reference config.json defaults before and after each test that touches Cfg."""
    pbr.cfg_reset()
    yield pbr
    pbr.cfg_reset()


@pytest.fixture(scope="session")
def gpu_device(pbr):
    """A live device context; GPU tests fail (not skip) when the HIP path is unavailable."""
    dev = pbr.Device(0)
    dev.close()
    return 0


def same_values(a, b):
    """Bit-for-bit parity compared numerically: -0 == +0 and NaN == NaN."""
    return np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


def describe_mismatch(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    bad = ~((a == b) | (np.isnan(a) & np.isnan(b)))
    idx = np.argwhere(bad)
    d = np.abs(a.astype(np.float64) - b.astype(np.float64))
    d = d[np.isfinite(d)]
    first = tuple(idx[0]) if len(idx) else None
    return "%d of %d values differ, max |d| = %.3g, first at %s: %r vs %r" % (
        bad.sum(), a.size, d.max() if d.size else 0.0, first,
        a[first] if first is not None else None, b[first] if first is not None else None)
