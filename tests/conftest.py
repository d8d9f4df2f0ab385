import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
if GOLDEN not in sys.path:
    sys.path.insert(0, GOLDEN)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """numpy.load with allow_pickle=False (the default): data only."""
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


def allclose(x, y, tol=1e-5):
    """The reference tests' own comparison (tests/test_nn/test_dense.py:11-12)."""
    return np.allclose(np.asarray(x), np.asarray(y), atol=tol, rtol=tol)


def allclose_scaled(x, ref, tol=1e-5):
    """1e-5 relative to the output scale: |x - ref| <= tol * max(1, rms(ref)) + tol * |ref|.

    Used only for whole-network outputs whose magnitude is far from 1 (the
    784-1200-1200-10 net at batch 512 has output rms 36, max 135): there the
    reference's own fp32 sgemm differs from exact accumulation by 7.6e-5, so a
    fixed atol of 1e-5 would reject the reference itself."""
    x = np.asarray(x, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    scale = max(1.0, float(np.sqrt((ref ** 2).mean())))
    return bool((np.abs(x - ref) <= tol * scale + tol * np.abs(ref)).all())


def assert_close_scaled(x, ref, tol=1e-5, what=""):
    """allclose_scaled that ASSERTS (a bare allclose_scaled(...) call checks nothing) and reports the worst element."""
    x = np.asarray(x, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert x.shape == ref.shape, (what, x.shape, ref.shape)
    scale = max(1.0, float(np.sqrt((ref ** 2).mean())))
    excess = np.abs(x - ref) - (tol * scale + tol * np.abs(ref))
    worst = int(np.argmax(excess))
    assert bool((excess <= 0).all()), "%s: max |err| %.3e (allowed %.3e at that element, output scale %.3g, %d of %d elements out)" % (
        what, float(np.abs(x - ref).reshape(-1)[worst]), float((tol * scale + tol * np.abs(ref)).reshape(-1)[worst]), scale,
        int((excess > 0).sum()), excess.size)
    return float(np.abs(x - ref).max() / scale)


@pytest.fixture(scope="session", autouse=True)
def _device_error_word_is_clear_at_the_end():
    """GPU sessions end by reading the device error word (bnn_check_device): a kernel whose bounded hand-off wait gave up skips
    its stores and says so there -- a test that passed on stale outputs must not pass silently.  CPU sessions: nothing."""
    yield
    import sys
    if "torch" not in sys.modules:
        return
    import torch
    if not torch.cuda.is_available():
        return
    from bayesianneuralnetworks_amd import _lib
    if _lib._lib is None:              # no GPU test loaded the library
        return
    for i in range(torch.cuda.device_count()):
        _lib.check_device(torch.device("cuda", i))
