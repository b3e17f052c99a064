import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests never run by accident on a CPU-only host
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _native_library():
    """The HIP library normally travels with the tree (built by __graft_entry__.build()); on a checkout without it, build it
    once (hipcc is part of the image).  Nothing here falls back to a CPU path: if the build fails the GPU tests fail."""
    from vae_equalizer_amd import _native as nat
    if not os.path.exists(nat.LIB_PATH):
        nat.build(verbose=True)
    yield


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


def relerr(a, b):
    """max |a-b| / max |b|  (error relative to the largest reference magnitude)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
