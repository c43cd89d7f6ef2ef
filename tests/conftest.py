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


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o


def spheres_from(centres, radii):
    """[N,J,>=3] centres + [J] radii -> [N,J,4] fp32 (x,y,z,r)."""
    c = np.asarray(centres, np.float32)[..., :3]
    r = np.broadcast_to(np.asarray(radii, np.float32)[None, :, None], c.shape[:2] + (1,))
    return np.ascontiguousarray(np.concatenate([c, r], -1), np.float32)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)
