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


def lookat(eye, c, up):
    eye, c, up = (np.asarray(a, np.float32) for a in (eye, c, up))
    f = c - eye
    f /= np.linalg.norm(f)
    s = np.cross(f, up)
    s /= np.linalg.norm(s)
    u = np.cross(s, f)
    V = np.eye(4, dtype=np.float32)
    V[0, :3], V[1, :3], V[2, :3] = s, u, -f
    V[0, 3], V[1, 3], V[2, 3] = -s @ eye, -u @ eye, f @ eye
    return V


def persp(fov, aspect, n, f, flip=False):
    t = np.tan(np.radians(fov) / 2)
    P = np.zeros((4, 4), np.float32)
    P[0, 0] = 1 / (aspect * t)
    P[1, 1] = (-1 if flip else 1) / t
    P[2, 2] = f / (n - f)
    P[3, 2] = -1
    P[2, 3] = -(f * n) / (f - n)
    return P


@pytest.fixture(scope="session")
def ob():
    from oracle import binding
    binding.build()
    return binding


@pytest.fixture(scope="session")
def golden_meta():
    import json
    return json.load(open(os.path.join(GOLDEN, "meta.json")))
