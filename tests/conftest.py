import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "point-plane-object-slam_amd"))
sys.path.insert(0, os.path.join(REPO, "oracle"))
sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle_py
    return oracle_py.load()


@pytest.fixture(scope="session")
def oracle():
    import oracle_py
    return oracle_py.backend()


@pytest.fixture(scope="session")
def hip():
    from ppo_ba import binding
    return binding.hip_backend()


def dptr(a):
    import ctypes
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


@pytest.fixture(scope="session")
def P():
    return dptr


def rot_angle(q1, q2):
    """Angle (rad) between two unit quaternions given as (x, y, z, w)."""
    d = abs(float(np.dot(q1, q2)))
    return 2.0 * np.arccos(min(1.0, d))
