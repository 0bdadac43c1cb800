import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def read_pgm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"P5"
        w, h = map(int, f.readline().split())
        assert int(f.readline()) == 255
        return np.frombuffer(f.read(), dtype=np.uint8).reshape(h, w).astype(np.float32)


def read_keypoints(path):
    """u32 n, then n x {x, y, scale, orientation} f32 (test/detector.cpp:52-63)."""
    raw = open(path, "rb").read()
    n = int(np.frombuffer(raw[:4], dtype="<u4")[0])
    return np.frombuffer(raw[4:], dtype="<f4").reshape(n, 4).copy()


@pytest.fixture(scope="session")
def gray1():
    return read_pgm(os.path.join(GOLDEN, "gray1.pgm"))


@pytest.fixture(scope="session")
def golden_check():
    return read_keypoints(os.path.join(GOLDEN, "cusift1_check.bin"))


@pytest.fixture(scope="session")
def golden_run2():
    return read_keypoints(os.path.join(GOLDEN, "cusift1.bin"))


@pytest.fixture(scope="session")
def oracle():
    from oracle_binding import Oracle

    return Oracle()


@pytest.fixture(scope="session")
def ctx():
    """A cusift context on GPU 0 (gpu tests only). Fails loudly if the HIP extension is missing."""
    from cusift_amd import capi

    capi.lib()  # raises CusiftError when libcusift_amd.so is absent -- no fallback
    if capi.device_count() < 1:
        pytest.fail("gpu test selected but no HIP device is visible")
    c = capi.Context(0)
    # CUSIFT_FORCE_GENERIC=1 pytest -m gpu ...: the whole GPU suite on the generic (any pitch / alignment) kernels
    if os.environ.get("CUSIFT_FORCE_GENERIC"):
        c.set_policy(capi.POLICY_GENERIC_KERNELS, 1)
    yield c
    c.close()
