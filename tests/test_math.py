"""The written-out transcendental functions shared by the HIP kernels and the CPU oracle
(cusift_amd/csrc/sift_math.h), bounded against float64 on the CPU.

The reference calls CUDA's libm (expf, exp2f, atan2f, sinf, cosf in cuSIFT_D.cu:209-210,233,330,349,507), which is
documented to be accurate to 1-2 ulp; the same bound is demanded of the shared evaluation here, so that sharing it
between product and oracle cannot hide an inaccurate function.  Bit identity of the device build is checked on the
GPU (tests/test_gpu_parity.py::test_math_device_equals_host).
"""
import numpy as np
import pytest


def ulp_err(got, want64):
    ulp = np.spacing(np.abs(want64.astype(np.float32))).astype(np.float64)
    return np.abs(got.astype(np.float64) - want64) / ulp


@pytest.fixture(scope="module")
def rng():
    return np.random.default_rng(2024)


def test_expf_within_one_ulp(oracle, rng):
    x = np.concatenate([rng.uniform(-104, 89, 400000), rng.uniform(-1, 1, 200000), rng.uniform(-20, 0, 400000),
                        np.linspace(-87.4, -87.2, 1001)]).astype(np.float32)
    got, want = oracle.math_eval("exp", x), np.exp(x.astype(np.float64))
    normal = (want > 1.2e-38) & (want < 3.0e38)
    assert ulp_err(got[normal], want[normal]).max() <= 1.0
    # gradual underflow: within one denormal step; beyond the range: 0 / inf; NaN stays NaN
    den = want <= 1.2e-38
    assert np.abs(got[den].astype(np.float64) - want[den]).max() <= 1.5e-45
    sp = oracle.math_eval("exp", np.array([-np.inf, -200.0, 200.0, np.inf, np.nan, 0.0, -0.0], dtype=np.float32))
    assert sp[0] == 0 and sp[1] == 0 and np.isinf(sp[2]) and np.isinf(sp[3]) and np.isnan(sp[4])
    assert sp[5] == 1.0 and sp[6] == 1.0


def test_exp2f_within_one_ulp(oracle, rng):
    x = np.concatenate([rng.uniform(-150, 128, 400000), rng.uniform(-1, 1, 400000)]).astype(np.float32)
    got, want = oracle.math_eval("exp2", x), np.exp2(x.astype(np.float64))
    normal = (want > 1.2e-38) & (want < 3.0e38)
    assert ulp_err(got[normal], want[normal]).max() <= 1.0
    k = np.arange(-149, 128, dtype=np.float32)
    np.testing.assert_array_equal(oracle.math_eval("exp2", k), np.exp2(k.astype(np.float64)).astype(np.float32))
    sp = oracle.math_eval("exp2", np.array([-np.inf, -200.0, 128.0, np.inf, np.nan], dtype=np.float32))
    assert sp[0] == 0 and sp[1] == 0 and np.isinf(sp[2]) and np.isinf(sp[3]) and np.isnan(sp[4])


def test_atan2f_within_two_ulp_and_special_cases(oracle, rng):
    n = 1000000
    for spread in (0.0, 20.0, 80.0):
        y = (rng.normal(0, 1, n) * np.exp(rng.uniform(-spread, spread, n))).astype(np.float32)
        x = (rng.normal(0, 1, n) * np.exp(rng.uniform(-spread, spread, n))).astype(np.float32)
        got, want = oracle.math_eval("atan2", y, x), np.arctan2(y.astype(np.float64), x.astype(np.float64))
        ok = np.abs(want) > 1.2e-38
        assert ulp_err(got[ok], want[ok]).max() <= 2.0
        assert np.array_equal(np.signbit(got), np.signbit(want))
    # IEEE-754 / C99 special cases (the descriptor's angle-index-8 path needs atan2f(+0, x < 0) == (float)pi exactly)
    sp = np.array([0.0, -0.0, np.inf, -np.inf, 1e-45, -1e-45, 1.0, -1.0, 3e38, -3e38], dtype=np.float32)
    Y, X = [a.ravel() for a in np.meshgrid(sp, sp, indexing="ij")]
    got, want = oracle.math_eval("atan2", Y, X), np.arctan2(Y.astype(np.float64), X.astype(np.float64))
    assert np.array_equal(np.signbit(got), np.signbit(want))
    assert np.abs(got.astype(np.float64) - want).max() <= 2.4e-7
    exact = (want == 0) | (np.abs(want) == np.pi)
    np.testing.assert_array_equal(got[exact], want[exact].astype(np.float32))
    nan = oracle.math_eval("atan2", np.array([np.nan, 1.0, np.nan], dtype=np.float32),
                           np.array([1.0, np.nan, np.nan], dtype=np.float32))
    assert np.isnan(nan).all()


def test_sincosf_within_two_ulp(oracle, rng):
    t = np.concatenate([rng.uniform(0, 6.3, 500000), rng.uniform(-1000, 1000, 300000), rng.uniform(-9e8, 9e8, 100000),
                        (np.arange(-64, 65) * (np.pi / 4)).astype(np.float32)]).astype(np.float32)
    s, c = oracle.math_eval("sincos", t)
    assert ulp_err(s, np.sin(t.astype(np.float64))).max() <= 2.0
    assert ulp_err(c, np.cos(t.astype(np.float64))).max() <= 2.0
    big = np.abs(t) >= 128.0  # the double-precision path
    assert ulp_err(s[big], np.sin(t[big].astype(np.float64))).max() <= 1.0
    s, c = oracle.math_eval("sincos", np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 2e9], dtype=np.float32))
    assert s[0] == 0 and c[0] == 1 and s[1] == 0 and c[1] == 1
    assert np.isnan(s[2:]).all() and np.isnan(c[2:]).all()
