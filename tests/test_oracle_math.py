"""The oracle's deterministic math layer against float64 numpy: it must be a conformant
implementation of the OpenCL builtins the reference kernels call (OpenCL 1.1 §7.4: pow <= 16 ulp,
acos <= 4 ulp, atan <= 5 ulp; native_* are implementation-defined, we hold them to ~1e-7 abs)."""
import numpy as np
import pytest


def ulp_error(got, ref64):
    ref32 = ref64.astype(np.float32)
    ulp = np.spacing(np.abs(ref32)).astype(np.float64)
    return np.abs(got.astype(np.float64) - ref64) / ulp


@pytest.fixture(scope="module")
def rng():
    return np.random.default_rng(1234)


@pytest.mark.parametrize("span", [10.0, 100.0, 2.0e4])
def test_sin_cos_absolute_error(oracle, rng, span):
    x = rng.uniform(-span, span, 400000).astype(np.float32)
    x64 = x.astype(np.float64)
    assert np.abs(oracle.math("sin", x) - np.sin(x64)).max() < 1.5e-7
    assert np.abs(oracle.math("cos", x) - np.cos(x64)).max() < 1.5e-7


def test_sin_cos_exact_points_and_guards(oracle):
    x = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 3.0e8, -1.0e30], np.float32)
    s, c = oracle.math("sin", x), oracle.math("cos", x)
    assert s[0] == 0.0 and c[0] == 1.0 and s[1] == 0.0
    assert np.isnan(s[2:5]).all() and np.isnan(c[2:5]).all()
    # beyond 1e8 the argument is replaced by +-0: defined, not garbage
    assert s[5] == 0.0 and c[5] == 1.0 and s[6] == 0.0 and c[6] == 1.0


def test_tan_ulp(oracle, rng):
    x = rng.uniform(0.0, 1.5707, 400000).astype(np.float32)
    assert ulp_error(oracle.math("tan", x), np.tan(x.astype(np.float64))).max() < 4.0


def test_acos_ulp(oracle, rng):
    x = np.concatenate([rng.uniform(-1, 1, 800000), [-1.0, 1.0, 0.0, 0.5, -0.5]]).astype(np.float32)
    assert ulp_error(oracle.math("acos", x), np.arccos(x.astype(np.float64))).max() <= 4.0
    assert np.isnan(oracle.math("acos", np.array([1.5, -1.5, np.nan], np.float32))).all()


def test_atan_ulp(oracle, rng):
    x = np.concatenate([rng.uniform(-10, 10, 400000), rng.uniform(-1e4, 1e4, 50000), rng.uniform(-1, 1, 400000)]).astype(np.float32)
    assert ulp_error(oracle.math("atan", x), np.arctan(x.astype(np.float64))).max() <= 5.0
    big = oracle.math("atan", np.array([np.inf, -np.inf, 1e30], np.float32))
    assert np.allclose(big, [np.pi / 2, -np.pi / 2, np.pi / 2], rtol=2e-7)


def test_pow_ulp_on_the_brdf_domain(oracle, rng):
    # pow( dotHN, ps_e ): base in [0,1], exponents from 1e-3 up to the 1e5 lobes of suzanne.mtl
    x = rng.uniform(0, 1, 800000).astype(np.float32)
    y = (10 ** rng.uniform(-3, 5.5, 800000)).astype(np.float32)
    got = oracle.math("pow", x, y)
    ref = np.power(x.astype(np.float64), y.astype(np.float64))
    normal = ref > 1.2e-38
    assert ulp_error(got[normal], ref[normal]).max() <= 1.0
    assert np.abs(got[~normal].astype(np.float64) - ref[~normal]).max() < 1e-38


def test_pow_general_and_special_cases(oracle, rng):
    x = rng.uniform(0, 30, 400000).astype(np.float32)
    y = rng.uniform(-20, 20, 400000).astype(np.float32)
    got = oracle.math("pow", x, y)
    ref = np.power(x.astype(np.float64), y.astype(np.float64))
    ok = (ref > 1.2e-38) & (ref < 3e38)
    assert ulp_error(got[ok], ref[ok]).max() <= 1.0

    sp = np.array([0, -0.0, 1, -1, np.inf, -np.inf, np.nan, 2, -2, 0.5, -0.5, 3, -3, 1e-40], np.float32)
    X, Y = [a.ravel() for a in np.meshgrid(sp, sp)]
    got = oracle.math("pow", X, Y)
    with np.errstate(all="ignore"):
        ref = np.power(X.astype(np.float64), Y.astype(np.float64)).astype(np.float32)
    same = (got == ref) | (np.isnan(got) & np.isnan(ref))
    assert same.all(), [(X[i], Y[i], got[i], ref[i]) for i in np.nonzero(~same)[0]]


def test_rand_hash_is_in_unit_interval_and_roughly_uniform(oracle, rng):
    # rand() = fract( native_sin( seed ) * 43758.5453123 ), pt_utils.cl:39-44
    seeds = rng.uniform(0, 3000, 500000).astype(np.float32)
    r = oracle.math("randhash", seeds)
    assert r.min() >= 0.0 and r.max() < 1.0
    hist, _ = np.histogram(r, bins=16, range=(0, 1))
    assert np.abs(hist / len(r) - 1 / 16).max() < 0.01
