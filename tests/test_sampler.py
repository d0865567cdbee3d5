"""The sampler's random-number layer on the host (no GPU needed): Philox4x32-10 known answers
(Random123 kat_vectors) and the exact binomial generator against scipy's pmf."""
import ctypes

import numpy as np
import pytest
from scipy import stats

from naqs_amd import _lib


@pytest.fixture(scope="module")
def lib():
    return _lib.load_library()


@pytest.mark.parametrize("ctr,key,want", [
    ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
    ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
    ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
     [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
])
def test_philox_known_answers(lib, ctr, key, want):
    c = np.array(ctr, dtype=np.uint32)
    k = np.array(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    _lib.check(lib.naqs_rng_philox_host(c.ctypes.data, k.ctypes.data, out.ctypes.data), "philox")
    assert out.tolist() == want


def _draw(lib, n, p, reps, seed):
    out = np.zeros(reps, dtype=np.int64)
    _lib.check(lib.naqs_rng_binomial_host(n, p, seed, reps, out.ctypes.data), "binomial")
    return out


# (n, p): inversion branch (n min(p,q) < 10), BTRS branch, the p > 1/2 reflection, and the extremes the sampler sees
CASES = [(5, 0.3), (3, 0.5), (20, 0.4), (100, 0.05), (100, 0.2), (1000, 0.013), (1000, 0.5), (50, 0.9),
         (10 ** 6, 1e-5), (10 ** 6, 1.2e-5), (10 ** 12, 1e-11), (10 ** 9, 2e-8), (200, 0.94)]


@pytest.mark.parametrize("n,p", CASES)
def test_binomial_matches_pmf(lib, n, p):
    reps = 400000
    x = _draw(lib, n, p, reps, seed=1234 + n % 97)
    assert x.min() >= 0 and x.max() <= n
    mu, sd = n * p, np.sqrt(n * p * (1 - p))
    lo, hi = int(max(0, np.floor(mu - 7 * sd))), int(min(n, np.ceil(mu + 7 * sd)))
    ks = np.arange(lo, hi + 1)
    expect = stats.binom.pmf(ks, n, p) * reps
    obs = np.bincount(np.clip(x - lo, 0, hi - lo), minlength=hi - lo + 1)[:hi - lo + 1]
    m = expect >= 5
    chi2 = ((obs[m] - expect[m]) ** 2 / expect[m]).sum()
    assert stats.chi2.sf(chi2, m.sum() - 1) > 1e-4, (chi2, m.sum() - 1)


@pytest.mark.parametrize("n,p", [(10 ** 6, 0.3), (10 ** 12, 0.25), (10 ** 12, 0.999), (2 ** 44, 0.5)])
def test_binomial_moments_at_large_n(lib, n, p):
    reps = 200000
    x = _draw(lib, n, p, reps, seed=99).astype(np.float64)
    mu, var = n * p, n * p * (1 - p)
    assert abs(x.mean() - mu) < 5 * np.sqrt(var / reps)
    assert abs(x.var() / var - 1) < 5 * np.sqrt(2 / reps)
    z = (x - mu) / np.sqrt(var)                                  # normal limit
    assert stats.kstest(z, "norm").pvalue > 1e-4


def test_binomial_edge_cases_and_determinism(lib):
    assert _draw(lib, 0, 0.5, 10, 1).tolist() == [0] * 10
    assert _draw(lib, 17, 0.0, 10, 1).tolist() == [0] * 10
    assert _draw(lib, 17, 1.0, 10, 1).tolist() == [17] * 10
    a, b = _draw(lib, 1000, 0.3, 1000, 5), _draw(lib, 1000, 0.3, 1000, 5)
    assert np.array_equal(a, b) and not np.array_equal(a, _draw(lib, 1000, 0.3, 1000, 6))
    assert lib.naqs_rng_binomial_host(-1, 0.5, 0, 1, np.zeros(1, dtype=np.int64).ctypes.data) == -1     # NAQS_ERR_INVALID


def test_generator_log_has_no_zero_argument(lib):
    """Round-5 advice: `log_fast` has no guard for x <= 0 (it sits on every draw's critical chain).  It needs none: the
    uniforms are on the OPEN interval — u01(0, 0) = 2^-54, never 0 — so the exact acceptance test's smallest argument
    v alpha us^2 / (a + b us^2) stays a normal float64; and over the range the generator uses them the three elementary
    functions are within a few ulp of libm's."""
    def call(fn, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty_like(x)
        _lib.check(lib.naqs_rng_math_host(fn, len(x), x.ctypes.data, y.ctypes.data), "rng_math")
        return y
    words = np.array([0, 0xFFFFFFFFFFFFFFFF, 0x00000020_00000000, 0x00000000_00000040], dtype=np.uint64)
    u = call(3, words.view(np.float64))
    assert u[0] == 2.0 ** -54 and 0.0 < u.min() and u.max() <= 1.0
    assert u[2] == (2 ** 26 + 0.5) * 2.0 ** -53 and u[3] == 1.5 * 2.0 ** -53          # hi >> 5 and lo >> 6 land where they should
    rs = np.random.RandomState(5)
    x = np.concatenate([np.exp(rs.uniform(-440, 60, 200000)), [2.0 ** -190, 2.0 ** -162, 0.5, 1.0, 2.0 ** 44]])
    ulp = lambda got, want: np.max(np.abs(got - want) / np.spacing(np.abs(want)))
    assert ulp(call(0, x), np.log(x)) <= 3
    p = np.concatenate([rs.uniform(0, 0.5, 100000), np.exp(rs.uniform(-40, np.log(0.5), 100000)), [0.5, 2.0 ** -60]])
    assert ulp(call(1, p), np.log1p(-p)) <= 4
    t = -np.concatenate([rs.uniform(0, 40, 100000), np.exp(rs.uniform(-30, np.log(700), 100000))])
    assert ulp(call(2, t), np.exp(t)) <= 2
