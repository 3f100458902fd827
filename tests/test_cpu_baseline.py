"""The optimised CPU baseline bench.py reports (oracle/cpd_baseline.c: -O3 -march=native -fopenmp, SIMD exponential) against
the strict checker (oracle/cpd_oracle.c) -- so that the number printed next to the GPU's is the number of a CORRECT program."""
import numpy as np

from oracle import c_baseline as cb
from oracle import c_oracle as co


def test_simd_exponential_against_libm():
    assert cb.exp_max_rel_error(2000001, -700.0) < 1e-12
    assert cb.exp_max_rel_error(200001, -1e-3) < 1e-12


def test_baseline_passes_match_the_checker():
    rng = np.random.default_rng(5)
    x = rng.normal(0, 50, (700, 3))
    y = x[rng.permutation(700)[:650]] + rng.normal(0, 2, (650, 3))
    for s2 in (2500.0, 9.0):
        den_c = co.cpd_colsum_partial(y, x, s2, 0, y.shape[0])
        den_b = cb.colsum(cb.soa(y), cb.soa(x), s2)
        assert np.allclose(den_b, den_c, rtol=1e-11, atol=1e-300)
        den = den_c + co.outlier_constant(y.shape[0], x.shape[0], s2, 0.1)
        P1_c, PX_c = co.cpd_rowstats_partial(y, x, s2, den, 0, y.shape[0])
        P1_b, PX_b = cb.rowstats(cb.soa(y), cb.soa(x), s2, 1.0 / den)
        assert np.allclose(P1_b, P1_c, rtol=1e-11) and np.allclose(PX_b.T, PX_c, rtol=1e-10, atol=1e-12)
    assert cb.num_threads() >= 1
