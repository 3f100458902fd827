"""Size-independent properties of the hot-path operators at the full BASELINE sizes (50k <-> 50k, the metric; 100k <-> 100k,
config 4), where the oracle can only check samples: conservation of assignment mass, invariance under reordering and rigid motion
of the inputs, linear response of P X to the targets, exactness properties of the nearest neighbour."""
import numpy as np
import pytest

from oracle import gingr_oracle as go

pytestmark = pytest.mark.gpu


def clouds(n, seed):
    rng = np.random.default_rng(seed)
    x = rng.normal(0, 50, (n, 3)).astype(np.float32).astype(np.float64)
    y = x[rng.permutation(n)] + rng.normal(0, 2, (n, 3))
    return y, x


@pytest.mark.parametrize("n,sigma2,w", [(50000, 5032.781, 0.1), (100000, 40.0, 0.3)])
def test_assignment_mass_is_conserved(ctx, n, sigma2, w):
    """sum_m P1 = sum_n Pt1 = Np (every entry of P is counted once by either marginal); Pt1_n = 1 - c / den_n; the sigma2 update
    is the P-weighted mean square distance: sigma2' = (xPx - 2 trPXY + yPy) / (3 Np)   (CPD.scala:133-147)."""
    y, x = clouds(n, 5)
    g = ctx.cpd_stats(y, x, sigma2, w)
    assert abs(g["P1"].sum() - g["Np"]) < 1e-9 * g["Np"] and abs(g["Pt1"].sum() - g["Np"]) < 1e-9 * g["Np"]
    assert np.allclose(g["Pt1"], 1.0 - g["c"] / g["den"], rtol=0, atol=1e-13)
    c = w / (1 - w) * (2 * np.pi * sigma2) ** 1.5 * 1.0
    assert abs(g["c"] - c) < 1e-12 * c
    s2 = (g["xPx"] - 2 * g["trPXY"] + g["yPy"]) / (3 * g["Np"])
    assert abs(g["sigma2_next"] - s2) < 1e-12 * s2
    assert abs(g["yPy"] - float(g["P1"] @ (y * y).sum(1))) < 1e-9 * g["yPy"]
    assert abs(g["trPXY"] - float((y * g["PX"]).sum())) < 1e-9 * abs(g["trPXY"])


def test_statistics_do_not_depend_on_the_order_of_the_inputs(ctx):
    """Shuffling either cloud permutes the outputs and leaves the sums alone (up to the order of the additions): the library's
    internal reordering (k-d leaves / Morton) must be invisible."""
    n = 50000
    y, x = clouds(n, 6)
    g = ctx.cpd_stats(y, x, 300.0, 0.1)
    rng = np.random.default_rng(60)
    py, px = rng.permutation(n), rng.permutation(n)
    h = ctx.cpd_stats(y[py], x[px], 300.0, 0.1)
    assert np.allclose(h["P1"], g["P1"][py], rtol=1e-11) and np.allclose(h["PX"], g["PX"][py], rtol=1e-10, atol=1e-9)
    assert np.allclose(h["den"], g["den"][px], rtol=1e-11) and abs(h["Np"] - g["Np"]) < 1e-10 * g["Np"]
    assert abs(h["sigma2_next"] - g["sigma2_next"]) < 1e-10 * g["sigma2_next"]


def test_rigid_motion_of_both_clouds(ctx):
    """P depends on distances only: moving both clouds by the same rigid motion leaves P1, den, Np, sigma2' alone and moves P X
    like the targets, P X -> (P X) R^T + P1 t (linear in the targets)."""
    n = 50000
    y, x = clouds(n, 7)
    R, t = go.euler_to_rot(0.3, -0.2, 0.5), np.array([40.0, -25.0, 10.0])
    g = ctx.cpd_stats(y, x, 200.0, 0.05)
    h = ctx.cpd_stats(y @ R.T + t, x @ R.T + t, 200.0, 0.05)
    assert np.allclose(h["P1"], g["P1"], rtol=1e-9) and np.allclose(h["den"], g["den"], rtol=1e-9)
    assert abs(h["sigma2_next"] - g["sigma2_next"]) < 1e-9 * g["sigma2_next"]
    assert np.allclose(h["PX"], g["PX"] @ R.T + g["P1"][:, None] * t, rtol=1e-9, atol=1e-7)
    # linearity in the target coordinates at fixed P: scaling everything by s and sigma2 by s^2 scales P X by s
    # (w = 0: the outlier constant grows with sigma^3 and would break the scale invariance)
    g0 = ctx.cpd_stats(y, x, 200.0, 0.0)
    k = ctx.cpd_stats(2.0 * y, 2.0 * x, 4.0 * 200.0, 0.0)
    assert np.allclose(k["P1"], g0["P1"], rtol=1e-11) and np.allclose(k["PX"], 2.0 * g0["PX"], rtol=1e-11, atol=1e-9)


def test_nearest_neighbour_properties_at_full_size(ctx):
    """idx is a true argmin (no other sampled target is closer), d2 is the squared distance to the reported target bit for bit,
    a rigid motion of both clouds keeps the assignment, duplicated targets resolve to the lowest index."""
    n = 100000
    y, x = clouds(n, 8)
    idx, d2, mean = ctx.nn(y, x)
    dd = x[idx] - y
    assert np.array_equal(d2, (dd[:, 0] * dd[:, 0] + dd[:, 1] * dd[:, 1]) + dd[:, 2] * dd[:, 2])
    assert abs(mean - np.sqrt(d2).mean()) < 1e-12 * mean
    rng = np.random.default_rng(80)
    for j in rng.choice(n, 5, replace=False):                       # against 5 random other targets for every query
        other = x[(idx + 1 + j) % n] - y
        assert np.all((other * other).sum(1) >= d2)
    # appending a copy of the targets changes nothing: ties go to the lowest index
    idx2, d22, _ = ctx.nn(y[:20000], np.concatenate([x, x]))
    assert np.array_equal(idx2, idx[:20000]) and np.array_equal(d22, d2[:20000])
    # mutual nearest neighbours are symmetric: if x_j is nearest to y_i and y_i is nearest to x_j, both distances agree exactly
    ridx, rd2, _ = ctx.nn(x, y)
    mutual = ridx[idx] == np.arange(n)
    assert mutual.sum() > n // 2 and np.array_equal(rd2[idx[mutual]], d2[mutual])
