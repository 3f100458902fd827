"""The ICP update's closest-point search (grid search + masked full scan, gingr_amd/csrc/nn_grid.hip) against the oracle's linear
argmin on geometries chosen to break a grid: exact ties (duplicates, lattices), queries on cell boundaries, flat and clustered target
clouds, queries far from every target (the masked full scan answers those), single points.

Bar: indices bit-exact, lowest original index on equal distances (G/api/registration/utils/ClosestPointRegistrator.scala:33-44).
"""
import numpy as np
import pytest

from oracle import gingr_oracle as go

pytestmark = pytest.mark.gpu


def tiny_model(ref, seed):
    rng = np.random.default_rng(seed)
    M = ref.shape[0]
    r = min(3, 3 * M)
    U, _ = np.linalg.qr(rng.normal(0, 1, (3 * M, r)))
    return go.PDM(ref=ref, mean=np.zeros_like(ref), U=U, lam=np.array([9.0, 4.0, 1.0])[:r])


def shell(n, seed, radius=40.0):
    rng = np.random.default_rng(seed)
    v = rng.normal(0, 1, (n, 3))
    return radius * v / np.linalg.norm(v, axis=1, keepdims=True)


def cases():
    rng = np.random.default_rng(5)
    out = {}
    t = shell(3000, 1)
    out["surface"] = (t[rng.permutation(3000)[:2500]] + rng.normal(0, 0.4, (2500, 3)), t)
    # every target twice (and a few three times), in shuffled order: the lower original index must win every tie
    t2 = np.concatenate([t[:800], t[:800], t[:50]])[rng.permutation(1650)]
    out["duplicates"] = (t[:800] + rng.normal(0, 0.2, (800, 3)), t2)
    # integer lattice, queries on cell centres / edges / vertices: 2, 4 or 8 targets at exactly the same distance
    g = np.stack(np.meshgrid(np.arange(8.0), np.arange(8.0), np.arange(8.0), indexing="ij"), -1).reshape(-1, 3)
    lat = g[rng.permutation(g.shape[0])]
    off = np.array([[0.5, 0, 0], [0.5, 0.5, 0], [0.5, 0.5, 0.5], [0, 0, 0], [0.25, 0.5, 0.0]])
    out["lattice_ties"] = (np.concatenate([g[:300] + o for o in off]), lat)
    flat = np.column_stack([rng.uniform(-50, 50, 2000), rng.uniform(-50, 50, 2000), np.zeros(2000)])
    out["flat_target"] = (flat[:1500] + np.array([0.0, 0.0, 1.5]) + rng.normal(0, 0.3, (1500, 3)), flat)
    line = np.column_stack([np.linspace(-100, 100, 900), np.zeros(900), np.zeros(900)])
    out["line_target"] = (line[::2] + rng.normal(0, 0.5, (450, 3)), line)
    out["far_queries"] = (t[:700] * 3.0 + 500.0, t)                       # nothing within reach of the grid search
    mixed = t[:1200] + rng.normal(0, 0.3, (1200, 3))
    mixed[::3] += 400.0
    out["near_and_far"] = (mixed, t)
    two = np.concatenate([rng.normal(0, 1.0, (600, 3)), rng.normal(0, 1.0, (600, 3)) + 5000.0])
    out["two_clusters"] = (two[rng.permutation(1200)[:900]] + rng.normal(0, 0.05, (900, 3)), two)
    out["identical_targets"] = (rng.normal(0, 1, (70, 3)), np.tile(np.array([[1.0, 2.0, 3.0]]), (33, 1)))
    out["on_the_targets"] = (t2[:500].copy(), t2)                          # distance 0, duplicates
    out["one_target"] = (rng.normal(0, 5, (40, 3)), np.array([[0.5, -1.0, 2.0]]))
    out["one_query"] = (np.array([[3.0, 1.0, -2.0]]), t[:300])
    out["two_points"] = (np.array([[0.0, 0.0, 0.0], [1.0, 1.0, 1.0]]), np.array([[1.0, 1.0, 1.0], [0.0, 0.0, 0.0]]))
    return out


CASES = cases()


@pytest.mark.parametrize("name", sorted(CASES))
def test_closest_point_indices_are_the_linear_argmin(ctx, name):
    import gingr_amd as ga
    ref, target = CASES[name]
    mo = tiny_model(ref, seed=len(name))
    algo = ga.IcpRegistration(ctx)
    cfg = ga.IcpConfiguration(maxIterations=10, initialSigma=4.0, endSigma=1.0, correspondenceMethod="PointcloudClosestPoint")
    state = algo.createInitialState(ga.PointDistributionModel(mo.ref, mo.mean, mo.U, mo.lam), target, cfg)
    for it in range(4):  # the first search starts cold, the later ones from the previous matches
        fit_before = np.array(state.general.fit)
        state = algo.update(state)
        want, _, _ = go.icp_closest_point(fit_before, target)
        got = algo.last_correspondence_indices()
        assert np.array_equal(got, want), (name, it, int(np.sum(got != want)))
        if state.general.status != 0:  # (a degenerate pair may stop the registration; the search above was still checked)
            break
    algo.close()


def test_the_grid_search_is_what_runs(ctx):
    """On a surface-like pair the search executes a small fraction of the all-pairs distance tests (the counter of
    gingr_ctx_nn_counting sees the grid kernel and the masked scan)."""
    import gingr_amd as ga
    ref, target = CASES["surface"]
    mo = tiny_model(ref, seed=3)
    algo = ga.IcpRegistration(ctx)
    cfg = ga.IcpConfiguration(maxIterations=10, initialSigma=4.0, endSigma=1.0, correspondenceMethod="PointcloudClosestPoint")
    state = algo.createInitialState(ga.PointDistributionModel(mo.ref, mo.mean, mo.U, mo.lam), target, cfg)
    state = algo.update(state)
    ctx.nn_counting(True)
    state = algo.update(state)
    ctx.synchronize()
    tests = ctx.nn_tests()
    ctx.nn_counting(False)
    algo.close()
    assert 0 < tests < 0.05 * ref.shape[0] * target.shape[0], tests


@pytest.mark.parametrize("seed", range(24))
def test_random_geometries(ctx, seed):
    """Random sizes, anisotropic extents, large coordinate offsets (rounding of the cell index), float32-rounded coordinates (many
    exact ties), clusters and a sprinkling of far outliers among the queries."""
    import gingr_amd as ga
    rng = np.random.default_rng(1000 + seed)
    N = int(rng.integers(1, 3000))
    M = int(rng.integers(1, 2500))
    scale = 10.0 ** rng.uniform(-3, 3, 3) if seed % 3 == 0 else np.full(3, 10.0 ** rng.uniform(-2, 3))
    offset = rng.normal(0, 1, 3) * (1e6 if seed % 4 == 1 else 1.0)
    kind = seed % 5
    if kind == 0:
        t = rng.normal(0, 1, (N, 3))
    elif kind == 1:
        t = shell(N, seed, radius=1.0)
    elif kind == 2:
        centres = rng.normal(0, 5, (4, 3))
        t = centres[rng.integers(0, 4, N)] + rng.normal(0, 0.05, (N, 3))
    elif kind == 3:
        t = np.round(rng.normal(0, 2, (N, 3)))            # a coarse lattice: many coincident targets and equidistant pairs
    else:
        t = rng.uniform(-1, 1, (N, 3)) * np.array([1.0, 1.0, 1e-7])   # nearly flat
    target = t * scale + offset
    if seed % 2:
        target = target.astype(np.float32).astype(np.float64)
    q = target[rng.integers(0, N, M)] + rng.normal(0, 1, (M, 3)) * scale * 10.0 ** rng.uniform(-3, 0)
    far = rng.random(M) < 0.02
    q[far] += rng.normal(0, 1, (int(far.sum()), 3)) * scale * 300.0
    if seed % 2:
        q = q.astype(np.float32).astype(np.float64)
    mo = tiny_model(q, seed)
    algo = ga.IcpRegistration(ctx)
    cfg = ga.IcpConfiguration(maxIterations=10, initialSigma=4.0, endSigma=1.0, correspondenceMethod="PointcloudClosestPoint")
    state = algo.createInitialState(ga.PointDistributionModel(mo.ref, mo.mean, mo.U, mo.lam), target, cfg)
    for it in range(3):
        fit_before = np.array(state.general.fit)
        state = algo.update(state)
        want, _, _ = go.icp_closest_point(fit_before, target)
        got = algo.last_correspondence_indices()
        assert np.array_equal(got, want), (seed, it, int(np.sum(got != want)), N, M)
        if state.general.status != 0:
            break
    algo.close()


def test_benchmark_size_against_the_stateless_tile_scan(ctx):
    """50k <-> 50k (bench.py's clouds): the fitter's search (grid + masked tile scan, warm-started) must give the indices of the stateless
    gingr_nn (tile scan from scratch, no grid) -- two exact searches of different structure, so equal indices including every tie."""
    import gingr_amd as ga
    from gingr_amd import _native as nat
    from bench import synth_clouds
    y, x = synth_clouds(50000)
    scan = ga.Context(0)                    # the stateless call: box-pruned tile scan over spatially ordered clouds, no grid
    grid = ga.Context(0)
    grid.set_option(nat.OPT_NN_GRID, 2)     # ... and with a grid of its own, built per call, cold start
    rng = np.random.default_rng(2)
    r = 8
    U, _ = np.linalg.qr(rng.normal(0, 1, (3 * y.shape[0], r)))
    model = ga.PointDistributionModel(y, np.zeros_like(y), U, np.linspace(400.0, 50.0, r))
    algo = ga.IcpRegistration(ctx)
    cfg = ga.IcpConfiguration(maxIterations=10, initialSigma=20.0, endSigma=1.0, correspondenceMethod="PointcloudClosestPoint")
    state = algo.createInitialState(model, x, cfg)
    for it in range(3):
        fit_before = np.array(state.general.fit)
        state = algo.update(state)
        want, wd2, _ = scan.nn(fit_before, x)
        got = algo.last_correspondence_indices()
        assert np.array_equal(got, want), (it, int(np.sum(got != want)))
        # ... and the stateless call with its own grid (built per call from the caller's arrays, no warm start): the same again
        again, ad2, _ = grid.nn(fit_before, x)
        assert np.array_equal(again, want) and np.array_equal(ad2, wd2), it
    algo.close()
    scan.close()
    grid.close()


@pytest.mark.parametrize("seed", range(6))
def test_stateless_grid_search_equals_the_tile_scan(ctx, seed):
    """gingr_nn as the plain scan of all pairs (GINGR_OPT_CULL = 0), as the box-pruned scan over ordered clouds (default) and with a
    per-call grid (GINGR_OPT_NN_GRID = 2) on awkward inputs (ties on a lattice, a flat cloud, far queries): identical indices, distances
    and mean distance."""
    import gingr_amd as ga
    from gingr_amd import _native as nat
    rng = np.random.default_rng(100 + seed)
    n = 4096
    if seed % 3 == 0:
        target = np.stack(np.meshgrid(np.arange(16.0), np.arange(16.0), np.arange(16.0), indexing="ij"), -1).reshape(-1, 3)   # lattice: ties
    elif seed % 3 == 1:
        target = np.concatenate([rng.normal(0, 30, (n, 2)), np.zeros((n, 1))], 1)                                             # flat
    else:
        target = rng.normal(0, 40, (n, 3)).astype(np.float32).astype(np.float64)
    query = np.concatenate([target[rng.integers(0, n, 1500)] + rng.normal(0, 0.7, (1500, 3)),      # near the cloud
                            np.round(rng.uniform(0, 15, (600, 3)) * 2) / 2,                        # half-integer points: exact ties on the lattice
                            rng.normal(0, 1, (50, 3)) * 1e4])                                      # far away
    brute = ga.Context(0)
    brute.set_option(nat.OPT_CULL, 0)
    grid = ga.Context(0)
    grid.set_option(nat.OPT_NN_GRID, 2)
    want, wd2, wmd = brute.nn(query, target)
    for c in (ctx, grid):
        got, gd2, gmd = c.nn(query, target)
        assert np.array_equal(got, want) and np.array_equal(gd2, wd2) and gmd == wmd
    brute.close()
    grid.close()


def test_heaps_of_coincident_targets(ctx):
    """Thousands of targets in one grid cell (a blob of coincident points plus a far outlier that stretches the grid): the row is too
    crowded for one lane, the query is handed to the tile scan -- still the linear argmin, lowest index among the coincident points."""
    import gingr_amd as ga
    rng = np.random.default_rng(11)
    blob = np.tile(np.array([[1.0, 2.0, 3.0]]), (3000, 1))
    blob[::7] += 1e-9                                   # nearly coincident too
    target = np.concatenate([blob, rng.normal(0, 1, (200, 3)) * 50.0 + 500.0, np.array([[1e4, 0.0, 0.0]])])
    ref = np.concatenate([rng.normal(0, 0.5, (300, 3)) + np.array([1.0, 2.0, 3.0]), rng.normal(0, 1, (100, 3)) * 50.0 + 500.0])
    mo = tiny_model(ref, seed=4)
    algo = ga.IcpRegistration(ctx)
    cfg = ga.IcpConfiguration(maxIterations=10, initialSigma=4.0, endSigma=1.0, correspondenceMethod="PointcloudClosestPoint")
    state = algo.createInitialState(ga.PointDistributionModel(mo.ref, mo.mean, mo.U, mo.lam), target, cfg)
    for it in range(3):
        fit_before = np.array(state.general.fit)
        state = algo.update(state)
        want, _, _ = go.icp_closest_point(fit_before, target)
        assert np.array_equal(algo.last_correspondence_indices(), want), it
        if state.general.status != 0:
            break
    algo.close()


def _argmin_lowest_index(q, t):
    """the reference's linear scan: d2 = dx*dx + dy*dy + dz*dz (separately rounded), strict <, targets in index order"""
    idx, d2 = np.empty(q.shape[0], dtype=np.int64), np.empty(q.shape[0])
    for i, p in enumerate(q):
        d = t - p
        v = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        ok = ~np.isnan(v)
        if not ok.any() or not (v[ok].min() < np.inf):
            idx[i], d2[i] = -1, np.inf
        else:
            j = int(np.flatnonzero(v == v[ok].min())[0])
            idx[i], d2[i] = j, v[j]
    return idx, d2


@pytest.mark.parametrize("name", sorted(CASES))
def test_stateless_all_pairs_form_on_the_tie_cases(ctx, name):
    """gingr_nn up to 2^26 pairs is the all-pairs form of round 4 (nn_small_kernel: the index is not tracked in the pair loop, the
    combination pass finds the first slice with the minimum and searches it again): same indices and the same distance BITS as the
    linear scan on every tie / degenerate case of this file."""
    q, t = CASES[name]
    idx, d2, md = ctx.nn(q, t)
    want, wd2 = _argmin_lowest_index(q, t)
    assert np.array_equal(idx, want), (name, int((idx != want).sum()))
    assert np.array_equal(d2, wd2)
    assert abs(md - np.sqrt(wd2).mean()) <= 1e-12 * max(1.0, md)


def test_stateless_all_pairs_form_shapes_and_non_finite_input(ctx):
    rng = np.random.default_rng(21)
    # one query against many targets (slices of at most 2 048 targets), many queries against one slice, sizes that are no multiples of anything
    for M, N in ((1, 100_003), (3, 40_000), (5_000, 33), (513, 2_049), (257, 6_151)):
        q, t = rng.normal(0, 10, (M, 3)), rng.normal(0, 10, (N, 3))
        t[N // 2] = t[0]                                  # an exact duplicate: the lower index wins
        q[0] = t[0]
        idx, d2, _ = ctx.nn(q, t)
        want, wd2 = _argmin_lowest_index(q, t)
        assert np.array_equal(idx, want) and np.array_equal(d2, wd2), (M, N)
        assert idx[0] == 0 and d2[0] == 0.0
    # NaN / infinite coordinates: a NaN distance never wins; a query with no finite distance gets -1 / +inf
    q, t = rng.normal(0, 1, (70, 3)), rng.normal(0, 1, (90, 3))
    t[5, 1] = np.nan
    t[17] = np.inf
    q[3, 0] = np.nan
    q[9] = np.inf
    idx, d2, _ = ctx.nn(q, t)
    want, wd2 = _argmin_lowest_index(q, t)
    assert np.array_equal(idx, want) and np.array_equal(d2, wd2)
    assert idx[3] == -1 and idx[9] == -1 and np.isinf(d2[3]) and not np.any(idx == 5)
