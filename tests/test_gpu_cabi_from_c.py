"""The boundary used the way the reference's host would use it: a plain C program (gcc, no Python, no torch in the process) linked
against libgingr_hip.so.  Its output is compared with the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import gingr_oracle as go

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_consumer(tmp_path):
    exe = str(tmp_path / "cabi_driver")
    libdir = os.path.join(ROOT, "gingr_amd")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "cabi_driver.c"),
                           "-o", exe, "-L", libdir, "-lgingr_hip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    rng = np.random.default_rng(3)
    q = rng.normal(0, 5, (300, 3))
    t = q[rng.permutation(300)[:250]] + rng.normal(0, 0.3, (250, 3))
    text = f"{q.shape[0]} {t.shape[0]}\n" + "\n".join(" ".join(repr(float(v)) for v in row) for row in np.concatenate([q, t]))
    out = subprocess.run([exe], input=text, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    idx = np.array([int(v) for v in lines[:300]])
    oidx, od2, omean = go.icp_closest_point(q, t)
    assert np.array_equal(idx, oidx)
    Np, s2n, mean = (float(v) for v in lines[300].split())
    st = go.cpd_stats_dense(q, t, 4.0, 0.1)
    assert abs(Np - st.Np) < 1e-9 * st.Np and abs(s2n - st.sigma2_next) < 1e-9 * abs(st.sigma2_next)
    assert abs(mean - omean) < 1e-12 * max(1.0, omean)


def _grid_mesh(n, size, height, seed):
    rng = np.random.default_rng(seed)
    xs = np.linspace(-size, size, n)
    X, Y = np.meshgrid(xs, xs, indexing="ij")
    Z = height * np.sin(X / size * 2.0) * np.cos(Y / size * 1.5) + rng.normal(0, 0.05, X.shape)
    v = np.stack([X.ravel(), Y.ravel(), Z.ravel()], 1)
    idx = np.arange(n * n).reshape(n, n)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, 1:].ravel()
    return v, np.concatenate([np.stack([a, b, c], 1), np.stack([b, d, c], 1)]).astype(np.int32)


def _parse(path):
    res = {}
    for line in open(path):
        name, *vals = line.split()
        if name != "build":
            res[name] = np.array([float(v) for v in vals])
    return res


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


def test_plain_c_consumer_of_the_fused_path(tmp_path):
    """tests/c/cabi_fitter_driver.c: model upload -> fitter -> fused updates -> state; the 3-phase / 2-segment protocol on two row
    shards with the exchange summed on the host in C; the in-library device group; stateless operators; the ICP flavours;
    probabilistic proposal; classic CPD; the per-coordinate GPMM builder, closest surface points, model transfer, the classic
    rigid ICP and the optimal-step non-rigid ICP -- all from a C program, compared with the oracle."""
    exe = str(tmp_path / "cabi_fitter_driver")
    libdir = os.path.join(ROOT, "gingr_amd")
    subprocess.check_call(["gcc", "-std=gnu99", "-O1", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                           os.path.join(ROOT, "tests", "c", "cabi_fitter_driver.c"), "-o", exe, "-L", libdir, "-lgingr_hip",
                           "-L", "/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    rng = np.random.default_rng(21)
    ref, cells = _grid_mesh(22, 40.0, 6.0, 1)                       # 484 vertices, open surface
    tv, tcells = _grid_mesh(20, 40.0, 6.5, 2)
    mo = go.build_gaussian_gpmm(ref, 60.0, 20.0, rel_tol=1e-9, max_rank=20)
    mo.mean = rng.normal(0, 0.05, ref.shape)
    target = tv @ go.euler_to_rot(0.02, -0.01, 0.015).T + np.array([0.5, -0.3, 0.2])
    M, N, r, n_iter = mo.M, target.shape[0], mo.rank, 3
    sigma2, w = go.cpd_initial_sigma2(mo.ref + mo.mean, target), 0.1
    inp, outp = tmp_path / "in.bin", tmp_path / "out.txt"
    with open(inp, "wb") as f:
        np.array([M, N, r, cells.shape[0], tcells.shape[0], n_iter], dtype=np.int64).tofile(f)
        for a in (mo.ref, mo.mean, np.asfortranarray(mo.U).ravel(order="F"), mo.lam, target, np.array([sigma2, w])):
            np.ascontiguousarray(a, dtype=np.float64).tofile(f)
        np.ascontiguousarray(cells, dtype=np.int32).tofile(f)
        np.ascontiguousarray(tcells, dtype=np.int32).tofile(f)
    run = subprocess.run([exe, str(inp), str(outp)], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-2000:]
    got = _parse(outp)

    # A / B / C: n_iter CPD updates = the oracle's trajectory, whichever way the C program drove them
    st = go.initial_state(mo, sigma2)
    for _ in range(n_iter):
        st = go.cpd_update(mo, target, st, w=w)
    for tag in ("A", "B", "C"):
        sc = got[f"{tag}_scalars"]
        assert int(sc[11]) == n_iter and int(sc[12]) == 0, tag
        assert rel(got[f"{tag}_fit"].reshape(M, 3), st.fit) < 1e-5, tag
        assert rel(got[f"{tag}_alpha"], st.alpha) < 1e-4 and abs(sc[10] - st.sigma2) < 1e-8 * st.sigma2, tag
    assert rel(got["B_fit"], got["A_fit"]) < 1e-9 and rel(got["C_fit"], got["A_fit"]) < 1e-9
    assert int(got["A_retry"][0]) == 10 and int(got["A_timing_update"][1]) == n_iter and got["A_timing_update"][0] > 0
    # the statistics of the last iteration's affinity evaluation
    stats = None
    s_prev = go.initial_state(mo, sigma2)
    for _ in range(n_iter - 1):
        s_prev = go.cpd_update(mo, target, s_prev, w=w)
    stats = go.cpd_stats_dense(s_prev.fit, target, s_prev.sigma2, w)
    assert np.allclose(got["A_P1"], stats.P1, rtol=1e-7, atol=1e-12) and abs(got["A_scalars6"][0] - stats.Np) < 1e-7 * stats.Np
    # group ICP step from the CPD result
    sti, _ = go.icp_update(mo, target, st, st.sigma2, 1.0, 10)
    assert rel(got["C_icp_fit"].reshape(M, 3), sti.fit) < 1e-5 and int(got["C_icp_scalars"][11]) == n_iter + 1

    # D: stateless operators
    alpha = 0.1 * ((np.arange(r) % 5) - 2.0)
    R, c, t = go.euler_to_rot(0.05, -0.02, 0.03), np.array([1.0, 2.0, 3.0]), np.array([0.5, -0.25, 0.75])
    inst = (mo.instance(alpha) - c) @ R.T + c + t
    assert rel(got["D_instance"].reshape(M, 3), inst) < 1e-12
    assert rel(got["D_coefficients"], alpha) < 1e-5
    posed = mo.transform(R, t, c)
    obs = np.flatnonzero(np.arange(M) % 3 != 0)
    _, a_post = posed.posterior_mean(obs, inst[obs], np.tile(np.eye(3) / 2.0, (obs.shape[0], 1, 1)))
    assert rel(got["D_posterior_coeffs"], a_post) < 1e-6
    assert abs(got["D_initial_sigma2"][0] - go.cpd_initial_sigma2(mo.ref, target)) < 1e-10 * got["D_initial_sigma2"][0]
    assert np.allclose(got["D_gauss_block"].reshape(4, 5), go.gauss_block(mo.ref[:4], target[:5], 30.0, 2.0), rtol=1e-12)
    assert np.allclose(got["D_extrema"], go.pointset_distance_extrema(mo.ref), rtol=1e-12)
    built = go.build_gaussian_gpmm(mo.ref, 60.0, 30.0, rel_tol=0.0, max_rank=12)
    assert np.allclose(got["D_built_variance"], built.lam, rtol=1e-8)

    # E: ICP flavours, probabilistic proposal
    s_icp = go.initial_state(mo, 25.0)
    for _ in range(2):
        s_before = s_icp
        s_icp, idx = go.icp_update(mo, target, s_icp, 25.0, 1.0, 10)
    assert rel(got["E_icp_fit"].reshape(M, 3), s_icp.fit) < 1e-5 and got["E_icp_scalars"][10] == s_icp.sigma2
    assert np.array_equal(got["E_icp_idx"].astype(np.int64), go.icp_closest_point(s_before.fit, target)[0])
    s0 = go.initial_state(mo, 25.0)
    ocp, ow, _ = go.surface_correspondence(s0.fit, cells, target, tcells)
    assert np.array_equal(got["E_surface_weights"], ow) and np.abs(got["E_surface_cp"].reshape(M, 3) - ocp).max() < 1e-9
    s_surf, _ = go.icp_surface_update(mo, cells, target, tcells, s0, 25.0, 1.0, 10)
    assert rel(got["E_surface_fit"].reshape(M, 3), s_surf.fit) < 1e-5
    want = go.surface_distance_stats(s_surf.fit, target, tcells, sdev=2.0)
    assert np.allclose(got["E_surface_stats"], want, rtol=1e-6)
    z = 0.3 * ((np.arange(r) * 7) % 5 - 2.0)
    s_samp = go.cpd_update(mo, target, go.initial_state(mo, sigma2), w=w, z=z)
    assert rel(got["E_sample_fit"].reshape(M, 3), s_samp.fit) < 1e-5
    pids, pts, var = go.cpd_observations(mo, target, go.initial_state(mo, sigma2), w=w)
    lp = go.posterior_logpdf_of_mesh(mo, go.initial_state(mo, sigma2), pids, pts, var, s_samp.fit)
    assert abs(got["E_logpdf"][0] - lp) < 1e-5 * abs(lp)

    # F: classic rigid CPD, stand-alone mesh statistics
    ty, s2c, iters, _ = go.classic_cpd_registration(mo.ref, target, "rigid", lam=2.0, beta=2.0, w=0.0, max_iteration=3, tolerance=0.0)
    assert iters == 3 and rel(got["F_classic_ty"].reshape(M, 3), ty) < 1e-8 and abs(got["F_classic_sigma2"][0] - s2c) < 1e-8 * s2c
    assert np.allclose(got["F_mesh_stats"], go.surface_distance_stats(mo.ref, target, tcells, sdev=0.0), rtol=1e-6)

    # G: per-coordinate GPMM kernels, closest surface points, model transfer, classic rigid ICP
    msym = go.build_gpmm_diagonal(mo.ref, go.symmetric_gauss_kernel_fun(mo.ref, 60.0, 30.0), 0.0, 14)
    assert np.allclose(got["G_sym_variance"], msym.lam, rtol=1e-8)
    mdot = go.build_gpmm_diagonal(mo.ref, go.dot_kernel_fun(mo.ref, 0.01), 0.0, 9)
    assert got["G_dot_variance"].shape[0] == mdot.rank and np.allclose(got["G_dot_variance"], mdot.lam, rtol=1e-7)
    ocp2, _ = go.mesh_closest_point(mo.ref, target, tcells)
    assert np.abs(got["G_cp"].reshape(M, 3) - ocp2).max() < 1e-9
    bary = got["G_bary"].reshape(M, 3)
    assert np.all(bary >= 0) and np.allclose(bary.sum(1), 1.0, atol=1e-13)
    Us = got["G_sym_basis"].reshape(14, 3 * M).T.reshape(M, 3, 14)              # column-major download
    ids = np.array([[(7 * i + 3 * k) % M for k in range(3)] for i in range(40)])
    want_nb = 0.5 * Us[ids[:, 0]] + 0.3 * Us[ids[:, 1]] + 0.2 * Us[ids[:, 2]]
    assert np.allclose(got["G_new_basis"].reshape(14, 120).T.reshape(40, 3, 14), want_nb, atol=1e-13)
    fit_i, dists = mo.ref, []
    for _ in range(2):
        fit_i, dd, _ = go.rigid_icp_iteration(fit_i, target)
        dists.append(dd)
    assert np.allclose(got["G_icp_dist"], dists, rtol=1e-11) and np.abs(got["G_icp_points"].reshape(M, 3) - fit_i).max() < 1e-9
    # diagnostics reached from C: the stateless scan executes every pair (queries rounded up to whole 64-lane waves), and the group
    # says what it exchanges through
    tests, md = got["G_nn_tests"]
    assert M * target.shape[0] <= tests <= (M + 63) // 64 * 64 * target.shape[0]
    assert abs(md - go.icp_closest_point(mo.ref, target)[2]) < 1e-12 * max(md, 1.0)
    assert got["C_exchange_info"][0] in (1.0, 2.0)
    # one N-ICP-T and one N-ICP-A iteration of the model reference (explicit points through gingr_fitter_set_fit_points)
    edges = go.nicp_edges(cells)
    lm_ids, ul = np.array([5, 17]), target[[11, 18]]
    _, ow, _ = go.surface_correspondence(mo.ref, cells, target, tcells)
    assert np.array_equal(got["G_nicp_w"], ow)
    want_t, _ = go.nicp_iteration_t(mo.ref, cells, target, tcells, edges, lm_ids, ul, 10.0, 5.0)
    want_a, _, want_lm = go.nicp_iteration_a(mo.ref, cells, target, tcells, edges, lm_ids, ul, 10.0, 5.0, 0.5)
    assert np.abs(got["G_nicp_t"].reshape(M, 3) - want_t).max() < 1e-7
    assert np.abs(got["G_nicp_a"].reshape(M, 3) - want_a).max() < 1e-7 and np.abs(got["G_nicp_lm"].reshape(2, 3) - want_lm).max() < 1e-7
