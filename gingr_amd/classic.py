"""Host mirror of the reference's classic Coherent Point Drift family (G/other/algorithms/: cpd/CPDFactory.scala, cpd/RigidCPD.scala,
cpd/AffineCPD.scala, cpd/NonRigidCPD.scala, CPDRegistration.scala) over the device-resident engine of csrc/classic_cpd.hip:

    cpd = CPDFactory(ctx, templatePoints, lambda_=2, beta=2, w=0)
    fit = cpd.registerNonRigidly(targetPoints).Registration(max_iteration=100, tolerance=0.001)

Expectation / Maximization run on the GPU (streaming statistics instead of the M x N matrix P; blocked Cholesky for the non-rigid
M x M system); this module keeps the Registration loop with its convergence test (RigidCPD.scala:59-83)."""
from __future__ import annotations

import ctypes
from ctypes import c_void_p
from typing import Optional, Tuple

import numpy as np

from .api import Context, _check
from ._native import f64, dptr

RIGID, AFFINE, NONRIGID = 0, 1, 2


class RigidCPD:
    """RigidCPD / AffineCPD / NonRigidCPD (one class, `kind` selects the Maximization)."""

    def __init__(self, factory: "CPDFactory", targetPoints, kind: int):
        self.cpd, self.kind = factory, kind
        self.target = f64(targetPoints)
        self.N = self.target.shape[0]
        self._lib = factory.ctx._lib
        h = c_void_p()
        _check(factory.ctx.handle, self._lib.gingr_classic_cpd_create(factory.ctx.handle, kind, factory.M, dptr(factory.template), self.N,
                                                                      dptr(self.target), float(factory.lambda_), float(factory.beta),
                                                                      float(factory.w), ctypes.byref(h)), "gingr_classic_cpd_create")
        self._h = h
        self._sigma2_init = self.sigma2()      # initializeGaussianKernel of the factory's template against this target

    def close(self):
        if self._h:
            self._lib.gingr_classic_cpd_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- device state ---------------------------------------------------------------------------------------------
    def state(self) -> Tuple[np.ndarray, float]:
        ty = np.empty((self.cpd.M, 3))
        s2 = ctypes.c_double()
        _check(self.cpd.ctx.handle, self._lib.gingr_classic_cpd_get(self._h, dptr(ty), ctypes.byref(s2), None, None), "gingr_classic_cpd_get")
        return ty, s2.value

    def sigma2(self) -> float:
        s2 = ctypes.c_double()
        _check(self.cpd.ctx.handle, self._lib.gingr_classic_cpd_get(self._h, None, ctypes.byref(s2), None, None), "gingr_classic_cpd_get")
        return s2.value

    def set_state(self, ty, sigma2: float):
        ty = f64(ty)
        _check(self.cpd.ctx.handle, self._lib.gingr_classic_cpd_set(self._h, dptr(ty), float(sigma2)), "gingr_classic_cpd_set")

    def transform(self) -> Tuple[float, np.ndarray, np.ndarray]:
        """(s, R or B, t) of the last rigid / affine Maximization: TY = s Y R^T + 1 t^T."""
        p = np.zeros(13)
        _check(self.cpd.ctx.handle, self._lib.gingr_classic_cpd_get(self._h, None, None, dptr(p), None), "gingr_classic_cpd_get")
        return float(p[0]), p[1:10].reshape(3, 3).copy(), p[10:13].copy()

    def W(self) -> np.ndarray:
        w = np.empty((self.cpd.M, 3))
        _check(self.cpd.ctx.handle, self._lib.gingr_classic_cpd_get(self._h, None, None, None, dptr(w)), "gingr_classic_cpd_get")
        return w

    # -- the reference surface ------------------------------------------------------------------------------------
    def Iteration(self, Y=None, sigma2: Optional[float] = None) -> Tuple[np.ndarray, float]:
        """Iteration(X, Y, sigma2) (RigidCPD.scala:85-88) = Maximization(Expectation(...)); without arguments it advances the
        device state."""
        if Y is not None:
            self.set_state(Y, sigma2)
        _check(self.cpd.ctx.handle, self._lib.gingr_classic_cpd_iterate(self._h, 1), "gingr_classic_cpd_iterate")
        return self.state()

    def Registration(self, max_iteration: int, tolerance: float = 0.001, verbose: bool = False) -> np.ndarray:
        """RigidCPD.Registration (:59-83): iterate until |sigma2' - sigma2| < tolerance or max_iteration non-converged iterations."""
        i, converged = 0, False
        # the reference always starts from cpd.template and the initial variance (RigidCPD.scala:59-62), whatever Iteration or an
        # earlier Registration left on the device
        self.set_state(self.cpd.template, self._sigma2_init)
        current = self._sigma2_init
        while i < max_iteration and not converged:
            if verbose:
                print(f"CPD, iteration: {i}, variance: {current}")
            _check(self.cpd.ctx.handle, self._lib.gingr_classic_cpd_iterate(self._h, 1), "gingr_classic_cpd_iterate")
            new = self.sigma2()
            if abs(new - current) < tolerance:
                if verbose:
                    print("Converged")
                converged = True
            else:
                i += 1
            current = new
        self.iterations, self.converged = i, converged
        return self.state()[0]


class CPDFactory:
    def __init__(self, ctx: Context, templatePoints, lambda_: float = 2.0, beta: float = 2.0, w: float = 0.0):
        if not (0.0 <= w <= 1.0) or not beta > 0 or not lambda_ > 0:
            raise ValueError("requirement failed")          # CPDFactory.scala:43-45
        self.ctx, self.lambda_, self.beta, self.w = ctx, lambda_, beta, w
        self.template = f64(templatePoints)
        self.M, self.dim = self.template.shape[0], 3

    def registerRigidly(self, targetPoints) -> RigidCPD:
        return RigidCPD(self, targetPoints, RIGID)

    def registerNonRigidly(self, targetPoints) -> RigidCPD:
        return RigidCPD(self, targetPoints, NONRIGID)

    def registerAffine(self, targetPoints) -> RigidCPD:
        return RigidCPD(self, targetPoints, AFFINE)


class RigidCPDRegistration:
    """RigidCPDRegistration / NonRigidCPDRegistration / AffineCPDRegistration (CPDRegistration.scala:23-73): register(target)
    returns the warped template points."""

    _kind = RIGID

    def __init__(self, ctx: Context, template, lambda_: float = 2.0, beta: float = 2.0, w: float = 0.0, max_iterations: int = 100):
        self.cpd, self.max_iterations = CPDFactory(ctx, template, lambda_, beta, w), max_iterations

    def register(self, target) -> np.ndarray:
        task = RigidCPD(self.cpd, target, self._kind)
        try:
            return task.Registration(self.max_iterations)
        finally:
            task.close()


class NonRigidCPDRegistration(RigidCPDRegistration):
    _kind = NONRIGID


class AffineCPDRegistration(RigidCPDRegistration):
    _kind = AFFINE


# ------------------------------------------------------------------------------------------------ classic rigid ICP
RIGID_REGISTRATOR, AFFINE_REGISTRATOR = 0, 1   # PoseRegistrator.RigidRegistrator3D / AffineRegistrator3D (= similarity)


class RigidICP:
    """G/other/algorithms/icp/RigidICP.scala:24-84 over csrc/rigid_icp.hip: closest points, Umeyama and the transform of the
    template run on the GPU; this class keeps `Registration`'s loop and convergence test."""

    def __init__(self, factory: "ICPFactory", targetPoints):
        self.icp = factory
        self.target = f64(targetPoints)
        self._lib = factory.ctx._lib
        h = c_void_p()
        _check(factory.ctx.handle, self._lib.gingr_rigid_icp_create(factory.ctx.handle, int(factory.registrator), factory.M,
                                                                    dptr(factory.template), self.target.shape[0], dptr(self.target),
                                                                    ctypes.byref(h)), "gingr_rigid_icp_create")
        self._h = h

    def close(self):
        if self._h:
            self._lib.gingr_rigid_icp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def points(self) -> np.ndarray:
        p = np.empty((self.icp.M, 3))
        _check(self.icp.ctx.handle, self._lib.gingr_rigid_icp_get(self._h, dptr(p), None), "gingr_rigid_icp_get")
        return p

    def transform(self) -> Tuple[float, np.ndarray, np.ndarray]:
        """(s, R, t) of the last iteration: p -> s R p + t"""
        t = np.empty(13)
        _check(self.icp.ctx.handle, self._lib.gingr_rigid_icp_get(self._h, None, dptr(t)), "gingr_rigid_icp_get")
        return float(t[0]), t[1:10].reshape(3, 3).copy(), t[10:13].copy()

    def Iteration(self, template=None) -> Tuple[np.ndarray, float]:
        """(registered points, mean closest-point distance BEFORE the move)  (:75-82)"""
        if template is not None:
            tp = f64(template)
            _check(self.icp.ctx.handle, self._lib.gingr_rigid_icp_set(self._h, dptr(tp)), "gingr_rigid_icp_set")
        d = np.empty(1)
        _check(self.icp.ctx.handle, self._lib.gingr_rigid_icp_iterate(self._h, 1, dptr(d)), "gingr_rigid_icp_iterate")
        return self.points(), float(d[0])

    def Registration(self, max_iteration: int, tolerance: float = 0.001, verbose: bool = False) -> np.ndarray:
        """:30-55: stop when the mean distance changes by less than `tolerance` between two iterations (first comparison against
        0.0); the converged iteration's points are the result."""
        _check(self.icp.ctx.handle, self._lib.gingr_rigid_icp_set(self._h, dptr(self.icp.template)), "gingr_rigid_icp_set")
        last = 0.0
        i, converged = 0, False
        self.iterations = 0
        d = np.empty(1)
        while i < max_iteration and not converged:
            _check(self.icp.ctx.handle, self._lib.gingr_rigid_icp_iterate(self._h, 1, dptr(d)), "gingr_rigid_icp_iterate")
            if verbose:
                print(f"ICP, iteration: {i}, distance: {d[0]}")
            if abs(d[0] - last) < tolerance:
                if verbose:
                    print("Converged")
                converged = True
            last = float(d[0])
            i += 1
        self.iterations, self.converged, self.distance = i, converged, last
        return self.points()


class ICPFactory:
    """G/other/algorithms/icp/ICPFactory.scala:28-38 (`registrator`: RIGID_REGISTRATOR or AFFINE_REGISTRATOR, the implicit
    Registrator of the reference)."""

    def __init__(self, ctx: Context, templatePoints, registrator: int = RIGID_REGISTRATOR):
        self.ctx, self.template, self.registrator = ctx, f64(templatePoints), registrator
        self.M = self.template.shape[0]

    def registerRigidly(self, targetPoints) -> RigidICP:
        return RigidICP(self, targetPoints)


class RigidICPRegistration:
    """G/other/algorithms/RigidICPRegistration.scala:24-46.  As in the reference the warp field pairs the TARGET's points with the
    registered template points by index (`target.points zip registration.points`, :40-43), i.e. the result is the registered
    template carried on the target's topology -- both point sets must have the same size for that to mean anything."""

    def __init__(self, ctx: Context, template, max_iterations: int = 100, registrator: int = RIGID_REGISTRATOR):
        self.icp, self.max_iterations = ICPFactory(ctx, template, registrator), max_iterations

    def register(self, target) -> np.ndarray:
        task = self.icp.registerRigidly(target)
        try:
            reg = task.Registration(self.max_iterations)
        finally:
            task.close()
        tgt = f64(target)
        n = min(tgt.shape[0], reg.shape[0])
        return tgt[:n] + (reg[:n] - tgt[:n])


# ------------------------------------------------------------------------------------------------ optimal-step non-rigid ICP
NICP_DEFAULT_ALPHA = [1e1] * 11   # NonRigidOptimalStepICP.scala:63-65: the scanLeft / reverse chain ends in `.map(_ => 1e1)`


def nicp_edges(cells) -> np.ndarray:
    """trianglesToEdges (:67-76): the unique sorted vertex pairs of the triangles, (E, 2) int32 with p1 < p2 (the reference keeps
    them in a Set's iteration order; the order of the rows of M does not change the least-squares solution)."""
    t = np.sort(np.asarray(cells, dtype=np.int64).reshape(-1, 3), axis=1)
    e = np.concatenate([t[:, [0, 1]], t[:, [0, 2]], t[:, [1, 2]]])
    return np.ascontiguousarray(np.unique(e, axis=0), dtype=np.int32)


class NonRigidOptimalStepICP:
    """G/other/algorithms/icp/NonRigidOptimalStepICP.scala:31-284 (Amberg et al., "Optimal Step Nonrigid ICP Algorithms for Surface
    Registration"): `kind` "T" = N-ICP-T (one displacement per vertex), "A" = N-ICP-A (one affine 4 x 3 map per vertex).
    Both halves of an iteration run on the GPU: the correspondence (closest target surface point + the three rejection tests of
    ClosestPointTriangleMesh3D: the surface-ICP query of the GiNGR path, asked for the current template through
    gingr_fitter_set_fit_points) and the least-squares step (gingr_nicp_solve: normal equations, blocked MFMA Cholesky).
    Landmarks: two mappings id -> point; the common ids are used (:45-55)."""

    def __init__(self, ctx: Context, templateMesh, targetMesh, templateLandmarks=None, targetLandmarks=None, gamma: float = 1.0,
                 kind: str = "T"):
        from . import api as ga
        if gamma < 0:
            raise ValueError("gamma >= 0 required")
        if kind not in ("T", "A"):
            raise ValueError("kind is 'T' or 'A'")
        self.ctx, self.kind, self.gamma = ctx, kind, float(gamma)
        self.template = f64(templateMesh[0])
        self.cells = np.ascontiguousarray(templateMesh[1], dtype=np.int32).reshape(-1, 3)
        self.target = f64(targetMesh[0])
        self.targetCells = np.ascontiguousarray(targetMesh[1], dtype=np.int32).reshape(-1, 3)
        self.n = self.template.shape[0]
        self.edges = nicp_edges(self.cells)
        tl, gl = dict(templateLandmarks or {}), dict(targetLandmarks or {})
        common = [k for k in tl if k in gl]
        if common:
            tp = f64(np.array([tl[k] for k in common]))
            gp = f64(np.array([gl[k] for k in common]))
            self.lmIdsOnTemplate = ctx.nn(tp, self.template)[0].astype(np.int32)       # closest template VERTEX of each landmark
            self.UL = self.target[ctx.nn(gp, self.target)[0]].copy()                    # closest target VERTEX (not the landmark)
        else:
            self.lmIdsOnTemplate, self.UL = np.zeros(0, dtype=np.int32), np.zeros((0, 3))
        # carrier of the correspondence query: a fitter over the template's topology (the basis is never used)
        M = self.n
        self._model = ga.PointDistributionModel(self.template, np.zeros((M, 3)), np.zeros((3 * M, 1)), np.ones(1), cells=self.cells)
        self._algo = ga.IcpRegistration(ctx)
        cfg = ga.IcpConfiguration(maxIterations=1, initialSigma=1.0, endSigma=1.0, correspondenceMethod="TriangularClosestPoint")
        self._state = self._algo.createInitialState(self._model, self.target, cfg, transform=ga.GlobalTranformationType.NoTransforms,
                                                    targetCells=self.targetCells)
        self._lib = ctx._lib

    def close(self):
        if self._algo is not None:
            self._algo.close()
            self._algo = None

    def getClosestPoints(self, template) -> Tuple[np.ndarray, np.ndarray, float]:
        """(:118-128) (closest target surface points, weights in {0, 1}, mean distance) of the given template points."""
        from . import _native as nat
        a, st = self._algo, self._state
        g, c = st.general, st.config
        pts = f64(template)
        a._bind(g, c.useLandmarkCorrespondence)
        a._push_state(g)
        a._device_state = None
        a._select_direction(c)
        a._select_surface_method(c)
        _check(self.ctx.handle, self._lib.gingr_fitter_set_fit_points(a._fitter, dptr(pts)), "gingr_fitter_set_fit_points")
        p = nat.IcpParams(c.initialSigma, c.endSigma, c.maxIterations)
        _check(self.ctx.handle, self._lib.gingr_fitter_icp_surface_phase_async(a._fitter, ctypes.byref(p), 0),
               "gingr_fitter_icp_surface_phase_async")
        cp, w = np.empty((self.n, 3)), np.empty(self.n)
        _check(self.ctx.handle, self._lib.gingr_fitter_get_surface_correspondence(a._fitter, dptr(cp), dptr(w)),
               "gingr_fitter_get_surface_correspondence")
        dd = cp - pts
        dist = float(np.sqrt((dd[:, 0] * dd[:, 0] + dd[:, 1] * dd[:, 1]) + dd[:, 2] * dd[:, 2]).sum() / self.n)
        return cp, w, dist

    def Iteration(self, template, alpha: float, beta: float):
        """(:151-190 / :241-283) -> (moved template points, mean distance BEFORE the move, moved landmark vertices)"""
        if alpha < 0 or beta < 0:
            raise ValueError("alpha, beta >= 0 required")
        from ._native import iptr
        pts = f64(template)
        cp, w, dist = self.getClosestPoints(pts)
        out, lm = np.empty((self.n, 3)), np.empty((self.lmIdsOnTemplate.shape[0], 3))
        _check(self.ctx.handle, self._lib.gingr_nicp_solve(
            self.ctx.handle, 0 if self.kind == "T" else 1, self.n, dptr(pts), self.edges.shape[0], iptr(self.edges), dptr(w), dptr(cp),
            self.lmIdsOnTemplate.shape[0], iptr(self.lmIdsOnTemplate) if len(self.lmIdsOnTemplate) else None,
            dptr(self.UL) if len(self.UL) else None, float(alpha), float(beta), self.gamma, dptr(out), dptr(lm) if len(lm) else None),
            "gingr_nicp_solve")
        return out, dist, lm

    def Registration(self, max_iteration: int, tolerance: float = 0.001, alpha=None, beta=None, verbose: bool = False) -> np.ndarray:
        """(:89-121): one stage per (alpha, beta) pair, each up to max_iteration steps or until the mean distance measured before
        a step is below the tolerance."""
        alpha = NICP_DEFAULT_ALPHA if alpha is None else list(alpha)
        beta = alpha if beta is None else list(beta)
        if len(alpha) != len(beta):
            raise ValueError("alpha and beta need the same length")
        fit = self.template
        self.iterations = 0
        for j, (a, b) in enumerate(zip(alpha, beta)):
            dist, i = float("inf"), 0
            while i < max_iteration and dist >= tolerance:
                fit, dist, _ = self.Iteration(fit, a, b)
                if verbose:
                    print(f"ICP, iteration: {j * max_iteration + i}/{max_iteration * len(alpha)}, alpha: {a}, beta: {b}, "
                          f"average distance to target: {dist}")
                i += 1
                self.iterations += 1
        return fit


def NonRigidOptimalStepICP_T(ctx, templateMesh, targetMesh, templateLandmarks=None, targetLandmarks=None, gamma: float = 1.0):
    return NonRigidOptimalStepICP(ctx, templateMesh, targetMesh, templateLandmarks, targetLandmarks, gamma, "T")


def NonRigidOptimalStepICP_A(ctx, templateMesh, targetMesh, templateLandmarks=None, targetLandmarks=None, gamma: float = 1.0):
    return NonRigidOptimalStepICP(ctx, templateMesh, targetMesh, templateLandmarks, targetLandmarks, gamma, "A")
