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
        current = self.sigma2()
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
