"""ctypes wrapper of oracle/cpd_oracle.c (streaming C restatement).  TEST INFRASTRUCTURE ONLY --
see the header of cpd_oracle.c; PARITY UNPINNED (no reference golden vectors exist for this path)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build() -> str:
    if os.environ.get("GINGR_ORACLE_SANITIZED") == "1":  # tests/test_oracle_sanitized.py (the sanitizer runtime must be preloaded)
        subprocess.check_call(["make", "-C", _HERE, "libcpd_oracle_asan.so"], stdout=subprocess.DEVNULL)
        return os.path.join(_HERE, "libcpd_oracle_asan.so")
    so = os.path.join(_HERE, "libcpd_oracle.so")
    src = os.path.join(_HERE, "cpd_oracle.c")
    if not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(so) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", _HERE, "libcpd_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        dp = ctypes.POINTER(ctypes.c_double)
        i64 = ctypes.c_int64
        L.oracle_cpd_stats.argtypes = [i64, i64, dp, dp, ctypes.c_double, ctypes.c_double, dp, dp, dp, dp, dp]
        L.oracle_cpd_stats.restype = None
        L.oracle_cpd_colsum_partial.argtypes = [i64, i64, i64, dp, dp, ctypes.c_double, dp]
        L.oracle_cpd_colsum_partial.restype = None
        L.oracle_cpd_rowstats_partial.argtypes = [i64, i64, i64, dp, dp, ctypes.c_double, dp, dp, dp]
        L.oracle_cpd_rowstats_partial.restype = None
        L.oracle_cpd_outlier_constant.argtypes = [i64, i64, ctypes.c_double, ctypes.c_double]
        L.oracle_cpd_outlier_constant.restype = ctypes.c_double
        L.oracle_cpd_initial_sigma2.argtypes = [i64, i64, dp, dp]
        L.oracle_cpd_initial_sigma2.restype = ctypes.c_double
        L.oracle_nn.argtypes = [i64, i64, dp, dp, ctypes.POINTER(ctypes.c_int32), dp]
        L.oracle_nn.restype = ctypes.c_double
        L.oracle_gauss_block.argtypes = [i64, i64, dp, dp, ctypes.c_double, ctypes.c_double, dp]
        L.oracle_gauss_block.restype = None
        L.oracle_num_threads.restype = ctypes.c_int
        L.oracle_mesh_closest_point.argtypes = [i64, dp, i64, dp, ctypes.POINTER(ctypes.c_int32), dp, dp, ctypes.POINTER(ctypes.c_int32)]
        L.oracle_mesh_closest_point.restype = None
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def num_threads() -> int:
    return int(lib().oracle_num_threads())


def cpd_stats(fit, target, sigma2: float, w: float):
    """Returns gingr_oracle.CpdStats computed by the streaming C restatement."""
    from .gingr_oracle import CpdStats
    y, x = _c(fit), _c(target)
    M, N = y.shape[0], x.shape[0]
    den = np.empty(N); P1 = np.empty(M); PX = np.empty((M, 3)); Pt1 = np.empty(N); sc = np.empty(6)
    lib().oracle_cpd_stats(M, N, _p(y), _p(x), sigma2, w, _p(den), _p(P1), _p(PX), _p(Pt1), _p(sc))
    return CpdStats(den=den, P1=P1, PX=PX, Pt1=Pt1, Np=float(sc[0]), sigma2_next=float(sc[4]))


def cpd_colsum_partial(fit, target, sigma2: float, i0: int, i1: int):
    y, x = _c(fit), _c(target)
    out = np.empty(x.shape[0])
    lib().oracle_cpd_colsum_partial(i0, i1, x.shape[0], _p(y), _p(x), sigma2, _p(out))
    return out


def cpd_rowstats_partial(fit, target, sigma2: float, den, i0: int, i1: int):
    y, x, den = _c(fit), _c(target), _c(den)
    M = y.shape[0]
    P1 = np.zeros(M); PX = np.zeros((M, 3))
    lib().oracle_cpd_rowstats_partial(i0, i1, x.shape[0], _p(y), _p(x), sigma2, _p(den), _p(P1), _p(PX))
    return P1[i0:i1].copy(), PX[i0:i1].copy()


def outlier_constant(M, N, sigma2, w) -> float:
    return float(lib().oracle_cpd_outlier_constant(M, N, sigma2, w))


def initial_sigma2(ref, target) -> float:
    y, x = _c(ref), _c(target)
    return float(lib().oracle_cpd_initial_sigma2(y.shape[0], x.shape[0], _p(y), _p(x)))


def nn(query, target):
    y, x = _c(query), _c(target)
    idx = np.empty(y.shape[0], dtype=np.int32); d2 = np.empty(y.shape[0])
    dist = lib().oracle_nn(y.shape[0], x.shape[0], _p(y), _p(x),
                           idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), _p(d2))
    return idx, d2, float(dist)


def gauss_block(A, B, sigma: float, scaling: float):
    A, B = _c(A), _c(B)
    out = np.empty((A.shape[0], B.shape[0]))
    lib().oracle_gauss_block(A.shape[0], B.shape[0], _p(A), _p(B), sigma, scaling, _p(out))
    return out


def mesh_closest_point(points, verts, tris):
    """closestPointOnSurface of every row of `points` on the mesh (verts, tris), brute force in C:
    (closest points (K,3), squared distances (K,), triangle index (K,)); ties -> lowest triangle index."""
    p, v = _c(points), _c(verts)
    t = np.ascontiguousarray(tris, dtype=np.int32).reshape(-1, 3)
    K = p.shape[0]
    cp, d2, tid = np.empty((K, 3)), np.empty(K), np.empty(K, dtype=np.int32)
    lib().oracle_mesh_closest_point(K, _p(p), t.shape[0], _p(v), t.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), _p(cp), _p(d2),
                                    tid.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    return cp, d2, tid
