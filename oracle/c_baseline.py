"""Run-time build + ctypes wrapper of oracle/cpd_baseline.c, the OPTIMISED CPU baseline (-O3 -march=native -fopenmp, SIMD
exponential) that bench.py times next to the GPU.  MEASUREMENT INFRASTRUCTURE ONLY (see the header of cpd_baseline.c); the
strict checker stays oracle/cpd_oracle.c.  The shared object is compiled on the host that runs it (native code must not travel
to another CPU) into a per-user temporary directory."""
from __future__ import annotations

import ctypes
import os
import subprocess
import tempfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
LAST_CALIBRATION = {}
CFLAGS = ["-O3", "-march=native", "-fopenmp", "-fPIC", "-shared"]


def build() -> str:
    if os.environ.get("GINGR_ORACLE_SANITIZED") == "1":  # tests/test_oracle_sanitized.py (the sanitizer runtime must be preloaded)
        subprocess.check_call(["make", "-C", _HERE, "libcpd_baseline_asan.so"], stdout=subprocess.DEVNULL)
        return os.path.join(_HERE, "libcpd_baseline_asan.so")
    out_dir = os.path.join(tempfile.gettempdir(), f"gingr_cpu_baseline_{os.getuid()}")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libcpd_baseline.so")
    src = os.path.join(_HERE, "cpd_baseline.c")
    subprocess.check_call([os.environ.get("CC", "gcc")] + CFLAGS + [src, "-o", so, "-lm"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        dp, i64, dbl = ctypes.POINTER(ctypes.c_double), ctypes.c_int64, ctypes.c_double
        L.baseline_num_threads.restype = ctypes.c_int
        L.baseline_set_num_threads.restype = None
        L.baseline_set_num_threads.argtypes = [ctypes.c_int]
        L.baseline_exp_max_rel_error.restype = dbl
        L.baseline_exp_max_rel_error.argtypes = [i64, dbl]
        L.baseline_cpd_colsum.restype = None
        L.baseline_cpd_colsum.argtypes = [i64, dp, dp, dp, i64, dp, dp, dp, dbl, dp]
        L.baseline_cpd_rowstats.restype = None
        L.baseline_cpd_rowstats.argtypes = [i64, dp, dp, dp, i64, dp, dp, dp, dbl, dp, dp, dp, dp, dp]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def soa(points) -> np.ndarray:
    """(n, 3) -> contiguous (3, n) planes"""
    return np.ascontiguousarray(np.asarray(points, dtype=np.float64).T)


def num_threads() -> int:
    return int(lib().baseline_num_threads())


def set_num_threads(n: int) -> None:
    lib().baseline_set_num_threads(int(n))


def cpu_quota():
    """the container's CPU quota in cores (cgroup v2 cpu.max / v1 cfs quota); None when unlimited or unknown"""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else float(q) / float(p)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / p
    except Exception:
        return None


def calibrate_threads(fit_soa: np.ndarray, target_soa: np.ndarray, sigma2: float, rows: int = 2048) -> int:
    """Pick the OpenMP thread count with the highest pair rate on THIS host (a container's CPU quota is often far below the
    number of hardware threads it can see; oversubscribed threads then make the baseline slower, not faster) and leave it
    set.  Every candidate (1, 2, 4, ... hardware threads) runs the column-sum pass over the same `rows` rows twice; the better
    of the two counts.  Returns the chosen count."""
    import time
    global LAST_CALIBRATION
    LAST_CALIBRATION = {}
    best, best_rate = 1, 0.0
    limit = os.cpu_count() or 1
    quota = cpu_quota()
    if quota:   # a burst of more threads than the quota looks fast for a few milliseconds and is throttled in a sustained run
        limit = max(1, min(limit, int(quota + 0.5)))
    m = min(fit_soa.shape[1], rows)
    sub = np.ascontiguousarray(fit_soa[:, :m])
    nt = 1
    while nt <= limit:
        set_num_threads(nt)
        rate = 0.0
        for _ in range(2):
            t0 = time.perf_counter()
            colsum(sub, target_soa, sigma2)
            rate = max(rate, m / (time.perf_counter() - t0))
        LAST_CALIBRATION[nt] = rate * target_soa.shape[1] / 1e9          # Gpair/s of the column-sum pass
        if rate > best_rate * 1.03:
            best, best_rate = nt, rate
        nt = nt * 2 if nt * 2 <= limit or nt == limit else limit
    set_num_threads(best)
    return best


def exp_max_rel_error(n: int = 1000001, lo: float = -700.0) -> float:
    return float(lib().baseline_exp_max_rel_error(n, lo))


def colsum(fit_soa: np.ndarray, target_soa: np.ndarray, sigma2: float) -> np.ndarray:
    """sum_i K_ij over the rows of fit_soa (3, m) for every target (3, N): the partial of den (no outlier constant)."""
    m, n = fit_soa.shape[1], target_soa.shape[1]
    out = np.empty(n)
    lib().baseline_cpd_colsum(m, _p(fit_soa[0]), _p(fit_soa[1]), _p(fit_soa[2]), n, _p(target_soa[0]), _p(target_soa[1]),
                              _p(target_soa[2]), float(sigma2), _p(out))
    return out


def rowstats(fit_soa: np.ndarray, target_soa: np.ndarray, sigma2: float, inv_den: np.ndarray):
    """P1 (m,) and PX (3, m) of the rows of fit_soa given 1/den of every target."""
    m, n = fit_soa.shape[1], target_soa.shape[1]
    inv = np.ascontiguousarray(inv_den, dtype=np.float64)
    P1, PX = np.empty(m), np.empty((3, m))
    lib().baseline_cpd_rowstats(m, _p(fit_soa[0]), _p(fit_soa[1]), _p(fit_soa[2]), n, _p(target_soa[0]), _p(target_soa[1]),
                                _p(target_soa[2]), float(sigma2), _p(inv), _p(P1), _p(PX[0]), _p(PX[1]), _p(PX[2]))
    return P1, PX
