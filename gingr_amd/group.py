"""Host-side handle of the in-library device group (include/gingr_hip.h, gingr_group_*): the row-sharded update across the
GPUs of one node driven from ONE process -- the path a C / JVM host uses (no torch.distributed).  Python is only a caller here;
the exchange (one-shot all-reduce over peer pointers) and the per-device worker threads live in libgingr_hip.so
(gingr_amd/csrc/group.hip)."""
from __future__ import annotations

import ctypes
from ctypes import c_int64, c_void_p
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _native as nat
from ._native import GingrNativeError, dptr, f64, iptr


class DeviceGroup:
    """gingr_group: one model shard + fitter per entry of `devices` (the same device may appear several times)."""

    def __init__(self, devices: Sequence[int]):
        self._lib = nat.load()
        dev = np.ascontiguousarray(devices, dtype=np.int32)
        h = c_void_p()
        rc = self._lib.gingr_group_create(int(dev.shape[0]), iptr(dev), ctypes.byref(h))
        if rc != nat.GINGR_OK:
            raise GingrNativeError(rc, "gingr_group_create", f"devices={list(devices)}")
        self.handle = h
        self.devices = [int(d) for d in devices]
        self.M = 0

    def _check(self, rc: int, where: str):
        if rc != nat.GINGR_OK:
            raise GingrNativeError(rc, where, (self._lib.gingr_group_last_error(self.handle) or b"").decode())

    @property
    def size(self) -> int:
        return int(self._lib.gingr_group_size(self.handle))

    @property
    def rank(self) -> int:
        return int(self._lib.gingr_group_model_rank(self.handle))

    def shard_rows(self, shard: int) -> Tuple[int, int]:
        b, e = c_int64(), c_int64()
        self._check(self._lib.gingr_group_shard_rows(self.handle, shard, ctypes.byref(b), ctypes.byref(e)), "gingr_group_shard_rows")
        return b.value, e.value

    def ctx_handle(self, shard: int = 0):
        return c_void_p(self._lib.gingr_group_ctx(self.handle, shard))

    # -- model --------------------------------------------------------------------------------------------------------
    def upload_model(self, reference, mean, basis, variance):
        ref, mu, var = f64(reference), f64(mean), f64(variance)
        U = np.asfortranarray(basis, dtype=np.float64)
        self.M = ref.shape[0]
        self._check(self._lib.gingr_group_model_upload(self.handle, self.M, var.shape[0], dptr(ref), dptr(mu),
                                                       U.ctypes.data_as(nat._dp), dptr(var)), "gingr_group_model_upload")

    def build_gaussian_gpmm(self, reference, sigmas: Sequence[float], scalings: Sequence[float], relative_tolerance: float,
                            max_rank: int = 0):
        ref, sg, sc = f64(reference), f64(sigmas), f64(scalings)
        self.M = ref.shape[0]
        self._check(self._lib.gingr_group_gpmm_build_gaussian(self.handle, self.M, dptr(ref), sg.shape[0], dptr(sg), dptr(sc),
                                                              float(relative_tolerance), int(max_rank)),
                    "gingr_group_gpmm_build_gaussian")

    # -- fitter -------------------------------------------------------------------------------------------------------
    def set_target(self, target):
        x = f64(target)
        self._check(self._lib.gingr_group_set_target(self.handle, x.shape[0], dptr(x)), "gingr_group_set_target")

    def set_landmarks(self, pids, points, covs):
        if pids is None or len(pids) == 0:
            self._check(self._lib.gingr_group_set_landmarks(self.handle, 0, None, None, None), "gingr_group_set_landmarks")
            return
        p = np.ascontiguousarray(pids, dtype=np.int32)
        x, c = f64(points), f64(covs)
        self._check(self._lib.gingr_group_set_landmarks(self.handle, p.shape[0], iptr(p), dptr(x), dptr(c)), "gingr_group_set_landmarks")

    def set_options(self, global_transform: int = 1, step_length: float = 1.0):
        self._check(self._lib.gingr_group_set_options(self.handle, int(global_transform), float(step_length)), "gingr_group_set_options")

    def set_state(self, alpha, sigma2: float, euler=(0.0, 0.0, 0.0), center=(0.0, 0.0, 0.0), translation=(0.0, 0.0, 0.0),
                  scale: float = 1.0, iteration: int = 0, status: int = 0):
        s = nat.StateScalars()
        s.euler[:] = list(euler)
        s.center[:] = list(center)
        s.translation[:] = list(translation)
        s.scale, s.sigma2, s.iteration, s.status = scale, sigma2, iteration, status
        a = f64(alpha)
        self._check(self._lib.gingr_group_set_state(self.handle, dptr(a), ctypes.byref(s)), "gingr_group_set_state")

    def get_state(self, fit: bool = True):
        alpha = np.empty(self.rank)
        out = np.empty((self.M, 3)) if fit else None
        s = nat.StateScalars()
        self._check(self._lib.gingr_group_get_state(self.handle, dptr(alpha), ctypes.byref(s), dptr(out)), "gingr_group_get_state")
        return alpha, s, out

    def update_cpd(self, w: float = 0.0, lambda_: float = 1.0, n_iterations: int = 1):
        p = nat.CpdParams(w, lambda_)
        self._check(self._lib.gingr_group_update_cpd_async(self.handle, ctypes.byref(p), int(n_iterations)), "gingr_group_update_cpd_async")

    def update_icp(self, initial_sigma: float, end_sigma: float, max_iterations: int, n_iterations: int = 1):
        p = nat.IcpParams(initial_sigma, end_sigma, max_iterations)
        self._check(self._lib.gingr_group_update_icp_async(self.handle, ctypes.byref(p), int(n_iterations)), "gingr_group_update_icp_async")

    # -- round 4: surface correspondence, sampled proposal, transition density through the group ------------------------
    def set_meshes(self, model_triangles, target_triangles, method: int = 0):
        """Triangles of the WHOLE template (vertex ids of the full model) and of the target (gingr_group_set_meshes)."""
        mt = np.ascontiguousarray(model_triangles, dtype=np.int32)
        tt = np.ascontiguousarray(target_triangles, dtype=np.int32)
        self._check(self._lib.gingr_group_set_meshes(self.handle, mt.shape[0], iptr(mt), tt.shape[0], iptr(tt)), "gingr_group_set_meshes")
        self._check(self._lib.gingr_group_set_surface_method(self.handle, int(method)), "gingr_group_set_surface_method")

    def set_correspondence_direction(self, reversed: bool):
        """IcpConfiguration.reverseCorrespondenceDirection (ICP.scala:46-48) for the ICP flavours of `update`; with more than one shard
        the meshes must have been set (the correspondence runs replicated against the gathered template)."""
        self._check(self._lib.gingr_group_set_correspondence_direction(self.handle, 1 if reversed else 0),
                    "gingr_group_set_correspondence_direction")

    @staticmethod
    def _params(flavour: int, params):
        cp = nat.CpdParams(*params) if flavour == nat.FLAVOUR_CPD else None
        ip = nat.IcpParams(*params) if flavour != nat.FLAVOUR_CPD else None
        return cp, ip

    def update(self, flavour: int, params, n_iterations: int = 1, z=None):
        """flavour 0 CPD (params = (w, lambda)), 1 ICP point cloud, 2 ICP surface (params = (initialSigma, endSigma, maxIterations));
        z: rank standard normals = the sampled proposal (one iteration)."""
        cp, ip = self._params(flavour, params)
        zz = None if z is None else f64(z)
        self._check(self._lib.gingr_group_update_async(self.handle, int(flavour), ctypes.byref(cp) if cp else None,
                                                       ctypes.byref(ip) if ip else None, int(n_iterations), dptr(zz)),
                    "gingr_group_update_async")

    def posterior_logpdf(self, flavour: int, params, mesh) -> float:
        cp, ip = self._params(flavour, params)
        m = f64(mesh)
        out = ctypes.c_double()
        self._check(self._lib.gingr_group_posterior_logpdf(self.handle, int(flavour), ctypes.byref(cp) if cp else None,
                                                           ctypes.byref(ip) if ip else None, dptr(m), ctypes.byref(out)),
                    "gingr_group_posterior_logpdf")
        return float(out.value)

    def synchronize(self):
        self._check(self._lib.gingr_group_synchronize(self.handle), "gingr_group_synchronize")

    def exchange_info(self) -> dict:
        """How the shards exchange: physical devices behind them, and whether the peer-read send buffers are fine-grained device
        memory (gingr_group_exchange_info; always true once shards sit on more than one device)."""
        nd, fg = ctypes.c_int32(0), ctypes.c_int32(0)
        self._check(self._lib.gingr_group_exchange_info(self.handle, ctypes.byref(nd), ctypes.byref(fg)), "gingr_group_exchange_info")
        return {"distinct_devices": int(nd.value), "fine_grained_send_buffers": bool(fg.value)}

    def close(self):
        if getattr(self, "handle", None):
            self._lib.gingr_group_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
