"""Row-sharded GiNGR update: one process per GPU, reference-point rows split across ranks.

The N x M affinity matrix is sharded by reference (fit) rows; the target cloud and all r-sized state are replicated
(SURVEY.md section 8e).  One iteration is the three phases of gingr_amd/csrc/fitter.hip; after phases 0 and 1 the partial
sums in exchange segment p are all-reduced (sum, float64) across ranks:

    segment 0  CPD column sums den_j (N doubles)          <- the column-sum exchange named in BASELINE.json north_star
    segment 1  weighted Gram (rp*rp) + rhs (rp) + 8 sigma^2 partial sums

Everything after the posterior solve is replicated O(r^2) algebra on one-off moments of the basis (all-reduced once at
model finalisation), so phase 2 needs no collective.

Two ways to run the collective:
  * native (`rccl=True`, what bench.py --gpus N uses): the context owns an RCCL communicator (Context.rccl_init) and the LIBRARY
    enqueues ncclAllReduce on its own stream between the phases (gingr_fitter_update_*_rccl_async) -- n iterations with no callback,
    no stream hop and no Python between the kernels;
  * host-driven (`all_reduce=callable`): `torch.distributed.all_reduce` (or anything else, e.g. gloo on host copies in the CPU test)
    on a tensor that aliases the library's exchange buffer, called back once per exchange.
The r x r solve and the 3x3 SVD are replicated on every rank (deterministic, so no broadcast is needed).
"""
from __future__ import annotations

import ctypes
from ctypes import c_int64, c_void_p
from typing import Callable, Optional, Tuple

import numpy as np

from . import _native as nat
from .api import (Context, DeviceModel, PointDistributionModel, _check, f64, dptr)

NUM_PHASES = nat.NUM_PHASES
NUM_SEGMENTS = nat.NUM_SEGMENTS


def shard_rows(M: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous balanced row ranges: the first (M mod world) ranks hold one extra row."""
    base, extra = divmod(int(M), int(world))
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


PHASE_GATHER = 3        # GINGR_PHASE_GATHER / GINGR_SEGMENT_FULLFIT of gingr_hip.h
SEGMENT_FULLFIT = nat.SEGMENT_FULLFIT
SEGMENT_REVSUM = nat.SEGMENT_REVSUM


def drive_update(run_phase: Callable[[int], None], all_reduce_segment: Callable[[int], None], world: int,
                 skip_segment0: bool = False, flavour: int = 0, reversed_direction: bool = False) -> None:
    """One iteration in the order of fitter_sharded_update (gingr_amd/csrc/fitter.hip): for the surface correspondence (flavour 2)
    the gather of the fit first; then phase p followed (for p < 2) by the all-reduce of exchange segment p -- the ICP flavours
    exchange nothing after phase 0 (their closest-point search is local to the shard's rows)."""
    if (flavour == 2 or (flavour == 1 and reversed_direction)) and world > 1:   # (the reversed direction works on the gathered template)
        run_phase(PHASE_GATHER)
        all_reduce_segment(SEGMENT_FULLFIT)
    for ph in range(NUM_PHASES):
        run_phase(ph)
        if world > 1 and ph < 2 and not ((skip_segment0 or flavour != 0) and ph == 0):
            all_reduce_segment(ph)
        if world > 1 and ph == 0 and flavour != 0 and reversed_direction:
            # every shard scanned its index range of the target queries: the per-template-vertex sums are totalled
            all_reduce_segment(SEGMENT_REVSUM)


class _DevArray:
    """Minimal __cuda_array_interface__ view of device memory owned by libgingr_hip."""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


def as_torch(ptr: int, count: int, device_index: int = 0):
    import torch
    return torch.as_tensor(_DevArray(ptr, count), device=f"cuda:{device_index}")


class ShardedFitter:
    """The native fitter of one rank (rows [begin, end) of the model) plus the exchange plumbing.

    all_reduce(tensor) must sum the float64 device tensor in place across all ranks; None means a single shard.

    Stream contract (world > 1): the phase kernels run on the context's stream, the collective on torch's CURRENT stream.  When
    the two are the same stream (bench.py: `ctx.set_stream(stream.cuda_stream)` + `with torch.cuda.stream(stream)`) stream
    order alone makes the exchange correct and nothing waits on the host.  Otherwise every exchange is bracketed by host
    synchronisation of both streams (correct, slower) -- checked per call, so a caller cannot race the collective against the
    column-sum / Gram kernels by accident.
    """

    def __init__(self, ctx: Context, model: PointDistributionModel, target, rank: int = 0, world: int = 1,
                 all_reduce: Optional[Callable] = None, global_transform: int = 1, step_length: float = 1.0,
                 defer_setup: bool = False, rccl: bool = False):
        self.ctx, self.rank, self.world = ctx, rank, world
        self._lib = ctx._lib
        self.all_reduce = all_reduce
        self.rccl = bool(rccl)  # the context's own RCCL communicator does the exchanges (Context.rccl_init was called)
        self.begin, self.end = shard_rows(model.numberOfPoints, world, rank)
        self.dev_model = DeviceModel(ctx, model, self.begin, self.end)
        self.model = model
        self._target, self._opts = target, (global_transform, step_length)
        self.handle = None
        if not defer_setup:
            if self.rccl:  # (also with a one-rank communicator: the native path end to end)
                ptr, n = self.dev_model.gram_exchange()
                ctx.rccl_allreduce(ptr, n)
                ctx.synchronize()
            elif world > 1:
                g = self.gram_tensor()
                ctx.synchronize()
                all_reduce(g)
                self._sync_torch()
            self.finish_setup()

    def gram_tensor(self):
        """Device tensor aliasing this shard's partial Q^T Q (to be summed across shards once, before finish_setup)."""
        ptr, n = self.dev_model.gram_exchange()
        return as_torch(ptr, n, self.ctx.device)

    def finish_setup(self):
        ctx, world = self.ctx, self.world
        target = self._target
        global_transform, step_length = self._opts
        if world > 1:
            self.dev_model.finalize()
        h = c_void_p()
        _check(ctx.handle, self._lib.gingr_fitter_create(ctx.handle, self.dev_model.handle, ctypes.byref(h)), "gingr_fitter_create")
        self.handle = h
        x = f64(target)
        _check(ctx.handle, self._lib.gingr_fitter_set_target(h, x.shape[0], dptr(x)), "gingr_fitter_set_target")
        _check(ctx.handle, self._lib.gingr_fitter_set_options(h, int(global_transform), float(step_length)), "set_options")
        p = c_void_p()
        offs = (c_int64 * NUM_SEGMENTS)()
        cnts = (c_int64 * NUM_SEGMENTS)()
        _check(ctx.handle, self._lib.gingr_fitter_exchange(h, ctypes.byref(p), offs, cnts), "gingr_fitter_exchange")
        self.offsets, self.counts = list(offs), list(cnts)
        self.xch = None
        if world > 1:
            total = self.offsets[-1] + self.counts[-1]
            self.xch = as_torch(p.value, total, ctx.device)

    def _sync_torch(self):
        import torch
        torch.cuda.synchronize(self.ctx.device)

    def set_state(self, alpha, sigma2: float, euler=(0.0, 0.0, 0.0), center=(0.0, 0.0, 0.0),
                  translation=(0.0, 0.0, 0.0), scale: float = 1.0, iteration: int = 0, status: int = 0):
        s = nat.StateScalars()
        s.euler[:] = list(euler)
        s.center[:] = list(center)
        s.translation[:] = list(translation)
        s.scale, s.sigma2, s.iteration, s.status = scale, sigma2, iteration, status
        a = f64(alpha)
        _check(self.ctx.handle, self._lib.gingr_fitter_set_state(self.handle, dptr(a), ctypes.byref(s)), "gingr_fitter_set_state")

    def get_state(self):
        r = self.model.rank
        M = self.end - self.begin
        alpha, fit = np.empty(r), np.empty((M, 3))
        s = nat.StateScalars()
        _check(self.ctx.handle, self._lib.gingr_fitter_get_state(self.handle, dptr(alpha), ctypes.byref(s), dptr(fit)),
               "gingr_fitter_get_state")
        return alpha, s, fit

    def get_cpd_stats(self) -> dict:
        """Affinity statistics of the LAST CPD evaluation of this shard (gingr_fitter_get_cpd_stats): P1 / PX of the local rows, den
        of every target (reduced), scalars {Np, xPx, trPXY, yPy, sigma2_next, c}."""
        M = self.end - self.begin
        N = self.counts[0]
        P1, PX, den, sc = np.empty(M), np.empty((M, 3)), np.empty(N), np.empty(6)
        _check(self.ctx.handle, self._lib.gingr_fitter_get_cpd_stats(self.handle, dptr(P1), dptr(PX), dptr(den), dptr(sc)),
               "gingr_fitter_get_cpd_stats")
        return {"P1": P1, "PX": PX, "den": den, "Np": float(sc[0]), "sigma2_next": float(sc[4]), "c": float(sc[5])}

    def _same_stream(self) -> bool:
        import torch
        return int(self.ctx.get_stream() or 0) == int(torch.cuda.current_stream(self.ctx.device).cuda_stream or 0)

    def _exchange(self, seg: int):
        """all-reduce of exchange segment `seg`, ordered against the library's kernels (see the class docstring)"""
        if self._same_stream():
            self.all_reduce(self._segment(seg))
            return
        import torch
        self.ctx.synchronize()                                     # the partial sums are complete
        self.all_reduce(self._segment(seg))
        torch.cuda.current_stream(self.ctx.device).synchronize()    # the sums are in place before the next phase reads them

    def _reduce_callback(self):
        """The gingr_allreduce_fn handed to the library (kept alive as long as the fitter): segment index -> all-reduce of that
        segment of the exchange buffer, ordered against the library's stream (_exchange).  An exception raised by the collective
        is kept and re-raised by the caller of the update; the library sees a non-zero status and stops enqueuing."""
        if getattr(self, "_reduce_cb", None) is None:
            def cb(_user, seg, _ptr, _count):
                try:
                    self._exchange(int(seg))
                    return 0
                except BaseException as e:  # must not propagate through the C frame
                    self._cb_error = e
                    return 1
            self._reduce_cb = nat.ALLREDUCE_FN(cb)
        self._cb_error = None
        return self._reduce_cb

    def _sharded_call(self, fn_name: str, params, n_iterations: int):
        cb = self._reduce_callback()
        rc = getattr(self._lib, fn_name)(self.handle, ctypes.byref(params), int(n_iterations), cb, None)
        if self._cb_error is not None:
            err, self._cb_error = self._cb_error, None
            raise err
        _check(self.ctx.handle, rc, fn_name)

    def update_cpd(self, w: float = 0.0, lambda_: float = 1.0, n_iterations: int = 1):
        """n iterations; world > 1: ONE library call that runs the three phases per iteration and calls back for the two all-reduces
        (gingr_fitter_update_cpd_sharded_async) -- no Python between the kernels of an iteration except inside the collective."""
        p = nat.CpdParams(w, lambda_)
        if self.rccl:
            _check(self.ctx.handle, self._lib.gingr_fitter_update_cpd_rccl_async(self.handle, ctypes.byref(p), n_iterations),
                   "gingr_fitter_update_cpd_rccl_async")
            return
        if self.world == 1:
            _check(self.ctx.handle, self._lib.gingr_fitter_update_cpd_async(self.handle, ctypes.byref(p), n_iterations),
                   "gingr_fitter_update_cpd_async")
            return
        self._sharded_call("gingr_fitter_update_cpd_sharded_async", p, n_iterations)

    def update_icp(self, initial_sigma: float, end_sigma: float, max_iterations: int, n_iterations: int = 1):
        p = nat.IcpParams(initial_sigma, end_sigma, max_iterations)
        if self.rccl:
            _check(self.ctx.handle, self._lib.gingr_fitter_update_icp_rccl_async(self.handle, ctypes.byref(p), n_iterations),
                   "gingr_fitter_update_icp_rccl_async")
            return
        if self.world == 1:
            _check(self.ctx.handle, self._lib.gingr_fitter_update_icp_async(self.handle, ctypes.byref(p), n_iterations),
                   "gingr_fitter_update_icp_async")
            return
        self._sharded_call("gingr_fitter_update_icp_sharded_async", p, n_iterations)

    # ---- round 4: any flavour / sampled proposal / transition density on a row shard -----------------------------------------
    def set_meshes(self, model_triangles, target_triangles, method: int = 0):
        """Triangles of the WHOLE template (vertex ids of the full model) and of the target; on every shard."""
        mt = np.ascontiguousarray(model_triangles, dtype=np.int32)
        tt = np.ascontiguousarray(target_triangles, dtype=np.int32)
        _check(self.ctx.handle, self._lib.gingr_fitter_set_meshes(self.handle, mt.shape[0], nat.iptr(mt), tt.shape[0], nat.iptr(tt)),
               "gingr_fitter_set_meshes")
        _check(self.ctx.handle, self._lib.gingr_fitter_set_surface_method(self.handle, int(method)), "gingr_fitter_set_surface_method")
        self._fullfit = None
        if self.world > 1 and not self.rccl:
            p, n = c_void_p(), c_int64()
            _check(self.ctx.handle, self._lib.gingr_fitter_fullfit_exchange(self.handle, ctypes.byref(p), ctypes.byref(n)),
                   "gingr_fitter_fullfit_exchange")
            self._fullfit = as_torch(p.value, n.value, self.ctx.device)

    def set_correspondence_direction(self, reversed: bool):
        """IcpConfiguration.reverseCorrespondenceDirection on this shard (after set_meshes): the shard scans its index range of the target
        queries against the gathered template, the per-template-vertex sums are all-reduced (SEGMENT_REVSUM) and the shard keeps the
        observations of its own rows."""
        _check(self.ctx.handle, self._lib.gingr_fitter_set_correspondence_direction(self.handle, 1 if reversed else 0),
               "gingr_fitter_set_correspondence_direction")
        self._revsum = None
        if reversed and self.world > 1 and not self.rccl:
            p, n = c_void_p(), c_int64()
            _check(self.ctx.handle, self._lib.gingr_fitter_reversal_exchange(self.handle, ctypes.byref(p), ctypes.byref(n)),
                   "gingr_fitter_reversal_exchange")
            self._revsum = as_torch(p.value, n.value, self.ctx.device)

    def _segment(self, seg: int):
        if seg == nat.SEGMENT_FULLFIT:
            return self._fullfit
        if seg == nat.SEGMENT_REVSUM:
            return self._revsum
        return self.xch[self.offsets[seg]: self.offsets[seg] + self.counts[seg]]

    @staticmethod
    def _params(flavour: int, params):
        cp = nat.CpdParams(*params) if flavour == nat.FLAVOUR_CPD else None
        ip = nat.IcpParams(*params) if flavour != nat.FLAVOUR_CPD else None
        return (ctypes.byref(cp) if cp else None), (ctypes.byref(ip) if ip else None), (cp, ip)

    def update(self, flavour: int, params, n_iterations: int = 1, z=None):
        """flavour 0 CPD (params = (w, lambda)), 1 ICP point cloud, 2 ICP surface (params = (initialSigma, endSigma, maxIterations));
        z: rank standard normals = update(current, probabilistic = true), one iteration, the same on every rank."""
        cpp, ipp, _keep = self._params(flavour, params)
        zz = None if z is None else f64(z)
        if self.rccl:
            _check(self.ctx.handle, self._lib.gingr_fitter_update_rccl_async(self.handle, int(flavour), cpp, ipp, int(n_iterations), dptr(zz)),
                   "gingr_fitter_update_rccl_async")
            return
        assert self.world > 1, "a single shard uses the plain entry points (gingr_amd.api)"
        cb = self._reduce_callback()
        rc = self._lib.gingr_fitter_update_sharded_async(self.handle, int(flavour), cpp, ipp, int(n_iterations), dptr(zz), cb, None)
        if self._cb_error is not None:
            err, self._cb_error = self._cb_error, None
            raise err
        _check(self.ctx.handle, rc, "gingr_fitter_update_sharded_async")

    def posterior_logpdf(self, flavour: int, params, mesh_full) -> float:
        """posterior(state).gp.logpdf(posterior.coefficients(mesh)); mesh_full = the FULL mesh [M_total, 3] on every rank."""
        cpp, ipp, _keep = self._params(flavour, params)
        m = f64(mesh_full)
        out = ctypes.c_double()
        if self.rccl:
            _check(self.ctx.handle, self._lib.gingr_fitter_posterior_logpdf_rccl(self.handle, int(flavour), cpp, ipp, dptr(m), ctypes.byref(out)),
                   "gingr_fitter_posterior_logpdf_rccl")
            return float(out.value)
        cb = self._reduce_callback()
        rc = self._lib.gingr_fitter_posterior_logpdf_sharded(self.handle, int(flavour), cpp, ipp, dptr(m), cb, None, ctypes.byref(out))
        if self._cb_error is not None:
            err, self._cb_error = self._cb_error, None
            raise err
        _check(self.ctx.handle, rc, "gingr_fitter_posterior_logpdf_sharded")
        return float(out.value)

    def update_cpd_by_phases(self, w: float = 0.0, lambda_: float = 1.0, n_iterations: int = 1):
        """The same iterations driven phase by phase from the host (three library calls + two collectives per iteration): the
        protocol spelled out, kept for tests and for hosts without callbacks."""
        p = nat.CpdParams(w, lambda_)
        for _ in range(n_iterations):
            drive_update(
                lambda ph: _check(self.ctx.handle, self._lib.gingr_fitter_cpd_phase_async(self.handle, ctypes.byref(p), ph),
                                  "gingr_fitter_cpd_phase_async"),
                self._exchange, self.world)

    def close(self):
        if getattr(self, "handle", None):
            if getattr(self.ctx, "handle", None):   # (never into a context that has been closed)
                self._lib.gingr_fitter_destroy(self.handle)
            self.handle = None
        if getattr(self, "dev_model", None) is not None:
            self.dev_model.close()
            self.dev_model = None
