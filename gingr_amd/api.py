"""Host-side mirror of GiNGR's plugin surface for the MI355X update path.

The reference's host language is Scala (no JVM exists in this image, see INTEGRATION.md for the Scala/JNI binding);
this module is the Python host layer over the same C ABI.  It keeps the reference's names, argument meaning and
error behaviour so that tests read like uses of the reference API:

  GingrConfig / GingrRegistrationState / GingrAlgorithm         G/api/GingrAlgorithm.scala:52-75,192-258
  GeneralRegistrationState, ModelFittingParameters              G/api/GeneralRegistrationState.scala:28-41,
                                                                G/api/ModelFittingParameters.scala:31-57
  CpdConfiguration / CpdRegistrationState / CpdRegistration     G/api/registration/config/CPD.scala:52-155
  IcpConfiguration / IcpRegistrationState / IcpRegistration     G/api/registration/config/ICP.scala:54-107
  GlobalTranformationType, FittingStatuses                      G/api/GlobalTranformationType.scala:20-24,
                                                                G/api/FittingStatuses.scala:22

(G/ = src/main/scala/gingr/ of unibas-gravis/GiNGR.)  All arithmetic runs in libgingr_hip.so on the GPU; nothing
here computes on the CPU and there is no fallback.
"""
from __future__ import annotations

import ctypes
import dataclasses
from ctypes import c_int64, c_void_p
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from . import _native as nat
from ._native import GingrNativeError, f64, dptr, iptr


# ----------------------------------------------------------------------------- enums
class GlobalTranformationType:  # sic: the reference spells it this way
    NoTransforms = 0
    RigidTransforms = 1
    SimilarityTransforms = 2


class FittingStatuses:
    None_ = 0
    Converged = 1
    MaxIteration = 2
    ModelFlexibilityError = 3


def _check(ctx_handle, code: int, where: str):
    if code != nat.GINGR_OK:
        text = ""
        if ctx_handle:
            text = (nat.load().gingr_last_error(ctx_handle) or b"").decode("utf-8", "replace")
        raise GingrNativeError(code, where, text)


# ----------------------------------------------------------------------------- context + operators
class Context:
    """One device + one stream (gingr_ctx).  Not thread-safe; one per chain / per GPU."""

    def __init__(self, device: int = 0):
        self._lib = nat.load()
        h = c_void_p()
        code = self._lib.gingr_ctx_create(int(device), ctypes.byref(h))
        if code != nat.GINGR_OK:
            raise GingrNativeError(code, "gingr_ctx_create", "(no usable GPU: this package has no CPU path)")
        self.handle = h
        self.device = device

    def close(self):
        if getattr(self, "handle", None):
            self._lib.gingr_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _check(self.handle, self._lib.gingr_ctx_synchronize(self.handle), "gingr_ctx_synchronize")

    def set_stream(self, hip_stream: Optional[int]):
        _check(self.handle, self._lib.gingr_ctx_set_stream(self.handle, c_void_p(hip_stream or 0)), "gingr_ctx_set_stream")

    # timing hooks (bench.py roofline)
    def get_stream(self) -> int:
        """The hipStream_t (as an integer) the context's kernels are enqueued on."""
        return int(self._lib.gingr_ctx_get_stream(self.handle) or 0)

    def timing_enable(self, on: bool = True):
        _check(self.handle, self._lib.gingr_ctx_timing_enable(self.handle, 1 if on else 0), "timing_enable")

    def timing_reset(self):
        _check(self.handle, self._lib.gingr_ctx_timing_reset(self.handle), "timing_reset")

    def timing_read(self, which: int) -> Tuple[float, int]:
        ms = ctypes.c_double()
        n = c_int64()
        _check(self.handle, self._lib.gingr_ctx_timing_read(self.handle, which, ctypes.byref(ms), ctypes.byref(n)), "timing_read")
        return ms.value, n.value

    def set_option(self, option: int, value: int):
        """gingr_ctx_set_option: nat.OPT_CULL / OPT_FINE_CULL / OPT_NN_GRID -- code paths with identical results (tests, A/B timing)."""
        _check(self.handle, self._lib.gingr_ctx_set_option(self.handle, int(option), int(value)), "gingr_ctx_set_option")

    def get_option(self, option: int) -> int:
        v = ctypes.c_int32()
        _check(self.handle, self._lib.gingr_ctx_get_option(self.handle, int(option), ctypes.byref(v)), "gingr_ctx_get_option")
        return int(v.value)

    # ---- native RCCL exchange (one process per GPU; the library enqueues ncclAllReduce on this context's stream)
    def rccl_load(self, path: Optional[str] = None):
        """Bind a specific librccl (default: the copy already loaded in the process, e.g. torch's, else the loader's search path)."""
        _check(self.handle, self._lib.gingr_rccl_load(self.handle, path.encode() if path else None), "gingr_rccl_load")

    def rccl_unique_id(self) -> bytes:
        """ncclGetUniqueId: rank 0 creates it, the host distributes the 128 bytes to every rank."""
        buf = ctypes.create_string_buffer(nat.RCCL_UNIQUE_ID_BYTES)
        _check(self.handle, self._lib.gingr_rccl_unique_id(self.handle, buf), "gingr_rccl_unique_id")
        return bytes(buf.raw)

    def rccl_init(self, unique_id: bytes, world: int, rank: int):
        """ncclCommInitRank on this context's device (collective over the `world` ranks); the communicator dies with the context."""
        assert len(unique_id) == nat.RCCL_UNIQUE_ID_BYTES
        buf = ctypes.create_string_buffer(unique_id, nat.RCCL_UNIQUE_ID_BYTES)
        _check(self.handle, self._lib.gingr_ctx_rccl_init(self.handle, buf, int(world), int(rank)), "gingr_ctx_rccl_init")

    def rccl_info(self) -> dict:
        w, r, v = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        path = ctypes.create_string_buffer(512)
        _check(self.handle, self._lib.gingr_ctx_rccl_info(self.handle, ctypes.byref(w), ctypes.byref(r), ctypes.byref(v), path, 512),
               "gingr_ctx_rccl_info")
        return {"world": int(w.value), "rank": int(r.value), "version": int(v.value), "library": path.value.decode()}

    def rccl_allreduce(self, device_ptr: int, count: int):
        """In-place float64 sum all-reduce of `count` elements at `device_ptr`, enqueued on this context's stream."""
        _check(self.handle, self._lib.gingr_ctx_rccl_allreduce_async(self.handle, c_void_p(int(device_ptr)), int(count)),
               "gingr_ctx_rccl_allreduce_async")

    def nn_counting(self, on: bool = True):
        """Diagnostics: count the distance tests the nearest-neighbour launches of this context really execute (clears the counter)."""
        _check(self.handle, self._lib.gingr_ctx_nn_counting(self.handle, 1 if on else 0), "nn_counting")

    def nn_tests(self) -> int:
        n = c_int64()
        _check(self.handle, self._lib.gingr_ctx_nn_tests(self.handle, ctypes.byref(n)), "nn_tests")
        return int(n.value)

    # ---- stateless all-pairs operators -------------------------------------------------
    def cpd_stats(self, fit, target, sigma2: float, w: float = 0.0) -> dict:
        """CPD statistics of one affinity evaluation (CPD.scala:54-75,36,133-147), P never materialised."""
        y, x = f64(fit), f64(target)
        M, N = y.shape[0], x.shape[0]
        den, Pt1 = np.empty(N), np.empty(N)
        P1, PX, sc = np.empty(M), np.empty((M, 3)), np.empty(6)
        _check(self.handle, self._lib.gingr_cpd_stats(self.handle, M, dptr(y), N, dptr(x), float(sigma2), float(w),
                                                      dptr(den), dptr(P1), dptr(PX), dptr(Pt1), dptr(sc)), "gingr_cpd_stats")
        return dict(den=den, P1=P1, PX=PX, Pt1=Pt1, Np=float(sc[0]), xPx=float(sc[1]), trPXY=float(sc[2]),
                    yPy=float(sc[3]), sigma2_next=float(sc[4]), c=float(sc[5]))

    def cpd_initial_sigma2(self, reference, target) -> float:
        y, x = f64(reference), f64(target)
        out = ctypes.c_double()
        _check(self.handle, self._lib.gingr_cpd_initial_sigma2(self.handle, y.shape[0], dptr(y), x.shape[0], dptr(x),
                                                               ctypes.byref(out)), "gingr_cpd_initial_sigma2")
        return out.value

    def nn(self, query, target) -> Tuple[np.ndarray, np.ndarray, float]:
        """Exact nearest neighbour, lowest index on ties (ClosestPointRegistrator.scala:133-148)."""
        y, x = f64(query), f64(target)
        idx = np.empty(y.shape[0], dtype=np.int32)
        d2 = np.empty(y.shape[0])
        md = ctypes.c_double()
        _check(self.handle, self._lib.gingr_nn(self.handle, y.shape[0], dptr(y), x.shape[0], dptr(x), iptr(idx), dptr(d2),
                                               ctypes.byref(md)), "gingr_nn")
        return idx, d2, md.value

    def gauss_block(self, A, B, sigma: float, scaling: float) -> np.ndarray:
        """scaling * exp(-|a-b|^2 / sigma^2)  (GPMMHelper.scala:99-102)."""
        A, B = f64(A), f64(B)
        out = np.empty((A.shape[0], B.shape[0]))
        _check(self.handle, self._lib.gingr_gauss_block(self.handle, A.shape[0], dptr(A), B.shape[0], dptr(B), float(sigma),
                                                        float(scaling), dptr(out)), "gingr_gauss_block")
        return out

    def mesh_distance_stats(self, points, vertices, cells, boundary_aware: bool = False, sdev: float = 0.0
                            ) -> Tuple[float, float, int, float]:
        """(sum, max, count, sum of log N(d; 0, sdev)) of d = |p - closestPointOnSurface(p)| for the rows of `points` against
        the triangle mesh (vertices, cells); boundary_aware as RegistrationComparison.avgDistanceBoundaryAware (:63-73)."""
        p, v = f64(points), f64(vertices)
        c = np.ascontiguousarray(cells, dtype=np.int32).reshape(-1, 3)
        out = np.zeros(4)
        _check(self.handle, self._lib.gingr_mesh_distance_stats(self.handle, p.shape[0], dptr(p), v.shape[0], dptr(v), c.shape[0],
                                                                iptr(c), int(bool(boundary_aware)), float(sdev), dptr(out)),
               "gingr_mesh_distance_stats")
        return float(out[0]), float(out[1]), int(out[2]), float(out[3])

    def mesh_closest_points(self, points, vertices, cells) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
        """mesh.operations.closestPointOnSurface for every row of `points`: (closest points (n,3), squared distances (n,), triangle
        ids (n,), barycentric weights of that triangle's corners (n,3)) -- gingr_mesh_closest_points."""
        p, v = f64(points).reshape(-1, 3), f64(vertices).reshape(-1, 3)
        c = np.ascontiguousarray(cells, dtype=np.int32).reshape(-1, 3)
        n = p.shape[0]
        cp, d2, tid, bary = np.empty((n, 3)), np.empty(n), np.empty(n, dtype=np.int32), np.empty((n, 3))
        _check(self.handle, self._lib.gingr_mesh_closest_points(self.handle, n, dptr(p), v.shape[0], dptr(v), c.shape[0], iptr(c),
                                                                dptr(cp), dptr(d2), iptr(tid), dptr(bary)),
               "gingr_mesh_closest_points")
        return cp, d2, tid, bary


# ----------------------------------------------------------------------------- model
@dataclasses.dataclass
class PointDistributionModel:
    """Numeric content of scalismo's PointDistributionModel: reference points, mean displacement, basisMatrix
    (3M x r, element (row, k) = U[row, k]) and variance."""
    reference: np.ndarray      # (M,3)
    mean: np.ndarray           # (M,3) displacement
    basis: np.ndarray          # (3M, r)
    variance: np.ndarray       # (r,)
    cells: Optional[np.ndarray] = None   # (T,3) int: triangulation of the reference mesh (TriangleMesh.triangulation)

    @property
    def rank(self) -> int:
        return int(self.variance.shape[0])

    @property
    def numberOfPoints(self) -> int:
        return int(self.reference.shape[0])


@dataclasses.dataclass(frozen=True)
class ScalarKernelSpec:
    """One scalar kernel of a DiagonalKernel (gingr_scalar_kernel): `kind` "gauss" (mixture of scaling_i exp(-|x-y|^2/sigma_i^2),
    optionally + mirror * the x-mirrored mixture), "dot" (scaling * x.y) or "lookup" (scaling * lookup[i, j])."""
    kind: str
    sigmas: Tuple[float, ...] = ()
    scalings: Tuple[float, ...] = ()
    mirror: float = 0.0
    scaling: float = 1.0
    lookup: Optional[np.ndarray] = None

    def _native(self, keep: list) -> "nat.ScalarKernel":
        k = nat.ScalarKernel()
        k.kind = {"gauss": nat.KERNEL_GAUSSIAN_MIXTURE, "dot": nat.KERNEL_DOT, "lookup": nat.KERNEL_LOOKUP}[self.kind]
        if self.kind == "gauss":
            sg, sc = f64(self.sigmas), f64(self.scalings)
            keep += [sg, sc]
            k.n_kernels, k.sigmas, k.scalings, k.mirror = len(self.sigmas), dptr(sg), dptr(sc), float(self.mirror)
        else:
            k.scaling = float(self.scaling)
            if self.kind == "lookup":
                m = f64(self.lookup)
                keep.append(m)
                k.lookup = dptr(m)
        return k


class DevicePointDistributionModel:
    """A PointDistributionModel whose basis lives in HBM: the result of the on-device GPMM construction
    (gingr_gpmm_build_gaussian / gingr_gpmm_build_diagonal).  Quacks like PointDistributionModel (reference / mean / basis /
    variance / rank / numberOfPoints); basis and variance are downloaded lazily, only if somebody asks for them."""

    def __init__(self, ctx: Context, reference, sigmas: Sequence[float], scalings: Sequence[float],
                 relativeTolerance: float, maxRank: int = 0, kernels: Optional[Sequence[ScalarKernelSpec]] = None,
                 cells: Optional[np.ndarray] = None):
        self.ctx = ctx
        self.reference = f64(reference)
        self._sig, self._sc = f64(sigmas), f64(scalings)
        self._tol, self._maxrank = float(relativeTolerance), int(maxRank)
        self._kernels = None if kernels is None else tuple(kernels)          # (k_x, k_y, k_z) of a DiagonalKernel
        self.cells = cells
        self._full: Optional["DeviceModel"] = None
        self._host: Optional[PointDistributionModel] = None

    def _build(self, ctx: Context, row_begin: int, row_end: int):
        h = c_void_p()
        if self._kernels is not None:
            keep: list = []
            uniq = {}
            nk = []
            for spec in self._kernels:                     # equal specs -> the SAME native struct (one factorisation)
                if id(spec) not in uniq:
                    uniq[id(spec)] = spec._native(keep)
                nk.append(uniq[id(spec)])
            _check(ctx.handle, ctx._lib.gingr_gpmm_build_diagonal(ctx.handle, self.numberOfPoints, dptr(self.reference),
                                                                  ctypes.byref(nk[0]), ctypes.byref(nk[1]), ctypes.byref(nk[2]),
                                                                  self._tol, self._maxrank, row_begin, row_end, ctypes.byref(h)),
                   "gingr_gpmm_build_diagonal")
            return h
        _check(ctx.handle, ctx._lib.gingr_gpmm_build_gaussian(ctx.handle, self.numberOfPoints, dptr(self.reference),
                                                              len(self._sig), dptr(self._sig), dptr(self._sc), self._tol,
                                                              self._maxrank, row_begin, row_end, ctypes.byref(h)),
               "gingr_gpmm_build_gaussian")
        return h

    def device(self) -> "DeviceModel":
        if self._full is None:
            M = self.numberOfPoints
            self._full = DeviceModel._adopt(self.ctx, self._build(self.ctx, 0, M), self, M)
        return self._full

    @property
    def numberOfPoints(self) -> int:
        return int(self.reference.shape[0])

    @property
    def rank(self) -> int:
        return self.device().rank

    @property
    def mean(self) -> np.ndarray:
        return np.zeros_like(self.reference)       # GaussianProcess(kernel): zero mean (GPMMHelper.scala:45)

    def to_host(self, basis: bool = True) -> PointDistributionModel:
        if self._host is None or (basis and self._host.basis is None):
            self._host = self.device().download(basis=basis)
        return self._host

    @property
    def variance(self) -> np.ndarray:
        return self.to_host(basis=False).variance

    @property
    def basis(self) -> np.ndarray:
        return self.to_host(basis=True).basis


class InterpolatedDevicePointDistributionModel(DevicePointDistributionModel):
    """model.newReference(newReference, interpolator) with the basis gathered in HBM (gingr_model_new_reference): every new point
    takes the fixed combination sum_k weights[i, k] * (value at source point vertex_ids[i, k]) of mean and basis functions;
    eigenvalues and rank are the source's."""

    def __init__(self, ctx: Context, source, new_reference, vertex_ids, weights, cells=None):
        self.ctx = ctx
        self.reference = f64(new_reference).reshape(-1, 3)
        self._ids = np.ascontiguousarray(vertex_ids, dtype=np.int32).reshape(-1, 3)
        self._w = f64(weights).reshape(-1, 3)
        if self._ids.shape[0] != self.reference.shape[0] or self._w.shape[0] != self.reference.shape[0]:
            raise ValueError("one (vertex_ids, weights) triple per new reference point")
        self._source = source
        # the source model resident on this context (a device-built model shares its handle; a host model is uploaded once)
        self._src_dev = DeviceModel(ctx, source)
        self.cells = None if cells is None else np.ascontiguousarray(cells, dtype=np.int32)
        self._kernels, self._full, self._host = None, None, None

    def _build(self, ctx: Context, row_begin: int, row_end: int):
        if ctx is not self.ctx:
            raise ValueError("an interpolated model lives on the context of its source; download it (to_host) for another device")
        h = c_void_p()
        _check(ctx.handle, ctx._lib.gingr_model_new_reference(ctx.handle, self._src_dev.handle, self.numberOfPoints, dptr(self.reference),
                                                              iptr(self._ids), dptr(self._w), row_begin, row_end, ctypes.byref(h)),
               "gingr_model_new_reference")
        return h

    @property
    def mean(self) -> np.ndarray:
        return self.to_host(basis=False).mean


@dataclasses.dataclass
class GaussianKernelParameters:
    """GPMMHelper.scala:94"""
    sigma: float
    scaling: float


class PointSetHelper:
    """GPMMHelper.scala:73-91: the O(n^2) scans behind the automatic kernel parameters, on the device."""

    def __init__(self, ctx: Context, reference):
        self.ctx, self.reference = ctx, f64(reference)
        self._ext: Optional[Tuple[float, float]] = None

    def _extrema(self) -> Tuple[float, float]:
        if self._ext is None:
            mx, mn = ctypes.c_double(), ctypes.c_double()
            _check(self.ctx.handle, self.ctx._lib.gingr_pointset_distance_extrema(
                self.ctx.handle, dptr(self.reference), self.reference.shape[0], ctypes.byref(mx), ctypes.byref(mn)),
                "gingr_pointset_distance_extrema")
            self._ext = (mx.value, mn.value)
        return self._ext

    def maximumPointDistance(self) -> float:
        return self._extrema()[0]

    def minimumPointDistance(self) -> float:
        return self._extrema()[1]


class GPMMTriangleMesh3D:
    """GPMMTriangleMesh3D(reference, relativeTolerance) (GPMMHelper.scala:96-153): every kernel of the reference, the low-rank
    factorisation built in HBM.  `cells` (the triangulation) is only needed by the Laplacian kernels; it is handed on to the model."""

    def __init__(self, ctx: Context, reference, relativeTolerance: float = 0.01, maxRank: int = 0, cells=None):
        self.ctx, self.reference, self.relativeTolerance, self.maxRank = ctx, f64(reference), relativeTolerance, maxRank
        self.cells = None if cells is None else np.ascontiguousarray(cells, dtype=np.int32)

    def Gaussian(self, sigma: float, scaling: float) -> DevicePointDistributionModel:
        return self.GaussianMixture([GaussianKernelParameters(sigma, scaling)])

    def GaussianMixture(self, pars: Sequence[GaussianKernelParameters]) -> DevicePointDistributionModel:
        return DevicePointDistributionModel(self.ctx, self.reference, [p.sigma for p in pars], [p.scaling for p in pars],
                                            self.relativeTolerance, self.maxRank, cells=self.cells)

    def AutomaticGaussian(self) -> DevicePointDistributionModel:
        mx = PointSetHelper(self.ctx, self.reference).maximumPointDistance()
        return self.GaussianMixture([GaussianKernelParameters(mx / 4.0, mx / 8.0), GaussianKernelParameters(mx / 8.0, mx / 16.0)])

    def _diagonal(self, kx: ScalarKernelSpec, ky: ScalarKernelSpec, kz: ScalarKernelSpec) -> DevicePointDistributionModel:
        return DevicePointDistributionModel(self.ctx, self.reference, [], [], self.relativeTolerance, self.maxRank,
                                            kernels=(kx, ky, kz), cells=self.cells)

    def GaussianDot(self, sigma: float, scaling: float) -> DevicePointDistributionModel:
        """GPMMHelper.scala:103-106: DotProductKernel(GaussianKernel(sigma), 1.0) * scaling.  DotProductKernel.k returns
        x.dot(y) and ignores both the wrapped kernel and gamma (KernelHelper.scala:43-51), so sigma has no effect -- mirrored."""
        k = ScalarKernelSpec("dot", scaling=scaling)
        return self._diagonal(k, k, k)

    def GaussianSymmetry(self, sigma: float, scaling: float) -> DevicePointDistributionModel:
        """GPMMHelper.scala:107-111 + KernelHelper.symmetrizeKernel (:25-31): DiagonalKernel(k, 3) + DiagonalKernel(-km, km, km)
        with km(x, y) = k((-x0, x1, x2), y): the x coordinate gets k - km, y and z get k + km."""
        kx = ScalarKernelSpec("gauss", (float(sigma),), (float(scaling),), mirror=-1.0)
        kyz = ScalarKernelSpec("gauss", (float(sigma),), (float(scaling),), mirror=1.0)
        return self._diagonal(kx, kyz, kyz)

    def InverseLaplacian(self, scaling: float) -> DevicePointDistributionModel:
        """GPMMHelper.scala:132-136: LookupKernel(reference, pinv(graph Laplacian)) * scaling.  The dense pseudo-inverse is one-off
        host work exactly as in the reference (LaplacianHelper, MatrixHelper.pinv); the low-rank factorisation runs on the device."""
        k = ScalarKernelSpec("lookup", scaling=scaling, lookup=LaplacianHelper(self.reference.shape[0], self._need_cells()).inverseLaplacianMatrix())
        return self._diagonal(k, k, k)

    def InverseLaplacianDot(self, scaling: float, gamma: float) -> DevicePointDistributionModel:
        """GPMMHelper.scala:138-142: DotProductKernel(LookupKernel(..), gamma) * scaling = scaling * x.y (see GaussianDot); the
        reference still computes the pseudo-inverse it never uses -- skipped."""
        return self.GaussianDot(0.0, scaling)

    def _need_cells(self) -> np.ndarray:
        if self.cells is None:
            raise ValueError("the Laplacian kernels need the triangulation: GPMMTriangleMesh3D(ctx, reference, tol, cells=...)")
        return self.cells

    def computeDistanceAbsMesh(self, model, lmId: int) -> np.ndarray:
        """GPMMHelper.scala:144-153: sum_i |cov(lmId, pid)_ii| per vertex, cov = U diag(variance) U^T (3 x 3 blocks)."""
        U = f64(model.basis).reshape(model.numberOfPoints, 3, -1)
        w = U[int(lmId)] * f64(model.variance)[None, :]                   # (3, r)
        return np.abs(np.einsum("pdr,dr->pd", U, w)).sum(axis=1)


class LaplacianHelper:
    """G/api/gpmm/LaplacianHelper.scala:25-55 and MatrixHelper.pinv (MatrixHelper.scala:21-26): combinatorial graph Laplacian of
    the triangulation (degree on the diagonal, -1 for adjacent vertices) and its SVD pseudo-inverse with the 1e-5 cut-off."""

    def __init__(self, n: int, cells):
        self.n = int(n)
        c = np.asarray(cells, dtype=np.int64).reshape(-1, 3)
        a = np.zeros((self.n, self.n), dtype=bool)
        for i, j in ((0, 1), (1, 2), (0, 2)):
            a[c[:, i], c[:, j]] = True
            a[c[:, j], c[:, i]] = True
        np.fill_diagonal(a, False)
        self.adjacency = a

    def laplacianMatrix(self, inverse: bool = False) -> np.ndarray:
        m = np.where(self.adjacency, -1.0, 0.0)
        m[np.arange(self.n), np.arange(self.n)] = self.adjacency.sum(axis=1).astype(np.float64)
        return self.pinv(m) if inverse else m

    def inverseLaplacianMatrix(self) -> np.ndarray:
        return self.pinv(self.laplacianMatrix())

    def laplacianNormalizedMatrix(self, inverse: bool = False) -> np.ndarray:
        deg = self.adjacency.sum(axis=1).astype(np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            m = np.where(self.adjacency, -1.0 / np.sqrt(deg[:, None] * deg[None, :]), 0.0)
        m[np.arange(self.n), np.arange(self.n)] = (deg != 0).astype(np.float64)
        return self.pinv(m) if inverse else m

    @staticmethod
    def pinv(m: np.ndarray, precision: float = 0.00001) -> np.ndarray:
        u, s, vt = np.linalg.svd(m)
        return (u * np.where(s > precision, 1.0 / np.where(s > precision, s, 1.0), 0.0)[None, :]) @ vt


def automaticGPMMfromTemplate(ctx: Context, template, relativeTolerance: float = 0.1) -> DevicePointDistributionModel:
    """registration/utils/GPMMHelper.scala:39-69 (on the model's own points the TriangleMeshInterpolator is the identity)."""
    h = PointSetHelper(ctx, template)
    mx, mn = h.maximumPointDistance(), h.minimumPointDistance()
    sig = [mx / 4, mx / 8, mn * 5]
    return DevicePointDistributionModel(ctx, template, sig, [v / 2 for v in sig], relativeTolerance)


class DeviceModel:
    """gingr_model: the (row shard of the) model resident in HBM.  `model` is either a host PointDistributionModel
    (uploaded) or a DevicePointDistributionModel (built on the device; full-row handles are shared with it)."""

    def __init__(self, ctx: Context, model, row_begin: int = 0, row_end: Optional[int] = None):
        self.ctx = ctx
        self._lib = ctx._lib
        self.host = model
        M = model.numberOfPoints
        row_end = M if row_end is None else row_end
        self._owner = True
        if isinstance(model, DevicePointDistributionModel):
            if int(row_begin) == 0 and int(row_end) == M and model.ctx is ctx:
                self.handle = model.device().handle     # shared, owned by the model
                self._owner = False
            else:
                self.handle = model._build(ctx, int(row_begin), int(row_end))
        else:
            ref, mean = f64(model.reference), f64(model.mean)
            basis = np.asfortranarray(model.basis, dtype=np.float64)   # column-major, as Breeze stores basisMatrix
            var = f64(model.variance)
            h = c_void_p()
            _check(ctx.handle, self._lib.gingr_model_upload(ctx.handle, M, model.rank, dptr(ref), dptr(mean),
                                                            basis.ctypes.data_as(nat._dp), dptr(var), int(row_begin),
                                                            int(row_end), ctypes.byref(h)), "gingr_model_upload")
            self.handle = h
        self.row_begin, self.row_end = int(row_begin), int(row_end)
        self.M_local = self.row_end - self.row_begin
        self.rank = int(self._lib.gingr_model_rank(self.handle))

    @classmethod
    def _adopt(cls, ctx: Context, handle, model, M: int) -> "DeviceModel":
        self = cls.__new__(cls)
        self.ctx, self._lib, self.host, self.handle, self._owner = ctx, ctx._lib, model, handle, True
        self.row_begin, self.row_end, self.M_local = 0, M, M
        self.rank = int(self._lib.gingr_model_rank(handle))
        return self

    def download(self, basis: bool = True) -> "PointDistributionModel":
        """Local rows back on the host in gingr_model_upload's layout (gingr_model_download)."""
        M, r = self.M_local, self.rank
        ref, mean, var = np.empty((M, 3)), np.empty((M, 3)), np.empty(r)
        U = np.empty((3 * M, r), order="F") if basis else None
        _check(self.ctx.handle, self._lib.gingr_model_download(self.ctx.handle, self.handle, dptr(ref), dptr(mean),
                                                               U.ctypes.data_as(nat._dp) if basis else None, dptr(var)),
               "gingr_model_download")
        return PointDistributionModel(reference=ref, mean=mean, basis=U, variance=var)

    def gram_exchange(self) -> Tuple[int, int]:
        p, n = c_void_p(), c_int64()
        _check(self.ctx.handle, self._lib.gingr_model_gram_exchange(self.handle, ctypes.byref(p), ctypes.byref(n)), "gram_exchange")
        return p.value, n.value

    def finalize(self):
        _check(self.ctx.handle, self._lib.gingr_model_finalize(self.ctx.handle, self.handle), "gingr_model_finalize")

    def close(self):
        if getattr(self, "handle", None):
            # (a model that outlives its context -- e.g. collected after Context.close() -- must not call into the freed context: its
            # device memory went with the context's process state; dropping the handle is all that is left to do)
            if self._owner and getattr(self.ctx, "handle", None):
                self._lib.gingr_model_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- stateless model operators ---------------------------------------------------------
    def instance(self, alpha, euler=(0, 0, 0), center=(0, 0, 0), translation=(0, 0, 0), scale: float = 1.0) -> np.ndarray:
        a, e, c, t = f64(alpha), f64(euler), f64(center), f64(translation)
        out = np.empty((self.M_local, 3))
        _check(self.ctx.handle, self._lib.gingr_model_instance(self.ctx.handle, self.handle, dptr(a), dptr(e), dptr(c), dptr(t),
                                                               float(scale), dptr(out)), "gingr_model_instance")
        return out

    def coefficients(self, mesh, euler=(0, 0, 0), center=(0, 0, 0), translation=(0, 0, 0)) -> np.ndarray:
        m, e, c, t = f64(mesh), f64(euler), f64(center), f64(translation)
        out = np.empty(self.rank)
        _check(self.ctx.handle, self._lib.gingr_model_coefficients(self.ctx.handle, self.handle, dptr(e), dptr(c), dptr(t),
                                                                   dptr(m), dptr(out)), "gingr_model_coefficients")
        return out

    def posterior_mean(self, obs_points, weights, euler=(0, 0, 0), center=(0, 0, 0), translation=(0, 0, 0),
                       landmarks: Optional["LandmarkCorrespondences"] = None) -> Tuple[np.ndarray, np.ndarray]:
        o, w, e, c, t = f64(obs_points), f64(weights), f64(euler), f64(center), f64(translation)
        mean = np.empty((self.M_local, 3))
        coeffs = np.empty(self.rank)
        if landmarks is not None and len(landmarks.pids) > 0:
            lp = np.ascontiguousarray(landmarks.pids, dtype=np.int32)
            lx, lc = f64(landmarks.points), f64(landmarks.covs)
            n = lp.shape[0]
        else:
            lp = lx = lc = None
            n = 0
        _check(self.ctx.handle, self._lib.gingr_model_posterior_mean(
            self.ctx.handle, self.handle, dptr(e), dptr(c), dptr(t), dptr(o), dptr(w), n, iptr(lp), dptr(lx), dptr(lc),
            dptr(mean), dptr(coeffs)), "gingr_model_posterior_mean")
        return mean, coeffs


# ----------------------------------------------------------------------------- state records
@dataclasses.dataclass(frozen=True)
class EulerAngles:
    phi: float = 0.0
    theta: float = 0.0
    psi: float = 0.0


@dataclasses.dataclass(frozen=True)
class ModelFittingParameters:
    """scale, pose = (translation, Euler rotation about a centre), shape  (ModelFittingParameters.scala:31-57)."""
    scale: float
    translation: Tuple[float, float, float]
    rotation: EulerAngles
    center: Tuple[float, float, float]
    shape: np.ndarray

    @staticmethod
    def zero(rank: int) -> "ModelFittingParameters":
        return ModelFittingParameters(1.0, (0.0, 0.0, 0.0), EulerAngles(), (0.0, 0.0, 0.0), np.zeros(rank))


@dataclasses.dataclass
class LandmarkCorrespondences:
    """(pid, target point, 3x3 covariance) triples (GeneralRegistrationState.scala:43-62)."""
    pids: np.ndarray
    points: np.ndarray
    covs: np.ndarray


@dataclasses.dataclass(frozen=True)
class CorrespondencePairs:
    """(PointId, Point) pairs (G/api/CorrespondencePairs.scala:23)."""
    pids: np.ndarray
    points: np.ndarray


@dataclasses.dataclass(frozen=True)
class GeneralRegistrationState:
    model: PointDistributionModel
    modelParameters: ModelFittingParameters
    target: np.ndarray
    fit: np.ndarray
    sigma2: float = 1.0
    globalTransformation: int = GlobalTranformationType.RigidTransforms
    stepLength: float = 1.0
    generatedBy: str = ""
    iteration: int = 0
    status: int = FittingStatuses.None_
    landmarkCorrespondences: Optional[LandmarkCorrespondences] = None
    targetCells: Optional[np.ndarray] = None   # (T,3) int: triangulation of the target mesh (surface ICP only)

    def updateStatus(self, status: int) -> "GeneralRegistrationState":
        return dataclasses.replace(self, status=status)

    def updateSigma2(self, s2: float) -> "GeneralRegistrationState":
        return dataclasses.replace(self, sigma2=s2)

    def statusText(self) -> str:
        """GeneralRegistrationState.printStatus (GeneralRegistrationState.scala:103-114)"""
        n = self.iteration + 1
        return {FittingStatuses.None_: "Initial state - no iterations performed!",
                FittingStatuses.Converged: f"Fitting converged after {n} accepted iterations!",
                FittingStatuses.MaxIteration: f"Fitting finished the MaxIterations with ({n}) accepted iterations!",
                FittingStatuses.ModelFlexibilityError:
                    f"Model not flexible enough to compute posterior model - finished after {n} accepted iterations!"}[self.status]

    def printStatus(self) -> None:
        print(self.statusText())


# ----------------------------------------------------------------------------- configs
def _cpd_converged(last: GeneralRegistrationState, current: GeneralRegistrationState, threshold: float) -> bool:
    return abs(last.sigma2 - current.sigma2) < threshold          # CPD.scala:108-110


def _never_converged(last, current, threshold) -> bool:
    return False                                                   # ICP.scala:57-58


@dataclasses.dataclass(frozen=True)
class CpdConfiguration:
    maxIterations: int = 100
    threshold: float = 1e-10
    converged: Callable = _cpd_converged
    useLandmarkCorrespondence: bool = True
    initialSigma: Optional[float] = None
    w: float = 0.0
    lambda_: float = 1.0


@dataclasses.dataclass(frozen=True)
class IcpConfiguration:
    maxIterations: int = 100
    threshold: float = 1e-10
    converged: Callable = _never_converged
    useLandmarkCorrespondence: bool = True
    initialSigma: float = 100.0
    endSigma: float = 1.0
    reverseCorrespondenceDirection: bool = False
    correspondenceMethod: str = "TriangularClosestPoint"    # the reference's default (ICP.scala:63); needs both triangulations

    @property
    def sigmaStep(self) -> float:
        return (self.initialSigma - self.endSigma) / float(self.maxIterations)


@dataclasses.dataclass(frozen=True)
class CpdRegistrationState:
    general: GeneralRegistrationState
    config: CpdConfiguration

    def updateGeneral(self, update: GeneralRegistrationState) -> "CpdRegistrationState":
        return dataclasses.replace(self, general=update)


@dataclasses.dataclass(frozen=True)
class IcpRegistrationState:
    general: GeneralRegistrationState
    config: IcpConfiguration

    def updateGeneral(self, update: GeneralRegistrationState) -> "IcpRegistrationState":
        return dataclasses.replace(self, general=update)


def _with_fields(obj, **changes):
    """dataclasses.replace for the plain frozen dataclasses above (no __post_init__, no slots), without its per-field machinery: 1 us
    instead of 7 for a GeneralRegistrationState -- the fused Metropolis-Hastings step builds two objects per step."""
    new = object.__new__(type(obj))
    d = new.__dict__
    d.update(obj.__dict__)
    d.update(changes)
    return new


# ----------------------------------------------------------------------------- the algorithm
class GingrAlgorithm:
    """HIP-backed GingrAlgorithm: `update` is ONE native call sequence per iteration
    (GingrAlgorithm.scala:192-254 + GingrGeneratorWrapper.propose).  Sub-classes supply the correspondence flavour."""

    name = "GiNGR"

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self._lib = ctx._lib
        self._dev_model: Optional[DeviceModel] = None
        self._fitter = None
        # The objects currently bound on the device.  Strong references compared with `is` (an id() of a collected object can be
        # reused by a new one of the same shape); arrays are treated as immutable, like the reference's case-class fields.
        self._bound_model = None
        self._bound_target = None
        self._bound_lm = None
        self._bound_mesh = None
        self._device_state = None         # the python state currently mirrored on the device (strong reference, compared with `is`)
        self._mh = None                   # fused Metropolis-Hastings steps: {"sdev", "points"} of the likelihood evaluated with them
        self._mh_last = None              # what the last fused step measured: {"from", "to", "stats", "fw", "bw"}
        self._mh_plan_memo = None         # (config object, what _mh_flavour said about it + ctypes pointers)
        self._mh_req = None               # the request / result structs of the fused step, reused (the call is synchronous)
        self._sel = {}                    # selections already made on the current fitter (options, direction, surface method): set again only when they change

    # -- native plumbing ------------------------------------------------------------------
    def _bind(self, general: GeneralRegistrationState, use_landmarks: bool):
        if self._bound_model is not general.model:
            self._release()
            self._dev_model = DeviceModel(self.ctx, general.model)
            h = c_void_p()
            _check(self.ctx.handle, self._lib.gingr_fitter_create(self.ctx.handle, self._dev_model.handle, ctypes.byref(h)),
                   "gingr_fitter_create")
            self._fitter = h
            self._sel = {}
            self._bound_model = general.model
            self._bound_target = None
            self._bound_lm = None
            self._bound_mesh = None
        if self._bound_target is not general.target:
            x = f64(general.target)
            _check(self.ctx.handle, self._lib.gingr_fitter_set_target(self._fitter, x.shape[0], dptr(x)), "gingr_fitter_set_target")
            self._bound_target = general.target
            self._device_state = None
            self._bound_mesh = None
            self._sel = {}
        mcells = getattr(general.model, "cells", None)
        if mcells is not None and general.targetCells is not None and not (
                self._bound_mesh is not None and self._bound_mesh[0] is mcells and self._bound_mesh[1] is general.targetCells):
            mt = np.ascontiguousarray(mcells, dtype=np.int32).reshape(-1, 3)
            tt = np.ascontiguousarray(general.targetCells, dtype=np.int32).reshape(-1, 3)
            _check(self.ctx.handle, self._lib.gingr_fitter_set_meshes(self._fitter, mt.shape[0], iptr(mt), tt.shape[0], iptr(tt)),
                   "gingr_fitter_set_meshes")
            self._bound_mesh = (mcells, general.targetCells)
            self._sel = {}                # (new meshes: the library starts from the forward direction again)
        lm = general.landmarkCorrespondences if use_landmarks else None
        if not (self._bound_lm is not None and self._bound_lm[0] is lm and self._bound_lm[1] == use_landmarks):
            if lm is not None and len(lm.pids) > 0:
                lp = np.ascontiguousarray(lm.pids, dtype=np.int32)
                lx, lc = f64(lm.points), f64(lm.covs)
                _check(self.ctx.handle, self._lib.gingr_fitter_set_landmarks(self._fitter, lp.shape[0], iptr(lp), dptr(lx), dptr(lc)),
                       "gingr_fitter_set_landmarks")
            else:
                _check(self.ctx.handle, self._lib.gingr_fitter_set_landmarks(self._fitter, 0, None, None, None),
                       "gingr_fitter_set_landmarks")
            self._bound_lm = (lm, use_landmarks)
        opts = (int(general.globalTransformation), float(general.stepLength))
        if self._sel.get("options") != opts:        # (one native call less per Metropolis-Hastings step)
            _check(self.ctx.handle, self._lib.gingr_fitter_set_options(self._fitter, opts[0], opts[1]), "gingr_fitter_set_options")
            self._sel["options"] = opts

    def _release(self):
        if self._fitter:
            if getattr(self.ctx, "handle", None):   # (never into a context that has been closed)
                self._lib.gingr_fitter_destroy(self._fitter)
            self._fitter = None
        if self._dev_model is not None:
            self._dev_model.close()
            self._dev_model = None
        self._bound_model = self._bound_target = self._bound_lm = self._bound_mesh = None
        self._device_state = None
        self._sel = {}

    def close(self):
        self._release()

    @property
    def retryCounter(self) -> int:
        """`retryCounter` of this algorithm instance (GingrAlgorithm.scala:69-70); it lives on the device next to the state."""
        if not self._fitter:
            return 10
        v = ctypes.c_int32()
        _check(self.ctx.handle, self._lib.gingr_fitter_retry_counter(self._fitter, -1, ctypes.byref(v)), "gingr_fitter_retry_counter")
        return int(v.value)

    def _push_state(self, general: GeneralRegistrationState):
        mp = general.modelParameters
        s = nat.StateScalars()
        s.euler[:] = [mp.rotation.phi, mp.rotation.theta, mp.rotation.psi]
        s.center[:] = list(mp.center)
        s.translation[:] = list(mp.translation)
        s.scale = mp.scale
        s.sigma2 = general.sigma2
        s.iteration = general.iteration
        s.status = general.status
        a = f64(mp.shape)
        _check(self.ctx.handle, self._lib.gingr_fitter_set_state(self._fitter, dptr(a), ctypes.byref(s)), "gingr_fitter_set_state")

    def _pull_state(self, general: GeneralRegistrationState) -> GeneralRegistrationState:
        r = general.model.rank
        M = general.model.numberOfPoints
        alpha = np.empty(r)
        fit = np.empty((M, 3))
        s = nat.StateScalars()
        _check(self.ctx.handle, self._lib.gingr_fitter_get_state(self._fitter, dptr(alpha), ctypes.byref(s), dptr(fit)),
               "gingr_fitter_get_state")
        mp = ModelFittingParameters(scale=s.scale, translation=tuple(s.translation),
                                    rotation=EulerAngles(*list(s.euler)), center=tuple(s.center), shape=alpha)
        return dataclasses.replace(general, modelParameters=mp, fit=fit, sigma2=s.sigma2, iteration=s.iteration,
                                   status=s.status, generatedBy=self.name)

    def _native_update(self, current, n: int):
        raise NotImplementedError

    # -- the reference surface ----------------------------------------------------------------
    def initializeState(self, general: GeneralRegistrationState, config):
        raise NotImplementedError

    def update(self, current, probabilistic: bool = False, rnd: Optional[np.random.Generator] = None):
        """One GiNGR iteration.  Failure handling as in the reference's Try(...) logic, decided on the device (post_solve_kernel):
        a failed posterior leaves the state unchanged at iteration 0, gives ModelFlexibilityError in a deterministic update,
        and in a probabilistic one uses up one of the instance's 10 retries (state unchanged) before it gives
        ModelFlexibilityError (:194-208; successes give retries back, :210); a failed coefficient projection is a
        ModelFlexibilityError at any iteration (:248-251).
        probabilistic=True proposes posterior.sample() instead of posterior.mean (:211); the standard-normal draws come from
        `rnd` (the reference's `implicit rnd: Random`)."""
        g = current.general
        if probabilistic and rnd is not None and self._mh_usable(current):
            return self._mh_step(current, 0, f64(rnd.standard_normal(g.model.rank)), None, self.name)
        self._bind(g, current.config.useLandmarkCorrespondence)
        if self._device_state is not current:
            self._push_state(g)
        if probabilistic:
            if rnd is None:
                raise ValueError("probabilistic update needs a random generator (rnd)")
            self._native_update_sample(current, f64(rnd.standard_normal(g.model.rank)))
        else:
            self._native_update(current, 1)
        new_general = self._pull_state(g)
        out = current.updateGeneral(new_general)
        self._device_state = out
        return out

    # -- one Metropolis-Hastings step per native call (gingr_fitter_mh_step) ---------------------------------------------------
    def enableFusedSteps(self, uncertainty: float, modelPointCount: int = 0):
        """From now on a probabilistic `update` / `proposeParameters` runs the WHOLE device side of a Metropolis-Hastings step in one
        native call -- proposal, model-to-target likelihood N(0, uncertainty) over the first modelPointCount vertices (0 = all), both
        transition densities of the informed proposal -- and the queries MetropolisHastings.next makes afterwards
        (`surfaceDistanceStats`, `logTransitionProbability`) are answered from what that call measured.  Same numbers, same order
        of random draws as the call-by-call path; states with stepLength != 1 or without meshes keep using that path."""
        self._mh = {"sdev": float(uncertainty), "points": int(modelPointCount or 0)}
        self._mh_last = None

    def disableFusedSteps(self):
        self._mh = self._mh_last = None

    def _mh_flavour(self, state):
        """(flavour, CpdParams | None, IcpParams | None) of gingr_mh_request; None when this configuration has no fused step"""
        return None

    def _mh_prepare(self, state):
        pass

    def _mh_plan(self, state):
        """_mh_flavour(state) with its ctypes pointers, remembered for the configuration OBJECT (frozen dataclasses shared by all the
        states of a chain): one Metropolis-Hastings step asks twice."""
        c = state.config
        memo = self._mh_plan_memo
        if memo is not None and memo[0] is c:
            return memo[1]
        fl = self._mh_flavour(state)
        plan = None
        if fl is not None:
            flavour, cp, ip = fl
            plan = (flavour, cp, ip, ctypes.pointer(cp) if cp is not None else None, ctypes.pointer(ip) if ip is not None else None)
        self._mh_plan_memo = (c, plan)
        return plan

    def _mh_usable(self, state) -> bool:
        g = state.general
        return (self._mh is not None and g.stepLength == 1.0 and getattr(g.model, "cells", None) is not None
                and g.targetCells is not None and self._mh_plan(state) is not None)

    def _adopt_state(self, old, new):
        """`new` is `old` with host-side fields rewritten (generatedBy): it stands for the same device state and measurements"""
        if self._device_state is old:
            self._device_state = new
        if self._mh_last is not None and self._mh_last["to"] is old:
            self._mh_last["to"] = new

    def _mh_step(self, current, kind: int, z, modelParameters, generatedBy: str):
        g = current.general
        self._bind(g, current.config.useLandmarkCorrespondence)
        last = self._mh_last
        if self._device_state is not current:
            if last is not None and last["from"] is current and self._device_state is last["to"]:
                _check(self.ctx.handle, self._lib.gingr_fitter_mh_restore(self._fitter), "gingr_fitter_mh_restore")   # rejected
            else:
                self._push_state(g)
            self._device_state = current
        self._mh_prepare(current)
        flavour, _, _, cpp, ipp = self._mh_plan(current)
        if self._mh_req is None:
            self._mh_req = (nat.MhRequest(), nat.MhResult())
        req, res = self._mh_req
        req.flavour, req.kind = flavour, kind
        req.cpd, req.icp = cpp, ipp
        req.z = req.alpha = None
        req.scalars = None
        req.eval_sdev, req.eval_points = self._mh["sdev"], self._mh["points"]
        # q(.|current) projects current.fit (step length 1): a number of `current` alone, known when a fused step produced or started
        # from this very state
        fw = last["bw"] if last is not None and last["to"] is current else (last["fw"] if last is not None and last["from"] is current else None)
        req.need_forward = 1 if fw is None else 0
        keep = None
        if kind == 0:
            req.z = dptr(z)
        else:
            mp = modelParameters
            keep = (f64(mp.shape), nat.StateScalars())
            sc = keep[1]
            sc.euler[:] = [mp.rotation.phi, mp.rotation.theta, mp.rotation.psi]
            sc.center[:] = list(mp.center)
            sc.translation[:] = list(mp.translation)
            sc.scale, sc.sigma2, sc.iteration, sc.status = mp.scale, g.sigma2, g.iteration + 1, g.status
            req.alpha, req.scalars = dptr(keep[0]), ctypes.pointer(sc)
        r, M = g.model.rank, g.model.numberOfPoints
        alpha, fit = np.empty(r), np.empty((M, 3))
        try:
            _check(self.ctx.handle, self._lib.gingr_fitter_mh_step(self._fitter, ctypes.byref(req), dptr(alpha), dptr(fit), ctypes.byref(res)),
                   "gingr_fitter_mh_step")
        except BaseException:
            # a step that failed half way leaves the device state undefined (the library refuses to continue from it): forget what
            # this object believes to be mirrored there, so that the next call pushes a state again
            self._device_state = None
            self._mh_last = None
            raise
        s = res.scalars
        e, t, c0 = s.euler, s.translation, s.center
        mp = ModelFittingParameters(scale=s.scale, translation=(t[0], t[1], t[2]), rotation=EulerAngles(e[0], e[1], e[2]),
                                    center=(c0[0], c0[1], c0[2]), shape=alpha)
        new_general = _with_fields(g, modelParameters=mp, fit=fit, sigma2=s.sigma2, iteration=s.iteration, status=s.status,
                                   generatedBy=generatedBy)
        out = _with_fields(current, general=new_general)
        self._device_state = out
        self._mh_last = {"from": current, "to": out, "sdev": self._mh["sdev"], "points": self._mh["points"],
                         "stats": (float(res.dist_sum), float(res.dist_max), int(res.count), float(res.log_value)),
                         "fw": float(res.log_q_forward) if fw is None else fw, "bw": float(res.log_q_backward)}
        return out

    def _ensure_device_state(self, state):
        """Make the fitter hold `state` (model, target, meshes, parameters; the fit is re-instantiated on the device)."""
        g = state.general
        self._bind(g, state.config.useLandmarkCorrespondence)
        if self._device_state is not state:
            self._push_state(g)
            self._device_state = state

    def proposeParameters(self, current, modelParameters: ModelFittingParameters, generatedBy: str):
        """GingrGeneratorWrapper.propose for a proposal that only rewrites the parameters (GingrGeneratorWrapper.scala:28-39):
        fit = modelInstanceShapePoseScale(model, parameters) -- instantiated on the device -- and iteration + 1."""
        if self._mh_usable(current):
            return self._mh_step(current, 1, None, modelParameters, generatedBy)
        g = dataclasses.replace(current.general, modelParameters=modelParameters, iteration=current.general.iteration + 1)
        self._bind(g, current.config.useLandmarkCorrespondence)
        self._push_state(g)
        new_general = dataclasses.replace(self._pull_state(g), generatedBy=generatedBy)
        out = current.updateGeneral(new_general)
        self._device_state = out
        return out

    def surfaceDistanceStats(self, state, direction: int, n_points: int = 0, points=None, boundary_aware: bool = False,
                             sdev: float = 0.0) -> Tuple[float, float, int, float]:
        """(sum, max, count, sum of log N(d; 0, sdev)) of the closest-point distances between the fit of `state` and the target
        surface: direction 0 = fit vertices -> target surface, 1 = target vertices (or `points`) -> fit surface."""
        g = state.general
        if getattr(g.model, "cells", None) is None or g.targetCells is None:
            raise ValueError("surface distances need model.cells and targetCells")
        last = self._mh_last
        if (last is not None and last["to"] is state and direction == 0 and points is None and not boundary_aware
                and float(sdev) == last["sdev"] and int(n_points or 0) == last["points"]):
            return last["stats"]          # measured by the fused step that produced this state
        self._ensure_device_state(state)
        out = np.zeros(4)
        pts = None if points is None else f64(points)
        n = int(n_points) if pts is None else pts.shape[0]
        _check(self.ctx.handle, self._lib.gingr_fitter_surface_distance_stats(
            self._fitter, int(direction), n, None if pts is None else dptr(pts), int(bool(boundary_aware)), float(sdev), dptr(out)),
            "gingr_fitter_surface_distance_stats")
        return float(out[0]), float(out[1]), int(out[2]), float(out[3])

    def logTransitionProbability(self, from_state, to_state) -> float:
        """GeneratorWrapperStochastic.logTransitionProbability (GeneratorWrapperStochastic.scala:42-63): log-density, under
        the posterior model of `from_state`, of the mesh the reference projects -- from.fit when stepLength == 1, otherwise
        the unposed instance of the step-compensated coefficients.  -inf when the posterior cannot be computed."""
        g = from_state.general
        last = self._mh_last
        if last is not None and g.stepLength == 1.0:   # both densities of the step that made `to` from `from` came with it
            if last["from"] is from_state and last["to"] is to_state:
                return last["fw"]
            if last["from"] is to_state and last["to"] is from_state:
                return last["bw"]
        if g.stepLength != 1.0:
            a0, a1 = f64(g.modelParameters.shape), f64(to_state.general.modelParameters.shape)
            comp = a0 + (a1 - a0) / g.stepLength
            dm = DeviceModel(self.ctx, g.model)
            try:
                mesh = dm.instance(comp)
            finally:
                dm.close()
        else:
            mesh = f64(g.fit)
        self._ensure_device_state(from_state)     # the query leaves the device state as it is: no push when it already holds `from`
        try:
            return self._native_logpdf(from_state, mesh)
        except GingrNativeError as e:
            if e.code in (nat.ERR_NOT_SPD, nat.ERR_NONFINITE):
                return float("-inf")
            raise

    def run(self, initialState, callBackLogger: Optional[Callable] = None, acceptRejectLogger=None, probabilisticSettings=None,
            generators=None, rnd=None):
        """GingrAlgorithm.run (:115-175).  Without probabilisticSettings: the deterministic registration loop -- the chain
        yields the initial state first, so take(maxIterations) performs maxIterations-1 updates; stops on convergence or
        ModelFlexibilityError.  With them: the Metropolis-Hastings chain of gingr_amd.sampling (informed + random-walk
        proposals, evaluator, best sample)."""
        if probabilisticSettings is not None or acceptRejectLogger is not None or generators is not None:
            from . import sampling
            return sampling.run(self, initialState, acceptRejectLogger, callBackLogger, probabilisticSettings, generators, rnd)
        state = initialState
        last_general = None
        converged = False
        k = 0
        if callBackLogger is None and state.config.converged in (_cpd_converged, _never_converged) \
                and state.general.status != FittingStatuses.ModelFlexibilityError and state.config.maxIterations > 1:
            return self._run_resident(state)
        while True:
            # dropWhile body (:142-153)
            if last_general is not None:
                converged = bool(state.config.converged(last_general, state.general, state.config.threshold))
            error = state.general.status == FittingStatuses.ModelFlexibilityError
            last_general = state.general
            if callBackLogger is not None:
                callBackLogger(state)
            k += 1
            if converged or error or k >= state.config.maxIterations:
                break
            state = self.update(state, False)
        if state.general.status == FittingStatuses.None_:
            state = state.updateGeneral(state.general.updateStatus(
                FittingStatuses.Converged if converged else FittingStatuses.MaxIteration))
        return state


def _run_resident_impl(self, state):
    """The deterministic loop of `run` with the state resident on the device: nobody watches the intermediate states (no
    call-back) and the convergence rule is one of the reference's own (|sigma2 - last sigma2| < threshold for CPD, never for ICP).
    Updates are enqueued in blocks and only the state a block ends in is looked at; coefficients and fit come back once, at the end.
    That is sound because the chain cannot run past its last state on the device: a failed fit stays as it is (post_solve_kernel commits
    nothing once the status is ModelFlexibilityError; GingrAlgorithm.scala:149-157 stops at that very state), and so does a state the
    CPD rule stopped at (gingr_fitter_set_stop_threshold: the comparison the dropWhile makes, made by the kernel that commits sigma2).
    ICP has no rule: the whole run is one block.  CPD: updates enqueued behind the stopping state are wasted work, so the block is
    one update where an update is long and a few where the synchronisation would be a fifth of it.
    Same states, same stopping iteration, same status as the generic loop."""
    g = state.general
    self._bind(g, state.config.useLandmarkCorrespondence)
    if self._device_state is not state:
        self._push_state(g)
    cpd_rule = state.config.converged is _cpd_converged
    left = state.config.maxIterations - 1
    if cpd_rule:
        pairs = float(g.model.numberOfPoints) * float(np.asarray(g.target).shape[0])
        block = 4 if pairs <= 4e6 else (2 if pairs <= 3e7 else 1)
    else:
        block = left
    hit, sc = ctypes.c_int32(0), nat.StateScalars()
    _check(self.ctx.handle, self._lib.gingr_fitter_set_stop_threshold(self._fitter, float(state.config.threshold) if cpd_rule else -1.0),
           "gingr_fitter_set_stop_threshold")
    try:
        while left > 0:
            n = min(block, left)
            self._native_update(state, n)
            left -= n
            _check(self.ctx.handle, self._lib.gingr_fitter_get_state(self._fitter, None, ctypes.byref(sc), None), "gingr_fitter_get_state")
            _check(self.ctx.handle, self._lib.gingr_fitter_stop_rule_hit(self._fitter, ctypes.byref(hit)), "gingr_fitter_stop_rule_hit")
            if hit.value or sc.status == FittingStatuses.ModelFlexibilityError:
                break
    finally:
        # (also clears the mark: the state on the device takes updates again)
        self._lib.gingr_fitter_set_stop_threshold(self._fitter, -1.0)
    out = state.updateGeneral(self._pull_state(g))
    self._device_state = out
    converged = bool(hit.value)
    if out.general.status == FittingStatuses.None_:
        out = out.updateGeneral(out.general.updateStatus(FittingStatuses.Converged if converged else FittingStatuses.MaxIteration))
    return out


GingrAlgorithm._run_resident = _run_resident_impl


def _initial_general(ctx: Context, model: PointDistributionModel, target: np.ndarray, sigma2: float,
                     transform: int, stepLength: float, landmarks: Optional[LandmarkCorrespondences],
                     initial_pose: Optional[Tuple[Sequence[float], Sequence[float]]] = None,
                     targetCells: Optional[np.ndarray] = None) -> GeneralRegistrationState:
    """GeneralRegistrationState.apply (:135-179): alpha = 0, optional initial pose, fit = instance."""
    mp = ModelFittingParameters.zero(model.rank)
    if initial_pose is not None:
        euler, t = initial_pose
        mp = dataclasses.replace(mp, rotation=EulerAngles(*euler), translation=tuple(t))
    dm = DeviceModel(ctx, model)
    try:
        fit = dm.instance(mp.shape, [mp.rotation.phi, mp.rotation.theta, mp.rotation.psi], mp.center, mp.translation, mp.scale)
    finally:
        dm.close()
    return GeneralRegistrationState(model=model, modelParameters=mp, target=f64(target), fit=fit, sigma2=sigma2,
                                    globalTransformation=transform, stepLength=stepLength,
                                    landmarkCorrespondences=landmarks, targetCells=targetCells)


class CpdRegistration(GingrAlgorithm):
    name = "CPD"

    def createInitialState(self, model: PointDistributionModel, target, config: CpdConfiguration,
                           transform: int = GlobalTranformationType.RigidTransforms, stepLength: float = 1.0,
                           landmarks: Optional[LandmarkCorrespondences] = None, initial_pose=None,
                           targetCells: Optional[np.ndarray] = None) -> CpdRegistrationState:
        """targetCells (with model.cells): the target's triangulation -- CPD itself works on the vertices; the surface
        likelihood of a probabilistic run (sampling.IndependentPointDistanceEvaluator) needs both meshes."""
        g = _initial_general(self.ctx, model, target, 1.0, transform, stepLength, landmarks, initial_pose, targetCells)
        return self.initializeState(g, config)

    def initializeState(self, general: GeneralRegistrationState, config: CpdConfiguration) -> CpdRegistrationState:
        # CpdRegistrationState.apply (CPD.scala:92-102): sigma2 = initialSigma or sum|x - y|^2/(3MN) over model MEAN vs target
        if config.initialSigma is not None:
            s2 = float(config.initialSigma)
        else:
            mean_pts = f64(general.model.reference) + f64(general.model.mean)
            s2 = self.ctx.cpd_initial_sigma2(mean_pts, general.target)
        return CpdRegistrationState(general.updateSigma2(s2), config)

    def _native_update(self, current: CpdRegistrationState, n: int):
        p = nat.CpdParams(current.config.w, current.config.lambda_)
        _check(self.ctx.handle, self._lib.gingr_fitter_update_cpd_async(self._fitter, ctypes.byref(p), n),
               "gingr_fitter_update_cpd_async")

    def _mh_flavour(self, state: CpdRegistrationState):
        return 0, nat.CpdParams(state.config.w, state.config.lambda_), None

    def _native_update_sample(self, current: CpdRegistrationState, z: np.ndarray):
        p = nat.CpdParams(current.config.w, current.config.lambda_)
        _check(self.ctx.handle, self._lib.gingr_fitter_update_cpd_sample_async(self._fitter, ctypes.byref(p), dptr(z)),
               "gingr_fitter_update_cpd_sample_async")

    def _native_logpdf(self, state: CpdRegistrationState, mesh: np.ndarray) -> float:
        p = nat.CpdParams(state.config.w, state.config.lambda_)
        out = ctypes.c_double()
        _check(self.ctx.handle, self._lib.gingr_fitter_posterior_logpdf_cpd(self._fitter, ctypes.byref(p), dptr(mesh),
                                                                             ctypes.byref(out)), "gingr_fitter_posterior_logpdf_cpd")
        return out.value

    # plugin accessors served from one streaming evaluation (the reference recomputes P for each of them)
    def _stats(self, state: CpdRegistrationState) -> dict:
        if getattr(self, "_stats_state", None) is not state:
            self._stats_cache = self.ctx.cpd_stats(state.general.fit, state.general.target, state.general.sigma2, state.config.w)
            self._stats_state = state
            self._stats_keep = state
        return self._stats_cache

    def getCorrespondence(self, state: CpdRegistrationState) -> CorrespondencePairs:
        st = self._stats(state)                                                       # CPD.scala:32-49
        y = f64(state.general.fit)
        with np.errstate(invalid="ignore", divide="ignore"):
            td = y + (st["PX"] * (1.0 / st["P1"])[:, None] - y)
        return CorrespondencePairs(np.arange(y.shape[0]), td)

    def getUncertainty(self, pid: int, state: CpdRegistrationState) -> np.ndarray:
        st = self._stats(state)                                                       # CPD.scala:120-128
        with np.errstate(invalid="ignore", divide="ignore"):
            return np.eye(3) * state.general.sigma2 * state.config.lambda_ * (1.0 / st["P1"][int(pid)])

    def updateSigma2(self, state: CpdRegistrationState) -> float:
        return self._stats(state)["sigma2_next"]                                      # CPD.scala:133-147


class IcpRegistration(GingrAlgorithm):
    name = "ICP"
    _METHODS = ("PointcloudClosestPoint", "TriangularClosestPoint", "AlongNormalClosestPoint")

    @staticmethod
    def _surface(config: IcpConfiguration) -> bool:
        return config.correspondenceMethod in ("TriangularClosestPoint", "AlongNormalClosestPoint")

    def _select_surface_method(self, config: IcpConfiguration):
        method = 1 if config.correspondenceMethod == "AlongNormalClosestPoint" else 0
        if self._sel.get("method") != method:
            _check(self.ctx.handle, self._lib.gingr_fitter_set_surface_method(self._fitter, method), "gingr_fitter_set_surface_method")
            self._sel["method"] = method

    def _select_direction(self, config: IcpConfiguration):
        rev = 1 if config.reverseCorrespondenceDirection else 0
        if self._sel.get("direction") != rev:
            _check(self.ctx.handle, self._lib.gingr_fitter_set_correspondence_direction(self._fitter, rev),
                   "gingr_fitter_set_correspondence_direction")
            self._sel["direction"] = rev

    def _phase0(self, state: "IcpRegistrationState"):
        g, c = state.general, state.config
        self._bind(g, c.useLandmarkCorrespondence)
        self._push_state(g)
        self._device_state = None
        self._select_direction(c)
        p = nat.IcpParams(c.initialSigma, c.endSigma, c.maxIterations)
        if self._surface(c):
            self._select_surface_method(c)
            _check(self.ctx.handle, self._lib.gingr_fitter_icp_surface_phase_async(self._fitter, ctypes.byref(p), 0),
                   "gingr_fitter_icp_surface_phase_async")
        else:
            _check(self.ctx.handle, self._lib.gingr_fitter_icp_phase_async(self._fitter, ctypes.byref(p), 0), "gingr_fitter_icp_phase_async")

    def reversedCorrespondence(self, state: "IcpRegistrationState") -> Tuple[np.ndarray, np.ndarray]:
        """closestPointCorrespondenceReversal (ClosestPointRegistrator.scala:34-49) for the state's fit: per TARGET vertex the
        template vertex it is assigned to and its weight in {0, 1}."""
        if not state.config.reverseCorrespondenceDirection:
            raise ValueError("the configuration does not reverse the correspondence direction")
        self._phase0(state)
        N = np.asarray(state.general.target).shape[0]
        tid, w = np.empty(N, dtype=np.int32), np.empty(N)
        _check(self.ctx.handle, self._lib.gingr_fitter_get_reversed_correspondence(self._fitter, iptr(tid), dptr(w)),
               "gingr_fitter_get_reversed_correspondence")
        return tid, w

    def createInitialState(self, model: PointDistributionModel, target, config: IcpConfiguration,
                           transform: int = GlobalTranformationType.RigidTransforms, stepLength: float = 1.0,
                           landmarks: Optional[LandmarkCorrespondences] = None, initial_pose=None,
                           targetCells: Optional[np.ndarray] = None) -> IcpRegistrationState:
        g = _initial_general(self.ctx, model, target, 1.0, transform, stepLength, landmarks, initial_pose, targetCells)
        return self.initializeState(g, config)

    def initializeState(self, general: GeneralRegistrationState, config: IcpConfiguration) -> IcpRegistrationState:
        if config.correspondenceMethod not in self._METHODS:
            raise NotImplementedError("ICP flavours: PointcloudClosestPoint, TriangularClosestPoint, AlongNormalClosestPoint (ICP.scala:32-44)")
        if self._surface(config) and (getattr(general.model, "cells", None) is None or general.targetCells is None):
            raise ValueError(config.correspondenceMethod + " needs the triangulations (model.cells and targetCells); for point "
                             "clouds choose correspondenceMethod=\"PointcloudClosestPoint\"")
        return IcpRegistrationState(general.updateSigma2(float(config.initialSigma)), config)   # ICP.scala:73-85

    def _native_update(self, current: IcpRegistrationState, n: int):
        c = current.config
        p = nat.IcpParams(c.initialSigma, c.endSigma, c.maxIterations)
        self._select_direction(c)
        if self._surface(c):
            self._select_surface_method(c)
            _check(self.ctx.handle, self._lib.gingr_fitter_update_icp_surface_async(self._fitter, ctypes.byref(p), n),
                   "gingr_fitter_update_icp_surface_async")
        else:
            _check(self.ctx.handle, self._lib.gingr_fitter_update_icp_async(self._fitter, ctypes.byref(p), n),
                   "gingr_fitter_update_icp_async")

    def surfaceCorrespondence(self, state: IcpRegistrationState) -> Tuple[np.ndarray, np.ndarray]:
        """ClosestPointTriangleMesh3D.closestPointCorrespondence(fit, target) (ClosestPointRegistrator.scala:75-100) for the
        state's fit: (closest surface points (M,3), weights in {0,1})."""
        g, c = state.general, state.config
        if c.reverseCorrespondenceDirection:
            raise ValueError("reversed direction: use reversedCorrespondence (entries are per target vertex)")
        if not self._surface(c):
            raise ValueError("surfaceCorrespondence needs a surface correspondenceMethod (the state uses " + c.correspondenceMethod + ")")
        self._phase0(state)
        M = g.model.numberOfPoints
        cp, w = np.empty((M, 3)), np.empty(M)
        _check(self.ctx.handle, self._lib.gingr_fitter_get_surface_correspondence(self._fitter, dptr(cp), dptr(w)),
               "gingr_fitter_get_surface_correspondence")
        return cp, w

    def _mh_flavour(self, state: IcpRegistrationState):
        c = state.config
        if c.reverseCorrespondenceDirection:
            return None
        return (2 if self._surface(c) else 1), None, nat.IcpParams(c.initialSigma, c.endSigma, c.maxIterations)

    def _mh_prepare(self, state: IcpRegistrationState):
        self._select_direction(state.config)
        if self._surface(state.config):
            self._select_surface_method(state.config)

    def _native_update_sample(self, current: IcpRegistrationState, z: np.ndarray):
        c = current.config
        p = nat.IcpParams(c.initialSigma, c.endSigma, c.maxIterations)
        self._select_direction(c)
        if self._surface(c):
            self._select_surface_method(c)
            _check(self.ctx.handle, self._lib.gingr_fitter_update_icp_surface_sample_async(self._fitter, ctypes.byref(p), dptr(z)),
                   "gingr_fitter_update_icp_surface_sample_async")
        else:
            _check(self.ctx.handle, self._lib.gingr_fitter_update_icp_sample_async(self._fitter, ctypes.byref(p), dptr(z)),
                   "gingr_fitter_update_icp_sample_async")

    def _native_logpdf(self, state: IcpRegistrationState, mesh: np.ndarray) -> float:
        c = state.config
        p = nat.IcpParams(c.initialSigma, c.endSigma, c.maxIterations)
        out = ctypes.c_double()
        self._select_direction(c)
        if self._surface(c):
            self._select_surface_method(c)
        fn = self._lib.gingr_fitter_posterior_logpdf_icp_surface if self._surface(c) else self._lib.gingr_fitter_posterior_logpdf_icp
        _check(self.ctx.handle, fn(self._fitter, ctypes.byref(p), dptr(mesh), ctypes.byref(out)), "gingr_fitter_posterior_logpdf_icp")
        return out.value

    def getCorrespondence(self, state: IcpRegistrationState) -> CorrespondencePairs:
        if state.config.reverseCorrespondenceDirection:                               # ICP.scala:46-50
            tid, w = self.reversedCorrespondence(state)
            keep = np.flatnonzero(w == 1.0)
            return CorrespondencePairs(tid[keep].astype(np.int64), f64(state.general.target)[keep])
        if self._surface(state.config):                                               # ICP.scala:40-41,50
            cp, w = self.surfaceCorrespondence(state)
            keep = np.flatnonzero(w == 1.0)
            return CorrespondencePairs(keep, cp[keep])
        idx, _, _ = self.ctx.nn(state.general.fit, state.general.target)              # ICP.scala:36-52
        return CorrespondencePairs(np.arange(idx.shape[0]), f64(state.general.target)[idx])

    def getUncertainty(self, pid: int, state: IcpRegistrationState) -> np.ndarray:
        return np.eye(3) * state.general.sigma2                                       # ICP.scala:90-92

    def updateSigma2(self, state: IcpRegistrationState) -> float:
        return max(state.general.sigma2 - state.config.sigmaStep, state.config.endSigma)   # ICP.scala:96-99

    def last_correspondence_indices(self) -> np.ndarray:
        M = self._dev_model.M_local
        idx = np.empty(M, dtype=np.int32)
        _check(self.ctx.handle, self._lib.gingr_fitter_get_icp_idx(self._fitter, iptr(idx), None), "gingr_fitter_get_icp_idx")
        return idx
