"""The reference's convenience layer on top of the registration path: `GingrInterface` (G/simple/GingrInterface.scala:22-65),
`SimpleRegistrator` (G/api/registration/SimpleRegistrator.scala:46-159) and `PointDistributionModel.newReference(_,
NearestNeighborInterpolator())` as `runDecimated` uses it -- every shipped demo enters through these (examples/DemoCPD.scala,
DemoICP.scala, DemoMultiResolution.scala).  Host control flow only: states are created, `GingrAlgorithm.run` drives the device
resident fitter, the final fit is instantiated on the FULL model and scored by `RegistrationComparison` on the GPU.

Mesh decimation: scalismo's `mesh.operations.decimate(n)` lives in the un-vendored dependency and is not restated.  `runDecimated`
takes a `decimate(vertices, cells, n) -> (vertices, cells)` callable; the default (`cluster_decimate`) is a deterministic vertex
clustering whose vertices are a subset of the input's.  A different coarse mesh changes the intermediate states of a coarse-to-fine
schedule, not what a stage computes from its inputs -- the decimated meshes are inputs of the path."""
from __future__ import annotations

import dataclasses
import math
from typing import Callable, Optional, Sequence, Tuple

import numpy as np

from .api import (GPMMTriangleMesh3D, InterpolatedDevicePointDistributionModel, Context, CpdConfiguration, CpdRegistration, DeviceModel, EulerAngles, FittingStatuses, GeneralRegistrationState,
                  GlobalTranformationType, IcpConfiguration, IcpRegistration, ModelFittingParameters, PointDistributionModel, f64)
from . import io as gio
from .sampling import (IndependentPoints, JSONStateLogger, ModelToTargetEvaluation, ProbabilisticSettings, Random,
                       RegistrationComparison, TriangleMesh3D)


# ------------------------------------------------------------------------------------------------ rotation convention
def euler_to_rotation_matrix(phi: float, theta: float, psi: float) -> np.ndarray:
    """scalismo Rotation(phi, theta, psi, centre) = Rz(phi) Ry(theta) Rx(psi)  [SCALISMO-RECALL, SURVEY A.6]; the same
    convention the device uses (gingr_amd/csrc/svd3.h: euler_to_rot)."""
    cps, sps = math.cos(psi), math.sin(psi)
    cth, sth = math.cos(theta), math.sin(theta)
    cph, sph = math.cos(phi), math.sin(phi)
    return np.array([[cth * cph, sps * sth * cph - cps * sph, sps * sph + cps * sth * cph],
                     [cth * sph, cps * cph + sps * sth * sph, cps * sth * sph - sps * cph],
                     [-sth, sps * cth, cps * cth]], dtype=np.float64)


def rotation_matrix_to_euler(R) -> Tuple[float, float, float]:
    """RotationSpace3D.rotMatrixToEulerAngles (Slabaugh's recipe) as GeneralRegistrationState.apply uses it for the
    initial model transform (GeneralRegistrationState.scala:144-150)."""
    R = np.asarray(R, dtype=np.float64)
    if abs(abs(R[2, 0]) - 1.0) > 0.0001:
        theta = math.asin(-R[2, 0])
        ct = math.cos(theta)
        return math.atan2(R[1, 0] / ct, R[0, 0] / ct), theta, math.atan2(R[2, 1] / ct, R[2, 2] / ct)
    if abs(R[2, 0] + 1.0) < 0.0001:
        return 0.0, math.pi / 2.0, math.atan2(R[0, 1], R[0, 2])
    return 0.0, -math.pi / 2.0, math.atan2(-R[0, 1], -R[0, 2])


@dataclasses.dataclass(frozen=True)
class TranslationAfterRotation:
    """scalismo TranslationAfterRotation: p -> R (p - centre) + centre + t.  Only R and t enter GeneralRegistrationState.apply; the
    rotation centre is REPLACED by the origin there (GeneralRegistrationState.scala:148), exactly as mirrored here."""
    translation: Tuple[float, float, float]
    rotation: np.ndarray                     # 3 x 3

    @staticmethod
    def fromEuler(translation, phi: float, theta: float, psi: float) -> "TranslationAfterRotation":
        return TranslationAfterRotation(tuple(float(v) for v in translation), euler_to_rotation_matrix(phi, theta, psi))


# ------------------------------------------------------------------------------------------------ model on a new reference
def new_reference_nearest_neighbor(ctx: Context, model, new_reference, new_cells=None):
    """model.newReference(newRef, NearestNeighborInterpolator())  (SimpleRegistrator.scala:89-90) [SCALISMO-RECALL]: the
    continuous GP takes mean and eigenfunctions of the closest OLD reference point; discretising it on the new points is a row
    gather -- eigenvalues and the number of components are unchanged, nothing is re-orthonormalised.  The closest-point search is
    the exact device search of the ICP path (lowest index on ties); the rows are gathered in HBM (gingr_model_new_reference)."""
    new_reference = f64(new_reference).reshape(-1, 3)
    idx, _, _ = ctx.nn(new_reference, f64(model.reference))
    ids = np.repeat(np.asarray(idx, dtype=np.int32)[:, None], 3, axis=1)
    w = np.tile(np.array([1.0, 0.0, 0.0]), (new_reference.shape[0], 1))
    return InterpolatedDevicePointDistributionModel(ctx, model, new_reference, ids, w, new_cells)


def new_reference_triangle_mesh(ctx: Context, model, new_reference, new_cells=None):
    """model.newReference(fullReference, TriangleMeshInterpolator3D())  (examples/DemoHelper/DemoDatasetLoader.scala:58-62: the
    demos build their model on a decimated reference and carry it to the full-resolution mesh) [SCALISMO-RECALL]: the field value
    at a new point is taken at its closest point on the OLD surface -- a vertex value, the linear blend along an edge, or the
    barycentric blend inside a triangle; in all three cases the barycentric combination of the triangle's corners."""
    if model.cells is None:
        raise ValueError("the source model needs its triangulation (model.cells)")
    new_reference = f64(new_reference).reshape(-1, 3)
    _, _, tid, bary = ctx.mesh_closest_points(new_reference, model.reference, model.cells)
    ids = np.ascontiguousarray(model.cells, dtype=np.int32).reshape(-1, 3)[tid]
    return InterpolatedDevicePointDistributionModel(ctx, model, new_reference, ids, bary, new_cells)


def cluster_decimate(vertices, cells, n_target: int) -> Tuple[np.ndarray, np.ndarray]:
    """Deterministic vertex clustering to about `n_target` vertices (NOT scalismo's decimation, see the module docstring): the
    bounding box is cut into cubes, every occupied cube keeps its vertex closest to the cube's mean (lowest index on ties), the
    triangles are re-indexed and the collapsed / duplicate ones dropped.  The cube size is bisected until the count is the
    closest reachable to `n_target` from above."""
    v = f64(vertices).reshape(-1, 3)
    c = None if cells is None else np.asarray(cells, dtype=np.int64).reshape(-1, 3)      # None: a point cloud
    if n_target >= v.shape[0]:
        return v.copy(), None if c is None else c.astype(np.int32)
    lo_corner = v.min(axis=0)
    extent = float(np.max(v.max(axis=0) - lo_corner)) or 1.0

    def clusters(h: float):
        key = np.floor((v - lo_corner) / h).astype(np.int64)
        _, inv = np.unique(key, axis=0, return_inverse=True)
        return inv.reshape(-1)

    lo, hi = extent * 1e-6, extent * 2.0          # cube sizes: lo -> every vertex alone, hi -> one cube
    best = None
    for _ in range(60):
        mid = math.sqrt(lo * hi)
        inv = clusters(mid)
        k = int(inv.max()) + 1
        if k >= n_target:
            best = inv
            lo = mid                               # still enough vertices: try coarser
        else:
            hi = mid
        if hi / lo < 1.0005:
            break
    inv = best if best is not None else clusters(lo)
    k = int(inv.max()) + 1
    cnt = np.bincount(inv, minlength=k).astype(np.float64)
    mean = np.stack([np.bincount(inv, weights=v[:, d], minlength=k) / cnt for d in range(3)], axis=1)
    d2 = np.sum((v - mean[inv]) ** 2, axis=1)
    order = np.lexsort((np.arange(v.shape[0]), d2, inv))          # by cluster, then distance, then index
    first = np.ones(order.shape[0], dtype=bool)
    first[1:] = inv[order][1:] != inv[order][:-1]
    rep = order[first]                                            # representative vertex of every cluster (cluster order)
    rep_sorted = np.sort(rep)                                     # keep the original vertex order
    new_id_of_cluster = np.empty(k, dtype=np.int64)
    new_id_of_cluster[inv[rep_sorted]] = np.arange(k)
    if c is None:
        return v[rep_sorted].copy(), None
    tri = new_id_of_cluster[inv[c]]
    keep = (tri[:, 0] != tri[:, 1]) & (tri[:, 1] != tri[:, 2]) & (tri[:, 0] != tri[:, 2])
    tri = tri[keep]
    # duplicates (same vertex set) -- keep the first occurrence
    _, uniq = np.unique(np.sort(tri, axis=1), axis=0, return_index=True)
    tri = tri[np.sort(uniq)]
    return v[rep_sorted].copy(), tri.astype(np.int32)


# ------------------------------------------------------------------------------------------------ SimpleRegistrator
class SimpleRegistrator:
    """G/api/registration/SimpleRegistrator.scala:46-159.  `target` is a TriangleMesh3D (points + cells; cells may be None for the
    point-cloud flavours); landmarks are `gingr_amd.io.Landmark` lists; `initialModelParameterTransform` a TranslationAfterRotation."""

    def __init__(self, algorithm, config, model: PointDistributionModel, target: TriangleMesh3D,
                 initialModelParameterTransform: Optional[TranslationAfterRotation] = None,
                 modelLandmarks: Optional[Sequence] = None, targetLandmarks: Optional[Sequence] = None,
                 evaluationMode: str = ModelToTargetEvaluation, evaluatorUncertainty: float = 1.0,
                 evaluatedPoints: Optional[int] = None, logFileFittingParameters: Optional[str] = None,
                 rnd: Optional[Random] = None, decimate: Callable = cluster_decimate, verbose: bool = True):
        self.algorithm, self.config, self.model, self.target = algorithm, config, model, target
        self.ctx: Context = algorithm.ctx
        self.initialModelParameterTransform = initialModelParameterTransform
        self.modelLandmarks, self.targetLandmarks = modelLandmarks, targetLandmarks
        self.evaluationMode, self.evaluatorUncertainty, self.evaluatedPoints = evaluationMode, evaluatorUncertainty, evaluatedPoints
        self.logFileFittingParameters = logFileFittingParameters
        self.rnd = rnd if rnd is not None else Random(0)
        self.decimate, self.verbose = decimate, verbose

    # -- helpers -----------------------------------------------------------------------------------------------------------
    def _landmarks(self, model: PointDistributionModel):
        # landmarkCorrespondences is a lazy val of the state: it follows the state's (possibly decimated) model
        # (GeneralRegistrationState.scala:43-62)
        if self.modelLandmarks and self.targetLandmarks:
            return gio.landmark_correspondences(model.reference, self.modelLandmarks, self.targetLandmarks)
        return None

    def _instance(self, model: PointDistributionModel, mp: ModelFittingParameters) -> np.ndarray:
        dm = DeviceModel(self.ctx, model)
        try:
            return dm.instance(mp.shape, [mp.rotation.phi, mp.rotation.theta, mp.rotation.psi], mp.center, mp.translation, mp.scale)
        finally:
            dm.close()

    def _combineStates(self, generalState: GeneralRegistrationState):
        """SimpleRegistrator.combineStates (:75-82): iteration and status cleared, then the algorithm's own initializeState
        (CPD: sigma2 from config.initialSigma or the mean-to-target formula; ICP: config.initialSigma)."""
        g = dataclasses.replace(generalState, iteration=0, status=FittingStatuses.None_)
        return self.algorithm.initializeState(g, self.config)

    def createInitialState(self, model: PointDistributionModel, target: TriangleMesh3D, globalTransformation: int,
                           modelTransform: Optional[TranslationAfterRotation] = None):
        """:108-126 + GeneralRegistrationState.apply (GeneralRegistrationState.scala:135-179): alpha = 0; the initial pose is the
        given translation and the Euler angles of the given rotation ABOUT THE ORIGIN; landmarks only when both lists are non-empty."""
        pose = None
        if modelTransform is not None:
            pose = (rotation_matrix_to_euler(modelTransform.rotation), tuple(modelTransform.translation))
        return self.algorithm.createInitialState(model, target.points, self.config, transform=globalTransformation,
                                                 landmarks=self._landmarks(model), initial_pose=pose, targetCells=target.cells)

    def _decimateState(self, generalState: Optional[GeneralRegistrationState], globalTransformation: int, modelPoints: int,
                       targetPoints: int) -> GeneralRegistrationState:
        """:84-106"""
        ref_v, ref_c = self.decimate(self.model.reference, self.model.cells, modelPoints)
        decimatedModel = new_reference_nearest_neighbor(self.ctx, self.model, ref_v, ref_c)
        tv, tc = self.decimate(self.target.points, self.target.cells, targetPoints)
        decimatedTarget = TriangleMesh3D(tv, tc)
        if generalState is not None:
            init = self._combineStates(generalState)
        else:
            init = self.createInitialState(decimatedModel, decimatedTarget, globalTransformation, self.initialModelParameterTransform)
        g = init.general
        return dataclasses.replace(g, model=decimatedModel, target=f64(decimatedTarget.points), targetCells=decimatedTarget.cells,
                                   fit=self._instance(decimatedModel, g.modelParameters),
                                   landmarkCorrespondences=self._landmarks(decimatedModel) if g.landmarkCorrespondences is not None
                                   else None)

    # -- the two entry points ----------------------------------------------------------------------------------------------
    def runDecimated(self, modelPoints: int, targetPoints: int, generalState: Optional[GeneralRegistrationState] = None,
                     globalTransformation: int = GlobalTranformationType.RigidTransforms, probabilistic: bool = False,
                     randomMixture: float = 0.5, callback: Optional[Callable] = None):
        """:62-73"""
        init = self._decimateState(generalState, globalTransformation, modelPoints, targetPoints)
        return self.run(init, globalTransformation, probabilistic, randomMixture, callback)

    def run(self, generalState: Optional[GeneralRegistrationState] = None,
            globalTransformation: int = GlobalTranformationType.RigidTransforms, probabilistic: bool = False,
            randomMixture: float = 0.5, callback: Optional[Callable] = None):
        """:128-158.  NOTE (as in the reference): a passed-in state keeps ITS globalTransformation -- the argument is only used
        when the state is created here."""
        if generalState is not None:
            state = self._combineStates(generalState)
        else:
            state = self.createInitialState(self.model, self.target, globalTransformation, self.initialModelParameterTransform)
        jsonLogger, settings = None, None
        if probabilistic:
            count, tpoints = None, None
            if self.evaluatedPoints is not None:
                # numberOfPointsForComparison (IndependentPointDistanceEvaluator.scala:44-50): instance and target are decimated;
                # the model side then compares the sample's vertices with the ids 0 .. n'-1 of the decimated instance, the target
                # side the decimated target's points
                g = state.general
                fv, _ = self.decimate(g.fit, g.model.cells, self.evaluatedPoints)
                tpoints, _ = self.decimate(g.target, g.targetCells, self.evaluatedPoints)
                count = int(fv.shape[0])
            evaluator = IndependentPoints(self.algorithm, state, self.evaluatorUncertainty, self.evaluationMode, None, count, tpoints)
            jsonLogger = JSONStateLogger(evaluator, self.logFileFittingParameters)
            settings = ProbabilisticSettings(evaluator, randomMixture=randomMixture)
        final = self.algorithm.run(state, callBackLogger=callback, acceptRejectLogger=jsonLogger, probabilisticSettings=settings,
                                   rnd=self.rnd)
        fit = self._instance(self.model, final.general.modelParameters)       # on the FULL model (:151)
        if jsonLogger is not None:
            if self.verbose:
                jsonLogger.printAcceptInfo()
            jsonLogger.writeLog()
        self.lastComparison = None
        if self.model.cells is not None and self.target.cells is not None:
            if self.verbose:
                print("Final registration with full resolution meshes:")
            self.lastComparison = RegistrationComparison(self.ctx, self.verbose).evaluateReconstruction2GroundTruthBoundaryAware(
                "", TriangleMesh3D(fit, self.model.cells), self.target)
        return final.updateGeneral(dataclasses.replace(final.general, fit=fit))


class GingrInterface:
    """G/simple/GingrInterface.scala:22-65: one model / target / landmark set, a SimpleRegistrator per configuration."""

    def __init__(self, ctx: Context, model: PointDistributionModel, target: TriangleMesh3D,
                 initialModelParameterTransform: Optional[TranslationAfterRotation] = None,
                 modelLandmarks: Optional[Sequence] = None, targetLandmarks: Optional[Sequence] = None,
                 evaluatorUncertainty: float = 1.0, evaluatedPoints: Optional[int] = None,
                 evaluationMode: str = ModelToTargetEvaluation, logFileFittingParameters: Optional[str] = None,
                 rnd: Optional[Random] = None, decimate: Callable = cluster_decimate, verbose: bool = True):
        self.ctx = ctx
        self._kw = dict(model=model, target=target, initialModelParameterTransform=initialModelParameterTransform,
                        modelLandmarks=modelLandmarks, targetLandmarks=targetLandmarks, evaluationMode=evaluationMode,
                        evaluatorUncertainty=evaluatorUncertainty, evaluatedPoints=evaluatedPoints,
                        logFileFittingParameters=logFileFittingParameters, rnd=rnd, decimate=decimate, verbose=verbose)

    def CPD(self, config: CpdConfiguration) -> SimpleRegistrator:
        return SimpleRegistrator(CpdRegistration(self.ctx), config, **self._kw)

    def ICP(self, config: IcpConfiguration) -> SimpleRegistrator:
        return SimpleRegistrator(IcpRegistration(self.ctx), config, **self._kw)


# ------------------------------------------------------------------------------------------------ SimpleModels
@dataclasses.dataclass(frozen=True)
class InvLapKernel:
    scaling: float
    name = "InvLap"

    @property
    def printpars(self) -> str:
        return str(self.scaling)


@dataclasses.dataclass(frozen=True)
class InvLapDotKernel:
    scaling: float
    gamma: float
    name = "InvLapDot"

    @property
    def printpars(self) -> str:
        return f"{self.scaling}_{self.gamma}"


@dataclasses.dataclass(frozen=True)
class GaussKernel:
    scaling: float
    sigma: float
    name = "Gauss"

    @property
    def printpars(self) -> str:
        return f"{self.scaling}_{self.sigma}"


@dataclasses.dataclass(frozen=True)
class GaussMixKernel:
    name = "GaussMix"
    printpars = ""


@dataclasses.dataclass(frozen=True)
class GaussDotKernel:
    scaling: float
    sigma: float
    name = "GaussDot"

    @property
    def printpars(self) -> str:
        return f"{self.scaling}_{self.sigma}"


@dataclasses.dataclass(frozen=True)
class GaussMirrorKernel:
    scaling: float
    sigma: float
    name = "GaussMirror"

    @property
    def printpars(self) -> str:
        return f"{self.scaling}_{self.sigma}"


class SimpleTriangleModels3D:
    """G/simple/SimpleModels.scala:52-75: one call from a kernel choice to a model, built in HBM."""

    @staticmethod
    def create(ctx: Context, reference: TriangleMesh3D, kernelSelect, relativeTolerance: float = 0.01, maxRank: int = 0):
        g = GPMMTriangleMesh3D(ctx, reference.points, relativeTolerance=relativeTolerance, maxRank=maxRank, cells=reference.cells)
        if isinstance(kernelSelect, InvLapKernel):
            return g.InverseLaplacian(scaling=kernelSelect.scaling)
        if isinstance(kernelSelect, InvLapDotKernel):
            return g.InverseLaplacianDot(scaling=kernelSelect.scaling, gamma=kernelSelect.gamma)
        if isinstance(kernelSelect, GaussKernel):
            return g.Gaussian(sigma=kernelSelect.sigma, scaling=kernelSelect.scaling)
        if isinstance(kernelSelect, GaussMixKernel):
            return g.AutomaticGaussian()
        if isinstance(kernelSelect, GaussDotKernel):
            return g.GaussianDot(sigma=kernelSelect.sigma, scaling=kernelSelect.scaling)
        if isinstance(kernelSelect, GaussMirrorKernel):
            return g.GaussianSymmetry(sigma=kernelSelect.sigma, scaling=kernelSelect.scaling)
        raise TypeError(f"unknown kernel choice {kernelSelect!r}")
