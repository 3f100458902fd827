"""Post-processing helpers of the reference's probabilistic demos (examples/DemoPosteriorVisualizationFemur.scala): the chain's JSON
log back into shapes, per-vertex variance maps of the sampled shapes, and the progress call-back.

  LogHelper.samplesFromLog / logSamples2shapes                      G/api/helper/LogHelper.scala:28-56
  JSONStateLogger.loadLog / jsonFormatToModelFittingParameters / getBestStateFromLog   G/api/sampling/loggers/JSONStateLogger.scala:205-236
  PosteriorHelper.computeDistanceMapFromMeshesTotal / ...Normal      G/api/helper/PosteriorHelper.scala:26-80
  CallBackFunctions.SimpleLogger                                     G/api/helper/CallBackFunctions.scala:23-44

The shapes are instantiated on the GPU (one model upload, one basis sweep per sample); the variance maps are O(samples x vertices)
reductions of arrays that are already on the host."""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import io as gio
from .api import Context, DeviceModel, EulerAngles, ModelFittingParameters
from .sampling import RegistrationComparison, TriangleMesh3D


# ---------------------------------------------------------------------------------------------------------------- log
def loadLog(path: str) -> List[gio.JsonLogEntry]:
    """JSONStateLogger.loadLog (:208-214)"""
    print(f"Loading JSON log file: {path}")
    return gio.read_log(path)


def jsonFormatToModelFittingParameters(e: gio.JsonLogEntry) -> ModelFittingParameters:
    """JSONStateLogger.jsonFormatToModelFittingParameters (:216-229)"""
    if len(e.rotation) != 3 or len(e.rotationCenter) != 3:
        raise ValueError("requirement failed")                       # a rejected entry carries no parameters
    return ModelFittingParameters(scale=float(e.scaling), translation=tuple(float(v) for v in e.translation),
                                  rotation=EulerAngles(*[float(v) for v in e.rotation]),
                                  center=tuple(float(v) for v in e.rotationCenter), shape=np.asarray(e.modelParameters, dtype=np.float64))


def getBestStateFromLog(log: Sequence[gio.JsonLogEntry]) -> gio.JsonLogEntry:
    """:231-234: sortBy(logvalue("product")).reverse.head -- the stable sort reversed, so of equal values the LAST one wins."""
    best = None
    for e in log:
        if best is None or e.logvalue["product"] >= best.logvalue["product"]:
            best = e
    if best is None:
        raise ValueError("empty log")
    return best


def samplesFromLog(log: Sequence[gio.JsonLogEntry], takeEveryN: int = 50, total: int = 100, burnIn: int = 0
                   ) -> List[Tuple[gio.JsonLogEntry, int]]:
    """LogHelper.samplesFromLog (:28-44): every takeEveryN-th entry from burnIn up to min(len, total), each replaced by the last
    ACCEPTED entry at or before it (a rejected entry means the chain stayed where it was)."""
    def last_accepted(i: int) -> int:
        while not log[i].status:
            i -= 1
            if i < 0:
                raise IndexError("no accepted entry before the requested sample")
        return i
    print("Log length: " + str(len(log)))
    idx = [last_accepted(i) for i in range(burnIn, min(len(log), total), takeEveryN)]
    out = [(log[i], i) for i in idx]
    return out[:min(total, len(out))]


def logSamples2shapes(ctx: Context, model, log: Sequence[gio.JsonLogEntry]) -> List[np.ndarray]:
    """LogHelper.logSamples2shapes (:46-55): ModelFittingParameters.modelInstanceShapePoseScale per entry, on the device."""
    dm = DeviceModel(ctx, model)
    try:
        shapes = []
        for e in log:
            mp = jsonFormatToModelFittingParameters(e)
            shapes.append(dm.instance(mp.shape, [mp.rotation.phi, mp.rotation.theta, mp.rotation.psi], mp.center, mp.translation,
                                      mp.scale))
        return shapes
    finally:
        dm.close()


# ---------------------------------------------------------------------------------------------------------------- posterior maps
def vertex_normals(vertices: np.ndarray, cells: np.ndarray) -> np.ndarray:
    """scalismo TriangleMesh.vertexNormals [SCALISMO-RECALL]: mean of the unit normals (b - a) x (c - a) of the adjacent cells
    (same rule as gingr_amd/csrc/surface.hip: vertex_normals_kernel)."""
    v, c = np.asarray(vertices, dtype=np.float64), np.asarray(cells, dtype=np.int64)
    n = np.cross(v[c[:, 1]] - v[c[:, 0]], v[c[:, 2]] - v[c[:, 0]])
    n = n / np.sqrt((n * n).sum(1))[:, None]
    acc, cnt = np.zeros_like(v), np.zeros(v.shape[0])
    for k in range(3):
        np.add.at(acc, c[:, k], n)
        np.add.at(cnt, c[:, k], 1.0)
    return acc / np.maximum(cnt, 1.0)[:, None]


def _unit(x: np.ndarray) -> np.ndarray:
    return x / np.sqrt((x * x).sum(-1))[..., None]


def computeDistanceMapFromMeshesTotal(meshes: Sequence[np.ndarray]) -> np.ndarray:
    """PosteriorHelper.computeDistanceMapFromMeshesTotal (:26-47): per vertex the trace of the sample covariance of its
    positions over the meshes (divisor samples - 1)."""
    X = np.stack([np.asarray(m, dtype=np.float64) for m in meshes])              # (S, M, 3)
    S = X.shape[0]
    mean = X.sum(axis=0) * (1.0 / S)
    return ((X - mean) ** 2).sum(axis=(0, 2)) * (1.0 / (S - 1))


def computeDistanceMapFromMeshesNormal(meshes: Sequence[np.ndarray], ref: TriangleMesh3D, sumNormals: bool = True) -> np.ndarray:
    """PosteriorHelper.computeDistanceMapFromMeshesNormal (:49-79): per vertex the variance of the samples along a normal -- the
    mean of the samples' unit vertex normals (not re-normalised, as in the reference) or the reference mesh's unit normal."""
    X = np.stack([np.asarray(m, dtype=np.float64) for m in meshes])
    S = X.shape[0]
    mean = X.sum(axis=0) * (1.0 / S)
    if sumNormals:
        n = sum(_unit(vertex_normals(m, ref.cells)) for m in X) * (1.0 / S)
    else:
        n = _unit(vertex_normals(ref.points, ref.cells))
    proj = ((X - mean) * n[None]).sum(-1)
    return (proj ** 2).sum(axis=0) * (1.0 / (S - 1))


# ---------------------------------------------------------------------------------------------------------------- call-back
class SimpleLogger:
    """CallBackFunctions.SimpleLogger (:23-44): every printUpdateFrequency-th state prints the acceptance statistics, flushes the
    JSON log and scores the current fit against the target (boundary-aware, on the GPU)."""

    def __init__(self, ctx: Context, jsonLogger=None, printUpdateFrequency: int = 100, verbose: bool = True):
        self.ctx, self.jsonLogger, self.printUpdateFrequency, self.verbose = ctx, jsonLogger, int(printUpdateFrequency), verbose
        self.counter = 0
        self.history: List[Tuple[int, float, float]] = []

    def __call__(self, sample) -> None:
        self.logState(sample)

    def logState(self, sample) -> None:
        self.counter += 1
        if self.counter % self.printUpdateFrequency == 0 and self.counter > 1:
            if self.jsonLogger is not None:
                if self.verbose:
                    self.jsonLogger.printAcceptInfo()
                if getattr(self.jsonLogger, "filePath", None):
                    self.jsonLogger.writeLog()
            g = sample.general
            if getattr(g.model, "cells", None) is not None and g.targetCells is not None:
                avg, mx = RegistrationComparison(self.ctx, self.verbose).evaluateReconstruction2GroundTruthBoundaryAware(
                    "", TriangleMesh3D(g.fit, g.model.cells), TriangleMesh3D(g.target, g.targetCells))
                self.history.append((self.counter, avg, mx))
