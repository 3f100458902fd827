//> using scala "3.3"
//> using dep "ch.unibas.cs.gravis::gingr:1.0-RC1"
//> using dep "ch.unibas.cs.gravis::scalismo:1.0-RC1"
/*
 * Reference dump for a maintainer WITH a JVM:   scala-cli run jvm/tools/RefDump.scala -- <data dir> <out dir> [--hot-path-only]
 * (--hot-path-only: the deterministic CPD and point-cloud-ICP cases only -- enough to pin the hot path; the surface-ICP case adds
 *  bit-sensitive accept / reject decisions that can only be agreement-rate-checked, INTEGRATION.md section 5)
 *
 * NOT RUN IN THIS REPOSITORY'S IMAGE (no JVM here or on the GPU boxes, SURVEY.md 8c) -- which is exactly why oracle/ says
 * "parity unpinned".  Running this once against the real GiNGR 1.0-RC1 / scalismo 1.0-RC1 produces the golden vectors that pin it:
 * copy <out dir>/*.json to tests/golden/reference/ and tests/test_reference_golden.py starts comparing the oracle (CPU) and the
 * HIP path (GPU) with them; no code change needed.  Until then that test SKIPS with the text "parity unpinned".
 *
 * What is dumped follows the path the registration really takes: every iteration is GingrGeneratorWrapper.propose
 * (gingr/api/sampling/generators/GingrGeneratorWrapper.scala:28-39) = algorithm.update followed by
 * fit = modelInstanceShapePoseScale and iteration + 1 -- NOT a bare algorithm.update (the fit would never be refreshed).
 * Everything a replay needs is in the file: the model (reference points in scalismo's vertex order, cells, mean, variance,
 * basis), the target, the configuration, the initial state, and per iteration the state plus the correspondence-level
 * quantities (CPD: P1 and the correspondence points of the PRE-update state; ICP: closest-point ids, surface weights).
 *
 * Schema "gingr-refdump-1" (tests/test_reference_golden.py reads exactly this):
 *  { schema, case, versions, config{algorithm,w,lambda,initialSigma,endSigma,maxIterations,method,globalTransformation,stepLength,
 *    useLandmarks}, model{reference[M][3],cells[T][3],mean[3M],variance[r],basis[3M][r]}, target{points[N][3],cells[T][3]},
 *    landmarks{pids[],points[][3],covs[][9]}, initial{sigma2,alpha[r],euler[3],center[3],translation[3],scale},
 *    iterations[{iteration,status,sigma2,alpha[r],euler[3],center[3],translation[3],scale,fit[M][3],
 *                P1[M]?,correspondence[M][3]?,closest_ids[M]?,surface_weights[M]?}], seconds_per_update }
 *
 * <data dir> must hold femur.stl, femur_target.stl, femur.json, femur_target.json (examples/data/femur of the reference).
 */
import java.io.{File, PrintWriter}

import gingr.api.registration.config.*
import gingr.api.sampling.generators.GeneratorWrapperDeterministic
import gingr.api.{GeneralRegistrationState, GingrAlgorithm, GingrRegistrationState, GlobalTranformationType, NoTransforms, RigidTransforms}
import gingr.api.gpmm.GPMMTriangleMesh3D
import scalismo.common.PointId
import scalismo.geometry.{_3D, Landmark, Point}
import scalismo.io.{LandmarkIO, MeshIO}
import scalismo.mesh.TriangleMesh
import scalismo.statisticalmodel.PointDistributionModel
import scalismo.utils.Random.implicits.*

object Json:
  def arr(xs: Iterable[Double]): String = xs.map(d => if d.isNaN || d.isInfinite then "null" else d.toString).mkString("[", ",", "]")
  def ints(xs: Iterable[Int]): String = xs.mkString("[", ",", "]")
  def pts(ps: Iterable[Point[_3D]]): String = ps.map(p => s"[${p.x},${p.y},${p.z}]").mkString("[", ",", "]")
  def cells(m: TriangleMesh[_3D]): String =
    m.triangulation.triangles.map(t => s"[${t.ptId1.id},${t.ptId2.id},${t.ptId3.id}]").mkString("[", ",", "]")

def modelJson(model: PointDistributionModel[_3D, TriangleMesh]): String =
  val b = model.gp.basisMatrix
  val rows = (0 until b.rows).map(i => Json.arr((0 until b.cols).map(j => b(i, j)))).mkString("[", ",", "]")
  s"""{"reference": ${Json.pts(model.reference.pointSet.points.toSeq)}, "cells": ${Json.cells(model.reference)},
     | "mean": ${Json.arr(model.gp.meanVector.toArray)}, "variance": ${Json.arr(model.gp.variance.toArray)}, "basis": $rows}""".stripMargin

def stateJson(g: GeneralRegistrationState, extra: String): String =
  val mp = g.modelParameters
  val a = mp.pose.rotation.angles; val c = mp.pose.rotation.center; val t = mp.pose.translation
  s"""{"iteration": ${g.iteration}, "status": "${g.status}", "sigma2": ${g.sigma2}, "alpha": ${Json.arr(mp.shape.parameters.toArray)},
     | "euler": [${a.phi},${a.theta},${a.psi}], "center": [${c.x},${c.y},${c.z}], "translation": [${t.x},${t.y},${t.z}],
     | "scale": ${mp.scale.s}, "fit": ${Json.pts(g.fit.pointSet.points.toSeq)}$extra}""".stripMargin

def dump[S <: GingrRegistrationState[S]](out: File, caseName: String, configJson: String, algorithm: GingrAlgorithm[S, ?], init: S,
                                         nIterations: Int, extras: S => String): Unit =
  val w = new PrintWriter(out)
  val g0 = init.general
  val lms = g0.landmarkCorrespondences
  w.println(s"""{"schema": "gingr-refdump-1", "case": "$caseName", "versions": "gingr 1.0-RC1, scalismo 1.0-RC1",""")
  w.println(s""" "config": $configJson,""")
  w.println(s""" "model": ${modelJson(g0.model)},""")
  w.println(s""" "target": {"points": ${Json.pts(g0.target.pointSet.points.toSeq)}, "cells": ${Json.cells(g0.target)}},""")
  w.println(s""" "landmarks": {"pids": ${Json.ints(lms.map(_._1.id))}, "points": ${Json.pts(lms.map(_._2))},
               | "covs": ${lms.map(l => Json.arr(l._3.cov.t.toArray)).mkString("[", ",", "]")}},""".stripMargin)
  w.println(s""" "initial": ${stateJson(g0, "")},""")
  // the path the registration takes: propose = update, then fit refresh and iteration + 1 (GingrGeneratorWrapper.scala:28-39)
  val wrapper = GeneratorWrapperDeterministic[S]((s: S, p: Boolean) => algorithm.update(s, p), algorithm.name)
  var state = init
  var seconds = 0.0
  w.println(""" "iterations": [""")
  for it <- 1 to nIterations do
    val pre = extras(state) // correspondence-level quantities of the state the update starts from
    val t0 = System.nanoTime()
    state = wrapper.propose(state)
    seconds += (System.nanoTime() - t0) / 1e9
    w.println("  " + stateJson(state.general, pre) + (if it < nIterations then "," else ""))
  w.println(" ],")
  w.println(s""" "seconds_per_update": ${seconds / nIterations}, "cores": ${Runtime.getRuntime.availableProcessors()}}""")
  w.close()
  println(s"wrote $out  (${seconds / nIterations} s per update)")

@main def RefDump(dataDir: String, outDir: String, flags: String*): Unit =
  val hotPathOnly = flags.contains("--hot-path-only")
  new File(outDir).mkdirs()
  val reference = MeshIO.readMesh(new File(dataDir, "femur.stl")).get
  val target = MeshIO.readMesh(new File(dataDir, "femur_target.stl")).get
  val refLm = LandmarkIO.readLandmarksJson3D(new File(dataDir, "femur.json")).get
  val tarLm = LandmarkIO.readLandmarksJson3D(new File(dataDir, "femur_target.json")).get
  // the femur demo kernel (examples/DemoHelper/DemoDatasetLoader.scala:113-114), truncated like examples/CreateArmadilloGPMM.scala:9
  val model = GPMMTriangleMesh3D(reference, relativeTolerance = 0.01).Gaussian(sigma = 70.0, scaling = 50.0).truncate(100)

  def general(t: GlobalTranformationType, withLm: Boolean) =
    // companion apply overloads: GeneralRegistrationState.scala:117-146 (modelTranform = None: identity initial pose)
    if withLm then GeneralRegistrationState(model, refLm, target, tarLm, t, None) else GeneralRegistrationState(model, target, t, None)

  // ---- CPD, rigid, w = 0 and w = 0.1, without / with landmarks
  for (w, withLm, tag) <- Seq((0.0, false, "cpd_w0"), (0.1, false, "cpd_w01"), (0.1, true, "cpd_w01_landmarks")) do
    val cfg = CpdConfiguration(maxIterations = 100, w = w, useLandmarkCorrespondence = withLm)
    val alg = new CpdRegistration()
    val init = alg.initializeState(general(RigidTransforms, withLm), cfg)
    dump[CpdRegistrationState](new File(outDir, s"femur_$tag.json"), s"femur_$tag",
      s"""{"algorithm": "cpd", "w": $w, "lambda": 1.0, "initialSigma": null, "globalTransformation": "RigidTransforms", "stepLength": 1.0, "useLandmarks": $withLm}""",
      alg, init, 5,
      s => {
        val p1 = breeze.linalg.sum(s.P, breeze.linalg.Axis._1).toArray
        val corr = alg.getCorrespondence(s).pairs.map(_._2)
        s""", "P1": ${Json.arr(p1)}, "correspondence": ${Json.pts(corr)}"""
      })

  // ---- ICP: point-cloud and surface correspondence, no global transform / rigid
  for (method, tr, tag) <- Seq((PointcloudClosestPoint, NoTransforms, "icp_pointcloud"), (TriangularClosestPoint, RigidTransforms, "icp_surface"))
      if !(hotPathOnly && method == TriangularClosestPoint) do
    val cfg = IcpConfiguration(maxIterations = 10, initialSigma = 100.0, endSigma = 1.0, correspondenceMethod = method, useLandmarkCorrespondence = false)
    val alg = new IcpRegistration()
    val init = alg.initializeState(general(tr, false), cfg)
    dump[IcpRegistrationState](new File(outDir, s"femur_$tag.json"), s"femur_$tag",
      s"""{"algorithm": "icp", "initialSigma": 100.0, "endSigma": 1.0, "maxIterations": 10, "method": "$method", "globalTransformation": "$tr", "stepLength": 1.0, "useLandmarks": false}""",
      alg, init, 3,
      s => {
        val pairs = alg.getCorrespondence(s).pairs
        val accepted = pairs.map(_._1.id).toSet
        val ids = pairs.map(p => s.general.target.pointSet.findClosestPoint(p._2).id.id)
        val weights = (0 until s.general.fit.pointSet.numberOfPoints).map(i => if accepted(i) then 1.0 else 0.0)
        s""", "closest_ids": ${Json.ints(ids)}, "correspondence_pids": ${Json.ints(pairs.map(_._1.id))}, "correspondence": ${Json.pts(pairs.map(_._2))}, "surface_weights": ${Json.arr(weights)}"""
      })
