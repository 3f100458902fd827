/*
 * HIP-backed CPD / ICP plugins for GiNGR: the reference-side binding a maintainer adds to drop the MI355X update path
 * in under the existing `run` / `update` API.  NOT COMPILED IN THIS REPOSITORY'S IMAGE (no JVM); see INTEGRATION.md.
 *
 * Lives in package gingr.api.registration.config next to CPD.scala / ICP.scala; it must be inside `gingr.api` because
 * the state updaters it needs (updateTranslation, updateRotation, updateScaling, updateShapeParameters, updateSigma2)
 * are `private[api]` (gingr/api/RegistrationState.scala:46-58).
 *
 * What changes w.r.t. the stock plugins:
 *  - HipCpdRegistrationState has NO `P` member (CPD.scala:54-75 builds an M x N DenseMatrix on every case-class copy) and the
 *    stock state class is never constructed, not even for the initial sigma2 (computed natively);
 *  - `update` (GingrAlgorithm.scala:192-254) is overridden: ONE native call per iteration;
 *  - getCorrespondence / getUncertainty / updateSigma2 stay available (served from one native affinity evaluation per
 *    state) so that computePosterior-based callers (GeneratorWrapperStochastic.logTransitionProbability) keep working.
 * `run`, the Metropolis-Hastings chain, loggers and evaluators are untouched.
 */
package gingr.api.registration.config

import breeze.linalg.{DenseMatrix, DenseVector}
import gingr.api._
import gingr.hip.GingrHipNative
import scalismo.common.PointId
import scalismo.geometry.{_3D, EuclideanVector, Point}
import scalismo.mesh.TriangleMesh
import scalismo.statisticalmodel.{MultivariateNormalDistribution, PointDistributionModel}
import scalismo.utils.Random

/** Flat-array views of scalismo objects in the layout of include/gingr_hip.h (x1x,x1y,x1z,x2x,...). */
private[config] object HipLayout {
  def points(ps: Iterator[Point[_3D]], n: Int): Array[Double] = {
    val a = new Array[Double](3 * n)
    var i = 0
    ps.foreach { p => a(i) = p.x; a(i + 1) = p.y; a(i + 2) = p.z; i += 3 }
    a
  }
  def mesh(m: TriangleMesh[_3D]): Array[Double] = points(m.pointSet.points, m.pointSet.numberOfPoints)
}

/** Device-resident model + fitter for one (model, target) pair; one instance per chain (not thread-safe). */
final class HipSession(device: Int) extends AutoCloseable {
  private val ctx = GingrHipNative.ctxCreate(device)
  require(ctx != 0L, "gingr_ctx_create failed: no usable GPU")
  private var model = 0L
  private var fitter = 0L
  private var boundModel: AnyRef = null
  private var boundTarget: AnyRef = null

  private def check(rc: Int, what: String): Unit =
    if (rc != 0) throw new RuntimeException(s"$what failed (gingr_status $rc): ${GingrHipNative.lastError(ctx)}")

  def bind(general: GeneralRegistrationState, useLandmarks: Boolean): Unit = {
    if (boundModel ne general.model) {
      close0()
      val pdm = general.model
      val m = pdm.reference.pointSet.numberOfPoints
      // gp.basisMatrix is a Breeze DenseMatrix (column-major data, 3M rows); meanVector / variance are DenseVectors
      model = GingrHipNative.modelUpload(ctx, m.toLong, pdm.rank, HipLayout.mesh(pdm.reference), pdm.gp.meanVector.toArray,
        pdm.gp.basisMatrix.toDenseMatrix.data, pdm.gp.variance.toArray, 0L, m.toLong)
      require(model != 0L, s"gingr_model_upload failed: ${GingrHipNative.lastError(ctx)}")
      fitter = GingrHipNative.fitterCreate(ctx, model)
      require(fitter != 0L, s"gingr_fitter_create failed: ${GingrHipNative.lastError(ctx)}")
      boundModel = pdm
      boundTarget = null
    }
    if (boundTarget ne general.target) {
      check(GingrHipNative.fitterSetTarget(fitter, flat(general.target)), "gingr_fitter_set_target")
      boundTarget = general.target
    }
    val lms = if (useLandmarks) general.landmarkCorrespondences else IndexedSeq()
    check(
      GingrHipNative.fitterSetLandmarks(fitter, lms.map(_._1.id).toArray, HipLayout.points(lms.iterator.map(_._2), lms.size),
        lms.flatMap(l => l._3.cov.t.toArray).toArray), // row-major 3x3 per landmark
      "gingr_fitter_set_landmarks")
    val gt = general.globalTransformation match {
      case NoTransforms         => 0
      case RigidTransforms      => 1
      case SimilarityTransforms => 2
    }
    check(GingrHipNative.fitterSetOptions(fitter, gt, general.stepLength), "gingr_fitter_set_options")
  }

  /** Pushes (alpha, pose, sigma2), runs ONE update, pulls the result.  Returns (alpha, pose11, status). */
  /** push the parameters of `general` (the device re-instantiates the fit) and run a query on the fitter */
  def withState(general: GeneralRegistrationState, run: Long => Int): Unit = {
    val mp = general.modelParameters
    val a = mp.pose.rotation.angles
    val c = mp.pose.rotation.center
    val t = mp.pose.translation
    val pose = Array(a.phi, a.theta, a.psi, c.x, c.y, c.z, t.x, t.y, t.z, mp.scale.s, general.sigma2)
    val status = if (general.status == FittingStatuses.ModelFlexibilityError) 3 else 0
    bind(general, useLandmarks = false)
    check(GingrHipNative.fitterSetState(fitter, mp.shape.parameters.toArray, pose, general.iteration, status), "gingr_fitter_set_state")
    check(run(fitter), "gingr_fitter query")
  }

  def updateOnce(general: GeneralRegistrationState, run: Long => Int): (Array[Double], Array[Double], Int) = {
    val mp = general.modelParameters
    val a = mp.pose.rotation.angles
    val c = mp.pose.rotation.center
    val t = mp.pose.translation
    val pose = Array(a.phi, a.theta, a.psi, c.x, c.y, c.z, t.x, t.y, t.z, mp.scale.s, general.sigma2)
    val status = if (general.status == FittingStatuses.ModelFlexibilityError) 3 else 0
    check(GingrHipNative.fitterSetState(fitter, mp.shape.parameters.toArray, pose, general.iteration, status), "gingr_fitter_set_state")
    check(run(fitter), "gingr_fitter_update")
    val alpha = new Array[Double](general.model.rank)
    val poseOut = new Array[Double](11)
    val iterStatus = new Array[Int](2)
    check(GingrHipNative.fitterGetState(fitter, alpha, poseOut, iterStatus, null), "gingr_fitter_get_state")
    (alpha, poseOut, iterStatus(1))
  }

  // flattened copies of the last meshes seen (a TriangleMesh is immutable: identity is enough); the target of a registration
  // never changes and a state's fit is asked for several times (getCorrespondence, getUncertainty, updateSigma2)
  private var flatCache: List[(AnyRef, Array[Double])] = Nil
  private def flat(m: TriangleMesh[_3D]): Array[Double] =
    flatCache.find(_._1 eq m).map(_._2).getOrElse {
      val a = HipLayout.mesh(m)
      flatCache = ((m, a) :: flatCache).take(4)
      a
    }

  def cpdStats(fit: TriangleMesh[_3D], target: TriangleMesh[_3D], sigma2: Double, w: Double)
    : (Array[Double], Array[Double], Double) = {
    val m = fit.pointSet.numberOfPoints
    val p1 = new Array[Double](m); val px = new Array[Double](3 * m); val sc = new Array[Double](6)
    check(GingrHipNative.cpdStats(ctx, flat(fit), flat(target), sigma2, w, null, p1, px, null, sc), "gingr_cpd_stats")
    (p1, px, sc(4))
  }

  /** sum_ij |x_j - y_i|^2 / (3 M N) (CpdRegistrationState.computeInitialSigma2, CPD.scala:81-90) on the GPU */
  def initialSigma2(reference: TriangleMesh[_3D], target: TriangleMesh[_3D]): Double = {
    val out = new Array[Double](1)
    check(GingrHipNative.cpdInitialSigma2(ctx, flat(reference), flat(target), out), "gingr_cpd_initial_sigma2")
    out(0)
  }

  def retryCounter: Int = {
    if (fitter == 0L) 10
    else {
      val out = new Array[Int](1)
      check(GingrHipNative.fitterRetryCounter(fitter, -1, out), "gingr_fitter_retry_counter")
      out(0)
    }
  }

  private var boundMeshes: (AnyRef, AnyRef) = (null, null)
  /** TriangleMesh.triangulation of both meshes as flat id triples (surface ICP, gingr_fitter_set_meshes). */
  def bindMeshes(general: GeneralRegistrationState): Unit = {
    val key = (general.model.reference.triangulation, general.target.triangulation)
    if ((boundMeshes._1 ne key._1) || (boundMeshes._2 ne key._2)) {
      def flat(m: TriangleMesh[_3D]): Array[Int] = m.triangulation.triangles.flatMap(t => Seq(t.ptId1.id, t.ptId2.id, t.ptId3.id)).toArray
      check(GingrHipNative.fitterSetMeshes(fitter, flat(general.model.reference), flat(general.target)), "gingr_fitter_set_meshes")
      boundMeshes = key
    }
  }

  def setDirection(reversed: Boolean): Unit =
    check(GingrHipNative.fitterSetCorrespondenceDirection(fitter, if (reversed) 1 else 0), "gingr_fitter_set_correspondence_direction")

  def nn(fit: TriangleMesh[_3D], target: TriangleMesh[_3D]): Array[Int] = {
    val idx = new Array[Int](fit.pointSet.numberOfPoints)
    check(GingrHipNative.nn(ctx, flat(fit), flat(target), idx, null, null), "gingr_nn")
    idx
  }

  private def close0(): Unit = {
    if (fitter != 0L) { GingrHipNative.fitterDestroy(fitter); fitter = 0L }
    if (model != 0L) { GingrHipNative.modelDestroy(model); model = 0L }
  }
  override def close(): Unit = { close0(); GingrHipNative.ctxDestroy(ctx) }
}

/** Applies a native result to the immutable Scala state exactly like GingrAlgorithm.update does (:239-246). */
private[config] object HipStateUpdate {
  def apply(general: GeneralRegistrationState, alpha: Array[Double], pose: Array[Double], status: Int): GeneralRegistrationState = {
    if (status == 3) general.updateStatus(FittingStatuses.ModelFlexibilityError)
    else
      general
        .updateTranslation(EuclideanVector(pose(6), pose(7), pose(8)))
        .updateRotation(EulerRotation(EulerAngles(pose(0), pose(1), pose(2)), Point(pose(3), pose(4), pose(5))))
        .updateScaling(ScaleParameter(pose(9)))
        .updateShapeParameters(ShapeParameters(DenseVector(alpha)))
        .updateSigma2(pose(10))
    // fit and iteration are refreshed by GingrGeneratorWrapper.propose exactly as for the stock plugins
  }
}

// ------------------------------------------------------------------------------------------------------------ CPD
case class HipCpdRegistrationState(general: GeneralRegistrationState, config: CpdConfiguration)
    extends GingrRegistrationState[HipCpdRegistrationState] {
  override def updateGeneral(update: GeneralRegistrationState): HipCpdRegistrationState = this.copy(general = update)
}

class HipCpdRegistration(device: Int = 0) extends GingrAlgorithm[HipCpdRegistrationState, CpdConfiguration] with AutoCloseable {
  private val session = new HipSession(device)
  // one streaming affinity evaluation per state serves getCorrespondence, getUncertainty and updateSigma2
  private var statsOf: HipCpdRegistrationState = null
  private var stats: (Array[Double], Array[Double], Double) = null
  private def statsFor(s: HipCpdRegistrationState) = {
    if (statsOf ne s) { stats = session.cpdStats(s.general.fit, s.general.target, s.general.sigma2, s.config.w); statsOf = s }
    stats
  }

  def name = "CPD-HIP"

  override val getCorrespondence: HipCpdRegistrationState => CorrespondencePairs = (s: HipCpdRegistrationState) => {
    val (p1, px, _) = statsFor(s)
    val pts = s.general.fit.pointSet.points.toIndexedSeq
    CorrespondencePairs(pts.indices.map { i =>
      val y = pts(i); val inv = 1.0 / p1(i)
      (PointId(i), Point(y.x + (px(3 * i) * inv - y.x), y.y + (px(3 * i + 1) * inv - y.y), y.z + (px(3 * i + 2) * inv - y.z)))
    })
  }
  override val getUncertainty: (PointId, HipCpdRegistrationState) => MultivariateNormalDistribution =
    (id: PointId, s: HipCpdRegistrationState) =>
      MultivariateNormalDistribution(DenseVector.zeros[Double](3),
        DenseMatrix.eye[Double](3) * s.general.sigma2 * s.config.lambda * (1.0 / statsFor(s)._1(id.id)))
  override def updateSigma2(current: HipCpdRegistrationState): Double = statsFor(current)._3

  /** The semantics of the stock companion's apply (CPD.scala:92-102): sigma2 = config.initialSigma, or else
    * sum_ij |x_j - y_i|^2 / (3 M N) over the model MEAN against the target (computeInitialSigma2, :81-90) -- evaluated on the
    * GPU.  The stock state class is never constructed: its eager `P` (CPD.scala:54-75) is a dense M x N matrix and
    * computeInitialSigma2 materialises an M*N Seq, both impossible at 50k x 50k (M*N > Int.MaxValue). */
  override def initializeState(general: GeneralRegistrationState, config: CpdConfiguration): HipCpdRegistrationState = {
    val sigma2 = config.initialSigma.getOrElse(session.initialSigma2(general.model.mean, general.target))
    HipCpdRegistrationState(general.copy(sigma2 = sigma2), config)
  }

  override def update(current: HipCpdRegistrationState, probabilistic: Boolean)(implicit rnd: Random): HipCpdRegistrationState = {
    session.bind(current.general, current.config.useLandmarkCorrespondence)
    val (alpha, pose, status) =
      if (probabilistic) {
        // posterior.sample(): the native side maps rank standard-normal draws of THIS chain's Random through L^-T
        val z = Array.fill(current.general.model.rank)(rnd.scalaRandom.nextGaussian())
        session.updateOnce(current.general, f => GingrHipNative.fitterUpdateCpdSample(f, current.config.w, current.config.lambda, z))
      } else
        session.updateOnce(current.general, f => GingrHipNative.fitterUpdateCpd(f, current.config.w, current.config.lambda, 1))
    // Failure handling is decided on the device exactly as in GingrAlgorithm.update (:194-210,248-251), including the retry
    // counter of the probabilistic proposal: the trait's `retryCounter` is private, so its stand-in lives next to the device
    // state (one fitter = one algorithm instance; GingrHipNative.fitterRetryCounter reads it).  A failed sampled posterior
    // therefore comes back as "state unchanged" (status 0, same parameters) up to 10 times in a row, like the reference.
    current.updateGeneral(HipStateUpdate(current.general, alpha, pose, status))
  }
  /** retryCounter of this instance (GingrAlgorithm.scala:69-70), read back from the device */
  def retryCounter: Int = session.retryCounter
  override def close(): Unit = session.close()
}

/** Multi-GPU from the JVM process itself: the same update on row shards over several devices of one node through the
  * in-library device group (gingr_group_*: one worker thread per device, one-shot all-reduce over peer pointers; no torch, no MPI).
  * `devices` lists one GPU per shard.  The deterministic `run` loop and the stock MH machinery work unchanged; the
  * probabilistic proposal and the transition density are single-GPU features (use HipCpdRegistration for those). */
class HipCpdGroupRegistration(devices: Seq[Int]) extends GingrAlgorithm[HipCpdRegistrationState, CpdConfiguration] with AutoCloseable {
  private val group = GingrHipNative.groupCreate(devices.toArray)
  require(group != 0L, "gingr_group_create failed (devices, peer access)")
  private val single = new HipSession(devices.head) // stateless helpers (initial sigma2, per-state statistics)
  private var boundModel: AnyRef = null
  private var boundTarget: AnyRef = null
  private def check(rc: Int, what: String): Unit =
    if (rc != 0) throw new RuntimeException(s"$what failed (gingr_status $rc): ${GingrHipNative.groupLastError(group)}")

  def name = "CPD-HIP-group"
  private var statsOf: HipCpdRegistrationState = null
  private var stats: (Array[Double], Array[Double], Double) = null
  private def statsFor(s: HipCpdRegistrationState) = {
    if (statsOf ne s) { stats = single.cpdStats(s.general.fit, s.general.target, s.general.sigma2, s.config.w); statsOf = s }
    stats
  }
  override val getCorrespondence: HipCpdRegistrationState => CorrespondencePairs = (s: HipCpdRegistrationState) => {
    val (p1, px, _) = statsFor(s)
    val pts = s.general.fit.pointSet.points.toIndexedSeq
    CorrespondencePairs(pts.indices.map { i =>
      val y = pts(i); val inv = 1.0 / p1(i)
      (PointId(i), Point(y.x + (px(3 * i) * inv - y.x), y.y + (px(3 * i + 1) * inv - y.y), y.z + (px(3 * i + 2) * inv - y.z)))
    })
  }
  override val getUncertainty: (PointId, HipCpdRegistrationState) => MultivariateNormalDistribution =
    (id: PointId, s: HipCpdRegistrationState) =>
      MultivariateNormalDistribution(DenseVector.zeros[Double](3),
        DenseMatrix.eye[Double](3) * s.general.sigma2 * s.config.lambda * (1.0 / statsFor(s)._1(id.id)))
  override def updateSigma2(current: HipCpdRegistrationState): Double = statsFor(current)._3
  override def initializeState(general: GeneralRegistrationState, config: CpdConfiguration): HipCpdRegistrationState = {
    val sigma2 = config.initialSigma.getOrElse(single.initialSigma2(general.model.mean, general.target))
    HipCpdRegistrationState(general.copy(sigma2 = sigma2), config)
  }

  override def update(current: HipCpdRegistrationState, probabilistic: Boolean)(implicit rnd: Random): HipCpdRegistrationState = {
    require(!probabilistic, "the device group runs the deterministic update; use HipCpdRegistration for sampled proposals")
    val general = current.general
    if (boundModel ne general.model) {
      val pdm = general.model
      val m = pdm.reference.pointSet.numberOfPoints
      check(GingrHipNative.groupModelUpload(group, m.toLong, pdm.rank, HipLayout.mesh(pdm.reference), pdm.gp.meanVector.toArray,
        pdm.gp.basisMatrix.toDenseMatrix.data, pdm.gp.variance.toArray), "gingr_group_model_upload")
      boundModel = pdm
      boundTarget = null
    }
    if (boundTarget ne general.target) {
      check(GingrHipNative.groupSetTarget(group, HipLayout.mesh(general.target)), "gingr_group_set_target")
      boundTarget = general.target
    }
    val lms = if (current.config.useLandmarkCorrespondence) general.landmarkCorrespondences else IndexedSeq()
    check(GingrHipNative.groupSetLandmarks(group, lms.map(_._1.id).toArray, HipLayout.points(lms.iterator.map(_._2), lms.size),
      lms.flatMap(l => l._3.cov.t.toArray).toArray), "gingr_group_set_landmarks")
    val gt = general.globalTransformation match {
      case NoTransforms         => 0
      case RigidTransforms      => 1
      case SimilarityTransforms => 2
    }
    check(GingrHipNative.groupSetOptions(group, gt, general.stepLength), "gingr_group_set_options")
    val mp = general.modelParameters
    val a = mp.pose.rotation.angles; val c = mp.pose.rotation.center; val t = mp.pose.translation
    val pose = Array(a.phi, a.theta, a.psi, c.x, c.y, c.z, t.x, t.y, t.z, mp.scale.s, general.sigma2)
    val status = if (general.status == FittingStatuses.ModelFlexibilityError) 3 else 0
    check(GingrHipNative.groupSetState(group, mp.shape.parameters.toArray, pose, general.iteration, status), "gingr_group_set_state")
    check(GingrHipNative.groupUpdateCpd(group, current.config.w, current.config.lambda, 1), "gingr_group_update_cpd_async")
    val alpha = new Array[Double](general.model.rank); val poseOut = new Array[Double](11); val iterStatus = new Array[Int](2)
    check(GingrHipNative.groupGetState(group, alpha, poseOut, iterStatus, null), "gingr_group_get_state")
    current.updateGeneral(HipStateUpdate(general, alpha, poseOut, iterStatus(1)))
  }
  override def close(): Unit = { GingrHipNative.groupDestroy(group); single.close() }
}

// ------------------------------------------------------------------------------------------------------------ ICP
case class HipIcpRegistrationState(general: GeneralRegistrationState, config: IcpConfiguration)
    extends GingrRegistrationState[HipIcpRegistrationState] {
  override def updateGeneral(update: GeneralRegistrationState): HipIcpRegistrationState = this.copy(general = update)
}

/** All three correspondence flavours (ICP.scala:32-44), both correspondence directions. */
class HipIcpRegistration(device: Int = 0) extends GingrAlgorithm[HipIcpRegistrationState, IcpConfiguration] with AutoCloseable {
  private val session = new HipSession(device)
  def name = "ICP-HIP"

  override val getCorrespondence: HipIcpRegistrationState => CorrespondencePairs = (s: HipIcpRegistrationState) => {
    val idx = session.nn(s.general.fit, s.general.target)
    val tp = s.general.target.pointSet.points.toIndexedSeq
    CorrespondencePairs(idx.indices.map(i => (PointId(i), tp(idx(i)))))
  }
  override val getUncertainty: (PointId, HipIcpRegistrationState) => MultivariateNormalDistribution =
    (_: PointId, s: HipIcpRegistrationState) =>
      MultivariateNormalDistribution(DenseVector.zeros[Double](3), DenseMatrix.eye[Double](3) * s.general.sigma2)
  override def updateSigma2(current: HipIcpRegistrationState): Double =
    math.max(current.general.sigma2 - current.config.sigmaStep, current.config.endSigma)

  override def initializeState(general: GeneralRegistrationState, config: IcpConfiguration): HipIcpRegistrationState = {
    HipIcpRegistrationState(IcpRegistrationState(general, config).general, config)
  }

  override def update(current: HipIcpRegistrationState, probabilistic: Boolean)(implicit rnd: Random): HipIcpRegistrationState = {
    session.bind(current.general, current.config.useLandmarkCorrespondence)
    val c = current.config
    if (c.correspondenceMethod != PointcloudClosestPoint) session.bindMeshes(current.general)
    session.setDirection(c.reverseCorrespondenceDirection) // after the target / meshes are bound
    val (alpha, pose, status) =
      if (probabilistic) {
        val z = Array.fill(current.general.model.rank)(rnd.scalaRandom.nextGaussian())
        if (c.correspondenceMethod != PointcloudClosestPoint) {
          session.bindMeshes(current.general)
          session.updateOnce(current.general, f => {
            GingrHipNative.fitterSetSurfaceMethod(f, if (c.correspondenceMethod == AlongNormalClosestPoint) 1 else 0)
            GingrHipNative.fitterUpdateIcpSurfaceSample(f, c.initialSigma, c.endSigma, c.maxIterations, z)
          })
        } else
          session.updateOnce(current.general, f => GingrHipNative.fitterUpdateIcpSample(f, c.initialSigma, c.endSigma, c.maxIterations, z))
      } else if (c.correspondenceMethod != PointcloudClosestPoint) {
        session.bindMeshes(current.general) // triangle lists of model.reference and target, once per (model, target)
        session.updateOnce(current.general, f => {
          GingrHipNative.fitterSetSurfaceMethod(f, if (c.correspondenceMethod == AlongNormalClosestPoint) 1 else 0)
          GingrHipNative.fitterUpdateIcpSurface(f, c.initialSigma, c.endSigma, c.maxIterations, 1)
        })
      } else
        session.updateOnce(current.general, f => GingrHipNative.fitterUpdateIcp(f, c.initialSigma, c.endSigma, c.maxIterations, 1))
    current.updateGeneral(HipStateUpdate(current.general, alpha, pose, status))
  }
  override def close(): Unit = session.close()
}

// ------------------------------------------------------------------------------------------------ GPMM construction
/** GPMMTriangleMesh3D(reference, relativeTolerance).Gaussian / GaussianMixture / AutomaticGaussian
  * (gingr/api/gpmm/GPMMHelper.scala:96-130) with the pivoted Cholesky, the eigen-decomposition and the O(n^2) distance
  * scans on the GPU.  Returns an ordinary scalismo PointDistributionModel (one download of the 3M x r basis); a session
  * that only registers can instead keep the native model handle and never materialise the basis on the JVM. */
object HipGPMM {
  import scalismo.common.DiscreteField
  import scalismo.statisticalmodel.DiscreteLowRankGaussianProcess

  def GaussianMixture(reference: TriangleMesh[_3D], pars: Seq[(Double, Double)], relativeTolerance: Double = 0.01, device: Int = 0)
    : PointDistributionModel[_3D, TriangleMesh] = {
    val ctx = GingrHipNative.ctxCreate(device)
    require(ctx != 0L, "gingr_ctx_create failed: no usable GPU")
    try {
      val m = reference.pointSet.numberOfPoints
      val h = GingrHipNative.gpmmBuildGaussian(ctx, m.toLong, HipLayout.mesh(reference), pars.map(_._1).toArray,
        pars.map(_._2).toArray, relativeTolerance, 0, 0L, 0L)
      require(h != 0L, s"gingr_gpmm_build_gaussian failed: ${GingrHipNative.lastError(ctx)}")
      try {
        val r = GingrHipNative.modelRank(h)
        val basis = new Array[Double](3 * m * r); val variance = new Array[Double](r); val mean = new Array[Double](3 * m)
        val rc = GingrHipNative.modelDownload(ctx, h, null, mean, basis, variance)
        require(rc == 0, s"gingr_model_download failed ($rc): ${GingrHipNative.lastError(ctx)}")
        val gp = new DiscreteLowRankGaussianProcess[_3D, TriangleMesh, EuclideanVector[_3D]](
          reference, DenseVector(mean), DenseVector(variance), new DenseMatrix(3 * m, r, basis)) // column-major, as Breeze
        PointDistributionModel(gp)
      } finally GingrHipNative.modelDestroy(h)
    } finally GingrHipNative.ctxDestroy(ctx)
  }

  def Gaussian(reference: TriangleMesh[_3D], sigma: Double, scaling: Double, relativeTolerance: Double = 0.01)
    : PointDistributionModel[_3D, TriangleMesh] = GaussianMixture(reference, Seq((sigma, scaling)), relativeTolerance)

  def AutomaticGaussian(reference: TriangleMesh[_3D], relativeTolerance: Double = 0.01): PointDistributionModel[_3D, TriangleMesh] = {
    val ctx = GingrHipNative.ctxCreate(0)
    val ext = new Array[Double](2)
    try require(GingrHipNative.pointsetDistanceExtrema(ctx, HipLayout.mesh(reference), ext) == 0)
    finally GingrHipNative.ctxDestroy(ctx)
    val maxDist = ext(0)
    GaussianMixture(reference, Seq((maxDist / 4.0, maxDist / 8.0), (maxDist / 8.0, maxDist / 16.0)), relativeTolerance)
  }
}

// ------------------------------------------------------------------------------------ surface likelihood and metrics
/** IndependentPointDistanceEvaluator (gingr/api/sampling/evaluators/IndependentPointDistanceEvaluator.scala:35-84) with the
  * closest-point scan and the reduction on the GPU.  Drop-in for the case the reference uses, likelihoodModel = Gaussian(0, sdev)
  * (gingr/api/sampling/Evaluator.scala:47) without decimation; the state is evaluated through the session of the algorithm
  * that produced it (model, target and meshes are already resident). */
case class HipIndependentPointDistanceEvaluator[State <: GingrRegistrationState[State]](
  session: HipSession,
  sdev: Double,
  evaluationMode: gingr.api.sampling.evaluators.EvaluationMode
) extends scalismo.sampling.DistributionEvaluator[State]
    with gingr.api.sampling.evaluators.EvaluationCaching[State] {
  import gingr.api.sampling.evaluators._
  private def stats(sample: State, direction: Int): Double = {
    session.bindMeshes(sample.general)
    val out = new Array[Double](4)
    session.withState(sample.general, f =>
      GingrHipNative.fitterSurfaceDistanceStats(f, direction, 0L, null, 0, sdev, out))
    out(3)
  }
  override def computeLogValue(sample: State): Double = evaluationMode match {
    case ModelToTargetEvaluation => stats(sample, 0)
    case TargetToModelEvaluation => stats(sample, 1)
    case SymmetricEvaluation     => 0.5 * stats(sample, 0) + 0.5 * stats(sample, 1)
  }
}

/** RegistrationComparison (gingr/api/helper/RegistrationComparison.scala:22-99) on the GPU. */
object HipRegistrationComparison {
  private def flat(m: TriangleMesh[_3D]): Array[Int] =
    m.triangulation.triangles.flatMap(t => Seq(t.ptId1.id, t.ptId2.id, t.ptId3.id)).toArray
  /** (sum, max, count) of the distances from m1's vertices to the surface of m2 */
  def stats(m1: TriangleMesh[_3D], m2: TriangleMesh[_3D], boundaryAware: Boolean = false, device: Int = 0): (Double, Double, Int) = {
    val ctx = GingrHipNative.ctxCreate(device)
    require(ctx != 0L, "gingr_ctx_create failed: no usable GPU")
    try {
      val out = new Array[Double](4)
      val rc = GingrHipNative.meshDistanceStats(ctx, HipLayout.mesh(m1), HipLayout.mesh(m2), flat(m2), if (boundaryAware) 1 else 0, 0.0, out)
      require(rc == 0, s"gingr_mesh_distance_stats failed: ${GingrHipNative.lastError(ctx)}")
      (out(0), out(1), out(2).toInt)
    } finally GingrHipNative.ctxDestroy(ctx)
  }
  def avgDistance(m1: TriangleMesh[_3D], m2: TriangleMesh[_3D]): Double = { val (s, _, n) = stats(m1, m2); s / n }
  def maxDistance(m1: TriangleMesh[_3D], m2: TriangleMesh[_3D]): Double = stats(m1, m2)._2
  def hausdorffDistance(m1: TriangleMesh[_3D], m2: TriangleMesh[_3D]): Double = math.max(maxDistance(m1, m2), maxDistance(m2, m1))
  def avgDistanceBoundaryAware(m1: TriangleMesh[_3D], m2: TriangleMesh[_3D]): (Double, Double) = {
    val (s, mx, n) = stats(m1, m2, boundaryAware = true); (s / n, mx)
  }
}

// ------------------------------------------------------------------------------------------------ classic CPD (other/)
/** Drop-in for gingr.other.algorithms.cpd.{RigidCPD, AffineCPD, NonRigidCPD} behind CPDFactory.register*
  * (gingr/other/algorithms/cpd/CPDFactory.scala:68-79): Expectation and Maximization run on the GPU (no M x N matrix; blocked
  * Cholesky for the non-rigid M x M system), the Registration loop with its tolerance test (RigidCPD.scala:59-83) stays here. */
final class HipClassicCPD(templatePoints: Seq[scalismo.geometry.Point[_3D]], targetPoints: Seq[scalismo.geometry.Point[_3D]], kind: Int,
                          lambda: Double = 2, beta: Double = 2, w: Double = 0, device: Int = 0) extends AutoCloseable {
  require(0.0 <= w && w <= 1.0); require(beta > 0); require(lambda > 0)
  private def flat(ps: Seq[scalismo.geometry.Point[_3D]]): Array[Double] = ps.flatMap(p => Seq(p.x, p.y, p.z)).toArray
  private val ctx = GingrHipNative.ctxCreate(device)
  require(ctx != 0L, "gingr_ctx_create failed: no usable GPU")
  private val h = GingrHipNative.classicCpdCreate(ctx, kind, flat(templatePoints), flat(targetPoints), lambda, beta, w)
  require(h != 0L, s"gingr_classic_cpd_create failed: ${GingrHipNative.lastError(ctx)}")
  private def sigma2(): Double = { val s = new Array[Double](1); GingrHipNative.classicCpdGet(h, null, s, null, null); s(0) }

  def Registration(max_iteration: Int, tolerance: Double = 0.001): Seq[scalismo.geometry.Point[_3D]] = {
    var i = 0; var converged = false; var current = sigma2()
    while (i < max_iteration && !converged) {
      println(s"CPD, iteration: ${i}, variance: ${current}")
      require(GingrHipNative.classicCpdIterate(h, 1) == 0, GingrHipNative.lastError(ctx))
      val next = sigma2()
      if (math.abs(next - current) < tolerance) { println("Converged"); converged = true } else i += 1
      current = next
    }
    val ty = new Array[Double](3 * templatePoints.length)
    GingrHipNative.classicCpdGet(h, ty, null, null, null)
    ty.grouped(3).map(a => scalismo.geometry.Point(a(0), a(1), a(2))).toIndexedSeq
  }
  override def close(): Unit = { GingrHipNative.classicCpdDestroy(h); GingrHipNative.ctxDestroy(ctx) }
}

// ------------------------------------------------------------------------------------------------ classic ICP baselines (other/)
/** Drop-in for gingr.other.algorithms.icp.RigidICP (RigidICP.scala:24-84) behind ICPFactory.registerRigidly: closest points,
  * the least-squares transform (kind 0 = PoseRegistrator.RigidRegistrator3D, 1 = AffineRegistrator3D) and the move of the template run
  * on the GPU; the Registration loop with its test on the change of the mean distance (:30-55) stays here. */
final class HipRigidICP(templatePoints: Seq[scalismo.geometry.Point[_3D]], targetPoints: Seq[scalismo.geometry.Point[_3D]], kind: Int = 0,
                        device: Int = 0) extends AutoCloseable {
  private def flat(ps: Seq[scalismo.geometry.Point[_3D]]): Array[Double] = ps.flatMap(p => Seq(p.x, p.y, p.z)).toArray
  private val ctx = GingrHipNative.ctxCreate(device)
  require(ctx != 0L, "gingr_ctx_create failed: no usable GPU")
  private val h = GingrHipNative.rigidIcpCreate(ctx, kind, flat(templatePoints), flat(targetPoints))
  require(h != 0L, s"gingr_rigid_icp_create failed: ${GingrHipNative.lastError(ctx)}")

  def Registration(max_iteration: Int, tolerance: Double = 0.001): Seq[scalismo.geometry.Point[_3D]] = {
    var i = 0; var converged = false; var last = 0.0
    val d = new Array[Double](1)
    while (i < max_iteration && !converged) {
      require(GingrHipNative.rigidIcpIterate(h, 1, d) == 0, GingrHipNative.lastError(ctx))
      println(s"ICP, iteration: ${i}, distance: ${d(0)}")
      if (math.abs(d(0) - last) < tolerance) { println("Converged"); converged = true }
      last = d(0); i += 1
    }
    val pts = new Array[Double](3 * templatePoints.length)
    GingrHipNative.rigidIcpGet(h, pts, null)
    pts.grouped(3).map(a => scalismo.geometry.Point(a(0), a(1), a(2))).toIndexedSeq
  }
  override def close(): Unit = { GingrHipNative.rigidIcpDestroy(h); GingrHipNative.ctxDestroy(ctx) }
}

/** Drop-in for gingr.other.algorithms.icp.NonRigidOptimalStepICP_T / _A (NonRigidOptimalStepICP.scala:31-284): per iteration the
  * correspondence of the current template (ClosestPointTriangleMesh3D: the surface query of the GiNGR ICP path, asked for explicit
  * points through gingr_fitter_set_fit_points) and the least-squares step (gingr_nicp_solve) run on the GPU; landmark bookkeeping,
  * edges and the stage / inner loops (:89-121) stay here.  affine = false: N-ICP-T, true: N-ICP-A. */
final class HipNonRigidOptimalStepICP(templateMesh: TriangleMesh[_3D], targetMesh: TriangleMesh[_3D],
                                      templateLandmarks: Seq[scalismo.geometry.Landmark[_3D]], targetLandmarks: Seq[scalismo.geometry.Landmark[_3D]],
                                      gamma: Double = 1.0, affine: Boolean = false, device: Int = 0) extends AutoCloseable {
  require(gamma >= 0)
  private val n = templateMesh.pointSet.numberOfPoints
  private def tri(m: TriangleMesh[_3D]): Array[Int] = m.triangulation.triangles.flatMap(t => Seq(t.ptId1.id, t.ptId2.id, t.ptId3.id)).toArray
  private val commonLmNames = templateLandmarks.map(_.id) intersect targetLandmarks.map(_.id)
  val lmIdsOnTemplate: Array[Int] = commonLmNames.map(name => templateLandmarks.find(_.id == name).get)
    .map(lm => templateMesh.pointSet.findClosestPoint(lm.point).id.id).toArray
  private val UL: Array[Double] = commonLmNames.map(name => targetLandmarks.find(_.id == name).get)
    .map(lm => targetMesh.pointSet.findClosestPoint(lm.point).point).flatMap(p => Seq(p.x, p.y, p.z)).toArray
  // trianglesToEdges (:67-76): unique sorted vertex pairs, flat (p1, p2, p1, p2, ...)
  private val edges: Array[Int] = templateMesh.triangulation.triangles.flatMap { t =>
    val s = t.pointIds.map(_.id).sorted
    Seq((s(0), s(1)), (s(0), s(2)), (s(1), s(2)))
  }.toSet.toArray.flatMap(e => Seq(e._1, e._2))

  private val ctx = GingrHipNative.ctxCreate(device)
  require(ctx != 0L, "gingr_ctx_create failed: no usable GPU")
  // carrier of the correspondence query: a rank-1 model over the template's points (its basis is never used)
  private val model = GingrHipNative.modelUpload(ctx, n.toLong, 1, HipLayout.mesh(templateMesh), new Array[Double](3 * n), new Array[Double](3 * n),
    Array(1.0), 0L, n.toLong)
  require(model != 0L, s"gingr_model_upload failed: ${GingrHipNative.lastError(ctx)}")
  private val fitter = GingrHipNative.fitterCreate(ctx, model)
  private def check(rc: Int, what: String): Unit =
    if (rc != 0) throw new RuntimeException(s"$what failed (gingr_status $rc): ${GingrHipNative.lastError(ctx)}")
  check(GingrHipNative.fitterSetTarget(fitter, HipLayout.mesh(targetMesh)), "gingr_fitter_set_target")
  check(GingrHipNative.fitterSetMeshes(fitter, tri(templateMesh), tri(targetMesh)), "gingr_fitter_set_meshes")
  check(GingrHipNative.fitterSetOptions(fitter, 0, 1.0), "gingr_fitter_set_options")

  private val defaultAlpha: Seq[Double] = Seq.fill(11)(1e1) // (:63-65): the scanLeft chain ends in `.map(_ => 1e1)`

  /** (closest points, weights, mean distance) of the given template points (:118-128) */
  def getClosestPoints(points: Array[Double]): (Array[Double], Array[Double], Double) = {
    val pose = Array(0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, 1.0)
    check(GingrHipNative.fitterSetState(fitter, Array(0.0), pose, 0, 0), "gingr_fitter_set_state")
    check(GingrHipNative.fitterSetFitPoints(fitter, points), "gingr_fitter_set_fit_points")
    check(GingrHipNative.fitterIcpSurfacePhase(fitter, 1.0, 1.0, 1, 0), "gingr_fitter_icp_surface_phase_async")
    val cp = new Array[Double](3 * n); val w = new Array[Double](n)
    check(GingrHipNative.fitterGetSurfaceCorrespondence(fitter, cp, w), "gingr_fitter_get_surface_correspondence")
    var dist = 0.0
    var i = 0
    while (i < n) {
      val dx = cp(3 * i) - points(3 * i); val dy = cp(3 * i + 1) - points(3 * i + 1); val dz = cp(3 * i + 2) - points(3 * i + 2)
      dist += math.sqrt(dx * dx + dy * dy + dz * dz); i += 1
    }
    (cp, w, dist / n)
  }

  /** one iteration: (moved points, mean distance before the move, moved landmark vertices) (:151-190 / :241-283) */
  def Iteration(points: Array[Double], alpha: Double, beta: Double): (Array[Double], Double, Array[Double]) = {
    require(alpha >= 0.0); require(beta >= 0.0)
    val (cp, w, dist) = getClosestPoints(points)
    val out = new Array[Double](3 * n); val lm = new Array[Double](3 * lmIdsOnTemplate.length)
    check(GingrHipNative.nicpSolve(ctx, if (affine) 1 else 0, points, edges, w, cp, lmIdsOnTemplate, UL, alpha, beta, gamma, out,
      if (lm.isEmpty) null else lm), "gingr_nicp_solve")
    (out, dist, lm)
  }

  def Registration(max_iteration: Int, tolerance: Double = 0.001, alpha: Seq[Double] = defaultAlpha, beta: Seq[Double] = defaultAlpha)
      : TriangleMesh[_3D] = {
    require(alpha.length == beta.length)
    var fit = HipLayout.mesh(templateMesh)
    alpha.zip(beta).zipWithIndex.foreach { case ((a, b), j) =>
      var dist = Double.PositiveInfinity
      var i = 0
      while (i < max_iteration && dist >= tolerance) {
        val (ty, d, _) = Iteration(fit, a, b)
        println(s"ICP, iteration: ${j * max_iteration + i}/${max_iteration * alpha.length}, alpha: ${a}, beta: ${b}, average distance to target: ${d}")
        fit = ty; dist = d; i += 1
      }
    }
    val pts = fit.grouped(3).map(a => scalismo.geometry.Point(a(0), a(1), a(2))).toIndexedSeq
    templateMesh.copy(pointSet = scalismo.common.UnstructuredPoints(pts))
  }
  override def close(): Unit = { GingrHipNative.fitterDestroy(fitter); GingrHipNative.modelDestroy(model); GingrHipNative.ctxDestroy(ctx) }
}
