//> using scala "3.3"
//> using dep "ch.unibas.cs.gravis::gingr:1.0-RC1"
//> using dep "ch.unibas.cs.gravis::scalismo:1.0-RC1"
/*
 * True JVM baseline + golden-vector dump for a maintainer WITH a JVM (scala-cli run jvm/tools/RefBench.scala -- <dir>).
 * NOT RUN IN THIS REPOSITORY'S IMAGE: there is no JVM here and on the GPU boxes (SURVEY.md 8c), which is why the oracle
 * under oracle/ is "parity unpinned".  This script (a) times CpdRegistration.update on the femur fixture -- the number
 * SURVEY.md 8(d) calls "the true JVM baseline" -- and (b) writes the states of the first five updates as JSON so they can
 * be dropped into tests/golden/ and pin the oracle for real.
 *
 * <dir> must hold femur.stl and femur_target.stl (examples/data/femur of the reference).
 */
import java.io.{File, PrintWriter}

import gingr.api.registration.config.{CpdConfiguration, CpdRegistration, CpdRegistrationState}
import gingr.api.{GeneralRegistrationState, RigidTransforms}
import gingr.api.gpmm.GPMMTriangleMesh3D
import scalismo.io.MeshIO
import scalismo.utils.Random.implicits._

@main def RefBench(dir: String): Unit =
  val reference = MeshIO.readMesh(new File(dir, "femur.stl")).get
  val target = MeshIO.readMesh(new File(dir, "femur_target.stl")).get
  // the femur demo kernel (examples/DemoHelper/DemoDatasetLoader.scala:113-114), truncated like the armadillo demo
  val model = GPMMTriangleMesh3D(reference, relativeTolerance = 0.01).Gaussian(sigma = 70.0, scaling = 50.0).truncate(100)
  val config = CpdConfiguration(maxIterations = 100, w = 0.0)
  val algorithm = new CpdRegistration()
  val general = GeneralRegistrationState(model, target, transform = RigidTransforms)
  var state: CpdRegistrationState = algorithm.initializeState(general, config)

  val out = new PrintWriter(new File(dir, "refbench_states.json"))
  out.println("[")
  for it <- 1 to 5 do
    state = algorithm.update(state, probabilistic = false)
    val g = state.general
    val fit = g.fit.pointSet.points.map(p => s"[${p.x},${p.y},${p.z}]").mkString("[", ",", "]")
    out.println(s"""{"iteration": $it, "sigma2": ${g.sigma2}, "alpha": ${g.modelParameters.shape.parameters.toArray.mkString("[", ",", "]")},""" +
      s""" "euler": [${g.modelParameters.pose.rotation.angles.phi},${g.modelParameters.pose.rotation.angles.theta},${g.modelParameters.pose.rotation.angles.psi}],""" +
      s""" "translation": [${g.modelParameters.pose.translation.x},${g.modelParameters.pose.translation.y},${g.modelParameters.pose.translation.z}],""" +
      s""" "fit": $fit}${if it < 5 then "," else ""}""")
  out.println("]")
  out.close()

  // timing: the stock path at 1 622 x 1 622 (every update materialises P four times, CPD.scala:54-77)
  val n = 5
  val t0 = System.nanoTime()
  for _ <- 0 until n do state = algorithm.update(state, probabilistic = false)
  val dt = (System.nanoTime() - t0) / 1e9 / n
  println(s"""{"kind": "reference", "config": "femur 1622 <-> 1622, rank ${model.rank}", "s_per_update": $dt, "cores": ${Runtime.getRuntime.availableProcessors()}}""")
