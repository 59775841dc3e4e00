// The C++ pose graph (coxgraph_amd/host/coxgraph_hip_posegraph.hpp) on a graph without registration constraints, so it
// runs without a GPU: prints the optimised poses for tests/test_host_logic.py to compare with coxgraph_amd/posegraph.py.
// With a GPU (argument "gpu") it adds a forced registration constraint between two fused submaps as well.
#include <cstdio>
#include <cstring>

#include "../../coxgraph_amd/host/coxgraph_hip_posegraph.hpp"

using namespace coxgraph_hip;

static Pose4 P(double x, double y, double z, double yaw) {
  Pose4 p;
  p.v[0] = x, p.v[1] = y, p.v[2] = z, p.v[3] = yaw;
  return p;
}

// ---- the server's flow with the reference's names -------------------------------------------------------------------
// CoxgraphServer::fuseMap (coxgraph/src/server/coxgraph_server.cpp:396-476) and the final-mesh what-if copy
// (src/server/visualizer/server_visualizer.cpp:28-56), on submaps fused on the GPU and handed over GPU to GPU.
// Scene: a room corner (walls x = 3, y = 2.5, floor z = -1.2) seen by a camera at the origin turning about z.
static void renderFrame(double yaw, Pointcloud* pts, Colors* cols, Transformation* T_G_C) {
  pts->clear();
  cols->clear();
  const double c = std::cos(yaw), s = std::sin(yaw);
  // optical frame: z forward, x right, y down; R_G_C = Rz(yaw) * [[0,0,1],[-1,0,0],[0,-1,0]]
  const double R[9] = {c * 0 + -s * -1, 0.0, c * 1, s * 0 + c * -1, 0.0, s * 1, 0.0, -1.0, 0.0};
  for (int v = 0; v < 96; ++v)
    for (int u = 0; u < 128; ++u) {
      const double dc[3] = {(u - 63.5) / 100.0, (v - 47.5) / 100.0, 1.0};
      const double d[3] = {R[0] * dc[0] + R[1] * dc[1] + R[2] * dc[2], R[3] * dc[0] + R[4] * dc[1] + R[5] * dc[2], R[6] * dc[0] + R[7] * dc[1] + R[8] * dc[2]};
      double t = 1e30;
      if (d[0] > 1e-9) t = std::min(t, 3.0 / d[0]);
      if (d[1] > 1e-9) t = std::min(t, 2.5 / d[1]);
      if (d[2] < -1e-9) t = std::min(t, -1.2 / d[2]);
      if (t > 20.0) continue;
      pts->push_back({{static_cast<float>(t * dc[0]), static_cast<float>(t * dc[1]), static_cast<float>(t * dc[2])}});
      cols->push_back(Color{static_cast<uint8_t>(u), static_cast<uint8_t>(v), 128, 255});
    }
  // quaternion of R_G_C: yaw about z composed with the fixed optical rotation (w,x,y,z) = (0.5,-0.5,0.5,-0.5)
  const double hw = std::cos(0.5 * yaw), hz = std::sin(0.5 * yaw);
  const double o[4] = {0.5, -0.5, 0.5, -0.5};
  T_G_C->q[0] = static_cast<float>(hw * o[0] - hz * o[3]);
  T_G_C->q[1] = static_cast<float>(hw * o[1] - hz * o[2]);
  T_G_C->q[2] = static_cast<float>(hw * o[2] + hz * o[1]);
  T_G_C->q[3] = static_cast<float>(hw * o[3] + hz * o[0]);
  T_G_C->t[0] = T_G_C->t[1] = T_G_C->t[2] = 0.0f;
}

static int serverFlow() {
  const float voxel = 0.10f;
  TsdfIntegratorConfig cfg;
  cfg.default_truncation_distance = 0.3f, cfg.use_const_weight = 1, cfg.max_ray_length_m = 10.0f, cfg.min_ray_length_m = 0.2f;
  // three client submaps: yaw ranges [-0.5, 0.1], [-0.1, 0.5] (client 0, consecutive) and [-0.2, 0.3] (client 1)
  const double ranges[3][2] = {{-0.5, 0.1}, {-0.1, 0.5}, {-0.2, 0.3}};
  std::vector<std::unique_ptr<TsdfLayer>> client_layers;
  std::vector<std::vector<StampedPose>> trajectories(3);
  for (int k = 0; k < 3; ++k) {
    client_layers.emplace_back(new TsdfLayer(voxel, 16, 0, 2048));
    auto integ = TsdfIntegrator::create("merged", cfg, client_layers.back().get());
    for (int f = 0; f < 7; ++f) {
      Pointcloud pts;
      Colors cols;
      Transformation T;
      renderFrame(ranges[k][0] + (ranges[k][1] - ranges[k][0]) * f / 6.0, &pts, &cols, &T);
      integ->integratePointCloud(T, pts, cols, false);
      trajectories[k].push_back(StampedPose{Time(100 + 10 * k + f, 0), T});
    }
  }
  VoxgraphSubmap::Config sm_cfg;
  sm_cfg.tsdf_voxel_size = voxel;
  sm_cfg.registration_filter.max_voxel_distance = 0.3;
  sm_cfg.esdf.max_distance_m = sm_cfg.esdf.default_distance_m = 2.0f;
  sm_cfg.esdf.min_distance_m = 0.15f;
  SubmapCollection::Ptr collection(new SubmapCollection(sm_cfg, 2));
  PoseGraphInterface pg(collection);
  // the server believes submap 1 sits 8 cm / 1.5 deg off and client 1's submap 6 cm / -2 deg off
  const double believed[3][4] = {{0, 0, 0, 0}, {0.08, -0.05, 0.03, 0.026}, {-0.06, 0.04, -0.02, -0.035}};
  // ---- fuseMap for (client 0 submap 0, client 1 submap 2) ----
  VoxgraphSubmap::Ptr sm0 = cliSubmapFromDevice(0, sm_cfg, *client_layers[0], trajectories[0], transformationFromPose4(believed[0]));
  VoxgraphSubmap::Ptr sm2 = cliSubmapFromDevice(2, sm_cfg, *client_layers[2], trajectories[2], transformationFromPose4(believed[2]));
  if (!sm0->isFinished() || sm0->getRegistrationPoints(RegistrationPointType::kIsosurfacePoints)->size() < 500) return 10;
  collection->addSubmap(sm0, 0, 0);
  pg.addSubmap(sm0->getID());
  collection->addSubmap(sm2, 1, 0);
  pg.addSubmap(sm2->getID());
  Transformation T_A_t1, T_B_t2;
  if (!sm0->lookupPoseByTime(Time(103, 500), &T_A_t1) || !sm2->lookupPoseByTime(Time(123, 0), &T_B_t2)) return 11;
  if (sm0->lookupPoseByTime(Time(99, 0), &T_A_t1)) return 12;  // before the submap was in use
  sm0->lookupPoseByTime(Time(103, 500), &T_A_t1);
  // both cameras stood at the same place: T_t1_t2 is the rotation between the two views; the measurement is what a place
  // recognition front end would report, i.e. the TRUE relative pose of the two map frames (identity) up to its own noise
  const Transformation T_t1_t2 = inverse(T_A_t1) * T_B_t2;
  const Transformation T_A_B = T_A_t1 * T_t1_t2 * inverse(T_B_t2);
  if (!pg.addLoopClosureMeasurement(sm0->getID(), sm2->getID(), T_A_B, false)) return 13;
  pg.addForceRegistrationConstraint(sm0->getID(), sm2->getID());
  pg.updateSubmapRPConstraints();
  if (!pg.checkLoopClosureCandidates()) return 14;
  PoseGraphInterface::RegistrationConfig rc = pg.getRegistrationConfig();
  if (rc.registration_point_type != RegistrationPointType::kIsosurfacePoints || !rc.use_esdf_distance || rc.sampling_ratio != 0.3) return 15;
  auto res = pg.optimize(true);
  const Pose4 p2 = pg.getPoseMap().at(2);
  std::printf("fuseMap: submap 2 pose after optimise %.4f %.4f %.4f %.5f (cost %.4g -> %.4g, %zu overlap constraints)\n", p2.v[0], p2.v[1], p2.v[2], p2.v[3],
              res.second.initial_cost, res.second.final_cost, pg.getOverlappingSubmapList().size());
  if (std::fabs(p2.v[0]) > 0.02 || std::fabs(p2.v[1]) > 0.02 || std::fabs(p2.v[2]) > 0.02 || std::fabs(p2.v[3]) > 0.006) return 16;
  if (pg.getOverlappingSubmapList().size() != 1) return 17;  // the two submaps see the same corner
  // forced registration constraints are optimised whatever the flag says: start again from the believed pose
  {
    PoseGraphInterface pg_off(pg);
    pg_off.poseGraph().poses[2] = Pose4();
    for (int k = 0; k < 4; ++k) pg_off.poseGraph().poses[2].v[k] = believed[2][k];
    pg_off.poseGraph().rel.clear();  // only the forced registration constraint is left
    pg_off.optimize(false);
    const Pose4 q = pg_off.getPoseMap().at(2);
    std::printf("forced registration only, enable_registration = false: submap 2 pose %.4f %.4f %.4f %.5f\n", q.v[0], q.v[1], q.v[2], q.v[3]);
    // submap 0 sees the wall x = 3 and the floor only: y is free to slide without the loop closure; x, z and yaw must come back
    if (std::fabs(q.v[0]) > 0.02 || std::fabs(q.v[2]) > 0.02 || std::fabs(q.v[3]) > 0.006) return 18;
  }
  // ---- final global mesh: a what-if copy with the remaining submap (server_visualizer.cpp:28-56) ----
  const PoseGraphInterface::PoseMap live_before = pg.getPoseMap();
  SubmapCollection::Ptr global_collection(new SubmapCollection(*collection));
  PoseGraphInterface global_pg(pg, global_collection);
  VoxgraphSubmap::Ptr sm1 = cliSubmapFromDevice(1, sm_cfg, *client_layers[1], trajectories[1], transformationFromPose4(believed[1]));
  global_collection->addSubmap(sm1, 0, 1);
  global_pg.addSubmap(sm1->getID());
  global_pg.updateSubmapRPConstraints();
  if (global_pg.evaluateResiduals(PoseGraphInterface::ConstraintType::SubmapRelPose).size() != 4) return 20;  // one constraint: submaps 0 -> 1 of client 0
  global_pg.optimize(true);
  global_pg.printResiduals(PoseGraphInterface::ConstraintType::RelPose);
  global_pg.printResiduals(PoseGraphInterface::ConstraintType::SubmapRelPose);
  global_pg.updateSubmapCollectionPoses();
  const Pose4 g1 = global_pg.getPoseMap().at(1);
  std::printf("final mesh what-if: submap 1 pose %.4f %.4f %.4f %.5f, %zu overlapping pairs\n", g1.v[0], g1.v[1], g1.v[2], g1.v[3], global_pg.getOverlappingSubmapList().size());
  // the consecutive-submap constraint (information 1000 / 2500) holds submap 1 at its believed offset against the overlap
  // registration (information ~ residual count): it moves towards the truth but not all the way; the live graph is untouched
  if (global_pg.getOverlappingSubmapList().size() != 3) return 21;
  if (pg.getPoseMap().size() != 2 || collection->size() != 2 || global_collection->size() != 3) return 22;
  for (const auto& kv : live_before)
    for (int k = 0; k < 4; ++k)
      if (pg.getPoseMap().at(kv.first).v[k] != kv.second.v[k]) return 23;
  double moved[4];
  pose4FromTransformation(global_collection->getSubmapPtr(1)->getPose(), moved);
  if (std::fabs(moved[0] - g1.v[0]) > 1e-6 || std::fabs(moved[3] - g1.v[3]) > 1e-6) return 24;
  if (!(std::fabs(g1.v[0]) < std::fabs(believed[1][0]))) return 25;
  // mergeToCliMap: the same submap sent again is merged into the stored one and finished again
  VoxgraphSubmap::Ptr again = cliSubmapFromDevice(0, sm_cfg, *client_layers[0], trajectories[0], transformationFromPose4(believed[0]));
  const size_t n_before = collection->getSubmapPtr(0)->getRegistrationPoints(RegistrationPointType::kVoxels)->size();
  collection->mergeToCliMap(again);
  if (collection->getSubmapPtr(0)->getRegistrationPoints(RegistrationPointType::kVoxels)->size() < n_before) return 26;
  // RCCL from the C++ host: a communicator of one rank (what this box can hold) sums the packed normal equations through
  // ncclAllReduce and must leave the solve unchanged; its all-gather hands a device buffer back unchanged
  {
    uint8_t id[COX_COMM_ID_BYTES];
    const int st = cox_comm_unique_id(id);
    if (st == COX_ERR_UNSUPPORTED) {
      std::printf("librccl.so not found: communicator path skipped\n");
    } else {
      if (st != COX_OK) return 30;
      cox_comm_t* comm = nullptr;
      if (cox_comm_init_rank(0, 0, 1, id, &comm) != COX_OK) return 31;
      PoseGraphInterface with(pg), without(pg);
      for (PoseGraphInterface* g : {&with, &without}) {
        g->poseGraph().poses[2] = Pose4();
        for (int k = 0; k < 4; ++k) g->poseGraph().poses[2].v[k] = believed[2][k];
        g->poseGraph().solve_counter = 77;
      }
      with.poseGraph().comm = comm;
      with.optimize(true);
      without.optimize(true);
      for (int k = 0; k < 4; ++k)
        if (with.getPoseMap().at(2).v[k] != without.getPoseMap().at(2).v[k]) return 32;
      double probe[5] = {1.5, -2.0, 3.25, 0.0, 1e-300};
      if (cox_comm_allreduce_f64(comm, probe, 5) != COX_OK || probe[0] != 1.5 || probe[2] != 3.25 || probe[4] != 1e-300) return 33;
      cox_comm_destroy(comm);
      std::printf("rccl communicator (1 rank): all-reduce of the normal equations leaves the solve bit-identical\n");
    }
  }
  std::printf("server flow ok\n");
  return 0;
}

// the trust-region minimiser on two problems whose iteration traces are derived by hand in tests/test_host_logic.py
static int solverKats() {
  {  // r = 2 x - 6 from x = 0
    std::vector<double> x{0.0};
    auto ev = [](const std::vector<double>& xx, std::vector<double>* g, std::vector<double>* H) {
      const double r = 2.0 * xx[0] - 6.0;
      g->assign(1, 2.0 * r);
      H->assign(1, 4.0);
      return 0.5 * r * r;
    };
    auto plus = [](const std::vector<double>& xx, const std::vector<double>& d) { return std::vector<double>{xx[0] + d[0]}; };
    const TrustRegionSummary S = trustRegionMinimize(ev, plus, &x);
    std::printf("linear %.17g %d %s|", x[0], S.iterations, S.message.c_str());
    for (const auto& it : S.trace) std::printf(" %d %.17g %.17g %.17g %.17g", it.iteration, it.cost, it.step_norm, it.relative_decrease, it.trust_region_radius);
    std::printf("\n");
    TrustRegionOptions o;
    o.parameter_tolerance = 0.0;
    x[0] = 0.0;
    const TrustRegionSummary S2 = trustRegionMinimize(ev, plus, &x, o);
    std::printf("linear_gradient %.17g %d %s\n", x[0], S2.iterations, S2.message.c_str());
  }
  {  // r = x^3 - 1 from x = 0.1: six rejected steps before the first accepted one
    std::vector<double> x{0.1};
    auto ev = [](const std::vector<double>& xx, std::vector<double>* g, std::vector<double>* H) {
      const double r = xx[0] * xx[0] * xx[0] - 1.0, J = 3.0 * xx[0] * xx[0];
      g->assign(1, J * r);
      H->assign(1, J * J);
      return 0.5 * r * r;
    };
    auto plus = [](const std::vector<double>& xx, const std::vector<double>& d) { return std::vector<double>{xx[0] + d[0]}; };
    TrustRegionOptions o;
    o.parameter_tolerance = 1e-12;
    const TrustRegionSummary S = trustRegionMinimize(ev, plus, &x, o);
    std::printf("cubic %.17g %d %d %d %s|", x[0], S.iterations, S.successful_steps, S.unsuccessful_steps, S.message.c_str());
    for (const auto& it : S.trace) std::printf(" %d %d %.17g %.17g", it.iteration, it.step_is_successful ? 1 : 0, it.cost, it.trust_region_radius);
    std::printf("\n");
  }
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 1 && std::strcmp(argv[1], "lm") == 0) return solverKats();
  // square-root information: S^T S == information, also for a semi-definite matrix (eigen branch)
  const double semi[16] = {4, 2, 0, 0, 2, 1, 0, 0, 0, 0, 9, 0, 0, 0, 0, 0};
  double S[16];
  sqrtInformation(semi, S);
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += S[4 * k + i] * S[4 * k + j];
      if (std::fabs(s - semi[4 * i + j]) > 1e-9) {
        std::printf("sqrtInformation mismatch at %d,%d: %g vs %g\n", i, j, s, semi[4 * i + j]);
        return 2;
      }
    }
  PoseGraphInterface pg;
  // two clients: submaps 0,1,2 (client 0) and 3,4 (client 1), odometry-like initial poses
  pg.addSubmap(0, P(0, 0, 0, 0), 0);
  pg.addSubmap(1, P(1.0, 0.1, 0.0, 0.05), 0);
  pg.addSubmap(2, P(2.1, 0.3, 0.02, 0.12), 0);
  pg.addSubmap(3, P(0.2, 1.9, -0.03, 1.50), 1);
  pg.addSubmap(4, P(0.1, 3.1, 0.01, 1.62), 1);
  pg.updateSubmapRPConstraints();
  const double lc1[4] = {0.15, 2.05, 0.0, 1.57}, lc2[4] = {-1.8, 2.9, 0.05, 1.49}, lc3[4] = {1.05, -0.05, 0.01, 0.02};
  pg.addLoopClosureMeasurement(0, 3, lc1);
  pg.addLoopClosureMeasurement(2, 4, lc2);
  pg.addLoopClosureMeasurement(1, 2, lc3);
  const auto res = pg.optimize(false);
  std::printf("summary %.12g %.12g %d %d\n", res.second.initial_cost, res.second.final_cost, res.first.iterations, res.second.iterations);
  for (const auto& kv : pg.getPoseMap()) std::printf("pose %d %.15g %.15g %.15g %.15g\n", kv.first, kv.second.v[0], kv.second.v[1], kv.second.v[2], kv.second.v[3]);
  if (argc > 1 && std::strcmp(argv[1], "gpu") == 0) {
    // a wall fused into two layers, the second shifted along its normal: the forced registration constraint pulls it back
    TsdfLayer la(0.10f, 16, 0, 2048), lb(0.10f, 16, 0, 2048);
    TsdfIntegratorConfig cfg;
    cfg.default_truncation_distance = 0.3f, cfg.use_const_weight = 1, cfg.max_ray_length_m = 10.0f, cfg.min_ray_length_m = 0.2f;
    auto fa = TsdfIntegrator::create("merged", cfg, &la), fb = TsdfIntegrator::create("merged", cfg, &lb);
    Pointcloud pts;
    Colors cols;
    for (int v = 0; v < 120; ++v)
      for (int u = 0; u < 160; ++u) {
        pts.push_back({{2.0f * (u - 80) / 120.0f, 2.0f * (v - 60) / 120.0f, 2.0f + 0.2f * (u - 80) / 120.0f}});
        cols.push_back(Color{128, 128, 128, 255});
      }
    Transformation T;
    for (int k = 0; k < 4; ++k) {
      T.t[0] = 0.02f * k;
      fa->integratePointCloud(T, pts, cols, false);
      fb->integratePointCloud(T, pts, cols, false);
    }
    uint64_t n = 0;
    check(cox_layer_registration_points(la.handle(), 1.0f, 0.3f, nullptr, 0, &n), "registration points");
    std::vector<float> raw(5 * n);
    check(cox_layer_registration_points(la.handle(), 1.0f, 0.3f, raw.data(), n, &n), "registration points");
    std::vector<RegistrationPoint> rp(n);
    for (uint64_t i = 0; i < n; ++i) rp[i] = RegistrationPoint{{raw[5 * i], raw[5 * i + 1], raw[5 * i + 2]}, raw[5 * i + 3], raw[5 * i + 4]};
    RegistrationCostFunction cost(rp, lb);
    PoseGraphInterface g2;
    g2.addSubmap(0, P(0, 0, 0, 0));
    g2.addSubmap(1, P(0.0, 0.0, 0.06, 0.0));  // the reading submap believed 6 cm too far along the wall's normal (camera z)
    g2.addForceRegistrationConstraint(0, 1, &cost);
    const auto r2 = g2.optimize(true);
    const Pose4& p1 = g2.getPoseMap().at(1);
    std::printf("registration %zu points: cost %.6g -> %.6g, pose1 z %.5f (%d evaluations)\n", (size_t)n, r2.second.initial_cost, r2.second.final_cost, p1.v[2],
                r2.second.evaluations);
    if (!(r2.second.final_cost < 1e-3 * r2.second.initial_cost) || std::fabs(p1.v[2]) > 5e-3) return 3;
    const int rc = serverFlow();
    if (rc) return rc;
  }
  return 0;
}
