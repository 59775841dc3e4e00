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

int main(int argc, char** argv) {
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
  }
  return 0;
}
