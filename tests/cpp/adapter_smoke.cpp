// Mirrors the only in-tree caller of the integrator, TsdfRecover::processMesh
// (coxgraph/include/coxgraph/map_comm/tsdf_recover.h:59-99): clear the layer, integrate a few point
// clouds, report the layer memory, serialise it -- through the C++ adapters.  Exit code 0 = all good;
// 77 = no GPU (the constructors fail with COX_ERR_NO_DEVICE, nothing falls back).
#include <cmath>
#include <cstdio>

#include "../../coxgraph_amd/host/coxgraph_hip_adapters.hpp"

using namespace coxgraph_hip;

int main() {
  if (cox_device_count() == 0) {
    try {
      TsdfLayer layer(0.10f);
    } catch (const std::runtime_error& e) {
      std::printf("no GPU: %s\n", e.what());
      return 77;
    }
    return 1;
  }
  TsdfLayer layer(0.10f, 16, 0, 2048);
  TsdfIntegratorConfig cfg;
  cfg.default_truncation_distance = 0.3f;
  cfg.use_const_weight = 1;
  cfg.max_ray_length_m = 10.0f;
  cfg.min_ray_length_m = 0.2f;
  auto integrator = TsdfIntegrator::create("merged", cfg, &layer);
  layer.removeAllBlocks();
  Pointcloud points_C;
  Colors colors;
  for (int v = 0; v < 60; ++v)
    for (int u = 0; u < 80; ++u) {  // a wall 2 m in front of the camera
      points_C.push_back({{2.0f * (u - 40) / 60.0f, 2.0f * (v - 30) / 60.0f, 2.0f}});
      colors.push_back(Color{static_cast<uint8_t>(u), static_cast<uint8_t>(v), 128, 255});
    }
  Transformation T_G_C;
  for (int i = 0; i < 3; ++i) {
    T_G_C.t[0] = 0.05f * i;
    integrator->integratePointCloud(T_G_C, points_C, colors, false);
  }
  const cox_frame_stats st = integrator->lastFrameStats();
  std::printf("layer memory: %zu bytes, %zu blocks; last frame: %llu rays, %llu updates\n", layer.getMemorySize(), layer.getNumberOfAllocatedBlocks(),
              (unsigned long long)st.n_rays, (unsigned long long)st.n_updates);
  LayerMsg msg;
  serializeLayerAsMsg(layer, false, &msg);
  if (msg.blocks.empty() || msg.blocks.size() != layer.getNumberOfAllocatedBlocks()) return 2;
  TsdfLayer copy(0.10f, 16, 0, 2048);
  if (!deserializeMsgToLayer(msg, &copy) || copy.getMemorySize() != layer.getMemorySize()) return 3;
  // registration of the wall against itself: zero residuals at identical poses
  std::vector<RegistrationPoint> pts;
  for (int i = 0; i < 500; ++i) pts.push_back(RegistrationPoint{{0.01f * (i % 50), 0.02f * (i / 50), 2.0f}, 0.0f, 1.0f});
  RegistrationCostFunction cost(pts, layer);
  std::vector<double> r(pts.size()), jf(4 * pts.size()), jr(4 * pts.size());
  const double pose[4] = {0, 0, 0, 0};
  const double* params[2] = {pose, pose};
  double* jac[2] = {jf.data(), jr.data()};
  if (!cost.Evaluate(params, r.data(), jac)) return 4;
  double worst = 0;
  for (double x : r) worst = std::fmax(worst, std::fabs(x));
  std::printf("registration max |r| = %g\n", worst);
  if (!(worst < 0.06)) return 5;
  {
    // jacobian_evaluation_method = numeric (voxgraph's debugging aid): central differences of the residuals agree with the analytic
    // rows at a pose where the points have moved off the wall (float interpolation inside: a 1e-4 step keeps its rounding noise at 1e-3)
    const double pa[4] = {0.01, -0.02, 0.03, 0.02}, pb[4] = {-0.02, 0.01, -0.01, -0.01};
    const double* p2[2] = {pa, pb};
    std::vector<double> r2(pts.size()), nf(4 * pts.size()), nr(4 * pts.size());
    double* njac[2] = {nf.data(), nr.data()};
    if (!cost.Evaluate(p2, r.data(), jac)) return 4;
    cost.setJacobianEvaluationMethod(RegistrationCostFunction::JacobianEvaluationMethod::kNumeric);
    if (!cost.Evaluate(p2, r2.data(), njac)) return 4;
    cost.setJacobianEvaluationMethod(RegistrationCostFunction::JacobianEvaluationMethod::kAnalytic);
    double dj = 0, dr = 0, jmax = 0;
    for (size_t i = 0; i < nf.size(); ++i) {
      dj = std::fmax(dj, std::fmax(std::fabs(nf[i] - jf[i]), std::fabs(nr[i] - jr[i])));
      jmax = std::fmax(jmax, std::fabs(jf[i]));
    }
    for (size_t i = 0; i < r.size(); ++i) dr = std::fmax(dr, std::fabs(r[i] - r2[i]));
    std::printf("numeric vs analytic Jacobian: max |difference| = %g (largest entry %g)\n", dj, jmax);
    if (dr != 0.0 || !(dj < 0.01 * std::fmax(1.0, jmax))) return 8;
  }
  {
    // the asynchronous host entry: two more frames without waiting, then sync -- the same layer as the synchronous call gives
    TsdfLayer la(0.10f, 16, 0, 2048), lb(0.10f, 16, 0, 2048);
    auto ia = TsdfIntegrator::create("merged", cfg, &la), ib = TsdfIntegrator::create("merged", cfg, &lb);
    for (int i = 0; i < 4; ++i) {
      T_G_C.t[0] = 0.03f * i;
      ia->integratePointCloudAsync(T_G_C, points_C, colors, false);
      ib->integratePointCloud(T_G_C, points_C, colors, false);
    }
    ia->waitInputs();
    ia->sync();
    LayerMsg ma, mb2;
    serializeLayerAsMsg(la, false, &ma);
    serializeLayerAsMsg(lb, false, &mb2);
    if (ma.blocks.size() != mb2.blocks.size() || ma.blocks.empty()) return 9;
    for (size_t k = 0; k < ma.blocks.size(); ++k)
      if (ma.blocks[k].data != mb2.blocks[k].data) return 10;
    std::printf("asynchronous host entry: %zu blocks identical\n", ma.blocks.size());
  }
  // recover mode: one mesh block (two triangles seen in frames 0..1) through MeshConverter + processMesh
  MeshMsg mesh;
  mesh.block_edge_length = 1.6f;
  MeshBlockMsg mb;
  mb.index[0] = 1, mb.index[1] = 0, mb.index[2] = 0;
  const uint16_t vy[6] = {0, 16384, 0, 16384, 16384, 0}, vz[6] = {0, 0, 16384, 0, 16384, 16384};
  for (int k = 0; k < 6; ++k) {
    mb.x.push_back(8192), mb.y.push_back(vy[k]), mb.z.push_back(vz[k]);
    mb.r.push_back(200), mb.g.push_back(static_cast<uint8_t>(40 * k)), mb.b.push_back(10);
  }
  mb.history = {{0, 1}, {1, 1}};
  mesh.mesh_blocks.push_back(mb);
  for (int k = 0; k < 2; ++k) {
    StampedTransformation p;
    p.sec = 100, p.nsec = 50000000u * k;
    p.T_G_C.t[0] = 0.02f * k;  // identity rotation: camera z looks along world z, the triangles sit at x = 2 m
    mesh.trajectory.push_back(p);
  }
  MeshConverter converter(0.1f);
  LayerMsg recovered_layer;
  std::vector<PointXYZRGB> recovered;
  processMesh(&converter, integrator.get(), &layer, mesh, &recovered_layer, &recovered);
  std::printf("recover: %zu mesh vertices -> %zu blocks\n", recovered.size(), recovered_layer.blocks.size());
  if (recovered.size() != 6 || recovered_layer.blocks.empty()) return 6;
  if (std::fabs(recovered[0].x - 2.0f) > 1e-3f) return 7;
  return 0;
}
