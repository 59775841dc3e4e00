// C++ adapters that give the C ABI of include/coxgraph_hip.h the signatures coxgraph's host code already
// uses (voxblox / voxgraph shapes), so a call site such as
//     tsdf_integrator_->integratePointCloud(T_G_C, *points_C, *colors, false);      // tsdf_recover.h:75
// keeps its form.  Header-only, dependency-free C++14 (the reference builds with -std=c++14,
// coxgraph/CMakeLists.txt:4); link with -lcoxgraph_hip.  Failures throw std::runtime_error where the reference
// would glog-CHECK-abort.  See INTEGRATION.md for the glue to the real voxblox / Eigen / ROS types.
#pragma once
#include <array>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/coxgraph_hip.h"

namespace coxgraph_hip {

inline void check(int status, const char* what) {
  if (status != COX_OK) throw std::runtime_error(std::string(what) + ": " + cox_status_string(status));
}

// voxblox::Transformation (kindr::minimal::QuatTransformationTemplate<float>): rotation (w,x,y,z) + position
struct Transformation {
  float q[4] = {1.f, 0.f, 0.f, 0.f};
  float t[3] = {0.f, 0.f, 0.f};
  void pack(float out[7]) const {
    for (int i = 0; i < 4; ++i) out[i] = q[i];
    for (int i = 0; i < 3; ++i) out[4 + i] = t[i];
  }
};
using Point = std::array<float, 3>;  // voxblox::Point (Eigen::Vector3f), tightly packed
struct Color {                       // voxblox::Color
  uint8_t r = 0, g = 0, b = 0, a = 0;
};
using Pointcloud = std::vector<Point>;  // voxblox::Pointcloud
using Colors = std::vector<Color>;      // voxblox::Colors
static_assert(sizeof(Point) == 12 && sizeof(Color) == 4, "packed layouts expected by the C ABI");

// voxblox_msgs/Layer (the part coxgraph touches: utils/msg_converter.h:149-167)
struct BlockMsg {
  int32_t x_index, y_index, z_index;
  std::vector<uint32_t> data;  // 3 words per voxel, 4096 voxels
};
struct LayerMsg {
  double voxel_size = 0.0;
  uint32_t voxels_per_side = 16;
  std::string layer_type = "tsdf";
  uint8_t action = 0;  // 0 update, 1 merge, 2 reset
  std::vector<BlockMsg> blocks;
};

// voxblox::Layer<TsdfVoxel>
class TsdfLayer {
 public:
  TsdfLayer(float voxel_size, size_t voxels_per_side = 16, int device = 0, uint64_t capacity_blocks = 0) : voxel_size_(voxel_size), vps_(voxels_per_side) {
    check(cox_layer_create(voxel_size, static_cast<int>(voxels_per_side), device, capacity_blocks, &h_), "Layer");
  }
  // an exact copy of `other` on `device` (the submap hand-over: block array + keys GPU to GPU, hash rebuilt there)
  TsdfLayer(const TsdfLayer& other, int device) : voxel_size_(other.voxel_size_), vps_(other.vps_) {
    check(cox_layer_clone_to_device(other.h_, device, 0, &h_), "Layer(copy)");
  }
  ~TsdfLayer() { cox_layer_destroy(h_); }
  TsdfLayer(const TsdfLayer&) = delete;
  TsdfLayer& operator=(const TsdfLayer&) = delete;

  void removeAllBlocks() { check(cox_layer_clear(h_), "removeAllBlocks"); }  // tsdf_recover.h:62
  size_t getNumberOfAllocatedBlocks() const {                                // map_server.h:142
    uint64_t n = 0;
    check(cox_layer_stats(h_, &n, nullptr), "getNumberOfAllocatedBlocks");
    return n;
  }
  size_t getMemorySize() const {  // tsdf_recover.h:92
    uint64_t b = 0;
    check(cox_layer_stats(h_, nullptr, &b), "getMemorySize");
    return b;
  }
  float voxel_size() const { return voxel_size_; }
  size_t voxels_per_side() const { return vps_; }
  float block_size() const { return voxel_size_ * static_cast<float>(vps_); }  // map_server.cpp:124
  cox_layer_t* handle() const { return h_; }

 private:
  cox_layer_t* h_ = nullptr;
  float voxel_size_;
  size_t vps_;
};

// voxblox::serializeLayerAsMsg<TsdfVoxel>(layer, only_updated = false, &msg)   (utils/msg_converter.h:49)
inline void serializeLayerAsMsg(const TsdfLayer& layer, bool /*only_updated*/, LayerMsg* msg) {
  uint64_t n = 0;
  check(cox_layer_download(layer.handle(), nullptr, nullptr, 0, &n), "serializeLayerAsMsg");
  std::vector<int32_t> idx(3 * n);
  std::vector<uint32_t> words(n * 4096 * 3);
  if (n) check(cox_layer_download(layer.handle(), idx.data(), words.data(), n, &n), "serializeLayerAsMsg");
  msg->voxel_size = layer.voxel_size();
  msg->voxels_per_side = static_cast<uint32_t>(layer.voxels_per_side());
  msg->layer_type = "tsdf";
  msg->action = 0;
  msg->blocks.resize(n);
  for (uint64_t i = 0; i < n; ++i) {
    BlockMsg& b = msg->blocks[i];
    b.x_index = idx[3 * i];
    b.y_index = idx[3 * i + 1];
    b.z_index = idx[3 * i + 2];
    b.data.assign(words.begin() + i * 12288, words.begin() + (i + 1) * 12288);
  }
}
// voxblox::deserializeMsgToLayer(msg, layer) -> bool   (utils/msg_converter.h:107)
inline bool deserializeMsgToLayer(const LayerMsg& msg, TsdfLayer* layer) {
  if (msg.layer_type != "tsdf" || msg.voxels_per_side != layer->voxels_per_side() ||
      static_cast<float>(msg.voxel_size) != layer->voxel_size())
    return false;
  const uint64_t n = msg.blocks.size();
  std::vector<int32_t> idx(3 * n);
  std::vector<uint32_t> words(n * 12288);
  for (uint64_t i = 0; i < n; ++i) {
    const BlockMsg& b = msg.blocks[i];
    if (b.data.size() != 12288) return false;
    idx[3 * i] = b.x_index;
    idx[3 * i + 1] = b.y_index;
    idx[3 * i + 2] = b.z_index;
    std::copy(b.data.begin(), b.data.end(), words.begin() + i * 12288);
  }
  return cox_layer_upload(layer->handle(), idx.data(), words.data(), n, msg.action) == COX_OK;
}

// voxblox::mergeLayerAintoLayerB(layer_A, layer_B) and (layer_A, T_B_A, layer_B)   (src/client/map_server.cpp:67-69,
// src/server/submap_collection.cpp:31-33)
inline void mergeLayerAintoLayerB(const TsdfLayer& layer_A, TsdfLayer* layer_B) { check(cox_layer_merge(layer_A.handle(), nullptr, layer_B->handle()), "mergeLayerAintoLayerB"); }
inline void mergeLayerAintoLayerB(const TsdfLayer& layer_A, const Transformation& T_B_A, TsdfLayer* layer_B) {
  float T[7];
  T_B_A.pack(T);
  check(cox_layer_merge(layer_A.handle(), T, layer_B->handle()), "mergeLayerAintoLayerB");
}

// voxblox::TsdfIntegratorBase::Config
struct TsdfIntegratorConfig : cox_tsdf_config {
  TsdfIntegratorConfig() { cox_tsdf_config_default(this); }
};

// voxblox::TsdfIntegratorBase + TsdfIntegratorFactory
class TsdfIntegrator {
 public:
  using Ptr = std::shared_ptr<TsdfIntegrator>;
  // TsdfIntegratorFactory::create(method, config, layer): method in {"simple", "merged", "fast", "projective"}
  static Ptr create(const std::string& method, const TsdfIntegratorConfig& config, TsdfLayer* layer) {
    int m = -1;
    if (method == "simple") m = COX_METHOD_SIMPLE;
    if (method == "merged") m = COX_METHOD_MERGED;
    if (method == "fast") m = COX_METHOD_FAST;
    if (method == "projective") m = COX_METHOD_PROJECTIVE;  // config/tsdf_server_default.yaml:6, tsdf_server_carla.yaml:6
    if (m < 0) throw std::runtime_error("Unknown TSDF integrator type: " + method);
    return Ptr(new TsdfIntegrator(m, config, layer));
  }
  ~TsdfIntegrator() { cox_integrator_destroy(h_); }
  // TsdfIntegratorBase::integratePointCloud(T_G_C, points_C, colors, freespace_points)
  void integratePointCloud(const Transformation& T_G_C, const Pointcloud& points_C, const Colors& colors, const bool freespace_points = false) {
    if (!colors.empty() && colors.size() != points_C.size()) throw std::runtime_error("integratePointCloud: points_C.size() != colors.size()");
    float T[7];
    T_G_C.pack(T);
    check(cox_integrate_points(h_, T, points_C.empty() ? nullptr : points_C[0].data(), colors.empty() ? nullptr : &colors[0].r, points_C.size(),
                               freespace_points ? 1 : 0),
          "integratePointCloud");
  }
  // The same call without waiting for the frame (cox_integrate_points_async): the clouds of a stream are copied to the GPU while the
  // frames before them are still being fused.  points_C / colors are free again when this returns unless they live in pinned memory
  // (then: after waitInputs()).  Errors surface at sync() -- voxblox has no return value here either; call sync() before reading the layer.
  void integratePointCloudAsync(const Transformation& T_G_C, const Pointcloud& points_C, const Colors& colors, const bool freespace_points = false) {
    if (!colors.empty() && colors.size() != points_C.size()) throw std::runtime_error("integratePointCloud: points_C.size() != colors.size()");
    float T[7];
    T_G_C.pack(T);
    check(cox_integrate_points_async(h_, T, points_C.empty() ? nullptr : points_C[0].data(), colors.empty() ? nullptr : &colors[0].r, points_C.size(),
                                     freespace_points ? 1 : 0),
          "integratePointCloudAsync");
  }
  void waitInputs() { check(cox_integrator_wait_inputs(h_), "waitInputs"); }
  void sync() { check(cox_integrator_sync(h_), "sync"); }
  cox_frame_stats lastFrameStats() const {
    cox_frame_stats s;
    check(cox_integrator_last_stats(h_, &s), "lastFrameStats");
    return s;
  }
  cox_integrator_t* handle() const { return h_; }

 private:
  TsdfIntegrator(int method, const TsdfIntegratorConfig& config, TsdfLayer* layer) { check(cox_integrator_create(layer->handle(), &config, method, &h_), "TsdfIntegrator"); }
  cox_integrator_t* h_ = nullptr;
};

// ---- recover mode ---------------------------------------------------------------------------------------------------
// voxblox_msgs/Mesh (+ history, + trajectory) the way recover mode receives it (mesh_converter.h:55-72)
struct MeshBlockMsg {
  int64_t index[3];
  std::vector<uint16_t> x, y, z;
  std::vector<uint8_t> r, g, b;
  std::vector<std::vector<uint32_t>> history;  // voxblox_msgs/ObsHistory per triangle; empty = block skipped
};
struct StampedTransformation {
  uint32_t sec = 0, nsec = 0;  // header.stamp
  Transformation T_G_C;        // pose after tf::poseMsgToKindr(...).cast<float>()
};
struct MeshMsg {
  float block_edge_length = 0.0f;
  std::vector<MeshBlockMsg> mesh_blocks;
  std::vector<StampedTransformation> trajectory;
};
struct PointXYZRGB {
  float x, y, z;
  uint8_t r, g, b;
};

// voxblox::MeshConverter (mesh_converter.h:22-289); clouds stay on the GPU
class MeshConverter {
 public:
  explicit MeshConverter(float interpolate_voxel_size = 0.20f, int device = 0) { check(cox_meshconv_create(device, interpolate_voxel_size, &h_), "MeshConverter"); }
  ~MeshConverter() { cox_meshconv_destroy(h_); }
  MeshConverter(const MeshConverter&) = delete;
  MeshConverter& operator=(const MeshConverter&) = delete;
  void setMesh(const MeshMsg& mesh) {
    Flat f(mesh);
    check(cox_meshconv_set_mesh(h_, &f.msg), "setMesh");
  }
  bool convertToPointCloud(std::vector<PointXYZRGB>* recovered_pointcloud) {
    uint64_t n = 0;
    int converted = 0;
    check(cox_meshconv_convert(h_, &n, &converted), "convertToPointCloud");
    fetchRecovered(recovered_pointcloud);
    return converted != 0;
  }
  // device-side variant of getNextPointcloud: the cloud is handed over as device pointers
  bool getNextPointcloud(int* i, Transformation* T_G_C, const float** points_C_dev, const uint8_t** colors_dev, uint64_t* n) {
    float T[7];
    int has_next = 0;
    int32_t ii = *i;
    check(cox_meshconv_next(h_, &ii, T, points_C_dev, colors_dev, n, &has_next), "getNextPointcloud");
    if (!has_next) return false;
    *i = ii;
    for (int k = 0; k < 4; ++k) T_G_C->q[k] = T[k];
    for (int k = 0; k < 3; ++k) T_G_C->t[k] = T[4 + k];
    return true;
  }
  void clear() { check(cox_meshconv_clear(h_), "clear"); }
  void fetchRecovered(std::vector<PointXYZRGB>* out) {
    if (!out) return;
    uint64_t n = 0;
    check(cox_meshconv_recovered(h_, nullptr, nullptr, 0, &n), "recovered");
    std::vector<float> xyz(3 * n);
    std::vector<uint8_t> rgb(3 * n);
    if (n) check(cox_meshconv_recovered(h_, xyz.data(), rgb.data(), n, &n), "recovered");
    out->resize(n);
    for (uint64_t k = 0; k < n; ++k) (*out)[k] = PointXYZRGB{xyz[3 * k], xyz[3 * k + 1], xyz[3 * k + 2], rgb[3 * k], rgb[3 * k + 1], rgb[3 * k + 2]};
  }
  cox_meshconv_t* handle() const { return h_; }

  // flattened view of a MeshMsg for the C ABI
  struct Flat {
    std::vector<int64_t> index;
    std::vector<uint64_t> vertex_begin, history_begin;
    std::vector<uint16_t> x, y, z;
    std::vector<uint8_t> r, g, b, has;
    std::vector<uint32_t> history, sec, nsec;
    std::vector<float> T;
    cox_mesh_msg msg;
    explicit Flat(const MeshMsg& m) {
      vertex_begin.push_back(0);
      history_begin.push_back(0);
      for (const MeshBlockMsg& mb : m.mesh_blocks) {
        if (mb.x.size() % 3 || (!mb.history.empty() && mb.history.size() != mb.x.size() / 3)) throw std::runtime_error("MeshConverter: x.size() / 3 != history.size()");
        index.insert(index.end(), mb.index, mb.index + 3);
        x.insert(x.end(), mb.x.begin(), mb.x.end()), y.insert(y.end(), mb.y.begin(), mb.y.end()), z.insert(z.end(), mb.z.begin(), mb.z.end());
        r.insert(r.end(), mb.r.begin(), mb.r.end()), g.insert(g.end(), mb.g.begin(), mb.g.end()), b.insert(b.end(), mb.b.begin(), mb.b.end());
        vertex_begin.push_back(x.size());
        has.push_back(mb.history.empty() ? 0 : 1);
        for (size_t t = 0; t < mb.x.size() / 3; ++t) {
          if (!mb.history.empty()) history.insert(history.end(), mb.history[t].begin(), mb.history[t].end());
          history_begin.push_back(history.size());
        }
      }
      for (const StampedTransformation& p : m.trajectory) {
        sec.push_back(p.sec), nsec.push_back(p.nsec);
        float t7[7];
        p.T_G_C.pack(t7);
        T.insert(T.end(), t7, t7 + 7);
      }
      msg.block_edge_length = m.block_edge_length;
      msg.n_blocks = m.mesh_blocks.size();
      msg.block_index = index.data(), msg.vertex_begin = vertex_begin.data();
      msg.x = x.data(), msg.y = y.data(), msg.z = z.data(), msg.r = r.data(), msg.g = g.data(), msg.b = b.data();
      msg.block_has_history = has.data(), msg.history_begin = history_begin.data(), msg.history = history.data();
      msg.n_poses = m.trajectory.size();
      msg.stamp_sec = sec.data(), msg.stamp_nsec = nsec.data(), msg.T_G_C = T.data();
    }
  };

 private:
  cox_meshconv_t* h_ = nullptr;
};

// TsdfRecover::processMesh(mesh_msg, &layer_msg, &recovered_pointcloud)  (tsdf_recover.h:59-99)
inline void processMesh(MeshConverter* mesh_converter, TsdfIntegrator* tsdf_integrator, TsdfLayer* layer, const MeshMsg& mesh_msg, LayerMsg* layer_msg,
                        std::vector<PointXYZRGB>* recovered_pointcloud) {
  MeshConverter::Flat f(mesh_msg);
  uint64_t n_rec = 0, n_int = 0;
  check(cox_recover_process_mesh(mesh_converter->handle(), tsdf_integrator->handle(), &f.msg, &n_rec, &n_int), "processMesh");
  mesh_converter->fetchRecovered(recovered_pointcloud);
  if (layer_msg) serializeLayerAsMsg(*layer, false, layer_msg);
}


// voxgraph::RegistrationPoint
struct RegistrationPoint {
  float position[3];
  float distance;
  float weight;
};
static_assert(sizeof(RegistrationPoint) == 20, "5 packed floats expected by the C ABI");

// voxgraph::RegistrationCostFunction for one (reference submap, reading submap) pair.  Derive from
// ceres::CostFunction in a Ceres build; Evaluate has exactly the Ceres signature.
class RegistrationCostFunction {
 public:
  RegistrationCostFunction(const std::vector<RegistrationPoint>& reference_points, const TsdfLayer& reading_layer, double no_correspondence_cost = 0.0,
                           int device = 0)
      : n_(reference_points.size()) {
    check(cox_regpoints_create(device, reference_points.empty() ? nullptr : reference_points[0].position, n_, &pts_), "RegistrationCostFunction");
    cox_reg_config cfg{no_correspondence_cost};
    check(cox_reg_create(pts_, reading_layer.handle(), &cfg, &reg_), "RegistrationCostFunction");
  }
  // reference points and reading layer that already live on the GPU (a finished submap's sets): nothing is copied, the
  // caller keeps both alive (`keepalive`: whatever owns them, typically the two submaps)
  RegistrationCostFunction(cox_regpoints_t* reference_points, cox_layer_t* reading_layer, double no_correspondence_cost = 0.0,
                           std::vector<std::shared_ptr<const void>> keepalive = std::vector<std::shared_ptr<const void>>())
      : owns_points_(false), pts_(reference_points), keepalive_(std::move(keepalive)) {
    uint64_t n = 0;
    check(cox_regpoints_size(pts_, &n), "RegistrationCostFunction");
    n_ = static_cast<size_t>(n);
    cox_reg_config cfg{no_correspondence_cost};
    check(cox_reg_create(pts_, reading_layer, &cfg, &reg_), "RegistrationCostFunction");
  }
  RegistrationCostFunction(const RegistrationCostFunction&) = delete;
  RegistrationCostFunction& operator=(const RegistrationCostFunction&) = delete;
  ~RegistrationCostFunction() {
    cox_reg_destroy(reg_);
    if (owns_points_) cox_regpoints_destroy(pts_);
  }
  int num_residuals() const { return static_cast<int>(drawn_ ? drawn_ : (sample_idx_.empty() ? n_ : sample_idx_.size())); }
  size_t num_points() const { return n_; }
  // WeightedSampler::getRandomItem for a whole evaluation, on the GPU: n draws with replacement, weight-proportional,
  // reproducible from the seed (they stay on the GPU and are used until the next draw / setSampleIndices)
  void drawSamples(uint64_t n, uint64_t seed) {
    check(cox_reg_draw_samples(reg_, n, seed), "drawSamples");
    sample_idx_.clear();
    drawn_ = static_cast<size_t>(n);
  }
  // the weighted sampler's draws (sampling_ratio > 0); empty = every point once (sampling_ratio = -1)
  // (the indices are uploaded once and stay on the GPU)
  void setSampleIndices(const std::vector<uint32_t>& idx) {
    sample_idx_ = idx;
    drawn_ = 0;
    check(cox_reg_set_samples(reg_, idx.empty() ? nullptr : idx.data(), idx.size()), "setSampleIndices");
  }
  // voxgraph's RegistrationCostFunction::Config::jacobian_evaluation_method (kAnalytic is voxgraph's default and what coxgraph's server
  // runs; kNumeric is its debugging aid: central differences of the residuals, here with a step of 1e-4 per parameter -- the
  // interpolation inside is float, so a residual carries ~1e-7 of rounding noise and a 1e-6 step would turn it into 0.05 of Jacobian)
  enum class JacobianEvaluationMethod { kAnalytic = 0, kNumeric };
  void setJacobianEvaluationMethod(JacobianEvaluationMethod m) { jacobian_method_ = m; }
  // ceres::CostFunction::Evaluate: parameters = {reference pose (x,y,z,yaw), reading pose}, jacobians row-major N x 4
  bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const {
    if (jacobian_method_ == JacobianEvaluationMethod::kAnalytic || !jacobians)
      return cox_reg_evaluate(reg_, parameters[0], parameters[1], nullptr, num_residuals(), residuals, jacobians ? jacobians[0] : nullptr,
                              jacobians ? jacobians[1] : nullptr) == COX_OK;
    const size_t n = static_cast<size_t>(num_residuals());
    if (residuals && cox_reg_evaluate(reg_, parameters[0], parameters[1], nullptr, n, residuals, nullptr, nullptr) != COX_OK) return false;
    std::vector<double> rp(n), rm(n);
    for (int blk = 0; blk < 2; ++blk) {
      if (!jacobians[blk]) continue;
      for (int k = 0; k < 4; ++k) {
        double p[2][4];
        for (int b2 = 0; b2 < 2; ++b2)
          for (int c = 0; c < 4; ++c) p[b2][c] = parameters[b2][c];
        const double h = 1e-4;
        p[blk][k] = parameters[blk][k] + h;
        if (cox_reg_evaluate(reg_, p[0], p[1], nullptr, n, rp.data(), nullptr, nullptr) != COX_OK) return false;
        p[blk][k] = parameters[blk][k] - h;
        if (cox_reg_evaluate(reg_, p[0], p[1], nullptr, n, rm.data(), nullptr, nullptr) != COX_OK) return false;
        for (size_t i = 0; i < n; ++i) jacobians[blk][4 * i + k] = (rp[i] - rm[i]) / (2.0 * h);
      }
    }
    return true;
  }
  // fused Gauss-Newton block: H = J^T J (8x8), b = J^T r, cost = |r|^2 / 2
  bool NormalEquations(const double ref[4], const double read[4], double H[64], double b[8], double* cost) const {
    return cox_reg_normal_eq(reg_, ref, read, nullptr, num_residuals(), H, b, cost, nullptr) == COX_OK;
  }
  // the same in two halves: a solver begins every constraint of an evaluation, then collects them (the kernels overlap)
  bool BeginNormalEquations(const double ref[4], const double read[4]) const { return cox_reg_normal_eq_begin(reg_, ref, read, nullptr, num_residuals()) == COX_OK; }
  bool FinishNormalEquations(double H[64], double b[8], double* cost) const { return cox_reg_normal_eq_finish(reg_, H, b, cost, nullptr) == COX_OK; }
  // every constraint of one pose-graph evaluation in ONE launch (cox_reg_normal_eq_batch): costs[c] with poses ref[4c..] / read[4c..];
  // each uses its drawn / set samples (or all its points).  All on one GPU.  H: 64 per constraint, b: 8, cost: 1.
  static bool NormalEquationsBatch(const std::vector<const RegistrationCostFunction*>& costs, const double* ref, const double* read, double* H, double* b,
                                   double* cost) {
    std::vector<cox_reg_t*> regs;
    for (const RegistrationCostFunction* c : costs) {
      regs.push_back(c->reg_);
    }
    return cox_reg_normal_eq_batch(regs.data(), regs.size(), ref, read, H, b, cost, nullptr) == COX_OK;
  }

 private:
  size_t n_ = 0;
  bool owns_points_ = true;
  cox_regpoints_t* pts_ = nullptr;
  cox_reg_t* reg_ = nullptr;
  std::vector<uint32_t> sample_idx_;
  size_t drawn_ = 0;
  JacobianEvaluationMethod jacobian_method_ = JacobianEvaluationMethod::kAnalytic;
  std::vector<std::shared_ptr<const void>> keepalive_;
};

}  // namespace coxgraph_hip
