// The submap seam in C++ (header-only, C++14, no dependencies): what coxgraph reaches through
// `using CliSm = voxgraph::VoxgraphSubmap` (coxgraph/include/coxgraph/common.h:23) and
// `SubmapCollection : voxgraph::VoxgraphSubmapCollection` (coxgraph/include/coxgraph/server/submap_collection.h:19),
// with the TSDF layer, the ESDF and both registration point sets resident on the GPU.
//
//   VoxgraphSubmap(T_M_S, id, config), getID, getPose / setPose, addPoseToHistory, getPoseHistory, getStartTime /
//   getEndTime, lookupPoseByTime, getTsdfMapPtr()->getTsdfLayerPtr(), finishSubmap(), getOdomFrameSurfaceAabb,
//   overlapsWith                                     coxgraph/include/coxgraph/utils/msg_converter.h:46-118,
//                                                    src/client/coxgraph_client.cpp:55-63, server/distribution/distribution_controller.h:84
//   SubmapCollection::addSubmap(ptr, cid, cli_sm_id), getSerSmIdsByCliId, getSerSmIdByCliSmId, getCliSmIdsByCliId, getOriPose,
//   updateOriPose, getCliIdPairBySsid, mergeToCliMap   coxgraph/include/coxgraph/server/submap_collection.h:19-93,
//                                                      src/server/submap_collection.cpp:10-37
//
// finishSubmap() is one call per received submap in the reference (msg_converter.h:113, submap_collection.cpp:35) and runs
// voxblox's ESDF integrator and mesher on the CPU; here it is cox_esdf_from_tsdf + cox_regpoints_from_layer +
// cox_regpoints_from_isosurface + cox_layer_surface_obb, all on the submap's GPU.
#pragma once
#include <algorithm>
#include <cmath>
#include <limits>
#include <map>
#include <memory>
#include <utility>
#include <vector>

#include "coxgraph_hip_adapters.hpp"

namespace coxgraph_hip {

typedef unsigned int SubmapID;  // cblox::SubmapID; coxgraph's SerSmId / CliSmId
typedef int CliId;

// ros::Time as the pose history's key
struct Time {
  uint32_t sec = 0, nsec = 0;
  Time() {}
  Time(uint32_t s, uint32_t ns) : sec(s), nsec(ns) {}
  double toSec() const { return static_cast<double>(sec) + 1e-9 * static_cast<double>(nsec); }
  bool operator<(const Time& o) const { return sec < o.sec || (sec == o.sec && nsec < o.nsec); }
  bool operator==(const Time& o) const { return sec == o.sec && nsec == o.nsec; }
};

// ---- kindr::minimal::QuatTransformationTemplate<float> arithmetic (Eigen's operation order, as in the kernels) -----------
inline void rotate(const float q[4], const float v[3], float out[3]) {
  const float qv[3] = {q[1], q[2], q[3]};
  float uv[3] = {qv[1] * v[2] - qv[2] * v[1], qv[2] * v[0] - qv[0] * v[2], qv[0] * v[1] - qv[1] * v[0]};
  for (int k = 0; k < 3; ++k) uv[k] = uv[k] + uv[k];
  const float c[3] = {qv[1] * uv[2] - qv[2] * uv[1], qv[2] * uv[0] - qv[0] * uv[2], qv[0] * uv[1] - qv[1] * uv[0]};
  for (int k = 0; k < 3; ++k) out[k] = (v[k] + q[0] * uv[k]) + c[k];
}
inline void transformPoint(const Transformation& T, const float p[3], float out[3]) {
  rotate(T.q, p, out);
  for (int k = 0; k < 3; ++k) out[k] += T.t[k];
}
inline Transformation operator*(const Transformation& A, const Transformation& B) {
  Transformation C;
  const float *a = A.q, *b = B.q;
  C.q[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  C.q[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  C.q[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  C.q[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  float r[3];
  rotate(A.q, B.t, r);
  for (int k = 0; k < 3; ++k) C.t[k] = A.t[k] + r[k];
  return C;
}
inline Transformation inverse(const Transformation& T) {
  Transformation I;
  I.q[0] = T.q[0];
  for (int k = 1; k < 4; ++k) I.q[k] = -T.q[k];
  float r[3];
  rotate(I.q, T.t, r);
  for (int k = 0; k < 3; ++k) I.t[k] = -r[k];
  return I;
}
// voxgraph's 4-DoF node pose of a transformation: translation + the z component of the rotation vector (log()[5]);
// roll and pitch are dropped (submaps are gravity aligned)
inline void pose4FromTransformation(const Transformation& T, double out[4]) {
  const double w = T.q[0], x = T.q[1], y = T.q[2], z = T.q[3];
  const double nv = std::sqrt(x * x + y * y + z * z);
  double yaw = 0.0;
  if (nv > 1e-12) {
    double angle = 2.0 * std::atan2(nv, w);
    if (angle > M_PI) angle -= 2.0 * M_PI;
    yaw = z / nv * angle;
  }
  out[0] = T.t[0];
  out[1] = T.t[1];
  out[2] = T.t[2];
  out[3] = yaw;
}
inline Transformation transformationFromPose4(const double p[4]) {
  Transformation T;
  T.q[0] = static_cast<float>(std::cos(0.5 * p[3]));
  T.q[1] = T.q[2] = 0.0f;
  T.q[3] = static_cast<float>(std::sin(0.5 * p[3]));
  for (int k = 0; k < 3; ++k) T.t[k] = static_cast<float>(p[k]);
  return T;
}

// voxgraph::BoundingBox
struct BoundingBox {
  float min[3], max[3];
  BoundingBox() {
    for (int k = 0; k < 3; ++k) {
      min[k] = std::numeric_limits<float>::infinity();
      max[k] = -std::numeric_limits<float>::infinity();
    }
  }
  bool empty() const { return min[0] > max[0] || min[1] > max[1] || min[2] > max[2]; }
  // the 8 corners through the pose, then min / max
  static BoundingBox getAabbFromObbAndPose(const BoundingBox& obb, const Transformation& pose) {
    BoundingBox out;
    if (obb.empty()) return out;
    for (int c = 0; c < 8; ++c) {
      const float p[3] = {(c & 1) ? obb.max[0] : obb.min[0], (c & 2) ? obb.max[1] : obb.min[1], (c & 4) ? obb.max[2] : obb.min[2]};
      float q[3];
      transformPoint(pose, p, q);
      for (int k = 0; k < 3; ++k) {
        out.min[k] = std::min(out.min[k], q[k]);
        out.max[k] = std::max(out.max[k], q[k]);
      }
    }
    return out;
  }
  // separation along any axis -> no overlap
  bool overlapsWith(const BoundingBox& o) const {
    if (empty() || o.empty()) return false;
    for (int k = 0; k < 3; ++k)
      if (max[k] < o.min[k] || min[k] > o.max[k]) return false;
    return true;
  }
};

enum class RegistrationPointType { kVoxels = 0, kIsosurfacePoints = 1 };  // "implicit_to_implicit" / "explicit_to_implicit"

// a registration point set on the GPU (shared between the copies of a submap and the constraints that read it)
class RegistrationPointSet {
 public:
  explicit RegistrationPointSet(cox_regpoints_t* h) : h_(h) { check(cox_regpoints_size(h_, &n_), "regpoints_size"); }
  ~RegistrationPointSet() { cox_regpoints_destroy(h_); }
  RegistrationPointSet(const RegistrationPointSet&) = delete;
  RegistrationPointSet& operator=(const RegistrationPointSet&) = delete;
  cox_regpoints_t* handle() const { return h_; }
  uint64_t size() const { return n_; }

 private:
  cox_regpoints_t* h_;
  uint64_t n_ = 0;
};

// a layer handle this file owns (the TSDF of a submap, its ESDF, or a copy that arrived from another GPU)
class LayerHandle {
 public:
  explicit LayerHandle(cox_layer_t* h) : h_(h) {}
  ~LayerHandle() { cox_layer_destroy(h_); }
  LayerHandle(const LayerHandle&) = delete;
  LayerHandle& operator=(const LayerHandle&) = delete;
  cox_layer_t* handle() const { return h_; }

 private:
  cox_layer_t* h_;
};

class VoxgraphSubmap {
 public:
  typedef std::shared_ptr<VoxgraphSubmap> Ptr;
  typedef std::shared_ptr<const VoxgraphSubmap> ConstPtr;
  typedef std::map<Time, Transformation> PoseHistoryMap;

  struct Config {  // voxgraph::VoxgraphSubmap::Config (cblox::TsdfEsdfSubmap::Config + the registration filter)
    float tsdf_voxel_size = 0.2f;
    size_t tsdf_voxels_per_side = 16;
    int device = 0;
    uint64_t capacity_blocks = 0;
    struct RegistrationFilter {
      double min_voxel_weight = 1.0;
      double max_voxel_distance = 0.3;
      bool use_esdf_distance = true;
    } registration_filter;
    cox_esdf_config esdf;
    float vertex_proximity_threshold_voxels = 0.5f;  // createConnectedMesh threshold, in voxels
    Config() { cox_esdf_config_default(&esdf); }
  };

  // cblox / voxgraph expose the layer as getTsdfMapPtr()->getTsdfLayerPtr()
  struct TsdfMap {
    TsdfLayer layer;
    TsdfMap(float voxel_size, size_t vps, int device, uint64_t cap) : layer(voxel_size, vps, device, cap) {}
    TsdfMap(const TsdfLayer& other, int device) : layer(other, device) {}
    TsdfLayer* getTsdfLayerPtr() { return &layer; }
    const TsdfLayer& getTsdfLayer() const { return layer; }
    float block_size() const { return layer.block_size(); }
    float voxel_size() const { return layer.voxel_size(); }
  };

  VoxgraphSubmap(const Transformation& T_M_S, SubmapID submap_id, const Config& config)
      : config_(config), id_(submap_id), T_M_S_(T_M_S),
        tsdf_map_(new TsdfMap(config.tsdf_voxel_size, config.tsdf_voxels_per_side, config.device, config.capacity_blocks)) {}

  // deep copy (the cblox fork's submap_deep_copy_constructors): an own copy of the layer; the finished products (ESDF, point
  // sets) are immutable and stay shared
  VoxgraphSubmap(const VoxgraphSubmap& rhs)
      : config_(rhs.config_), id_(rhs.id_), T_M_S_(rhs.T_M_S_), pose_history_(rhs.pose_history_), finished_(rhs.finished_), esdf_(rhs.esdf_),
        relevant_voxels_(rhs.relevant_voxels_), isosurface_vertices_(rhs.isosurface_vertices_), surface_obb_(rhs.surface_obb_),
        mesh_pointcloud_(rhs.mesh_pointcloud_) {
    tsdf_map_.reset(new TsdfMap(rhs.tsdf_map_->layer, config_.device));
  }
  // a submap around a copy of a layer that lives on a GPU of this process (the client's): peer copy instead of a ROS message
  VoxgraphSubmap(const Transformation& T_M_S, SubmapID submap_id, const Config& config, const TsdfLayer& layer_to_copy)
      : config_(config), id_(submap_id), T_M_S_(T_M_S), tsdf_map_(new TsdfMap(layer_to_copy, config.device)) {}
  VoxgraphSubmap& operator=(const VoxgraphSubmap&) = delete;

  SubmapID getID() const { return id_; }
  const Config& getConfig() const { return config_; }
  TsdfMap* getTsdfMapPtr() { return tsdf_map_.get(); }
  const TsdfMap& getTsdfMap() const { return *tsdf_map_; }

  const Transformation& getPose() const { return T_M_S_; }
  void setPose(const Transformation& T_M_S) { T_M_S_ = T_M_S; }

  void addPoseToHistory(const Time& timestamp, const Transformation& T_submap_base) { pose_history_[timestamp] = T_submap_base; }
  const PoseHistoryMap& getPoseHistory() const { return pose_history_; }
  Time getStartTime() const { return pose_history_.empty() ? Time() : pose_history_.begin()->first; }
  Time getEndTime() const { return pose_history_.empty() ? Time() : pose_history_.rbegin()->first; }
  // the last pose at or before the timestamp; false when the submap was not in use yet
  bool lookupPoseByTime(const Time& timestamp, Transformation* T_submap_robot) const {
    PoseHistoryMap::const_iterator it = pose_history_.upper_bound(timestamp);
    if (it == pose_history_.begin()) return false;
    --it;
    *T_submap_robot = it->second;
    return true;
  }

  // generateEsdf + findRelevantVoxelIndices + findIsosurfaceVertices + the surface box, on the GPU
  void finishSubmap() {
    cox_layer_t* L = tsdf_map_->layer.handle();
    const float vs = tsdf_map_->layer.voxel_size();
    cox_layer_t* e = nullptr;
    check(cox_esdf_from_tsdf(L, &config_.esdf, &e), "finishSubmap: ESDF");
    esdf_.reset(new LayerHandle(e));
    cox_regpoints_t* r = nullptr;
    check(cox_regpoints_from_layer(L, static_cast<float>(config_.registration_filter.min_voxel_weight), static_cast<float>(config_.registration_filter.max_voxel_distance), &r),
          "finishSubmap: relevant voxels");
    relevant_voxels_.reset(new RegistrationPointSet(r));
    r = nullptr;
    check(cox_regpoints_from_isosurface(L, static_cast<float>(config_.registration_filter.min_voxel_weight), config_.vertex_proximity_threshold_voxels * vs, &r, nullptr, nullptr),
          "finishSubmap: isosurface vertices");
    isosurface_vertices_.reset(new RegistrationPointSet(r));
    uint64_t n = 0;
    check(cox_layer_surface_obb(L, surface_obb_.min, surface_obb_.max, &n), "finishSubmap: surface box");
    finished_ = true;
  }
  bool isFinished() const { return finished_; }

  const std::shared_ptr<RegistrationPointSet>& getRegistrationPoints(RegistrationPointType type) const {
    if (!finished_) throw std::runtime_error("The cached registration points are only available for finished submaps");
    return type == RegistrationPointType::kVoxels ? relevant_voxels_ : isosurface_vertices_;
  }
  // the distance field a registration constraint reads from this submap
  cox_layer_t* getReadingLayer(bool use_esdf_distance) const {
    if (use_esdf_distance) {
      if (!finished_) throw std::runtime_error("The ESDF is only available for finished submaps");
      return esdf_->handle();
    }
    return tsdf_map_->layer.handle();
  }
  const BoundingBox& getSubmapFrameSurfaceObb() const { return surface_obb_; }
  BoundingBox getMissionFrameSurfaceAabb() const { return BoundingBox::getAabbFromObbAndPose(surface_obb_, T_M_S_); }
  BoundingBox getOdomFrameSurfaceAabb() const { return getMissionFrameSurfaceAabb(); }  // distribution_controller.h:84
  bool overlapsWith(const VoxgraphSubmap& other) const { return getMissionFrameSurfaceAabb().overlapsWith(other.getMissionFrameSurfaceAabb()); }

 private:
  Config config_;
  SubmapID id_;
  Transformation T_M_S_;
  PoseHistoryMap pose_history_;
  std::unique_ptr<TsdfMap> tsdf_map_;
  bool finished_ = false;
  std::shared_ptr<LayerHandle> esdf_;
  std::shared_ptr<RegistrationPointSet> relevant_voxels_, isosurface_vertices_;
  BoundingBox surface_obb_;

 public:
  // the fork's public member: the client's mesh as a sensor_msgs/PointCloud2, carried through untouched (msg_converter.h:72,114)
  std::shared_ptr<std::vector<uint8_t>> mesh_pointcloud_ = std::make_shared<std::vector<uint8_t>>();
};

// coxgraph::server::SubmapCollection
class SubmapCollection {
 public:
  typedef std::shared_ptr<SubmapCollection> Ptr;
  typedef std::pair<CliId, SubmapID> CIdCSIdPair;

  explicit SubmapCollection(const VoxgraphSubmap::Config& submap_config, int client_number = 3) : submap_config_(submap_config), client_number_(client_number) {}
  // copy: submaps are deep-copied (the final-mesh what-if collection, src/server/visualizer/server_visualizer.cpp:28-31)
  SubmapCollection(const SubmapCollection& rhs)
      : submap_config_(rhs.submap_config_), client_number_(rhs.client_number_), sm_cli_id_map_(rhs.sm_cli_id_map_), cli_ser_sm_id_map_(rhs.cli_ser_sm_id_map_),
        sm_id_ori_pose_map_(rhs.sm_id_ori_pose_map_) {
    for (const auto& kv : rhs.submaps_) submaps_[kv.first] = std::make_shared<VoxgraphSubmap>(*kv.second);
  }
  SubmapCollection& operator=(const SubmapCollection&) = delete;

  int getClientNumber() const { return client_number_; }
  const VoxgraphSubmap::Config& getConfig() const { return submap_config_; }

  // submap_collection.cpp:10-22
  Transformation addSubmap(const VoxgraphSubmap::Ptr& submap_ptr, CliId cid, SubmapID cli_sm_id) {
    if (!submap_ptr) throw std::runtime_error("addSubmap: null submap");
    const SubmapID id = submap_ptr->getID();
    if (submaps_.count(id)) throw std::runtime_error("addSubmap: submap id exists");
    submaps_[id] = submap_ptr;
    sm_cli_id_map_.emplace(id, CIdCSIdPair(cid, cli_sm_id));
    cli_ser_sm_id_map_[cid].push_back(id);
    sm_id_ori_pose_map_.emplace(id, submap_ptr->getPose());
    return Transformation();
  }
  bool exists(SubmapID id) const { return submaps_.count(id) != 0; }
  size_t size() const { return submaps_.size(); }
  VoxgraphSubmap::Ptr getSubmapPtr(SubmapID id) const {
    const auto it = submaps_.find(id);
    return it == submaps_.end() ? nullptr : it->second;
  }
  VoxgraphSubmap::ConstPtr getSubmapConstPtr(SubmapID id) const { return getSubmapPtr(id); }
  std::vector<VoxgraphSubmap::ConstPtr> getSubmapConstPtrs() const {
    std::vector<VoxgraphSubmap::ConstPtr> v;
    for (const auto& kv : submaps_) v.push_back(kv.second);
    return v;
  }
  std::vector<SubmapID> getIDs() const {
    std::vector<SubmapID> v;
    for (const auto& kv : submaps_) v.push_back(kv.first);
    return v;
  }
  bool getSubmapPose(SubmapID id, Transformation* pose) const {
    const auto it = submaps_.find(id);
    if (it == submaps_.end()) return false;
    *pose = it->second->getPose();
    return true;
  }
  bool setSubmapPose(SubmapID id, const Transformation& pose) {
    const auto it = submaps_.find(id);
    if (it == submaps_.end()) return false;
    it->second->setPose(pose);
    return true;
  }
  // submap_collection.h:44-93
  bool getSerSmIdsByCliId(CliId cid, std::vector<SubmapID>* ser_sids) const {
    const auto it = cli_ser_sm_id_map_.find(cid);
    if (it == cli_ser_sm_id_map_.end()) return false;
    *ser_sids = it->second;
    return true;
  }
  bool getSerSmIdByCliSmId(CliId cid, SubmapID cli_sm_id, SubmapID* ser_sm_id) const {
    const auto it = cli_ser_sm_id_map_.find(cid);
    if (it == cli_ser_sm_id_map_.end()) return false;
    for (SubmapID s : it->second)
      if (sm_cli_id_map_.at(s).second == cli_sm_id) {
        *ser_sm_id = s;
        return true;
      }
    return false;
  }
  bool getCliSmIdsByCliId(CliId cid, std::vector<SubmapID>* cli_sids) const {
    cli_sids->clear();
    const auto it = cli_ser_sm_id_map_.find(cid);
    if (it != cli_ser_sm_id_map_.end())
      for (SubmapID s : it->second) cli_sids->push_back(sm_cli_id_map_.at(s).second);
    return !cli_sids->empty();
  }
  void updateOriPose(SubmapID id, const Transformation& pose) { sm_id_ori_pose_map_[id] = pose; }
  Transformation getOriPose(SubmapID id) const { return sm_id_ori_pose_map_.at(id); }
  CIdCSIdPair getCliIdPairBySsid(SubmapID id) const { return sm_cli_id_map_.at(id); }
  // submap_collection.cpp:24-37: merge a re-sent submap into the one the server already holds, finish it again
  Transformation mergeToCliMap(const VoxgraphSubmap::Ptr& submap_ptr) {
    const VoxgraphSubmap::Ptr cli_map_ptr = getSubmapPtr(submap_ptr->getID());
    if (!cli_map_ptr) throw std::runtime_error("mergeToCliMap: unknown submap");
    mergeLayerAintoLayerB(submap_ptr->getTsdfMap().getTsdfLayer(), cli_map_ptr->getTsdfMapPtr()->getTsdfLayerPtr());
    cli_map_ptr->finishSubmap();
    return submap_ptr->getPose() * inverse(cli_map_ptr->getPose());
  }

 private:
  VoxgraphSubmap::Config submap_config_;
  int client_number_;
  std::map<SubmapID, VoxgraphSubmap::Ptr> submaps_;
  std::map<SubmapID, CIdCSIdPair> sm_cli_id_map_;
  std::map<CliId, std::vector<SubmapID>> cli_ser_sm_id_map_;
  std::map<SubmapID, Transformation> sm_id_ori_pose_map_;
};

// coxgraph::utils::cliSubmapFromMsg (utils/msg_converter.h:85-118): the wire layer + trajectory of a ClientSubmap message
// become a finished submap on the server's GPU
struct StampedPose {
  Time stamp;
  Transformation T_submap_base_link;
};
inline VoxgraphSubmap::Ptr cliSubmapFromMsg(SubmapID ser_sm_id, const VoxgraphSubmap::Config& submap_config, const LayerMsg& layer_msg,
                                            const std::vector<StampedPose>& trajectory, const Transformation& map_pose) {
  VoxgraphSubmap::Ptr submap_ptr(new VoxgraphSubmap(Transformation(), ser_sm_id, submap_config));
  for (const StampedPose& p : trajectory) submap_ptr->addPoseToHistory(p.stamp, p.T_submap_base_link);
  if (!submap_ptr->getPoseHistory().empty()) {
    submap_ptr->setPose(map_pose);
    if (!deserializeMsgToLayer(layer_msg, submap_ptr->getTsdfMapPtr()->getTsdfLayerPtr())) throw std::runtime_error("Received a submap msg with an invalid TSDF");
    submap_ptr->finishSubmap();
  }
  return submap_ptr;
}
// the same when the client's layer lives on a GPU of this process (one client per GPU): block array + keys go peer to peer
inline VoxgraphSubmap::Ptr cliSubmapFromDevice(SubmapID ser_sm_id, const VoxgraphSubmap::Config& submap_config, const TsdfLayer& client_layer,
                                               const std::vector<StampedPose>& trajectory, const Transformation& map_pose) {
  if (trajectory.empty()) return VoxgraphSubmap::Ptr(new VoxgraphSubmap(Transformation(), ser_sm_id, submap_config));
  VoxgraphSubmap::Ptr submap_ptr(new VoxgraphSubmap(map_pose, ser_sm_id, submap_config, client_layer));
  for (const StampedPose& p : trajectory) submap_ptr->addPoseToHistory(p.stamp, p.T_submap_base_link);
  submap_ptr->finishSubmap();
  return submap_ptr;
}

}  // namespace coxgraph_hip
