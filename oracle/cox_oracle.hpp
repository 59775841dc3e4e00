// CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.
//
// Plain single-file C++17 restatement of the algorithms on coxgraph's hot path
// (voxblox-style TSDF integration, voxgraph-style submap registration cost).
// It is the *checker* for the HIP engine in coxgraph_amd/csrc; only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, link, load
// or call anything in this directory.  Nothing here ships in the product path.
//
// "PARITY UNPINNED": the arithmetic restated here lives in un-vendored third-party
// forks that are NOT under /root/reference (LXYYY/voxblox@feature/multi_robots,
// LXYYY/voxgraph@feature/multi_robots, LXYYY/cblox@feature/submap_deep_copy_constructors,
// pinned by branch name only in coxgraph_ssh.rosinstall:1-8,55-58), the reference
// ships no tests, fixtures or golden vectors, and it cannot be compiled offline.
// The restatement follows the published upstream algorithms (voxblox
// integrator_utils / tsdf_integrator / interpolator, voxgraph
// registration_cost_function) and is anchored on coxgraph's own call sites:
//   integratePointCloud(T_G_C, points_C, colors, false)  coxgraph/include/coxgraph/map_comm/tsdf_recover.h:75
//   removeAllBlocks / getMemorySize / serializeLayerAsMsg coxgraph/include/coxgraph/map_comm/tsdf_recover.h:62,92,95
//   deserializeMsgToLayer                                 coxgraph/include/coxgraph/utils/msg_converter.h:107
//   addForceRegistrationConstraint                        coxgraph/src/server/pose_graph_interface.cpp:88-105
//   two-stage optimize                                    coxgraph/src/server/pose_graph_interface.cpp:32-49
//   RelativePoseCostFunction / sqrt information           coxgraph/include/coxgraph/server/backend/relative_pose_constraint.h:28-61,114-119
// The only independent checks are the hand-derivable known-answer tests in
// tests/test_oracle_kat.py.
//
// All float arithmetic is written with explicit operation order and must be
// compiled with -ffp-contract=off so that index-deciding results are
// reproducible bit-for-bit by the HIP kernels.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <tuple>
#include <unordered_map>
#include <vector>

namespace coxo {

constexpr float kEps = 1e-6f;  // voxblox kEpsilon == kCoordinateEpsilon == kFloatEpsilon

// ---------------------------------------------------------------------------------------------
// A.1 rigid transform (minkindr QuatTransformationTemplate<float>, Eigen quaternion op order)
// ---------------------------------------------------------------------------------------------
struct V3 {
  float x, y, z;
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }  // Eigen linear redux order
inline float norm(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
// Eigen MatrixBase::normalized(): z = squaredNorm; if (z > 0) return n / sqrt(z); else return n;
inline V3 normalized(V3 a) {
  const float z = dot(a, a);
  if (z > 0.0f) return a / std::sqrt(z);
  return a;
}

struct Transform {
  float qw, qx, qy, qz;  // unit quaternion (w,x,y,z)
  V3 t;
};
// Eigen Quaternion::_transformVector: uv = q.vec x v; uv += uv; v + w*uv + q.vec x uv
inline V3 rotate(const Transform& T, V3 v) {
  const V3 qv{T.qx, T.qy, T.qz};
  V3 uv = cross(qv, v);
  uv = uv + uv;
  const V3 c = cross(qv, uv);
  return {(v.x + T.qw * uv.x) + c.x, (v.y + T.qw * uv.y) + c.y, (v.z + T.qw * uv.z) + c.z};
}
inline V3 transform(const Transform& T, V3 p) { return rotate(T, p) + T.t; }

// ---------------------------------------------------------------------------------------------
// A.2 index math
// ---------------------------------------------------------------------------------------------
struct GIdx {
  int64_t x, y, z;
  bool operator==(const GIdx& o) const { return x == o.x && y == o.y && z == o.z; }
  bool operator!=(const GIdx& o) const { return !(*this == o); }
};
struct BIdx {
  int x, y, z;
  bool operator==(const BIdx& o) const { return x == o.x && y == o.y && z == o.z; }
  bool operator!=(const BIdx& o) const { return !(*this == o); }
};
struct AnyIndexHash {  // voxblox AnyIndexHash: unsigned int(x + y*17191 + z*17191^2)
  size_t operator()(const BIdx& i) const {
    constexpr size_t sl = 17191, sl2 = sl * sl;
    return static_cast<unsigned int>(i.x + i.y * sl + i.z * sl2);
  }
};
struct LongIndexHash {  // voxblox LongIndexHash
  size_t operator()(const GIdx& i) const {
    constexpr size_t sl = 17191, sl2 = sl * sl;
    return static_cast<size_t>(i.x + i.y * sl + i.z * sl2);
  }
};
inline GIdx gridIndexFromPoint(V3 p, float inv) {
  return {static_cast<int64_t>(std::floor(p.x * inv + kEps)), static_cast<int64_t>(std::floor(p.y * inv + kEps)),
          static_cast<int64_t>(std::floor(p.z * inv + kEps))};
}
inline GIdx gridIndexFromScaledPoint(V3 p) {
  return {static_cast<int64_t>(std::floor(p.x + kEps)), static_cast<int64_t>(std::floor(p.y + kEps)),
          static_cast<int64_t>(std::floor(p.z + kEps))};
}
inline BIdx blockIndexFromPoint(V3 p, float inv) {
  return {static_cast<int>(std::floor(p.x * inv + kEps)), static_cast<int>(std::floor(p.y * inv + kEps)),
          static_cast<int>(std::floor(p.z * inv + kEps))};
}
// (float(idx) + 0.5) * size : 0.5 is a double literal -> evaluated in double, narrowed to float
inline float centerCoord(int64_t idx, float size) {
  return static_cast<float>((static_cast<float>(idx) + 0.5) * size);
}
inline V3 centerPointFromGridIndex(GIdx g, float size) { return {centerCoord(g.x, size), centerCoord(g.y, size), centerCoord(g.z, size)}; }
inline BIdx blockFromGlobal(GIdx g, float vps_inv) {
  return {static_cast<int>(std::floor(static_cast<float>(g.x) * vps_inv)), static_cast<int>(std::floor(static_cast<float>(g.y) * vps_inv)),
          static_cast<int>(std::floor(static_cast<float>(g.z) * vps_inv))};
}
inline int localFromGlobal(int64_t g, int vps) { return static_cast<int>((g + (int64_t(1) << 31)) & (vps - 1)); }
inline int linearIndex(int lx, int ly, int lz, int vps) { return lx + vps * (ly + lz * vps); }

// ---------------------------------------------------------------------------------------------
// A.3 voxel / block / layer
// ---------------------------------------------------------------------------------------------
struct Color {
  uint8_t r = 0, g = 0, b = 0, a = 0;
};
inline Color blendTwoColors(Color c1, float w1, Color c2, float w2) {
  const float tot = w1 + w2;
  w1 /= tot;
  w2 /= tot;
  Color o;
  o.r = static_cast<uint8_t>(std::round(c1.r * w1 + c2.r * w2));
  o.g = static_cast<uint8_t>(std::round(c1.g * w1 + c2.g * w2));
  o.b = static_cast<uint8_t>(std::round(c1.b * w1 + c2.b * w2));
  o.a = static_cast<uint8_t>(std::round(c1.a * w1 + c2.a * w2));
  return o;
}
struct TsdfVoxel {
  float distance = 0.0f;
  float weight = 0.0f;
  Color color;
};
static_assert(sizeof(TsdfVoxel) == 12, "TsdfVoxel must be 12 bytes");

struct Block {
  BIdx index;
  V3 origin;
  std::vector<TsdfVoxel> voxels;
  bool updated = false;
  Block(BIdx idx, int vps, float block_size)
      : index(idx), origin{static_cast<float>(idx.x) * block_size, static_cast<float>(idx.y) * block_size, static_cast<float>(idx.z) * block_size},
        voxels(static_cast<size_t>(vps) * vps * vps) {}
};

struct Layer {
  float voxel_size;
  int vps;
  float voxel_size_inv, block_size, block_size_inv, vps_inv;
  std::unordered_map<BIdx, std::shared_ptr<Block>, AnyIndexHash> blocks;
  Layer(float vs, int v) : voxel_size(vs), vps(v) {
    voxel_size_inv = static_cast<float>(1.0 / voxel_size);
    block_size = voxel_size * static_cast<float>(vps);
    block_size_inv = static_cast<float>(1.0 / block_size);
    vps_inv = 1.0f / static_cast<float>(vps);
  }
  Block* getBlockPtr(const BIdx& b) const {
    auto it = blocks.find(b);
    return it == blocks.end() ? nullptr : it->second.get();
  }
  Block* allocateBlock(const BIdx& b) {
    auto it = blocks.find(b);
    if (it != blocks.end()) return it->second.get();
    auto p = std::make_shared<Block>(b, vps, block_size);
    blocks.emplace(b, p);
    return p.get();
  }
  void removeAllBlocks() { blocks.clear(); }
  size_t numBlocks() const { return blocks.size(); }
  size_t memorySize() const { return blocks.size() * (static_cast<size_t>(vps) * vps * vps * sizeof(TsdfVoxel)); }
};

// ---------------------------------------------------------------------------------------------
// A.4 RayCaster (Amanatides-Woo with voxblox's quirks kept: the |ray|<0 guard never fires so
// zero ray components give -inf/NaN t values; argmin is Eigen's "first strictly smaller" scan)
// ---------------------------------------------------------------------------------------------
inline int signum(float x) { return (x == 0.0f) ? 0 : (x < 0.0f ? -1 : 1); }

struct RayCaster {
  GIdx curr{0, 0, 0};
  int64_t sgn[3] = {0, 0, 0};
  float t_to_next[3] = {0, 0, 0};
  float t_step[3] = {0, 0, 0};
  uint64_t current_step = 0;
  uint64_t ray_length_in_steps = 0;
  bool nan_ray = false;

  RayCaster(V3 origin, V3 point_G, bool is_clearing, bool carving, float max_len, float inv, float trunc, bool cast_from_origin = true) {
    const V3 d = point_G - origin;
    const V3 unit = normalized(d);
    V3 ray_start, ray_end;
    if (is_clearing) {
      float len = norm(d);
      len = std::min(std::max(len - trunc, 0.0f), max_len);
      ray_end = origin + unit * len;
      ray_start = carving ? origin : ray_end;
    } else {
      ray_end = point_G + unit * trunc;
      ray_start = carving ? origin : (point_G - unit * trunc);
    }
    const V3 s = ray_start * inv;
    const V3 e = ray_end * inv;
    if (cast_from_origin)
      setup(s, e);
    else
      setup(e, s);
  }
  // RayCaster(start_scaled, end_scaled): the constructor the projective integrator uses on block-scaled points
  RayCaster(V3 start_scaled, V3 end_scaled) { setup(start_scaled, end_scaled); }
  void setup(V3 start, V3 end) {
    if (std::isnan(start.x) || std::isnan(start.y) || std::isnan(start.z) || std::isnan(end.x) || std::isnan(end.y) || std::isnan(end.z)) {
      ray_length_in_steps = 0;
      nan_ray = true;  // upstream would emit one (uninitialised) index; we emit none and flag it
      return;
    }
    curr = gridIndexFromScaledPoint(start);
    const GIdx endi = gridIndexFromScaledPoint(end);
    current_step = 0;
    ray_length_in_steps = static_cast<uint64_t>(std::llabs(endi.x - curr.x) + std::llabs(endi.y - curr.y) + std::llabs(endi.z - curr.z));
    const float ray[3] = {end.x - start.x, end.y - start.y, end.z - start.z};
    const float st[3] = {start.x, start.y, start.z};
    const int64_t ci[3] = {curr.x, curr.y, curr.z};
    for (int k = 0; k < 3; ++k) {
      const int sg = signum(ray[k]);
      sgn[k] = sg;
      const int corrected = std::max(0, sg);
      const float shifted = st[k] - static_cast<float>(ci[k]);
      const float dist = static_cast<float>(corrected) - shifted;
      // (std::abs(ray) < 0.0) ? 2.0 : dist/ray  -- the guard is never true
      t_to_next[k] = dist / ray[k];
      t_step[k] = static_cast<float>(sg) / ray[k];
    }
  }
  bool next(GIdx* out) {
    if (nan_ray) return false;
    if (current_step++ > ray_length_in_steps) return false;
    *out = curr;
    int k = 0;
    float best = t_to_next[0];
    if (t_to_next[1] < best) {
      best = t_to_next[1];
      k = 1;
    }
    if (t_to_next[2] < best) {
      best = t_to_next[2];
      k = 2;
    }
    if (k == 0) curr.x += sgn[0];
    if (k == 1) curr.y += sgn[1];
    if (k == 2) curr.z += sgn[2];
    t_to_next[k] += t_step[k];
    return true;
  }
};

// ---------------------------------------------------------------------------------------------
// A.5 integrator config + common pieces
// ---------------------------------------------------------------------------------------------
struct TsdfConfig {
  float default_truncation_distance = 0.1f;
  float max_weight = 10000.0f;
  bool voxel_carving_enabled = true;
  float min_ray_length_m = 0.1f;
  float max_ray_length_m = 5.0f;
  bool use_const_weight = false;
  bool allow_clear = true;
  bool use_weight_dropoff = true;
  bool use_sparsity_compensation_factor = false;
  float sparsity_compensation_factor = 1.0f;
  int integrator_threads = 1;
  int integration_order_mode = 0;  // 0 = "mixed" (reference default), 2 = plain index order (not a reference mode)
  bool enable_anti_grazing = false;
  float start_voxel_subsampling_factor = 2.0f;
  int max_consecutive_ray_collisions = 2;
  int clear_checks_every_n_frames = 1;
  float max_integration_time_s = 3.4e38f;
  // oracle-only: which order merged bundles are integrated in. 0 = canonical (order of first
  // visit, what the HIP engine implements); 1 = libstdc++ unordered_map iteration order, i.e.
  // what a single-threaded upstream build linked against libstdc++ does.
  int merged_bundle_order = 0;
  // oracle-only: 0 = ApproxHashSet (reference behaviour, lossy), 1 = exact sets ("fast-exact")
  int fast_exact_sets = 0;
};

struct FrameStats {
  uint64_t n_points = 0, n_valid = 0, n_rays = 0, n_updates = 0, n_touched_voxels = 0, n_touched_blocks = 0, n_new_blocks = 0;
  uint64_t max_bundle_points = 0, max_voxel_updates = 0;
};

// voxblox MixedThreadSafeIndex::getNextIndexImpl (step_size_ = 1024, number_of_groups_ = N / 1024)
inline size_t mixedIndex(size_t seq, size_t n_points) {
  constexpr size_t step = 1024;
  const size_t groups = n_points / step;
  if (groups * step <= seq) return seq;
  return (seq % groups) * step + seq / groups;
}

// ApproxHashSet<20, 10000, GlobalIndex, LongIndexHash>
struct ApproxHashSet {
  static constexpr size_t kBits = 20, kSize = size_t(1) << kBits, kMask = kSize - 1, kFullReset = 10000;
  size_t offset = 0;
  std::unique_ptr<std::atomic<size_t>[]> table;
  ApproxHashSet() : table(new std::atomic<size_t>[kSize]) {
    for (size_t i = 0; i < kSize; ++i) table[i].store(0, std::memory_order_relaxed);
  }
  bool replaceHash(size_t h) {
    const size_t slot = (h + offset) & kMask;
    if (table[slot].load(std::memory_order_relaxed) == h + offset) return false;
    table[slot].store(h + offset, std::memory_order_relaxed);
    return true;
  }
  void reset() {
    if (++offset >= kFullReset) {
      for (size_t i = 0; i < kSize; ++i) table[i].store(0, std::memory_order_relaxed);
      offset = 0;
    }
  }
};

class Integrator {
 public:
  enum Method { kSimple = 0, kMerged = 1, kFast = 2 };
  Integrator(Layer* layer, const TsdfConfig& cfg, int method) : layer_(layer), cfg_(cfg), method_(method) {
    voxel_size_ = layer->voxel_size;
    vps_ = layer->vps;
    voxel_size_inv_ = layer->voxel_size_inv;
    vps_inv_ = layer->vps_inv;
    block_size_ = layer->block_size;
    if (cfg_.integrator_threads < 1) cfg_.integrator_threads = 1;
  }
  FrameStats last_stats;
  bool count_touched = true;  // distinct-voxel counting (off while timing the CPU baseline)

  void integratePointCloud(const Transform& T_G_C, const V3* points_C, const Color* colors, size_t n, bool freespace) {
    stats_ = FrameStats();
    stats_.n_points = n;
    touched_.clear();
    const size_t blocks_before = layer_->numBlocks();
    if (method_ == kMerged)
      integrateMerged(T_G_C, points_C, colors, n, freespace);
    else
      integrateSimpleOrFast(T_G_C, points_C, colors, n, freespace);
    updateLayerWithStoredBlocks();
    stats_.n_new_blocks = layer_->numBlocks() - blocks_before;
    stats_.n_touched_voxels = touched_.size();
    for (const auto& kv : touched_) stats_.max_voxel_updates = std::max<uint64_t>(stats_.max_voxel_updates, kv.second);
    last_stats = stats_;
  }

 Layer* layer() const { return layer_; }

 private:
  Layer* layer_;
  TsdfConfig cfg_;
  int method_;
  float voxel_size_, voxel_size_inv_, vps_inv_, block_size_;
  int vps_;
  FrameStats stats_;
  std::unordered_map<GIdx, uint32_t, LongIndexHash> touched_;
  std::unordered_map<BIdx, std::shared_ptr<Block>, AnyIndexHash> temp_blocks_;
  std::mutex temp_block_mutex_, stats_mutex_;
  std::mutex voxel_mutexes_[4096];  // voxblox ApproxHashArray<12, std::mutex, GlobalIndex, LongIndexHash>
  ApproxHashSet start_set_, observed_set_;
  std::unordered_map<GIdx, uint8_t, LongIndexHash> start_exact_, observed_exact_;
  int64_t reset_counter_ = 0;
  std::chrono::steady_clock::time_point t_start_;

  size_t orderIndex(size_t seq, size_t n) const { return cfg_.integration_order_mode == 0 ? mixedIndex(seq, n) : seq; }

  // Non-finite points: upstream would cast NaN to an int64 voxel index (undefined); voxblox_ros drops such points before
  // the integrator sees them.  Defined here (and in the engine) as invalid.
  bool isPointValid(V3 p, bool freespace, bool* is_clearing) const {
    const float r = norm(p);
    if (!(r <= 3.0e38f)) return false;
    if (r < cfg_.min_ray_length_m) return false;
    if (r > cfg_.max_ray_length_m) {
      if (cfg_.allow_clear || freespace) {
        *is_clearing = true;
        return true;
      }
      return false;
    }
    *is_clearing = freespace;
    return true;
  }
  float getVoxelWeight(V3 p) const {
    if (cfg_.use_const_weight) return 1.0f;
    const float dz = std::abs(p.z);
    if (dz > kEps) return 1.0f / (dz * dz);
    return 0.0f;
  }
  static float computeDistance(V3 origin, V3 point_G, V3 center) {
    const V3 v = center - origin;
    const V3 d = point_G - origin;
    const float dist = norm(d);
    const float proj = dot(v, d) / dist;
    return dist - proj;
  }
  TsdfVoxel* allocateStorageAndGetVoxelPtr(const GIdx& g, Block** last_block, BIdx* last_idx) {
    const BIdx b = blockFromGlobal(g, vps_inv_);
    if (*last_block == nullptr || b != *last_idx) {
      *last_block = layer_->getBlockPtr(b);
      *last_idx = b;
    }
    if (*last_block == nullptr) {
      std::lock_guard<std::mutex> lock(temp_block_mutex_);
      auto it = temp_blocks_.find(b);
      if (it != temp_blocks_.end()) {
        *last_block = it->second.get();
      } else {
        auto p = std::make_shared<Block>(b, vps_, block_size_);
        temp_blocks_.emplace(b, p);
        *last_block = p.get();
      }
    }
    (*last_block)->updated = true;
    const int lin = linearIndex(localFromGlobal(g.x, vps_), localFromGlobal(g.y, vps_), localFromGlobal(g.z, vps_), vps_);
    return &(*last_block)->voxels[lin];
  }
  void updateLayerWithStoredBlocks() {
    for (auto& kv : temp_blocks_) layer_->blocks.emplace(kv.first, kv.second);
    temp_blocks_.clear();
  }
  void updateTsdfVoxel(V3 origin, V3 point_G, const GIdx& g, Color color, float weight, TsdfVoxel* v, bool threaded) {
    const V3 c = centerPointFromGridIndex(g, voxel_size_);
    const float sdf = computeDistance(origin, point_G, c);
    float uw = weight;
    const float T = cfg_.default_truncation_distance;
    const float eps = voxel_size_;
    if (cfg_.use_weight_dropoff && sdf < -eps) {
      uw = weight * (T + sdf) / (T - eps);
      uw = std::max(uw, 0.0f);
    }
    if (cfg_.use_sparsity_compensation_factor) {
      if (std::abs(sdf) < T) uw *= cfg_.sparsity_compensation_factor;
    }
    std::unique_lock<std::mutex> lock;
    if (threaded) lock = std::unique_lock<std::mutex>(voxel_mutexes_[LongIndexHash()(g) & 4095]);
    const float nw = v->weight + uw;
    if (nw < kEps) return;
    const float nsdf = (sdf * uw + v->distance * v->weight) / nw;
    if (std::abs(sdf) < T) v->color = blendTwoColors(v->color, v->weight, color, uw);
    v->distance = (nsdf > 0.0f) ? std::min(T, nsdf) : std::max(-T, nsdf);
    v->weight = std::min(cfg_.max_weight, nw);
  }
  void noteUpdate(const GIdx& g, uint64_t* local_updates) {
    ++*local_updates;
    if (count_touched) ++touched_[g];  // single-threaded use only
  }

  // -------- simple (A.6) and fast (A.7): one ray per point ----------
  void integrateSimpleOrFast(const Transform& T, const V3* pts, const Color* cols, size_t n, bool freespace) {
    const bool fast = (method_ == kFast);
    if (fast) {
      t_start_ = std::chrono::steady_clock::now();
      if (++reset_counter_ >= cfg_.clear_checks_every_n_frames) {
        reset_counter_ = 0;
        start_set_.reset();
        observed_set_.reset();
        start_exact_.clear();
        observed_exact_.clear();
      }
    }
    const int nthreads = cfg_.integrator_threads;
    std::atomic<size_t> next_seq{0};
    auto worker = [&](bool threaded) {
      uint64_t n_valid = 0, n_rays = 0, n_updates = 0;
      for (;;) {
        const size_t seq = next_seq.fetch_add(1, std::memory_order_relaxed);
        if (seq >= n) break;
        if (fast && cfg_.max_integration_time_s < 3.0e38f) {
          const double us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_start_).count();
          if (!(us < cfg_.max_integration_time_s * 1000000.0)) break;
        }
        const size_t idx = orderIndex(seq, n);
        const V3 p_C = pts[idx];
        const Color color = cols ? cols[idx] : Color{};
        bool is_clearing = false;
        if (!isPointValid(p_C, freespace, &is_clearing)) continue;
        ++n_valid;
        const V3 origin = T.t;
        const V3 p_G = transform(T, p_C);
        if (fast) {
          const GIdx key = gridIndexFromPoint(p_G, cfg_.start_voxel_subsampling_factor * voxel_size_inv_);
          bool fresh;
          if (cfg_.fast_exact_sets)
            fresh = start_exact_.emplace(key, 1).second;
          else
            fresh = start_set_.replaceHash(LongIndexHash()(key));
          if (!fresh) continue;
        }
        ++n_rays;
        RayCaster rc(origin, p_G, is_clearing, cfg_.voxel_carving_enabled, cfg_.max_ray_length_m, voxel_size_inv_, cfg_.default_truncation_distance,
                     /*cast_from_origin=*/!fast);
        int64_t collisions = 0;
        Block* block = nullptr;
        BIdx bidx{0, 0, 0};
        GIdx g;
        while (rc.next(&g)) {
          if (fast) {
            bool fresh;
            if (cfg_.fast_exact_sets)
              fresh = observed_exact_.emplace(g, 1).second;
            else
              fresh = observed_set_.replaceHash(LongIndexHash()(g));
            if (!fresh)
              ++collisions;
            else
              collisions = 0;
            if (collisions > cfg_.max_consecutive_ray_collisions) break;
          }
          TsdfVoxel* v = allocateStorageAndGetVoxelPtr(g, &block, &bidx);
          const float w = getVoxelWeight(p_C);
          updateTsdfVoxel(origin, p_G, g, color, w, v, threaded);
          if (!threaded)
            noteUpdate(g, &n_updates);
          else
            ++n_updates;
        }
      }
      std::lock_guard<std::mutex> lock(stats_mutex_);
      stats_.n_valid += n_valid;
      stats_.n_rays += n_rays;
      stats_.n_updates += n_updates;
    };
    if (nthreads == 1) {
      worker(false);
    } else {
      std::vector<std::thread> th;
      for (int i = 0; i < nthreads; ++i) th.emplace_back(worker, true);
      for (auto& t : th) t.join();
    }
  }

  // -------- merged (A.6) ----------
  struct Bundle {
    GIdx key;
    std::vector<size_t> pts;
  };
  void integrateBundle(const Transform& T, const V3* pts, const Color* cols, bool clearing, const Bundle& b,
                       const std::unordered_map<GIdx, uint32_t, LongIndexHash>& voxel_map_keys, bool threaded, uint64_t* n_updates) {
    if (b.pts.empty()) return;
    const V3 origin = T.t;
    Color merged_color;
    V3 m{0.0f, 0.0f, 0.0f};
    float W = 0.0f;
    for (size_t pi : b.pts) {
      const V3 p = pts[pi];
      const Color c = cols ? cols[pi] : Color{};
      const float w = getVoxelWeight(p);
      if (w < kEps) continue;
      const float den = W + w;
      m = {(m.x * W + p.x * w) / den, (m.y * W + p.y * w) / den, (m.z * W + p.z * w) / den};
      merged_color = blendTwoColors(merged_color, W, c, w);
      W += w;
      if (clearing) break;  // only take first point when clearing
    }
    const V3 m_G = transform(T, m);
    RayCaster rc(origin, m_G, clearing, cfg_.voxel_carving_enabled, cfg_.max_ray_length_m, voxel_size_inv_, cfg_.default_truncation_distance);
    GIdx g;
    while (rc.next(&g)) {
      if (cfg_.enable_anti_grazing) {
        if ((clearing || g != b.key) && voxel_map_keys.find(g) != voxel_map_keys.end()) continue;
      }
      Block* block = nullptr;
      BIdx bidx{0, 0, 0};
      TsdfVoxel* v = allocateStorageAndGetVoxelPtr(g, &block, &bidx);
      updateTsdfVoxel(origin, m_G, g, merged_color, W, v, threaded);
      if (!threaded)
        noteUpdate(g, n_updates);
      else
        ++*n_updates;
    }
  }
  void integrateMerged(const Transform& T, const V3* pts, const Color* cols, size_t n, bool freespace) {
    // bundleRays (single-threaded upstream as well): key = terminal voxel, members in visiting order
    std::unordered_map<GIdx, uint32_t, LongIndexHash> voxel_map, clear_map;  // key -> bundle ordinal
    std::vector<Bundle> voxel_bundles, clear_bundles;
    for (size_t seq = 0; seq < n; ++seq) {
      const size_t idx = orderIndex(seq, n);
      const V3 p_C = pts[idx];
      bool is_clearing = false;
      if (!isPointValid(p_C, freespace, &is_clearing)) continue;
      ++stats_.n_valid;
      const V3 p_G = transform(T, p_C);
      const GIdx key = gridIndexFromPoint(p_G, voxel_size_inv_);
      auto& map = is_clearing ? clear_map : voxel_map;
      auto& vec = is_clearing ? clear_bundles : voxel_bundles;
      auto it = map.find(key);
      if (it == map.end()) {
        map.emplace(key, static_cast<uint32_t>(vec.size()));
        vec.push_back(Bundle{key, {idx}});
      } else {
        vec[it->second].pts.push_back(idx);
      }
    }
    stats_.n_rays = voxel_bundles.size() + clear_bundles.size();
    for (const auto* vec : {&voxel_bundles, &clear_bundles})
      for (const Bundle& b : *vec) stats_.max_bundle_points = std::max<uint64_t>(stats_.max_bundle_points, b.pts.size());
    // integrateRays(non-clearing) then integrateRays(clearing)
    for (int pass = 0; pass < 2; ++pass) {
      const bool clearing = (pass == 1);
      auto& map = clearing ? clear_map : voxel_map;
      auto& vec = clearing ? clear_bundles : voxel_bundles;
      std::vector<uint32_t> order;
      order.reserve(vec.size());
      if (cfg_.merged_bundle_order == 1) {
        for (auto& kv : map) order.push_back(kv.second);  // libstdc++ unordered_map iteration order
      } else {
        for (uint32_t i = 0; i < vec.size(); ++i) order.push_back(i);  // canonical: order of first visit
      }
      const int nthreads = cfg_.integrator_threads;
      if (nthreads == 1) {
        uint64_t nu = 0;
        for (uint32_t bi : order) integrateBundle(T, pts, cols, clearing, vec[bi], voxel_map, false, &nu);
        stats_.n_updates += nu;
      } else {
        std::vector<std::thread> th;
        std::vector<uint64_t> nus(nthreads, 0);
        for (int t = 0; t < nthreads; ++t) {
          th.emplace_back([&, t]() {
            for (size_t i = 0; i < order.size(); ++i)
              if (((i + t + 1) % nthreads) == 0) integrateBundle(T, pts, cols, clearing, vec[order[i]], voxel_map, true, &nus[t]);
          });
        }
        for (auto& t : th) t.join();
        for (auto v : nus) stats_.n_updates += v;
      }
    }
  }
};

// ---------------------------------------------------------------------------------------------
// A.9 wire format (voxblox_msgs/Block: int32 x,y,z + uint32[] data, 3 words per TsdfVoxel)
// ---------------------------------------------------------------------------------------------
inline void voxelToWords(const TsdfVoxel& v, uint32_t* w) {
  std::memcpy(&w[0], &v.distance, 4);
  std::memcpy(&w[1], &v.weight, 4);
  w[2] = static_cast<uint32_t>(v.color.a) | (static_cast<uint32_t>(v.color.b) << 8) | (static_cast<uint32_t>(v.color.g) << 16) |
         (static_cast<uint32_t>(v.color.r) << 24);
}
inline void wordsToVoxel(const uint32_t* w, TsdfVoxel* v) {
  std::memcpy(&v->distance, &w[0], 4);
  std::memcpy(&v->weight, &w[1], 4);
  v->color.a = static_cast<uint8_t>(w[2] & 0xFF);
  v->color.b = static_cast<uint8_t>((w[2] >> 8) & 0xFF);
  v->color.g = static_cast<uint8_t>((w[2] >> 16) & 0xFF);
  v->color.r = static_cast<uint8_t>((w[2] >> 24) & 0xFF);
}
// voxblox mergeVoxelAIntoVoxelB (TSDF), A.8
inline void mergeVoxelAIntoVoxelB(const TsdfVoxel& a, TsdfVoxel* b) {
  const float cw = a.weight + b->weight;
  if (cw > 0.0f) {
    b->distance = (a.distance * a.weight + b->distance * b->weight) / cw;
    b->color = blendTwoColors(a.color, a.weight, b->color, b->weight);
    b->weight = cw;
  }
}

// ---------------------------------------------------------------------------------------------
// A.10 trilinear interpolation (voxblox Interpolator<TsdfVoxel>::getVoxelsAndQVector)
// ---------------------------------------------------------------------------------------------
static const float kInterpTable[8][8] = {{1, 0, 0, 0, 0, 0, 0, 0},   {-1, 0, 0, 0, 1, 0, 0, 0},   {-1, 0, 1, 0, 0, 0, 0, 0},
                                         {-1, 1, 0, 0, 0, 0, 0, 0},  {1, 0, -1, 0, -1, 0, 1, 0},  {1, -1, -1, 1, 0, 0, 0, 0},
                                         {1, -1, 0, 0, -1, 1, 0, 0}, {-1, 1, 1, -1, 1, -1, -1, 1}};

struct Interp {
  bool ok = false;
  float d[8];    // neighbour distances, column order (0,0,0),(0,0,1),(0,1,0),(0,1,1),(1,0,0),(1,0,1),(1,1,0),(1,1,1)
  float w[8];    // neighbour weights
  float off[3];  // (pos - centre0) / voxel_size
};
inline Interp getVoxelsAndQVector(const Layer& L, V3 pos) {
  Interp r;
  BIdx bi = blockIndexFromPoint(pos, L.block_size_inv);
  const Block* blk = L.getBlockPtr(bi);
  if (!blk) return r;
  // computeTruncatedVoxelIndexFromCoordinates
  const V3 rel = pos - blk->origin;
  GIdx gi = gridIndexFromPoint(rel, L.voxel_size_inv);
  int vi[3] = {static_cast<int>(gi.x), static_cast<int>(gi.y), static_cast<int>(gi.z)};
  for (int k = 0; k < 3; ++k) vi[k] = std::max(std::min(vi[k], L.vps - 1), 0);
  // centre of that voxel: origin + centerPoint(idx, voxel_size)
  const float c[3] = {blk->origin.x + centerCoord(vi[0], L.voxel_size), blk->origin.y + centerCoord(vi[1], L.voxel_size),
                      blk->origin.z + centerCoord(vi[2], L.voxel_size)};
  const float p[3] = {pos.x, pos.y, pos.z};
  int b[3] = {bi.x, bi.y, bi.z};
  for (int k = 0; k < 3; ++k) {
    if (p[k] - c[k] < 0.0f) {
      vi[k]--;
      if (vi[k] < 0) {
        b[k]--;
        vi[k] += L.vps;
      }
    }
  }
  const BIdx base{b[0], b[1], b[2]};
  const Block* base_blk = L.getBlockPtr(base);
  if (!base_blk) return r;
  static const int offs[8][3] = {{0, 0, 0}, {0, 0, 1}, {0, 1, 0}, {0, 1, 1}, {1, 0, 0}, {1, 0, 1}, {1, 1, 0}, {1, 1, 1}};
  for (int i = 0; i < 8; ++i) {
    int v[3] = {vi[0] + offs[i][0], vi[1] + offs[i][1], vi[2] + offs[i][2]};
    BIdx nb = base;
    int* nbp[3] = {&nb.x, &nb.y, &nb.z};
    for (int k = 0; k < 3; ++k)
      if (v[k] >= L.vps) {
        (*nbp[k])++;
        v[k] -= L.vps;
      }
    const Block* bp = (nb == base) ? base_blk : L.getBlockPtr(nb);
    if (!bp) return r;
    if (i == 0) {
      const float c0[3] = {bp->origin.x + centerCoord(v[0], L.voxel_size), bp->origin.y + centerCoord(v[1], L.voxel_size),
                           bp->origin.z + centerCoord(v[2], L.voxel_size)};
      for (int k = 0; k < 3; ++k) r.off[k] = (p[k] - c0[k]) * L.voxel_size_inv;
    }
    const TsdfVoxel& vox = bp->voxels[linearIndex(v[0], v[1], v[2], L.vps)];
    r.d[i] = vox.distance;
    r.w[i] = vox.weight;
    if (!(vox.weight > 0.0f)) return r;  // Interpolator<TsdfVoxel>::isVoxelValid
  }
  r.ok = true;
  return r;
}
// value = q . (M . d), gradient wrt position (per metre) = (1/voxel_size) Dq . (M . d); sequential float sums
inline void interpValueAndGrad(const Interp& it, float voxel_size_inv, float* value, float grad[3]) {
  float md[8];
  for (int r = 0; r < 8; ++r) {
    float s = 0.0f;
    for (int c = 0; c < 8; ++c) s += kInterpTable[r][c] * it.d[c];
    md[r] = s;
  }
  const float dx = it.off[0], dy = it.off[1], dz = it.off[2];
  const float q[8] = {1.0f, dx, dy, dz, dx * dy, dy * dz, dz * dx, dx * dy * dz};
  float v = 0.0f;
  for (int i = 0; i < 8; ++i) v += q[i] * md[i];
  *value = v;
  const float qx[8] = {0, 1, 0, 0, dy, 0, dz, dy * dz};
  const float qy[8] = {0, 0, 1, 0, dx, dz, 0, dz * dx};
  const float qz[8] = {0, 0, 0, 1, 0, dy, dx, dx * dy};
  float gx = 0.0f, gy = 0.0f, gz = 0.0f;
  for (int i = 0; i < 8; ++i) {
    gx += qx[i] * md[i];
    gy += qy[i] * md[i];
    gz += qz[i] * md[i];
  }
  grad[0] = gx * voxel_size_inv;
  grad[1] = gy * voxel_size_inv;
  grad[2] = gz * voxel_size_inv;
}

// ---------------------------------------------------------------------------------------------
// A.8 layer merge / rigid resample (voxblox merge_integration.h), as coxgraph calls it:
//   mergeLayerAintoLayerB(layer, T, layer)   coxgraph/src/client/map_server.cpp:67-69, src/server/submap_collection.cpp:31-33
// ---------------------------------------------------------------------------------------------
inline Transform inverse(const Transform& T) {
  Transform I;
  I.qw = T.qw;
  I.qx = -T.qx;
  I.qy = -T.qy;
  I.qz = -T.qz;
  I.t = {0.0f, 0.0f, 0.0f};
  const V3 r = rotate(I, T.t);
  I.t = {-r.x, -r.y, -r.z};
  return I;
}
// Interpolator<TsdfVoxel>::interpVoxel: distance, weight and the four colour channels each through the same
// q . (table . data) form; colour channels are narrowed to uint8 by truncation
inline float interpMember(const float q[8], const float data[8]) {
  float md[8];
  for (int r = 0; r < 8; ++r) {
    float s = 0.0f;
    for (int c = 0; c < 8; ++c) s += kInterpTable[r][c] * data[c];
    md[r] = s;
  }
  float v = 0.0f;
  for (int i = 0; i < 8; ++i) v += q[i] * md[i];
  return v;
}
// Interpolator::getVoxel(pos, &voxel, interpolate = true) then, if that fails, interpolate = false (nearest)
inline bool resampleVoxel(const Layer& L, V3 pos, TsdfVoxel* out) {
  // ---- trilinear ----
  {
    BIdx bi = blockIndexFromPoint(pos, L.block_size_inv);
    const Block* blk = L.getBlockPtr(bi);
    if (blk) {
      const V3 rel = pos - blk->origin;
      GIdx gi = gridIndexFromPoint(rel, L.voxel_size_inv);
      int vi[3] = {static_cast<int>(gi.x), static_cast<int>(gi.y), static_cast<int>(gi.z)};
      for (int k = 0; k < 3; ++k) vi[k] = std::max(std::min(vi[k], L.vps - 1), 0);
      const float c[3] = {blk->origin.x + centerCoord(vi[0], L.voxel_size), blk->origin.y + centerCoord(vi[1], L.voxel_size),
                          blk->origin.z + centerCoord(vi[2], L.voxel_size)};
      const float p[3] = {pos.x, pos.y, pos.z};
      int b[3] = {bi.x, bi.y, bi.z};
      for (int k = 0; k < 3; ++k)
        if (p[k] - c[k] < 0.0f) {
          vi[k]--;
          if (vi[k] < 0) {
            b[k]--;
            vi[k] += L.vps;
          }
        }
      const BIdx base{b[0], b[1], b[2]};
      const Block* base_blk = L.getBlockPtr(base);
      bool ok = base_blk != nullptr;
      float d[8], w[8], ch[4][8], off[3] = {0, 0, 0};
      static const int offs[8][3] = {{0, 0, 0}, {0, 0, 1}, {0, 1, 0}, {0, 1, 1}, {1, 0, 0}, {1, 0, 1}, {1, 1, 0}, {1, 1, 1}};
      for (int i = 0; i < 8 && ok; ++i) {
        int v[3] = {vi[0] + offs[i][0], vi[1] + offs[i][1], vi[2] + offs[i][2]};
        BIdx nb = base;
        int* nbp[3] = {&nb.x, &nb.y, &nb.z};
        for (int k = 0; k < 3; ++k)
          if (v[k] >= L.vps) {
            (*nbp[k])++;
            v[k] -= L.vps;
          }
        const Block* bp = (nb == base) ? base_blk : L.getBlockPtr(nb);
        if (!bp) {
          ok = false;
          break;
        }
        if (i == 0) {
          const float c0[3] = {bp->origin.x + centerCoord(v[0], L.voxel_size), bp->origin.y + centerCoord(v[1], L.voxel_size),
                               bp->origin.z + centerCoord(v[2], L.voxel_size)};
          for (int k = 0; k < 3; ++k) off[k] = (p[k] - c0[k]) * L.voxel_size_inv;
        }
        const TsdfVoxel& vox = bp->voxels[linearIndex(v[0], v[1], v[2], L.vps)];
        if (!(vox.weight > 0.0f)) {
          ok = false;
          break;
        }
        d[i] = vox.distance;
        w[i] = vox.weight;
        ch[0][i] = vox.color.r;
        ch[1][i] = vox.color.g;
        ch[2][i] = vox.color.b;
        ch[3][i] = vox.color.a;
      }
      if (ok) {
        const float dx = off[0], dy = off[1], dz = off[2];
        const float q[8] = {1.0f, dx, dy, dz, dx * dy, dy * dz, dz * dx, dx * dy * dz};
        out->distance = interpMember(q, d);
        out->weight = interpMember(q, w);
        out->color.r = static_cast<uint8_t>(static_cast<int>(interpMember(q, ch[0])));
        out->color.g = static_cast<uint8_t>(static_cast<int>(interpMember(q, ch[1])));
        out->color.b = static_cast<uint8_t>(static_cast<int>(interpMember(q, ch[2])));
        out->color.a = static_cast<uint8_t>(static_cast<int>(interpMember(q, ch[3])));
        return true;
      }
    }
  }
  // ---- nearest: Block::getVoxelByCoordinates ----
  const BIdx bi = blockIndexFromPoint(pos, L.block_size_inv);
  const Block* blk = L.getBlockPtr(bi);
  if (!blk) return false;
  const V3 rel = pos - blk->origin;
  const GIdx gi = gridIndexFromPoint(rel, L.voxel_size_inv);
  int vi[3] = {static_cast<int>(gi.x), static_cast<int>(gi.y), static_cast<int>(gi.z)};
  for (int k = 0; k < 3; ++k) vi[k] = std::max(std::min(vi[k], L.vps - 1), 0);
  *out = blk->voxels[linearIndex(vi[0], vi[1], vi[2], L.vps)];
  return true;
}
// candidate output blocks of transformLayer: input blocks approximated by spheres of diameter sqrt(3) * block_size
inline std::vector<BIdx> transformCandidateBlocks(const Layer& in, const Transform& T_out_in, float block_size_out) {
  std::unordered_map<BIdx, int, AnyIndexHash> set;
  std::vector<BIdx> order;
  const float kUnitCubeDiagonalLength = static_cast<float>(1.7320508075688772);
  const float inv = 1.0f / block_size_out;
  std::vector<BIdx> in_blocks;
  for (auto& kv : in.blocks) in_blocks.push_back(kv.first);
  std::sort(in_blocks.begin(), in_blocks.end(), [](const BIdx& a, const BIdx& b) { return std::tie(a.z, a.y, a.x) < std::tie(b.z, b.y, b.x); });
  for (const BIdx& bi : in_blocks) {
    const V3 c_in = {centerCoord(bi.x, in.block_size), centerCoord(bi.y, in.block_size), centerCoord(bi.z, in.block_size)};
    const V3 c_out = transform(T_out_in, c_in);
    const float offset = static_cast<float>(kUnitCubeDiagonalLength * in.block_size * 0.5);
    for (float x = c_out.x - offset; x < c_out.x + offset; x += block_size_out)
      for (float y = c_out.y - offset; y < c_out.y + offset; y += block_size_out)
        for (float z = c_out.z - offset; z < c_out.z + offset; z += block_size_out) {
          const BIdx idx = blockIndexFromPoint(V3{x, y, z}, inv);
          if (set.emplace(idx, 1).second) order.push_back(idx);
        }
  }
  return order;
}
inline void transformLayer(const Layer& in, const Transform& T_out_in, Layer* out) {
  const Transform T_in_out = inverse(T_out_in);
  for (const BIdx& bi : transformCandidateBlocks(in, T_out_in, out->block_size)) {
    Block* blk = out->allocateBlock(bi);
    bool has_data = false;
    const int nv = out->vps * out->vps * out->vps;
    for (int lin = 0; lin < nv; ++lin) {
      const int lx = lin % out->vps, ly = (lin / out->vps) % out->vps, lz = lin / (out->vps * out->vps);
      const V3 centre = {blk->origin.x + centerCoord(lx, out->voxel_size), blk->origin.y + centerCoord(ly, out->voxel_size),
                         blk->origin.z + centerCoord(lz, out->voxel_size)};
      if (resampleVoxel(in, transform(T_in_out, centre), &blk->voxels[lin])) has_data = true;
    }
    if (!has_data) out->blocks.erase(bi);
  }
}
inline void mergeLayerAintoLayerB(const Layer& A, Layer* B) {
  for (auto& kv : A.blocks) {
    Block* bb = B->allocateBlock(kv.first);
    for (size_t i = 0; i < bb->voxels.size(); ++i) mergeVoxelAIntoVoxelB(kv.second->voxels[i], &bb->voxels[i]);
  }
}
inline void mergeLayerAintoLayerB(const Layer& A, const Transform& T_B_A, Layer* B) {
  Layer At(B->voxel_size, B->vps);
  transformLayer(A, T_B_A, &At);
  mergeLayerAintoLayerB(At, B);
}

// ---------------------------------------------------------------------------------------------
// A.11 registration cost, 4-DoF (x, y, z, yaw) poses, pure-yaw rotations
// ---------------------------------------------------------------------------------------------
struct RegPoint {
  float x, y, z, distance, weight;
};
struct RegConfig {
  double no_correspondence_cost = 0.0;
};
// relative transform reading<-reference in double, narrowed to float for the per-point transform
struct RelPose {
  float R[9];  // row-major R_read^T R_ref
  float t[3];  // R_read^T (t_ref - t_read)
  double cf, sf, cr, sr;
  double tf[3], tr[3];
};
inline RelPose makeRelPose(const double ref[4], const double read[4]) {
  RelPose P;
  P.cf = std::cos(ref[3]);
  P.sf = std::sin(ref[3]);
  P.cr = std::cos(read[3]);
  P.sr = std::sin(read[3]);
  for (int k = 0; k < 3; ++k) {
    P.tf[k] = ref[k];
    P.tr[k] = read[k];
  }
  // R_rel = Rz(yaw_r)^T Rz(yaw_f) = Rz(yaw_f - yaw_r) built from the four cos/sin values
  const double c = P.cr * P.cf + P.sr * P.sf, s = P.cr * P.sf - P.sr * P.cf;
  const double Rd[9] = {c, -s, 0, s, c, 0, 0, 0, 1};
  for (int i = 0; i < 9; ++i) P.R[i] = static_cast<float>(Rd[i]);
  const double dx = ref[0] - read[0], dy = ref[1] - read[1], dz = ref[2] - read[2];
  P.t[0] = static_cast<float>(P.cr * dx + P.sr * dy);
  P.t[1] = static_cast<float>(-P.sr * dx + P.cr * dy);
  P.t[2] = static_cast<float>(dz);
  return P;
}
// One residual (unscaled by N/sum(w)) and its two 1x4 Jacobian rows (unscaled); returns reference weight.
inline float regResidual(const Layer& reading, const RelPose& P, const RegPoint& rp, const RegConfig& cfg, double* r, double Jf[4], double Jr[4],
                         bool* has_corr) {
  const float px = rp.x, py = rp.y, pz = rp.z;
  const V3 pr{(P.R[0] * px + P.R[1] * py) + P.R[2] * pz + P.t[0], (P.R[3] * px + P.R[4] * py) + P.R[5] * pz + P.t[1],
              (P.R[6] * px + P.R[7] * py) + P.R[8] * pz + P.t[2]};
  const Interp it = getVoxelsAndQVector(reading, pr);
  *has_corr = it.ok;
  if (!it.ok) {
    *r = static_cast<double>(rp.weight) * cfg.no_correspondence_cost;
    for (int k = 0; k < 4; ++k) Jf[k] = Jr[k] = 0.0;
    return rp.weight;
  }
  float val, g[3];
  interpValueAndGrad(it, reading.voxel_size_inv, &val, g);
  *r = static_cast<double>((rp.distance - val) * rp.weight);
  const double w = rp.weight, gx = g[0], gy = g[1], gz = g[2];
  // pr = Rr^T (Rf p + tf - tr)
  // d pr/d tf = Rr^T ; d pr/d tr = -Rr^T ; d pr/d yaw_f = Rr^T dRf p ; d pr/d yaw_r = dRr^T (Rf p + tf - tr)
  const double gRx = gx * P.cr - gy * P.sr, gRy = gx * P.sr + gy * P.cr, gRz = gz;  // g * Rr^T  (row vector)
  const double dRfp_x = -P.sf * px - P.cf * py, dRfp_y = P.cf * px - P.sf * py;      // dRz(yaw_f)/dyaw * p
  const double qx = (P.cf * px - P.sf * py) + P.tf[0] - P.tr[0], qy = (P.sf * px + P.cf * py) + P.tf[1] - P.tr[1];
  // dRr^T/dyaw = [[-s, c, 0], [-c, -s, 0], [0,0,0]]
  const double dpr_dyr_x = -P.sr * qx + P.cr * qy, dpr_dyr_y = -P.cr * qx - P.sr * qy;
  Jf[0] = -w * gRx;
  Jf[1] = -w * gRy;
  Jf[2] = -w * gRz;
  Jf[3] = -w * (gRx * dRfp_x + gRy * dRfp_y);
  Jr[0] = w * gRx;
  Jr[1] = w * gRy;
  Jr[2] = w * gRz;
  Jr[3] = -w * (gx * dpr_dyr_x + gy * dpr_dyr_y);
  return rp.weight;
}

}  // namespace coxo
