// CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see cox_oracle.hpp header).
//
// Restatement of coxgraph's recover-mode front end, the caller of the integrator hot path:
//   voxblox::MeshConverter            coxgraph/include/coxgraph/map_comm/mesh_converter.h:22-289
//   TsdfRecover::processMesh loop     coxgraph/include/coxgraph/map_comm/tsdf_recover.h:59-99
// This part of the path IS in the reference tree (unlike voxblox itself), so it is restated from that source; the
// reference holds no test vectors for it, hence still "parity unpinned".
//
// Behaviours kept on purpose:
//   * vertex decode  (float(x) * (2.0f/65535) + float(index)) * block_edge_length            mesh_converter.h:96-112
//   * the per-frame clouds live in a std::map<uint8_t, ...>: frame ids alias modulo 256       mesh_converter.h:287
//   * history runs are inclusive [first, last]; overlapping runs append a triangle twice      mesh_converter.h:137-142
//   * edge p0-p2 blends colors[0] with colors[1] (not colors[2])                              mesh_converter.h:236-238
//   * frame id of pose i = round((stamp_i - stamp_0).toSec() / 0.05), 0 if stamps are equal   mesh_converter.h:191-196
//   * T_odom_submap_ is identity, so T_Submap_C == T_G_C; the cloud is moved by its inverse   mesh_converter.h:200-203
#pragma once
#include "cox_oracle.hpp"

#include <map>

namespace coxo {

struct MeshBlockMsg {
  int64_t index[3];
  std::vector<uint16_t> x, y, z;
  std::vector<uint8_t> r, g, b;
  std::vector<std::vector<uint32_t>> history;  // one ObsHistory per triangle
};
struct StampedPose {
  uint32_t sec, nsec;
  Transform T;  // already cast to float
};
struct MeshMsg {
  float block_edge_length = 0.0f;
  std::vector<MeshBlockMsg> mesh_blocks;
  std::vector<StampedPose> trajectory;
};
struct Cloud {
  std::vector<V3> pts;
  std::vector<Color> colors;
};
struct RecoveredPoint {
  float x, y, z;
  uint8_t r, g, b;
};

// ros::Time - ros::Time -> ros::Duration (normalised sec/nsec), Duration::toSec() = sec + 1e-9 * nsec
inline double stampDiffSec(uint32_t sec_a, uint32_t nsec_a, uint32_t sec_b, uint32_t nsec_b) {
  const int64_t d = (static_cast<int64_t>(sec_a) * 1000000000ll + nsec_a) - (static_cast<int64_t>(sec_b) * 1000000000ll + nsec_b);
  int64_t s = d / 1000000000ll, ns = d % 1000000000ll;
  if (ns < 0) {
    ns += 1000000000ll;
    s -= 1;
  }
  return static_cast<double>(static_cast<int32_t>(s)) + 1e-9 * static_cast<double>(static_cast<int32_t>(ns));
}
// what "double -> std::map<uint8_t,...>::key_type" does on x86-64: truncate to int32, keep the low byte
inline uint8_t frameKey(double id) {
  if (!(id > -2147483649.0 && id < 2147483648.0)) return 0;
  return static_cast<uint8_t>(static_cast<int32_t>(id));
}

class MeshConverter {
 public:
  explicit MeshConverter(float interpolate_voxel_size) : voxel_size_(interpolate_voxel_size) {}

  // mesh_converter.h:55-72
  void setMesh(const MeshMsg& mesh) {
    if (mesh.trajectory.empty()) return;
    mesh_ = mesh;
    for (const StampedPose& p : mesh.trajectory) T_G_C_.push_back(p);
  }

  // mesh_converter.h:212-277
  void interpolateTriangle(const V3 tri[3], const Color col[3], std::vector<V3>* pts, std::vector<Color>* colors) const {
    const V3 p0 = tri[0], p1 = tri[1], p2 = tri[2];
    const V3 t01 = p1 - p0, t02 = p2 - p0, t12 = p2 - p1;
    std::vector<V3> e02, e12;
    std::vector<Color> c02, c12;
    auto edge = [&](V3 from, V3 t, Color ca, Color cb, std::vector<V3>* op, std::vector<Color>* oc) {
      const float len = norm(t);
      for (float dist = voxel_size_; dist < len; dist += voxel_size_) {
        const V3 dir = t / len;
        op->push_back({from.x + dir.x * dist, from.y + dir.y * dist, from.z + dir.z * dist});
        oc->push_back(blendTwoColors(ca, 1 - dist / len, cb, dist / len));
      }
    };
    edge(p0, t01, col[0], col[1], pts, colors);
    edge(p0, t02, col[0], col[1], &e02, &c02);  // reference blends colors[0] and colors[1] on this edge too
    edge(p1, t12, col[1], col[2], &e12, &c12);
    const V3 sum = (p0 + p1) + p2;
    pts->push_back(sum / 3.0f);
    colors->push_back(blendTwoColors(col[2], static_cast<float>(1 / 3.0), blendTwoColors(col[0], 0.5f, col[1], 0.5f), static_cast<float>(2 / 3.0)));
    pts->insert(pts->end(), e02.begin(), e02.end());
    pts->insert(pts->end(), e12.begin(), e12.end());
    colors->insert(colors->end(), c02.begin(), c02.end());
    colors->insert(colors->end(), c12.begin(), c12.end());
  }

  // mesh_converter.h:74-168
  bool convertToPointCloud(std::vector<RecoveredPoint>* recovered) {
    recovered->clear();
    if (mesh_.mesh_blocks.empty()) return false;
    constexpr float point_conv_factor = 2.0f / 65535;
    std::vector<V3> triangle;
    std::vector<Color> colors;
    for (const MeshBlockMsg& mb : mesh_.mesh_blocks) {
      if (mb.history.empty()) continue;
      for (size_t i = 0; i < mb.x.size(); ++i) {
        const float mx = (static_cast<float>(mb.x[i]) * point_conv_factor + static_cast<float>(mb.index[0])) * mesh_.block_edge_length;
        const float my = (static_cast<float>(mb.y[i]) * point_conv_factor + static_cast<float>(mb.index[1])) * mesh_.block_edge_length;
        const float mz = (static_cast<float>(mb.z[i]) * point_conv_factor + static_cast<float>(mb.index[2])) * mesh_.block_edge_length;
        const std::vector<uint32_t>& history = mb.history[i / 3];
        triangle.push_back({mx, my, mz});
        Color c;
        c.r = mb.r[i];
        c.g = mb.g[i];
        c.b = mb.b[i];
        c.a = 255;  // voxblox Color(r, g, b) sets a = 255
        colors.push_back(c);
        recovered->push_back({mx, my, mz, mb.r[i], mb.g[i], mb.b[i]});
        if (triangle.size() == 3) {
          std::vector<V3> interp_pts;
          std::vector<Color> interp_colors;
          interpolateTriangle(triangle.data(), colors.data(), &interp_pts, &interp_colors);
          std::vector<size_t> frames;
          for (size_t k = 0; k + 1 < history.size(); k += 2)
            for (size_t j = history[k]; j <= history[k + 1]; ++j) frames.push_back(j);
          for (size_t f : frames) {
            Cloud& c = pointcloud_[static_cast<uint8_t>(f)];
            c.pts.insert(c.pts.end(), triangle.begin(), triangle.end());
            c.pts.insert(c.pts.end(), interp_pts.begin(), interp_pts.end());
            c.colors.insert(c.colors.end(), colors.begin(), colors.end());
            c.colors.insert(c.colors.end(), interp_colors.begin(), interp_colors.end());
          }
          triangle.clear();
          colors.clear();
        }
      }
    }
    return true;
  }

  // mesh_converter.h:183-210
  bool getNextPointcloud(int* i, Transform* T_G_C, std::vector<V3>* points_C, std::vector<Color>* colors) {
    if (*i < 0 || static_cast<size_t>(*i) >= T_G_C_.size()) return false;
    const StampedPose& p = T_G_C_[*i];
    const StampedPose& p0 = T_G_C_[0];
    *T_G_C = p.T;
    const double id = (p.sec == p0.sec && p.nsec == p0.nsec) ? 0.0 : std::round(stampDiffSec(p.sec, p.nsec, p0.sec, p0.nsec) / 0.05);
    const Cloud& c = pointcloud_[frameKey(id)];  // creates an empty cloud when there is none, like the reference
    const Transform T_C_Submap = inverse(*T_G_C);
    points_C->resize(c.pts.size());
    for (size_t k = 0; k < c.pts.size(); ++k) (*points_C)[k] = transform(T_C_Submap, c.pts[k]);
    *colors = c.colors;
    (*i)++;
    return true;
  }

  // mesh_converter.h:170-181
  void clear() {
    pointcloud_.clear();
    T_G_C_.clear();
    mesh_.mesh_blocks.clear();
  }

  size_t numPoses() const { return T_G_C_.size(); }

 private:
  float voxel_size_;
  MeshMsg mesh_;
  std::vector<StampedPose> T_G_C_;
  std::map<uint8_t, Cloud> pointcloud_;
};

// TsdfRecover::processMesh (tsdf_recover.h:59-99) up to the serialisation of the layer
inline void processMesh(MeshConverter* conv, Integrator* integ, Layer* layer, const MeshMsg& mesh, std::vector<RecoveredPoint>* recovered,
                        size_t* n_integrated) {
  layer->removeAllBlocks();
  conv->setMesh(mesh);
  conv->convertToPointCloud(recovered);
  int i = 0;
  Transform T_G_C;
  std::vector<V3> points_C;
  std::vector<Color> colors;
  *n_integrated = 0;
  while (conv->getNextPointcloud(&i, &T_G_C, &points_C, &colors)) {
    if (points_C.empty()) continue;
    integ->integratePointCloud(T_G_C, points_C.data(), colors.data(), points_C.size(), false);
    ++*n_integrated;
  }
  conv->clear();
}

}  // namespace coxo
