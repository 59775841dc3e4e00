// CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see cox_oracle.hpp header).
//
// voxblox ProjectiveTsdfIntegrator<InterpolationScheme::kAdaptive>, the integrator `method: "projective"` selects
// (coxgraph/config/tsdf_server_default.yaml:6-9, tsdf_server_carla.yaml:6-9: 1280 x 960 range image, 360 degrees of
// vertical field of view).  The source is in the un-vendored voxblox fork (SURVEY.md section 0.2, Appendix A.7b marks this
// integrator as recalled with low confidence); this file restates upstream's
// voxblox/include/voxblox/integrator/projective_tsdf_integrator{.h,_inl.h}:
//   parsePointcloud    every point -> bearing (altitude = asin(z / r), azimuth = atan2(y, x)) -> range-image pixel, the
//                      pixel keeps the smallest range; points within [min_ray, max_ray] mark the blocks on the ray from
//                      (r + truncation) * bearing back to the sensor (RayCaster at block scale)
//   updateTsdfBlocks   every voxel of every marked block: project the voxel centre into the range image, interpolate the
//                      range, sdf = range - |voxel|; skip sdf < -truncation (and sdf > truncation without carving);
//                      observation weight 1 (x drop-off behind the surface, / r^2 without const weight);
//                      w' = min(w + obs, max_weight); d' = (d w + min(truncation, sdf) obs) / w'
// Decisions where the recollection is not certain (each is a parameter or stated here, none is verifiable offline):
//   * bearingToImage for the integer pixel of a POINT truncates (C++ float -> int conversion), for a VOXEL keeps the
//     fraction; rows outside [0, rows - 1] and columns outside (0, cols - 1) are rejected;
//   * adaptive interpolation = bilinear over the 2 x 2 neighbourhood unless it spans more than `adaptive_gap_m`
//     (then the smallest of the four) or holds an empty pixel (then the nearest pixel);
//   * a voxel whose new weight would be < 1e-6 is left alone (upstream would divide by it);
//   * asin / atan2 are the shared plain-float versions of include/coxgraph_hip_math.h (libm's last bit is not portable).
#pragma once
#include "../include/coxgraph_hip_math.h"
#include "cox_oracle.hpp"

namespace coxo {

struct ProjectiveConfig {
  int horizontal_resolution = 0, vertical_resolution = 0;
  float vertical_fov_deg = 0.0f;
  int interpolation_scheme = 3;  // 0 nearest, 1 min neighbour, 2 bilinear, 3 adaptive (TsdfIntegratorFactory's choice)
  float adaptive_gap_m = 0.5f;
};

class ProjectiveIntegrator {
 public:
  ProjectiveIntegrator(Layer* layer, const TsdfConfig& cfg, const ProjectiveConfig& pc)
      : layer_(layer), cfg_(cfg), pc_(pc), range_(static_cast<size_t>(pc.horizontal_resolution) * pc.vertical_resolution, 0.0f) {
    vertical_fov_rad_ = static_cast<float>(static_cast<double>(pc.vertical_fov_deg) * M_PI / 180.0);
  }
  FrameStats last_stats;
  Layer* layer() const { return layer_; }

  // rows / cols of a bearing: false when it falls outside the image
  bool bearingToImage(float altitude, float azimuth, float* h, float* w) const {
    const double H = pc_.vertical_resolution, W = pc_.horizontal_resolution;
    const float hh = static_cast<float>((H - 1.0) * (0.5 - static_cast<double>(altitude) / static_cast<double>(vertical_fov_rad_)));
    if (hh < 0.0f || static_cast<float>(pc_.vertical_resolution - 1) < hh) return false;
    float ww = static_cast<float>(W * static_cast<double>(azimuth) / (2.0 * M_PI));
    if (ww < 0.0f) ww += static_cast<float>(pc_.horizontal_resolution);
    *h = hh;
    *w = ww;
    return 0.0f < ww && ww < static_cast<float>(pc_.horizontal_resolution - 1);
  }
  float& px(int h, int w) { return range_[static_cast<size_t>(h) * pc_.horizontal_resolution + w]; }
  float px(int h, int w) const { return range_[static_cast<size_t>(h) * pc_.horizontal_resolution + w]; }

  float interpolate(float h, float w) const {
    const int H = pc_.vertical_resolution, W = pc_.horizontal_resolution;
    const int hr = static_cast<int>(std::round(h)), wr = static_cast<int>(std::round(w));
    const int h0 = static_cast<int>(std::floor(h)), w0 = static_cast<int>(std::floor(w));
    if (pc_.interpolation_scheme == 0) return px(std::min(hr, H - 1), std::min(wr, W - 1));
    if (h0 + 1 >= H || w0 + 1 >= W) return px(h0, w0);  // on the edge: no 2 x 2 neighbourhood
    const float a = px(h0, w0), b = px(h0, w0 + 1), c = px(h0 + 1, w0), d = px(h0 + 1, w0 + 1);
    const float mn = std::min(std::min(a, b), std::min(c, d)), mx = std::max(std::max(a, b), std::max(c, d));
    if (pc_.interpolation_scheme == 1) return mn;
    if (mn < kEps) return px(std::min(hr, H - 1), std::min(wr, W - 1));  // an empty neighbour: nearest pixel
    if (pc_.interpolation_scheme == 3 && mx - mn > pc_.adaptive_gap_m) return mn;  // depth discontinuity: the front surface
    const float dh = h - static_cast<float>(h0), dw = w - static_cast<float>(w0);
    return (a * (1.0f - dh) + c * dh) * (1.0f - dw) + (b * (1.0f - dh) + d * dh) * dw;
  }

  void integratePointCloud(const Transform& T_G_C, const V3* points_C, size_t n, bool deintegrate) {
    FrameStats st;
    st.n_points = n;
    std::fill(range_.begin(), range_.end(), 0.0f);
    const size_t blocks_before = layer_->numBlocks();
    std::unordered_map<BIdx, int, AnyIndexHash> touched;
    const V3 t_scaled = T_G_C.t * layer_->block_size_inv;
    for (size_t i = 0; i < n; ++i) {
      const V3 p = points_C[i];
      const float distance = norm(p);
      if (!(distance <= 3.0e38f)) continue;  // non-finite point: invalid (as in the ray-casting integrators)
      if (std::abs(distance) < kEps) continue;
      const float altitude = cox_asinf(p.z / distance), azimuth = cox_atan2f(p.y, p.x);
      float hf, wf;
      if (!bearingToImage(altitude, azimuth, &hf, &wf)) continue;
      const int h = static_cast<int>(hf), w = static_cast<int>(wf);
      float& r = px(h, w);
      r = (r < kEps) ? distance : std::min(r, distance);
      ++st.n_valid;
      if (cfg_.min_ray_length_m <= distance && distance <= cfg_.max_ray_length_m) {
        ++st.n_rays;
        const float scale = (distance + cfg_.default_truncation_distance) / distance;
        const V3 far_G = transform(T_G_C, V3{p.x * scale, p.y * scale, p.z * scale});
        RayCaster rc(far_G * layer_->block_size_inv, t_scaled);
        GIdx g;
        while (rc.next(&g)) touched.emplace(BIdx{static_cast<int>(g.x), static_cast<int>(g.y), static_cast<int>(g.z)}, 1);
      }
    }
    for (auto& kv : touched) layer_->allocateBlock(kv.first);
    const Transform T_C_G = inverse(T_G_C);
    const int nv = layer_->vps * layer_->vps * layer_->vps;
    for (auto& kv : touched) {
      Block* b = layer_->getBlockPtr(kv.first);
      for (int lin = 0; lin < nv; ++lin) {
        const int lx = lin % layer_->vps, ly = (lin / layer_->vps) % layer_->vps, lz = lin / (layer_->vps * layer_->vps);
        const V3 centre{b->origin.x + centerCoord(lx, layer_->voxel_size), b->origin.y + centerCoord(ly, layer_->voxel_size),
                        b->origin.z + centerCoord(lz, layer_->voxel_size)};
        if (updateTsdfVoxel(transform(T_C_G, centre), &b->voxels[lin], deintegrate)) ++st.n_updates;
      }
    }
    st.n_touched_voxels = st.n_updates;
    st.n_new_blocks = layer_->numBlocks() - blocks_before;
    st.max_voxel_updates = st.n_updates ? 1 : 0;
    n_touched_blocks = touched.size();
    last_stats = st;
  }
  uint64_t n_touched_blocks = 0;

 private:
  bool updateTsdfVoxel(V3 t_C_voxel, TsdfVoxel* voxel, bool deintegrate) const {
    const float distance_to_voxel = norm(t_C_voxel);
    if (distance_to_voxel < cfg_.min_ray_length_m || distance_to_voxel > cfg_.max_ray_length_m) return false;
    float h, w;
    if (!bearingToImage(cox_asinf(t_C_voxel.z / distance_to_voxel), cox_atan2f(t_C_voxel.y, t_C_voxel.x), &h, &w)) return false;
    const float distance_to_surface = interpolate(h, w);
    const float sdf = distance_to_surface - distance_to_voxel;
    const float trunc = cfg_.default_truncation_distance;
    if (sdf < -trunc) return false;
    if (!cfg_.voxel_carving_enabled && sdf > trunc) return false;
    float obs = deintegrate ? -1.0f : 1.0f;
    const float eps = layer_->voxel_size;
    if (cfg_.use_weight_dropoff && sdf < -eps) {
      obs = obs * ((trunc + sdf) / (trunc - eps));
      obs = std::max(obs, 0.0f);
    }
    if (!cfg_.use_const_weight) obs = obs / (distance_to_voxel * distance_to_voxel);
    const float new_weight = std::min(voxel->weight + obs, cfg_.max_weight);
    if (deintegrate && new_weight < 1.0f) {  // back to an unobserved voxel
      voxel->distance = 0.0f;
      voxel->weight = 0.0f;
      return true;
    }
    if (new_weight < kEps) return false;
    voxel->distance = (voxel->distance * voxel->weight + std::min(trunc, sdf) * obs) / new_weight;
    voxel->weight = new_weight;
    return true;
  }
  Layer* layer_;
  TsdfConfig cfg_;
  ProjectiveConfig pc_;
  std::vector<float> range_;
  float vertical_fov_rad_ = 0.0f;
};

}  // namespace coxo
