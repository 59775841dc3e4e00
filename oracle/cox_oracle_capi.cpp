// CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see cox_oracle.hpp header).
//
// C ABI of the oracle: the same entry points as include/coxgraph_hip.h with the prefix
// `coxo_` instead of `cox_`, so tests can drive the oracle and the HIP engine through one
// Python wrapper.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load this.
#include "../include/coxgraph_hip.h"
#include "cox_oracle.hpp"
#include "cox_oracle_mesh.hpp"
#include "cox_oracle_submap.hpp"
#include "cox_oracle_projective.hpp"

#include <map>
#include <tuple>

using namespace coxo;

struct coxo_layer {
  Layer layer;
  coxo_layer(float vs, int vps) : layer(vs, vps) {}
};
struct coxo_integrator {
  std::unique_ptr<Integrator> integ;
  std::unique_ptr<ProjectiveIntegrator> proj;  // method 3
};
struct coxo_regpoints {
  std::vector<RegPoint> pts;
};
struct coxo_reg {
  const coxo_regpoints* ref;
  const coxo_layer* reading;
  RegConfig cfg;
  std::vector<uint32_t> stored;  // cox_reg_set_samples
  bool has_stored = false, pending = false;
  double H[64], b[8], cost = 0.0;
  uint64_t n_corr = 0;
};

static TsdfConfig toCfg(const cox_tsdf_config* c) {
  TsdfConfig o;
  o.default_truncation_distance = c->default_truncation_distance;
  o.max_weight = c->max_weight;
  o.voxel_carving_enabled = c->voxel_carving_enabled != 0;
  o.min_ray_length_m = c->min_ray_length_m;
  o.max_ray_length_m = c->max_ray_length_m;
  o.use_const_weight = c->use_const_weight != 0;
  o.allow_clear = c->allow_clear != 0;
  o.use_weight_dropoff = c->use_weight_dropoff != 0;
  o.use_sparsity_compensation_factor = c->use_sparsity_compensation_factor != 0;
  o.sparsity_compensation_factor = c->sparsity_compensation_factor;
  o.integrator_threads = c->integrator_threads;
  o.integration_order_mode = c->integration_order_mode;
  o.enable_anti_grazing = c->enable_anti_grazing != 0;
  o.start_voxel_subsampling_factor = c->start_voxel_subsampling_factor;
  o.max_consecutive_ray_collisions = c->max_consecutive_ray_collisions;
  o.clear_checks_every_n_frames = c->clear_checks_every_n_frames;
  o.max_integration_time_s = c->max_integration_time_s;
  o.merged_bundle_order = c->merged_bundle_order;
  o.fast_exact_sets = c->fast_exact_sets;
  return o;
}

extern "C" {

void coxo_tsdf_config_default(cox_tsdf_config* c) {
  TsdfConfig d;
  c->default_truncation_distance = d.default_truncation_distance;
  c->max_weight = d.max_weight;
  c->voxel_carving_enabled = d.voxel_carving_enabled;
  c->min_ray_length_m = d.min_ray_length_m;
  c->max_ray_length_m = d.max_ray_length_m;
  c->use_const_weight = d.use_const_weight;
  c->allow_clear = d.allow_clear;
  c->use_weight_dropoff = d.use_weight_dropoff;
  c->use_sparsity_compensation_factor = d.use_sparsity_compensation_factor;
  c->sparsity_compensation_factor = d.sparsity_compensation_factor;
  c->integrator_threads = 1;
  c->integration_order_mode = 0;
  c->enable_anti_grazing = 0;
  c->start_voxel_subsampling_factor = d.start_voxel_subsampling_factor;
  c->max_consecutive_ray_collisions = d.max_consecutive_ray_collisions;
  c->clear_checks_every_n_frames = d.clear_checks_every_n_frames;
  c->max_integration_time_s = d.max_integration_time_s;
  c->merged_bundle_order = 0;
  c->fast_exact_sets = 0;
  const ProjectiveConfig pc;
  c->sensor_horizontal_resolution = pc.horizontal_resolution;
  c->sensor_vertical_resolution = pc.vertical_resolution;
  c->sensor_vertical_field_of_view_degrees = pc.vertical_fov_deg;
  c->projective_interpolation_scheme = pc.interpolation_scheme;
  c->projective_adaptive_gap_m = pc.adaptive_gap_m;
}

int coxo_layer_create(float voxel_size, int vps, int /*device*/, uint64_t /*capacity*/, coxo_layer** out) {
  if (!out || !(voxel_size > 0.0f) || vps <= 0 || (vps & (vps - 1))) return COX_ERR_INVALID_ARG;
  *out = new coxo_layer(voxel_size, vps);
  return COX_OK;
}
void coxo_layer_destroy(coxo_layer* l) { delete l; }
int coxo_layer_clear(coxo_layer* l) {
  l->layer.removeAllBlocks();
  return COX_OK;
}
int coxo_layer_stats(coxo_layer* l, uint64_t* n_blocks, uint64_t* bytes) {
  if (n_blocks) *n_blocks = l->layer.numBlocks();
  if (bytes) *bytes = l->layer.memorySize();
  return COX_OK;
}
int coxo_layer_download(coxo_layer* l, int32_t* idx, uint32_t* vox, uint64_t cap, uint64_t* n_blocks) {
  const uint64_t n = l->layer.numBlocks();
  if (n_blocks) *n_blocks = n;
  if (cap == 0 && !idx && !vox) return COX_OK;
  if (cap < n) return COX_ERR_BUFFER_TOO_SMALL;
  std::map<std::tuple<int, int, int>, const Block*> sorted;  // (z,y,x) order
  for (auto& kv : l->layer.blocks) sorted[std::make_tuple(kv.first.z, kv.first.y, kv.first.x)] = kv.second.get();
  const size_t nv = static_cast<size_t>(l->layer.vps) * l->layer.vps * l->layer.vps;
  uint64_t i = 0;
  for (auto& kv : sorted) {
    const Block* b = kv.second;
    idx[3 * i + 0] = b->index.x;
    idx[3 * i + 1] = b->index.y;
    idx[3 * i + 2] = b->index.z;
    for (size_t v = 0; v < nv; ++v) voxelToWords(b->voxels[v], &vox[(i * nv + v) * 3]);
    ++i;
  }
  return COX_OK;
}
int coxo_layer_upload(coxo_layer* l, const int32_t* idx, const uint32_t* vox, uint64_t n, int action) {
  if (action < 0 || action > 2) return COX_ERR_INVALID_ARG;
  if (action == 2) l->layer.removeAllBlocks();
  const size_t nv = static_cast<size_t>(l->layer.vps) * l->layer.vps * l->layer.vps;
  for (uint64_t i = 0; i < n; ++i) {
    Block* b = l->layer.allocateBlock(BIdx{idx[3 * i], idx[3 * i + 1], idx[3 * i + 2]});
    for (size_t v = 0; v < nv; ++v) {
      TsdfVoxel in;
      wordsToVoxel(&vox[(i * nv + v) * 3], &in);
      if (action == 1)
        mergeVoxelAIntoVoxelB(in, &b->voxels[v]);
      else
        b->voxels[v] = in;
    }
  }
  return COX_OK;
}

// mergeLayerAintoLayerB(A, [T_B_A,] B): T == nullptr merges on the same grid
int coxo_layer_merge(const coxo_layer* a, const float T[7], coxo_layer* b) {
  if (!a || !b) return COX_ERR_INVALID_ARG;
  if (!T) {
    if (a->layer.voxel_size != b->layer.voxel_size || a->layer.vps != b->layer.vps) return COX_ERR_INVALID_ARG;
    mergeLayerAintoLayerB(a->layer, &b->layer);
  } else {
    Transform Tr{T[0], T[1], T[2], T[3], {T[4], T[5], T[6]}};
    mergeLayerAintoLayerB(a->layer, Tr, &b->layer);
  }
  return COX_OK;
}

int coxo_integrator_create(coxo_layer* l, const cox_tsdf_config* cfg, int method, coxo_integrator** out) {
  if (!l || !cfg || !out || method < 0 || method > 3) return COX_ERR_INVALID_ARG;
  auto* h = new coxo_integrator();
  if (method == 3) {
    if (cfg->sensor_horizontal_resolution <= 1 || cfg->sensor_vertical_resolution <= 1 || !(cfg->sensor_vertical_field_of_view_degrees > 0.0f)) {
      delete h;
      return COX_ERR_INVALID_ARG;
    }
    ProjectiveConfig pc;
    pc.horizontal_resolution = cfg->sensor_horizontal_resolution;
    pc.vertical_resolution = cfg->sensor_vertical_resolution;
    pc.vertical_fov_deg = cfg->sensor_vertical_field_of_view_degrees;
    pc.interpolation_scheme = cfg->projective_interpolation_scheme;
    pc.adaptive_gap_m = cfg->projective_adaptive_gap_m;
    h->proj.reset(new ProjectiveIntegrator(&l->layer, toCfg(cfg), pc));
    *out = h;
    return COX_OK;
  }
  h->integ.reset(new Integrator(&l->layer, toCfg(cfg), method));
  *out = h;
  return COX_OK;
}
void coxo_integrator_destroy(coxo_integrator* h) { delete h; }
int coxo_integrator_set_count_touched(coxo_integrator* h, int on) {
  if (h->integ) h->integ->count_touched = on != 0;
  return COX_OK;
}
int coxo_integrate_points(coxo_integrator* h, const float T[7], const float* xyz, const uint8_t* rgba, uint64_t n, int freespace) {
  Transform Tr{T[0], T[1], T[2], T[3], {T[4], T[5], T[6]}};
  static_assert(sizeof(V3) == 12 && sizeof(Color) == 4, "layout");
  if (h->proj) {
    h->proj->integratePointCloud(Tr, reinterpret_cast<const V3*>(xyz), n, false);
    return COX_OK;
  }
  h->integ->integratePointCloud(Tr, reinterpret_cast<const V3*>(xyz), reinterpret_cast<const Color*>(rgba), n, freespace != 0);
  return COX_OK;
}
int coxo_integrate_points_ex(coxo_integrator* h, const float T[7], const float* xyz, const uint8_t* rgba, uint64_t n, int freespace, int deintegrate) {
  if (!deintegrate) return coxo_integrate_points(h, T, xyz, rgba, n, freespace);
  if (!h->proj) return COX_ERR_UNSUPPORTED;
  Transform Tr{T[0], T[1], T[2], T[3], {T[4], T[5], T[6]}};
  h->proj->integratePointCloud(Tr, reinterpret_cast<const V3*>(xyz), n, true);
  return COX_OK;
}
int coxo_integrator_sync(coxo_integrator*) { return COX_OK; }
int coxo_integrator_last_stats(coxo_integrator* h, cox_frame_stats* s) {
  const FrameStats& f = h->proj ? h->proj->last_stats : h->integ->last_stats;
  s->n_points = f.n_points;
  s->n_valid = f.n_valid;
  s->n_rays = f.n_rays;
  s->n_updates = f.n_updates;
  s->n_touched_voxels = f.n_touched_voxels;
  s->n_touched_blocks = h->proj ? h->proj->n_touched_blocks : 0;
  s->n_new_blocks = f.n_new_blocks;
  s->max_bundle_points = f.max_bundle_points;
  s->max_voxel_updates = f.max_voxel_updates;
  return COX_OK;
}

// ---- small probes used by the known-answer tests ----------------------------------------
int coxo_grid_index(const float p[3], float inv, int64_t out[3]) {
  const GIdx g = gridIndexFromPoint(V3{p[0], p[1], p[2]}, inv);
  out[0] = g.x;
  out[1] = g.y;
  out[2] = g.z;
  return COX_OK;
}
int coxo_block_local(const int64_t g[3], int vps, int32_t block[3], int32_t local[3], int32_t* linear) {
  const BIdx b = blockFromGlobal(GIdx{g[0], g[1], g[2]}, 1.0f / static_cast<float>(vps));
  block[0] = b.x;
  block[1] = b.y;
  block[2] = b.z;
  local[0] = localFromGlobal(g[0], vps);
  local[1] = localFromGlobal(g[1], vps);
  local[2] = localFromGlobal(g[2], vps);
  *linear = linearIndex(local[0], local[1], local[2], vps);
  return COX_OK;
}
// cast one ray; writes up to cap indices (3 int64 each); returns the count via *n
int coxo_raycast(const float origin[3], const float point_G[3], int is_clearing, int carving, float max_len, float inv, float trunc,
                 int cast_from_origin, int64_t* out, uint64_t cap, uint64_t* n) {
  RayCaster rc(V3{origin[0], origin[1], origin[2]}, V3{point_G[0], point_G[1], point_G[2]}, is_clearing != 0, carving != 0, max_len, inv, trunc,
               cast_from_origin != 0);
  uint64_t i = 0;
  GIdx g;
  while (rc.next(&g)) {
    if (i < cap) {
      out[3 * i] = g.x;
      out[3 * i + 1] = g.y;
      out[3 * i + 2] = g.z;
    }
    ++i;
  }
  *n = i;
  return COX_OK;
}
int coxo_transform_point(const float T[7], const float p[3], float out[3]) {
  Transform Tr{T[0], T[1], T[2], T[3], {T[4], T[5], T[6]}};
  const V3 r = transform(Tr, V3{p[0], p[1], p[2]});
  out[0] = r.x;
  out[1] = r.y;
  out[2] = r.z;
  return COX_OK;
}
uint64_t coxo_mixed_index(uint64_t seq, uint64_t n) { return mixedIndex(seq, n); }
// trilinear probe: returns 1 if interpolation possible
int coxo_interp(coxo_layer* l, const float pos[3], float* value, float grad[3]) {
  const Interp it = getVoxelsAndQVector(l->layer, V3{pos[0], pos[1], pos[2]});
  if (!it.ok) return 0;
  interpValueAndGrad(it, l->layer.voxel_size_inv, value, grad);
  return 1;
}

// ---- registration -------------------------------------------------------------------------
int coxo_regpoints_create(int /*device*/, const float* p, uint64_t n, coxo_regpoints** out) {
  auto* h = new coxo_regpoints();
  h->pts.resize(n);
  std::memcpy(h->pts.data(), p, n * sizeof(RegPoint));
  *out = h;
  return COX_OK;
}
void coxo_regpoints_destroy(coxo_regpoints* p) { delete p; }
int coxo_regpoints_size(const coxo_regpoints* p, uint64_t* n) {
  *n = p->pts.size();
  return COX_OK;
}
// VoxgraphSubmap::finishSubmap -> findRelevantVoxelIndices (called at coxgraph utils/msg_converter.h:113):
// voxels with weight > min and |distance| < max, position = Block::computeCoordinatesFromLinearIndex,
// blocks in (z,y,x) order, voxels in linear order.
static void relevantPoints(const Layer& L, float min_w, float max_d, std::vector<RegPoint>* out) {
  std::map<std::tuple<int, int, int>, const Block*> sorted;
  for (auto& kv : L.blocks) sorted[std::make_tuple(kv.first.z, kv.first.y, kv.first.x)] = kv.second.get();
  for (auto& kv : sorted) {
    const Block* b = kv.second;
    for (int lin = 0; lin < L.vps * L.vps * L.vps; ++lin) {
      const TsdfVoxel& v = b->voxels[lin];
      if (v.weight > min_w && std::abs(v.distance) < max_d) {
        const int lx = lin % L.vps, ly = (lin / L.vps) % L.vps, lz = lin / (L.vps * L.vps);
        out->push_back(RegPoint{b->origin.x + centerCoord(lx, L.voxel_size), b->origin.y + centerCoord(ly, L.voxel_size),
                                b->origin.z + centerCoord(lz, L.voxel_size), v.distance, v.weight});
      }
    }
  }
}
int coxo_layer_registration_points(coxo_layer* l, float min_w, float max_d, float* out, uint64_t cap, uint64_t* n) {
  std::vector<RegPoint> pts;
  relevantPoints(l->layer, min_w, max_d, &pts);
  *n = pts.size();
  if (out && !pts.empty()) {
    if (cap < pts.size()) return COX_ERR_BUFFER_TOO_SMALL;
    std::memcpy(out, pts.data(), pts.size() * sizeof(RegPoint));
  }
  return COX_OK;
}
int coxo_regpoints_from_layer(coxo_layer* l, float min_w, float max_d, coxo_regpoints** out) {
  auto* h = new coxo_regpoints();
  relevantPoints(l->layer, min_w, max_d, &h->pts);
  *out = h;
  return COX_OK;
}
int coxo_reg_create(const coxo_regpoints* ref, const coxo_layer* reading, const cox_reg_config* cfg, coxo_reg** out) {
  auto* h = new coxo_reg();
  h->ref = ref;
  h->reading = reading;
  h->cfg.no_correspondence_cost = cfg ? cfg->no_correspondence_cost : 0.0;
  *out = h;
  return COX_OK;
}
void coxo_reg_destroy(coxo_reg* r) { delete r; }

static int regRun(coxo_reg* reg, const double pose_ref[4], const double pose_read[4], const uint32_t* sample_idx, uint64_t n_res, double* residuals,
                  double* jf, double* jr, double* H, double* b, double* cost, uint64_t* n_corr) {
  const size_t npts = reg->ref->pts.size();
  if (!sample_idx && reg->has_stored) {  // cox_reg_set_samples
    if (n_res != reg->stored.size()) return COX_ERR_INVALID_ARG;
    sample_idx = reg->stored.data();
  }
  if (!sample_idx && n_res != npts) return COX_ERR_INVALID_ARG;
  const RelPose P = makeRelPose(pose_ref, pose_read);
  double sum_w = 0.0;
  uint64_t nc = 0;
  double Hs[64] = {0}, bs[8] = {0}, c = 0.0;
  for (uint64_t i = 0; i < n_res; ++i) {
    const size_t pi = sample_idx ? sample_idx[i] : i;
    if (pi >= npts) return COX_ERR_INVALID_ARG;
    double r, Jf[4], Jr[4];
    bool has;
    const float w = regResidual(reg->reading->layer, P, reg->ref->pts[pi], reg->cfg, &r, Jf, Jr, &has);
    sum_w += static_cast<double>(w);
    nc += has ? 1 : 0;
    if (residuals) residuals[i] = r;
    if (jf)
      for (int k = 0; k < 4; ++k) jf[4 * i + k] = Jf[k];
    if (jr)
      for (int k = 0; k < 4; ++k) jr[4 * i + k] = Jr[k];
    if (H) {
      double J[8] = {Jf[0], Jf[1], Jf[2], Jf[3], Jr[0], Jr[1], Jr[2], Jr[3]};
      for (int a = 0; a < 8; ++a) {
        for (int bb = 0; bb < 8; ++bb) Hs[8 * a + bb] += J[a] * J[bb];
        bs[a] += J[a] * r;
      }
      c += r * r;
    }
  }
  // "finally scale all by N / sum(w)"
  const double scale = (sum_w > 0.0) ? static_cast<double>(n_res) / sum_w : 0.0;
  if (residuals)
    for (uint64_t i = 0; i < n_res; ++i) residuals[i] *= scale;
  if (jf)
    for (uint64_t i = 0; i < 4 * n_res; ++i) jf[i] *= scale;
  if (jr)
    for (uint64_t i = 0; i < 4 * n_res; ++i) jr[i] *= scale;
  if (H) {
    for (int i = 0; i < 64; ++i) H[i] = Hs[i] * scale * scale;
    for (int i = 0; i < 8; ++i) b[i] = bs[i] * scale * scale;
    *cost = 0.5 * c * scale * scale;
  }
  if (n_corr) *n_corr = nc;
  return COX_OK;
}
int coxo_reg_evaluate(coxo_reg* reg, const double pose_ref[4], const double pose_read[4], const uint32_t* sample_idx, uint64_t n_res,
                      double* residuals, double* jac_ref, double* jac_read) {
  return regRun(reg, pose_ref, pose_read, sample_idx, n_res, residuals, jac_ref, jac_read, nullptr, nullptr, nullptr, nullptr);
}
int coxo_reg_normal_eq(coxo_reg* reg, const double pose_ref[4], const double pose_read[4], const uint32_t* sample_idx, uint64_t n_res, double H[64],
                       double b[8], double* cost, uint64_t* n_corr) {
  return regRun(reg, pose_ref, pose_read, sample_idx, n_res, nullptr, nullptr, nullptr, H, b, cost, n_corr);
}
// begin / finish: the oracle has nothing to overlap, begin computes and finish hands the result over
int coxo_reg_normal_eq_begin(coxo_reg* reg, const double pose_ref[4], const double pose_read[4], const uint32_t* sample_idx, uint64_t n_res) {
  if (!reg || reg->pending) return COX_ERR_INVALID_ARG;
  const int rc = regRun(reg, pose_ref, pose_read, sample_idx, n_res, nullptr, nullptr, nullptr, reg->H, reg->b, &reg->cost, &reg->n_corr);
  reg->pending = rc == COX_OK;
  return rc;
}
int coxo_reg_normal_eq_finish(coxo_reg* reg, double H[64], double b[8], double* cost, uint64_t* n_corr) {
  if (!reg || !reg->pending) return COX_ERR_INVALID_ARG;
  reg->pending = false;
  for (int i = 0; i < 64; ++i) H[i] = reg->H[i];
  for (int i = 0; i < 8; ++i) b[i] = reg->b[i];
  *cost = reg->cost;
  if (n_corr) *n_corr = reg->n_corr;
  return COX_OK;
}
// all constraints of one pose-graph evaluation (cox_reg_normal_eq_batch): the oracle takes them one after the other
int coxo_reg_normal_eq_batch(coxo_reg* const* regs, uint64_t n, const double* poses_ref, const double* poses_read, double* H, double* b, double* cost,
                             uint64_t* n_corr) {
  if (!regs || !n || !poses_ref || !poses_read || !H || !b || !cost) return COX_ERR_INVALID_ARG;
  for (uint64_t c = 0; c < n; ++c) {
    coxo_reg* reg = regs[c];
    if (!reg || reg->pending) return COX_ERR_INVALID_ARG;
    const uint64_t nr = reg->has_stored ? reg->stored.size() : reg->ref->pts.size();
    const int rc = regRun(reg, poses_ref + 4 * c, poses_read + 4 * c, nullptr, nr, nullptr, nullptr, nullptr, H + 64 * c, b + 8 * c, cost + c, n_corr ? n_corr + c : nullptr);
    if (rc != COX_OK) return rc;
  }
  return COX_OK;
}
int coxo_reg_set_samples(coxo_reg* reg, const uint32_t* sample_idx, uint64_t n_res) {
  if (!reg || reg->pending) return COX_ERR_INVALID_ARG;
  reg->has_stored = sample_idx != nullptr;
  reg->stored.assign(sample_idx ? sample_idx : nullptr, sample_idx ? sample_idx + n_res : nullptr);
  for (uint32_t i : reg->stored)
    if (i >= reg->ref->pts.size()) {
      reg->has_stored = false;
      reg->stored.clear();
      return COX_ERR_INVALID_ARG;
    }
  return COX_OK;
}

}  // extern "C"

// ---- recover mode: MeshConverter + processMesh (cox_oracle_mesh.hpp) ----------------------------------------------------
struct coxo_meshconv {
  MeshConverter conv;
  std::vector<RecoveredPoint> recovered;
  std::vector<V3> cur_pts;
  std::vector<Color> cur_colors;
  explicit coxo_meshconv(float vs) : conv(vs) {}
};
static int toMeshMsg(const cox_mesh_msg* m, MeshMsg* out) {
  if (!m) return COX_ERR_INVALID_ARG;
  out->block_edge_length = m->block_edge_length;
  out->mesh_blocks.resize(m->n_blocks);
  for (uint64_t b = 0; b < m->n_blocks; ++b) {
    MeshBlockMsg& mb = out->mesh_blocks[b];
    for (int k = 0; k < 3; ++k) mb.index[k] = m->block_index[3 * b + k];
    const uint64_t v0 = m->vertex_begin[b], v1 = m->vertex_begin[b + 1];
    if (v1 < v0 || v0 % 3 || v1 % 3) return COX_ERR_INVALID_ARG;
    mb.x.assign(m->x + v0, m->x + v1);
    mb.y.assign(m->y + v0, m->y + v1);
    mb.z.assign(m->z + v0, m->z + v1);
    mb.r.assign(m->r + v0, m->r + v1);
    mb.g.assign(m->g + v0, m->g + v1);
    mb.b.assign(m->b + v0, m->b + v1);
    if (m->block_has_history[b]) {
      for (uint64_t t = v0 / 3; t < v1 / 3; ++t) {
        const uint64_t h0 = m->history_begin[t], h1 = m->history_begin[t + 1];
        if (h1 < h0 || (h1 - h0) % 2) return COX_ERR_INVALID_ARG;
        mb.history.emplace_back(m->history + h0, m->history + h1);
      }
    }
  }
  for (uint64_t i = 0; i < m->n_poses; ++i) {
    const float* T = m->T_G_C + 7 * i;
    out->trajectory.push_back({m->stamp_sec[i], m->stamp_nsec[i], Transform{T[0], T[1], T[2], T[3], {T[4], T[5], T[6]}}});
  }
  return COX_OK;
}

extern "C" {
int coxo_meshconv_create(int, float interpolate_voxel_size, coxo_meshconv** out) {
  if (!out || !(interpolate_voxel_size > 0.0f)) return COX_ERR_INVALID_ARG;
  *out = new coxo_meshconv(interpolate_voxel_size);
  return COX_OK;
}
void coxo_meshconv_destroy(coxo_meshconv* c) { delete c; }
int coxo_meshconv_set_mesh(coxo_meshconv* c, const cox_mesh_msg* mesh) {
  if (!c) return COX_ERR_INVALID_ARG;
  MeshMsg m;
  const int rc = toMeshMsg(mesh, &m);
  if (rc != COX_OK) return rc;
  c->conv.setMesh(m);
  return COX_OK;
}
int coxo_meshconv_convert(coxo_meshconv* c, uint64_t* n_recovered, int* converted) {
  if (!c) return COX_ERR_INVALID_ARG;
  const bool ok = c->conv.convertToPointCloud(&c->recovered);
  if (n_recovered) *n_recovered = c->recovered.size();
  if (converted) *converted = ok ? 1 : 0;
  return COX_OK;
}
int coxo_meshconv_recovered(coxo_meshconv* c, float* xyz, uint8_t* rgb, uint64_t cap, uint64_t* n) {
  if (!c || !n) return COX_ERR_INVALID_ARG;
  *n = c->recovered.size();
  if (!xyz && !rgb) return COX_OK;
  if (cap < *n) return COX_ERR_BUFFER_TOO_SMALL;
  for (size_t i = 0; i < c->recovered.size(); ++i) {
    const RecoveredPoint& p = c->recovered[i];
    if (xyz) xyz[3 * i] = p.x, xyz[3 * i + 1] = p.y, xyz[3 * i + 2] = p.z;
    if (rgb) rgb[3 * i] = p.r, rgb[3 * i + 1] = p.g, rgb[3 * i + 2] = p.b;
  }
  return COX_OK;
}
// pose i's cloud as getNextPointcloud would yield it (i is not advanced here)
static bool meshconvCloud(coxo_meshconv* c, int32_t i, Transform* T, std::vector<V3>* pts, std::vector<Color>* colors) {
  int ii = i;
  return c->conv.getNextPointcloud(&ii, T, pts, colors);
}
// "device" pointers are host pointers here (valid until the next call on this handle)
int coxo_meshconv_next(coxo_meshconv* c, int32_t* i, float T_G_C[7], const float** xyz, const uint8_t** rgba, uint64_t* n, int* has_next) {
  if (!c || !i || !n || !has_next) return COX_ERR_INVALID_ARG;
  Transform T;
  *n = 0;
  if (!meshconvCloud(c, *i, &T, &c->cur_pts, &c->cur_colors)) {
    *has_next = 0;
    return COX_OK;
  }
  *has_next = 1;
  *n = c->cur_pts.size();
  if (T_G_C) {
    T_G_C[0] = T.qw, T_G_C[1] = T.qx, T_G_C[2] = T.qy, T_G_C[3] = T.qz;
    T_G_C[4] = T.t.x, T_G_C[5] = T.t.y, T_G_C[6] = T.t.z;
  }
  if (xyz) *xyz = c->cur_pts.empty() ? nullptr : &c->cur_pts[0].x;
  if (rgba) *rgba = c->cur_colors.empty() ? nullptr : &c->cur_colors[0].r;
  ++*i;
  return COX_OK;
}
int coxo_meshconv_download(coxo_meshconv* c, int32_t i, float* xyz, uint8_t* rgba, uint64_t cap, uint64_t* n) {
  if (!c || !n) return COX_ERR_INVALID_ARG;
  Transform T;
  std::vector<V3> pts;
  std::vector<Color> colors;
  if (!meshconvCloud(c, i, &T, &pts, &colors)) return COX_ERR_INVALID_ARG;
  *n = pts.size();
  if (!xyz && !rgba) return COX_OK;
  if (cap < pts.size()) return COX_ERR_BUFFER_TOO_SMALL;
  for (size_t k = 0; k < pts.size(); ++k) {
    if (xyz) xyz[3 * k] = pts[k].x, xyz[3 * k + 1] = pts[k].y, xyz[3 * k + 2] = pts[k].z;
    if (rgba) rgba[4 * k] = colors[k].r, rgba[4 * k + 1] = colors[k].g, rgba[4 * k + 2] = colors[k].b, rgba[4 * k + 3] = colors[k].a;
  }
  return COX_OK;
}
int coxo_meshconv_clear(coxo_meshconv* c) {
  if (!c) return COX_ERR_INVALID_ARG;
  c->conv.clear();  // the caller's recovered_pointcloud is not touched by clear()
  return COX_OK;
}
int coxo_recover_process_mesh(coxo_meshconv* c, coxo_integrator* integ, const cox_mesh_msg* mesh, uint64_t* n_recovered, uint64_t* n_integrated) {
  if (!c || !integ) return COX_ERR_INVALID_ARG;
  if (!integ->integ) return COX_ERR_UNSUPPORTED;  // recover mode is configured with a ray-casting integrator (tsdf_recover.yaml:6)
  MeshMsg m;
  const int rc = toMeshMsg(mesh, &m);
  if (rc != COX_OK) return rc;
  size_t ni = 0;
  processMesh(&c->conv, integ->integ.get(), integ->integ->layer(), m, &c->recovered, &ni);
  if (n_recovered) *n_recovered = c->recovered.size();
  if (n_integrated) *n_integrated = ni;
  return COX_OK;
}
}  // extern "C"

// ---- growth, device-resident uploads and the submap hand-over: in the oracle "device" pointers are host pointers and a
// clone is a deep copy ----------------------------------------------------------------------------------------------
extern "C" {
int coxo_layer_reserve(coxo_layer* l, uint64_t) { return l ? COX_OK : COX_ERR_INVALID_ARG; }
int coxo_layer_capacity(coxo_layer* l, uint64_t* cap) {
  if (!l || !cap) return COX_ERR_INVALID_ARG;
  *cap = ~0ull;  // unordered_map: unbounded
  return COX_OK;
}
int coxo_layer_set_auto_grow(coxo_layer* l, int) { return l ? COX_OK : COX_ERR_INVALID_ARG; }
int coxo_layer_upload_dev(coxo_layer* l, const int32_t* idx, const uint32_t* vox, uint64_t n, int action) { return coxo_layer_upload(l, idx, vox, n, action); }
int coxo_layer_export_dev(coxo_layer* l, int32_t* idx, uint32_t* vox, uint64_t cap, uint64_t* n) { return coxo_layer_download(l, idx, vox, cap, n); }
int coxo_layer_clone_to_device(const coxo_layer* src, int, uint64_t, coxo_layer** out) {
  if (!src || !out) return COX_ERR_INVALID_ARG;
  auto* d = new coxo_layer(src->layer.voxel_size, src->layer.vps);
  for (auto& kv : src->layer.blocks) {
    Block* b = d->layer.allocateBlock(kv.first);
    b->voxels = kv.second->voxels;
  }
  *out = d;
  return COX_OK;
}
int coxo_regpoints_create_dev(int device, const float* p, uint64_t n, coxo_regpoints** out) { return coxo_regpoints_create(device, p, n, out); }
int coxo_regpoints_data_dev(const coxo_regpoints* p, const float** data, uint64_t* n) {
  if (!p || !data) return COX_ERR_INVALID_ARG;
  *data = p->pts.empty() ? nullptr : &p->pts[0].x;
  if (n) *n = p->pts.size();
  return COX_OK;
}
int coxo_regpoints_download(const coxo_regpoints* p, float* out, uint64_t cap, uint64_t* n) {
  if (!p) return COX_ERR_INVALID_ARG;
  if (n) *n = p->pts.size();
  if (!out) return COX_OK;
  if (cap < p->pts.size()) return COX_ERR_BUFFER_TOO_SMALL;
  if (!p->pts.empty()) std::memcpy(out, p->pts.data(), p->pts.size() * sizeof(RegPoint));
  return COX_OK;
}
int coxo_regpoints_clone_to_device(const coxo_regpoints* src, int, coxo_regpoints** out) {
  if (!src || !out) return COX_ERR_INVALID_ARG;
  auto* h = new coxo_regpoints();
  h->pts = src->pts;
  *out = h;
  return COX_OK;
}
}  // extern "C"

// ---- finishSubmap(): surface box, isosurface registration points, ESDF, reproducible weighted sampler -----------------
extern "C" {
void coxo_esdf_config_default(cox_esdf_config* c) {
  const EsdfConfig d;
  c->max_distance_m = d.max_distance_m;
  c->min_distance_m = d.min_distance_m;
  c->default_distance_m = d.default_distance_m;
  c->min_weight = d.min_weight;
}
int coxo_layer_surface_obb(coxo_layer* l, float mn[3], float mx[3], uint64_t* n) {
  if (!l || !mn || !mx) return COX_ERR_INVALID_ARG;
  const Box b = surfaceObb(l->layer);
  for (int k = 0; k < 3; ++k) {
    mn[k] = b.min[k];
    mx[k] = b.max[k];
  }
  if (n) *n = b.count;
  return COX_OK;
}
int coxo_regpoints_from_isosurface(coxo_layer* l, float min_weight, float threshold, coxo_regpoints** out, uint64_t* n_mesh, uint64_t* n_connected) {
  if (!l || !out || !(threshold > 0.0f)) return COX_ERR_INVALID_ARG;
  auto* h = new coxo_regpoints();
  IsoStats st;
  h->pts = isosurfacePoints(l->layer, min_weight, threshold, &st);
  if (n_mesh) *n_mesh = st.n_mesh_vertices;
  if (n_connected) *n_connected = st.n_connected;
  *out = h;
  return COX_OK;
}
int coxo_esdf_from_tsdf(const coxo_layer* tsdf, const cox_esdf_config* cfg, coxo_layer** out) {
  if (!tsdf || !out) return COX_ERR_INVALID_ARG;
  EsdfConfig c;
  if (cfg) {
    c.max_distance_m = cfg->max_distance_m;
    c.min_distance_m = cfg->min_distance_m;
    c.default_distance_m = cfg->default_distance_m;
    c.min_weight = cfg->min_weight;
  }
  auto* e = new coxo_layer(tsdf->layer.voxel_size, tsdf->layer.vps);
  esdfFromTsdf(tsdf->layer, c, &e->layer);
  *out = e;
  return COX_OK;
}
int coxo_reg_draw_samples(coxo_reg* reg, uint64_t n_res, uint64_t seed) {
  if (!reg || reg->pending) return COX_ERR_INVALID_ARG;
  if (reg->ref->pts.empty()) return n_res == 0 ? COX_OK : COX_ERR_INVALID_ARG;
  reg->stored = drawWeightedSamples(reg->ref->pts, n_res, seed);
  reg->has_stored = true;
  return COX_OK;
}
int coxo_reg_get_samples(coxo_reg* reg, uint32_t* out, uint64_t cap, uint64_t* n) {
  if (!reg || !n) return COX_ERR_INVALID_ARG;
  *n = reg->has_stored ? reg->stored.size() : 0;
  if (!out || !reg->has_stored) return COX_OK;
  if (cap < reg->stored.size()) return COX_ERR_BUFFER_TOO_SMALL;
  std::copy(reg->stored.begin(), reg->stored.end(), out);
  return COX_OK;
}
}  // extern "C"

// probes for the shared plain-float asin / atan2 (tests/test_oracle_projective.py checks them against float64)
extern "C" {
int coxo_math_probe(const float* a, const float* b, uint64_t n, float* asin_out, float* atan2_out) {
  for (uint64_t i = 0; i < n; ++i) {
    asin_out[i] = cox_asinf(a[i]);
    atan2_out[i] = cox_atan2f(a[i], b[i]);
  }
  return COX_OK;
}
}
