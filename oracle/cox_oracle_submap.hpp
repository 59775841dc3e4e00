// CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see cox_oracle.hpp header).
//
// VoxgraphSubmap::finishSubmap() as coxgraph triggers it for every submap it receives
// (coxgraph/include/coxgraph/utils/msg_converter.h:113, coxgraph/src/server/submap_collection.cpp:35): the ESDF of the
// submap, the two registration point sets (relevant voxels: cox_oracle_capi.cpp; isosurface vertices: here) and the
// surface bounding box that the overlap test behind updateRegistrationConstraints()
// (coxgraph/src/server/pose_graph_interface.cpp:38) reads.  The arithmetic lives in voxgraph / voxblox / cblox forks that
// are not in the reference tree; this file restates the published upstream algorithms:
//   voxblox  mesh/marching_cubes.h, mesh/mesh_integrator.h (extractBlockMesh order, getSdfIfValid), mesh/mesh_utils.h
//            (createConnectedMesh), integrator/esdf_integrator.cc (batch update: fixed band + quasi-Euclidean wavefront)
//   voxgraph frontend/submap_collection/voxgraph_submap.cpp (findIsosurfaceVertices, getSubmapFrameSurfaceObb,
//            overlapsWith), tools/... bounding_box.cpp (getAabbFromObbAndPose), weighted_sampler.h
#pragma once
#include <deque>

#include "cox_oracle.hpp"
#include "cox_oracle_mc_table.hpp"

namespace coxo {

inline std::vector<const Block*> blocksInZyxOrder(const Layer& L) {
  std::vector<const Block*> out;
  for (auto& kv : L.blocks) out.push_back(kv.second.get());
  std::sort(out.begin(), out.end(), [](const Block* a, const Block* b) { return std::tie(a->index.z, a->index.y, a->index.x) < std::tie(b->index.z, b->index.y, b->index.x); });
  return out;
}

// ---- VoxgraphSubmap::getSubmapFrameSurfaceObb -----------------------------------------------------------------------
// observed voxels (weight > 1e-6) within one voxel of the surface (|distance| <= voxel_size); the box is grown by half a voxel
struct Box {
  float min[3] = {INFINITY, INFINITY, INFINITY}, max[3] = {-INFINITY, -INFINITY, -INFINITY};
  uint64_t count = 0;
};
inline Box surfaceObb(const Layer& L) {
  Box bx;
  const float half = 0.5f * L.voxel_size;
  for (auto& kv : L.blocks) {
    const Block* b = kv.second.get();
    for (int lin = 0; lin < L.vps * L.vps * L.vps; ++lin) {
      const TsdfVoxel& v = b->voxels[lin];
      if (v.weight > 1e-6f && std::abs(v.distance) <= L.voxel_size) {
        const int l[3] = {lin % L.vps, (lin / L.vps) % L.vps, lin / (L.vps * L.vps)};
        const float o[3] = {b->origin.x, b->origin.y, b->origin.z};
        for (int k = 0; k < 3; ++k) {
          const float c = o[k] + centerCoord(l[k], L.voxel_size);
          bx.min[k] = std::min(bx.min[k], c - half);
          bx.max[k] = std::max(bx.max[k], c + half);
        }
        ++bx.count;
      }
    }
  }
  return bx;
}
// BoundingBox::getAabbFromObbAndPose: the 8 corners through the pose, then min / max
inline Box aabbFromObbAndPose(const Box& obb, const Transform& T) {
  Box out;
  out.count = obb.count;
  for (int c = 0; c < 8; ++c) {
    const V3 p{(c & 1) ? obb.max[0] : obb.min[0], (c & 2) ? obb.max[1] : obb.min[1], (c & 4) ? obb.max[2] : obb.min[2]};
    const V3 q = transform(T, p);
    const float v[3] = {q.x, q.y, q.z};
    for (int k = 0; k < 3; ++k) {
      out.min[k] = std::min(out.min[k], v[k]);
      out.max[k] = std::max(out.max[k], v[k]);
    }
  }
  return out;
}
// VoxgraphSubmap::overlapsWith: separation along any axis -> no overlap
inline bool boxesOverlap(const Box& a, const Box& b) {
  for (int k = 0; k < 3; ++k)
    if (a.max[k] < b.min[k] || a.min[k] > b.max[k]) return false;
  return true;
}

// ---- isosurface vertices (VoxgraphSubmap::findIsosurfaceVertices) ----------------------------------------------------
// MeshIntegrator::extractBlockMesh over every block, MarchingCubes::meshCube per cube, createConnectedMesh with a vertex
// proximity threshold, then the TSDF's interpolated (distance, weight) at every surviving vertex.
// Canonical order (upstream iterates hash maps): blocks by (z, y, x); inside a block upstream's own cube order.
inline V3 mcInterpolateVertex(V3 v1, V3 v2, float sdf1, float sdf2) {
  const float diff = sdf1 - sdf2;
  if (std::abs(diff) >= 1e-6f) {
    const float t = sdf1 / diff;
    return V3{v1.x + t * (v2.x - v1.x), v1.y + t * (v2.y - v1.y), v1.z + t * (v2.z - v1.z)};
  }
  return V3{0.5f * (v1.x + v2.x), 0.5f * (v1.y + v2.y), 0.5f * (v1.z + v2.z)};
}
inline void mcMeshCube(const V3 corner[8], const float sdf[8], std::vector<V3>* out) {
  int cfg = 0;
  for (int i = 0; i < 8; ++i)
    if (sdf[i] < 0.0f) cfg |= 1 << i;
  if (cfg == 0) return;
  V3 edge[12];
  for (int e = 0; e < 12; ++e) {
    const int a = kMcEdgePairs[e][0], b = kMcEdgePairs[e][1];
    if ((sdf[a] < 0.0f && sdf[b] >= 0.0f) || (sdf[a] >= 0.0f && sdf[b] < 0.0f)) edge[e] = mcInterpolateVertex(corner[a], corner[b], sdf[a], sdf[b]);
  }
  const signed char* row = kMcTriangleTable[cfg];
  for (int c = 0; row[c] != -1; c += 3) {
    out->push_back(edge[row[c + 2]]);
    out->push_back(edge[row[c + 1]]);
    out->push_back(edge[row[c]]);
  }
}
// the cube whose lower corner is voxel (x, y, z) of block b: corners may lie in the +x / +y / +z neighbour blocks
inline void mcCube(const Layer& L, const Block* b, int x, int y, int z, float min_weight, std::vector<V3>* out) {
  V3 corner[8];
  float sdf[8];
  const V3 base{b->origin.x + centerCoord(x, L.voxel_size), b->origin.y + centerCoord(y, L.voxel_size), b->origin.z + centerCoord(z, L.voxel_size)};
  for (int i = 0; i < 8; ++i) {
    const int ox = (i ^ (i >> 1)) & 1, oy = (i >> 1) & 1, oz = i >> 2;
    int v[3] = {x + ox, y + oy, z + oz};
    BIdx nb = b->index;
    int* nbp[3] = {&nb.x, &nb.y, &nb.z};
    for (int k = 0; k < 3; ++k)
      if (v[k] >= L.vps) {
        (*nbp[k])++;
        v[k] -= L.vps;
      }
    const Block* bp = (nb == b->index) ? b : L.getBlockPtr(nb);
    if (!bp) return;
    const TsdfVoxel& vox = bp->voxels[linearIndex(v[0], v[1], v[2], L.vps)];
    if (vox.weight <= min_weight) return;  // utils::getSdfIfValid
    sdf[i] = vox.distance;
    corner[i] = V3{base.x + static_cast<float>(ox) * L.voxel_size, base.y + static_cast<float>(oy) * L.voxel_size, base.z + static_cast<float>(oz) * L.voxel_size};
  }
  mcMeshCube(corner, sdf, out);
}
inline void mcBlock(const Layer& L, const Block* b, float min_weight, std::vector<V3>* out) {
  const int n = L.vps;
  for (int x = 0; x < n - 1; ++x)
    for (int y = 0; y < n - 1; ++y)
      for (int z = 0; z < n - 1; ++z) mcCube(L, b, x, y, z, min_weight, out);
  for (int z = 0; z < n; ++z)  // max X plane
    for (int y = 0; y < n; ++y) mcCube(L, b, n - 1, y, z, min_weight, out);
  for (int z = 0; z < n; ++z)  // max Y plane
    for (int x = 0; x < n - 1; ++x) mcCube(L, b, x, n - 1, z, min_weight, out);
  for (int y = 0; y < n - 1; ++y)  // max Z plane
    for (int x = 0; x < n - 1; ++x) mcCube(L, b, x, y, n - 1, min_weight, out);
}
struct CellKey {
  int64_t x, y, z;
  bool operator==(const CellKey& o) const { return x == o.x && y == o.y && z == o.z; }
};
struct CellKeyHash {
  size_t operator()(const CellKey& k) const { return static_cast<size_t>(k.x + k.y * 17191 + k.z * 17191 * 17191); }
};
// interpolated TSDF voxel at pos (Interpolator::getVoxel(pos, &voxel, true)): false when any of the 8 neighbours is missing
inline bool interpDistanceWeight(const Layer& L, V3 pos, float* d, float* w) {
  const Interp it = getVoxelsAndQVector(L, pos);
  if (!it.ok) return false;
  const float dx = it.off[0], dy = it.off[1], dz = it.off[2];
  const float q[8] = {1.0f, dx, dy, dz, dx * dy, dy * dz, dz * dx, dx * dy * dz};
  *d = interpMember(q, it.d);
  *w = interpMember(q, it.w);
  return true;
}
struct IsoStats {
  uint64_t n_mesh_vertices = 0, n_connected = 0;
};
inline std::vector<RegPoint> isosurfacePoints(const Layer& L, float min_weight, float proximity_threshold, IsoStats* stats = nullptr) {
  std::vector<V3> verts;
  for (const Block* b : blocksInZyxOrder(L)) mcBlock(L, b, min_weight, &verts);
  const double inv = 1.0 / static_cast<double>(proximity_threshold);
  std::unordered_map<CellKey, size_t, CellKeyHash> uniques;
  std::vector<RegPoint> out;
  uint64_t connected = 0;
  for (const V3& v : verts) {
    const CellKey key{static_cast<int64_t>(std::round(static_cast<double>(v.x) * inv)), static_cast<int64_t>(std::round(static_cast<double>(v.y) * inv)),
                      static_cast<int64_t>(std::round(static_cast<double>(v.z) * inv))};
    if (!uniques.emplace(key, connected).second) continue;  // a vertex closer than the threshold to an earlier one is merged into it
    ++connected;
    float d, w;
    if (interpDistanceWeight(L, v, &d, &w)) out.push_back(RegPoint{v.x, v.y, v.z, d, w});
  }
  if (stats) {
    stats->n_mesh_vertices = verts.size();
    stats->n_connected = connected;
  }
  return out;
}

// ---- ESDF (voxblox EsdfIntegrator::updateFromTsdfLayerBatch) ---------------------------------------------------------
// Every TSDF block gets an ESDF block.  A TSDF voxel with weight >= min_weight is observed; if |distance| < min_distance_m
// it is FIXED (ESDF distance = TSDF distance) and seeds the wavefront, otherwise it starts at sign(distance) *
// default_distance_m.  The wavefront lowers |distance| through the 26-neighbourhood (quasi-Euclidean: a step costs
// voxel_size * {1, sqrt 2, sqrt 3}, accumulated in float), never crosses the sign of the source, never updates a fixed
// voxel, and does not propagate from voxels at or beyond max_distance_m.  Label-correcting relaxation converges to the
// least fixed point whatever the queue order (float addition is monotone), so any processing order gives these bits.
// The result is kept in a TSDF-layout layer: distance, weight = 1 for observed voxels (what the interpolator's validity
// test reads), colour word a = 1 for fixed voxels.
struct EsdfConfig {
  float max_distance_m = 2.0f, min_distance_m = 0.2f, default_distance_m = 2.0f, min_weight = 1e-6f;
};
inline void esdfFromTsdf(const Layer& tsdf, const EsdfConfig& cfg, Layer* esdf) {
  esdf->removeAllBlocks();
  const int n = tsdf.vps;
  std::deque<std::pair<Block*, int>> open;
  for (auto& kv : tsdf.blocks) {
    Block* eb = esdf->allocateBlock(kv.first);
    for (int lin = 0; lin < n * n * n; ++lin) {
      const TsdfVoxel& t = kv.second->voxels[lin];
      TsdfVoxel& e = eb->voxels[lin];
      if (t.weight < cfg.min_weight) continue;  // unobserved
      e.weight = 1.0f;
      if (std::abs(t.distance) < cfg.min_distance_m) {
        e.distance = t.distance;
        e.color.a = 1;
        open.emplace_back(eb, lin);
      } else {
        e.distance = (t.distance > 0.0f) ? cfg.default_distance_m : -cfg.default_distance_m;
      }
    }
  }
  const float s1 = 1.0f * tsdf.voxel_size, s2 = std::sqrt(2.0f) * tsdf.voxel_size, s3 = std::sqrt(3.0f) * tsdf.voxel_size;
  while (!open.empty()) {
    Block* b = open.front().first;
    const int lin = open.front().second;
    open.pop_front();
    const float d = b->voxels[lin].distance;
    if (std::abs(d) >= cfg.max_distance_m) continue;
    const int l[3] = {lin % n, (lin / n) % n, lin / (n * n)};
    for (int dz = -1; dz <= 1; ++dz)
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          const int m = std::abs(dx) + std::abs(dy) + std::abs(dz);
          if (m == 0) continue;
          int v[3] = {l[0] + dx, l[1] + dy, l[2] + dz};
          BIdx nb = b->index;
          int* nbp[3] = {&nb.x, &nb.y, &nb.z};
          for (int k = 0; k < 3; ++k) {
            if (v[k] < 0) {
              (*nbp[k])--;
              v[k] += n;
            } else if (v[k] >= n) {
              (*nbp[k])++;
              v[k] -= n;
            }
          }
          Block* nbk = (nb == b->index) ? b : esdf->getBlockPtr(nb);
          if (!nbk) continue;
          const int nlin = linearIndex(v[0], v[1], v[2], n);
          TsdfVoxel& t = nbk->voxels[nlin];
          if (!(t.weight > 0.0f) || t.color.a) continue;  // unobserved or fixed
          const float step = (m == 1) ? s1 : (m == 2) ? s2 : s3;
          if (d > 0.0f) {
            const float cand = d + step;
            if (t.distance > cand) {
              t.distance = cand;
              open.emplace_back(nbk, nlin);
            }
          } else {
            const float cand = d - step;
            if (t.distance < cand) {
              t.distance = cand;
              open.emplace_back(nbk, nlin);
            }
          }
        }
  }
}

// ---- WeightedSampler<RegistrationPoint>::getRandomItem made reproducible ----------------------------------------------
// voxgraph draws uniform in [0, sum w) and takes upper_bound on the cumulative weights, with an unseeded std::mt19937.  Here
// the weights are taken in fixed point (2^-20 units, so the cumulative sums are exact integers whatever the summation
// order) and draw i of a set is splitmix64(seed, i) scaled to [0, total).
inline uint64_t weightFixed(float w) {
  if (!(w > 0.0f)) return 0;
  const double s = static_cast<double>(w) * 1048576.0;
  return s >= 1.8e19 ? ~0ull : static_cast<uint64_t>(s);
}
inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
inline std::vector<uint32_t> drawWeightedSamples(const std::vector<RegPoint>& pts, uint64_t n_res, uint64_t seed) {
  std::vector<uint64_t> cum(pts.size());
  uint64_t run = 0;
  for (size_t i = 0; i < pts.size(); ++i) {
    run += weightFixed(pts[i].weight);
    cum[i] = run;
  }
  std::vector<uint32_t> out(n_res, 0);
  if (run == 0) return out;
  for (uint64_t i = 0; i < n_res; ++i) {
    const uint64_t r = splitmix64(seed * 0x9E3779B97F4A7C15ull + i);
    const uint64_t u = static_cast<uint64_t>((static_cast<unsigned __int128>(r) * run) >> 64);  // uniform in [0, total)
    out[i] = static_cast<uint32_t>(std::upper_bound(cum.begin(), cum.end(), u) - cum.begin());
  }
  return out;
}

}  // namespace coxo
