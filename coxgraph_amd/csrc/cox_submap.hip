// MI355X (gfx950): VoxgraphSubmap::finishSubmap() on the GPU.
//
// coxgraph finishes every submap it receives (coxgraph/include/coxgraph/utils/msg_converter.h:113,
// coxgraph/src/server/submap_collection.cpp:35): voxgraph then builds the submap's ESDF, its two registration point sets
// and its bounding boxes on the CPU (voxblox's mesher + ESDF integrator).  The server is configured with
// registration_method "explicit_to_implicit" (coxgraph/config/server.yaml:28-31): reference = isosurface vertices, reading
// = the other submap's distance field.  Everything stays in HBM here:
//
//   cox_layer_surface_obb          getSubmapFrameSurfaceObb: box of the observed voxels within one voxel of the surface
//                                  (read by overlapsWith -> updateRegistrationConstraints, pose_graph_interface.cpp:38)
//   cox_regpoints_from_isosurface  findIsosurfaceVertices: marching cubes over every block (one workgroup per block, the
//                                  17^3 corner samples staged in LDS), vertices merged at the proximity threshold through a
//                                  hash on their grid cell ("first in mesh order wins" = atomicMin of the sequence number),
//                                  TSDF distance / weight interpolated at the survivors, compacted in mesh order
//   cox_esdf_from_tsdf             EsdfIntegrator::updateFromTsdfLayerBatch: fixed band + quasi-Euclidean 26-neighbourhood
//                                  wavefront, as block-parallel relaxation sweeps over 18^3 LDS tiles until nothing moves
//
// The relaxation reaches the same least fixed point as the reference's queue (float addition is monotone), min / max and
// the mesh order are order-free by construction, so all three are bit-identical to the oracle.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <new>
#include <vector>

#include "../../include/coxgraph_hip.h"
#include "cox_device.hpp"
#include "cox_internal.hpp"
#include "cox_mc_table.hpp"
#include "cox_sort.hpp"

using namespace cox;

namespace {

struct TsdfView {
  const u32* voxels;
  const u64* ht_keys;
  const u32* ht_vals;
  u32 ht_mask;
  float voxel_size, voxel_size_inv, block_size, block_size_inv;
};
TsdfView tsdf_view(const cox_layer* L) {
  return TsdfView{L->voxels, L->ht_keys, L->ht_vals, L->ht_cap - 1, L->voxel_size, L->voxel_size_inv, L->block_size, L->block_size_inv};
}
__device__ __forceinline__ u32 find_pool(const TsdfView& L, int bx, int by, int bz) {
  const u32 slot = ht_find(L.ht_keys, L.ht_mask, pack_key(bx, by, bz));
  return slot == kInvalid ? kInvalid : L.ht_vals[slot];
}

#define COX_TRY(expr)              \
  do {                             \
    int st_ = (expr);              \
    if (st_ != COX_OK) return st_; \
  } while (0)

template <typename T>
struct DevBuf {  // frees on scope exit
  T* p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  int alloc(size_t count) {
    if (count == 0) count = 1;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T));
    if (e != hipSuccess) {
      (void)hipGetLastError();
      p = nullptr;
      return e == hipErrorOutOfMemory ? COX_ERR_OUT_OF_MEMORY : COX_ERR_NO_DEVICE;
    }
    return COX_OK;
  }
};

// block count + pool indices in (z, y, x) order of the block index (the packed key orders that way)
int sorted_blocks(cox_layer* L, u32* nb_out, std::vector<u64>* keys_sorted, DevBuf<u32>* d_order) {
  COX_HIP(hipSetDevice(L->device));
  COX_HIP(hipDeviceSynchronize());
  u32 nb = 0;
  COX_HIP(hipMemcpy(&nb, L->d_nblocks, sizeof(u32), hipMemcpyDeviceToHost));
  if (nb > L->capacity) nb = static_cast<u32>(L->capacity);
  *nb_out = nb;
  if (nb == 0) return COX_OK;
  std::vector<u64> keys(nb);
  COX_HIP(hipMemcpy(keys.data(), L->block_keys, sizeof(u64) * nb, hipMemcpyDeviceToHost));
  std::vector<u32> order(nb);
  for (u32 i = 0; i < nb; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](u32 a, u32 b) { return keys[a] < keys[b]; });
  keys_sorted->resize(nb);
  for (u32 i = 0; i < nb; ++i) (*keys_sorted)[i] = keys[order[i]];
  COX_TRY(d_order->alloc(nb));
  COX_HIP(hipMemcpy(d_order->p, order.data(), sizeof(u32) * nb, hipMemcpyHostToDevice));
  return COX_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// surface OBB
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_surface_obb(const u32* __restrict__ voxels, const u64* __restrict__ block_keys, float voxel_size, float block_size,
                                                     float* __restrict__ partial /*[nb][6]*/, u32* __restrict__ counts) {
  __shared__ float red[4][6];
  __shared__ u32 cred[4];
  const u32 pool = blockIdx.x;
  const u32* blk = voxels + static_cast<size_t>(pool) * kVoxelsPerBlock * kWordsPerVoxel;
  int bx, by, bz;
  unpack_key(block_keys[pool], &bx, &by, &bz);
  const float o[3] = {static_cast<float>(bx) * block_size, static_cast<float>(by) * block_size, static_cast<float>(bz) * block_size};
  const float half = 0.5f * voxel_size;
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  u32 cnt = 0;
  for (u32 v = threadIdx.x; v < kVoxelsPerBlock; v += 256) {
    const float d = __uint_as_float(blk[3 * v]), w = __uint_as_float(blk[3 * v + 1]);
    if (w > 1e-6f && fabsf(d) <= voxel_size) {
      const int l[3] = {static_cast<int>(v & 15u), static_cast<int>((v >> 4) & 15u), static_cast<int>(v >> 8)};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float c = o[k] + center_coord(l[k], voxel_size);
        mn[k] = fminf(mn[k], c - half);
        mx[k] = fmaxf(mx[k], c + half);
      }
      ++cnt;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      mn[k] = fminf(mn[k], __shfl_xor(mn[k], off, 64));
      mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], off, 64));
    }
    cnt += __shfl_xor(cnt, off, 64);
  }
  const u32 wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63u) == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      red[wave][k] = mn[k];
      red[wave][3 + k] = mx[k];
    }
    cred[wave] = cnt;
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const bool is_min = threadIdx.x < 3;
    float r = red[0][threadIdx.x];
    for (int w = 1; w < 4; ++w) r = is_min ? fminf(r, red[w][threadIdx.x]) : fmaxf(r, red[w][threadIdx.x]);
    partial[6 * pool + threadIdx.x] = r;
  }
  if (threadIdx.x == 0) counts[pool] = cred[0] + cred[1] + cred[2] + cred[3];
}

// ---------------------------------------------------------------------------------------------------------------------
// marching cubes
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kCorner = 17;                                // corner samples per axis of one block's cubes
constexpr int kCorners = kCorner * kCorner * kCorner;      // 4913
// cube sequence number within a block (MeshIntegrator::extractBlockMesh order) -> lower-corner voxel
__device__ __forceinline__ void cube_of_seq(u32 seq, int* x, int* y, int* z) {
  if (seq < 3375u) {  // inside: x outer, y, z inner, each 0..14
    *x = static_cast<int>(seq / 225u);
    *y = static_cast<int>((seq / 15u) % 15u);
    *z = static_cast<int>(seq % 15u);
  } else if (seq < 3631u) {  // max X plane: z outer, y inner, each 0..15
    const u32 s = seq - 3375u;
    *x = 15;
    *z = static_cast<int>(s / 16u);
    *y = static_cast<int>(s % 16u);
  } else if (seq < 3871u) {  // max Y plane: z outer 0..15, x inner 0..14
    const u32 s = seq - 3631u;
    *y = 15;
    *z = static_cast<int>(s / 15u);
    *x = static_cast<int>(s % 15u);
  } else {  // max Z plane: y outer 0..14, x inner 0..14
    const u32 s = seq - 3871u;
    *z = 15;
    *y = static_cast<int>(s / 15u);
    *x = static_cast<int>(s % 15u);
  }
}
__device__ __forceinline__ F3 mc_interpolate_vertex(F3 v1, F3 v2, float sdf1, float sdf2) {
  const float diff = sdf1 - sdf2;
  if (fabsf(diff) >= 1e-6f) {
    const float t = sdf1 / diff;
    return F3{v1.x + t * (v2.x - v1.x), v1.y + t * (v2.y - v1.y), v1.z + t * (v2.z - v1.z)};
  }
  return F3{0.5f * (v1.x + v2.x), 0.5f * (v1.y + v2.y), 0.5f * (v1.z + v2.z)};
}
// One workgroup per block (i-th in (z,y,x) order).  kWrite = false: vertex count of the block; true: the vertices, at
// block_offset[i] + (offset of the cube within the block, in extractBlockMesh order).
template <bool kWrite>
__global__ void __launch_bounds__(256) k_mc_block(TsdfView L, const u32* __restrict__ order, const u64* __restrict__ block_keys, float min_weight,
                                                  u32* __restrict__ block_count, const u64* __restrict__ block_offset, float* __restrict__ verts) {
  __shared__ float sdf[kCorners];
  __shared__ unsigned char ok[kCorners];
  __shared__ u32 nbr_pool[8];
  __shared__ u32 scan_lds[4];
  const u32 i = blockIdx.x;
  const u32 pool = order[i];
  int bx, by, bz;
  unpack_key(block_keys[pool], &bx, &by, &bz);
  if (threadIdx.x < 8) {
    const int dx = threadIdx.x & 1, dy = (threadIdx.x >> 1) & 1, dz = threadIdx.x >> 2;
    nbr_pool[threadIdx.x] = (threadIdx.x == 0) ? pool : find_pool(L, bx + dx, by + dy, bz + dz);
  }
  __syncthreads();
  for (u32 c = threadIdx.x; c < kCorners; c += 256) {
    const u32 cx = c % kCorner, cy = (c / kCorner) % kCorner, cz = c / (kCorner * kCorner);
    const u32 sel = (cx == 16u ? 1u : 0u) | (cy == 16u ? 2u : 0u) | (cz == 16u ? 4u : 0u);
    const u32 p = nbr_pool[sel];
    float d = 0.0f;
    bool valid = false;
    if (p != kInvalid) {
      const u32 lin = (cx & 15u) | ((cy & 15u) << 4) | ((cz & 15u) << 8);
      const u32* vw = L.voxels + (static_cast<size_t>(p) * kVoxelsPerBlock + lin) * kWordsPerVoxel;
      d = __uint_as_float(vw[0]);
      valid = __uint_as_float(vw[1]) > min_weight;  // utils::getSdfIfValid: weight <= min_weight is invalid
    }
    sdf[c] = d;
    ok[c] = valid ? 1 : 0;
  }
  __syncthreads();
  // thread t owns the 16 consecutive cubes [16 t, 16 t + 16) of the block's sequence
  auto cube_config = [&](u32 seq, int* x, int* y, int* z) -> u32 {
    cube_of_seq(seq, x, y, z);
    u32 cfg = 0;
    bool all = true;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int ox = (c ^ (c >> 1)) & 1, oy = (c >> 1) & 1, oz = c >> 2;
      const u32 ci = static_cast<u32>((*x + ox) + kCorner * ((*y + oy) + kCorner * (*z + oz)));
      all = all && ok[ci] != 0;
      if (sdf[ci] < 0.0f) cfg |= 1u << c;
    }
    return all ? cfg : 0u;
  };
  u32 mine = 0;
#pragma unroll 1
  for (u32 k = 0; k < 16; ++k) {
    int x, y, z;
    const u32 cfg = cube_config(threadIdx.x * 16u + k, &x, &y, &z);
    u32 n = 0;
    while (n < 16 && kMcTriangleTable[cfg][n] != -1) ++n;
    mine += n;
  }
  u32 total;
  u32 off = block_exclusive_scan<4>(mine, &total, scan_lds);
  if (!kWrite) {
    if (threadIdx.x == 0) block_count[i] = total;
    return;
  }
  const float ox0 = static_cast<float>(bx) * L.block_size, oy0 = static_cast<float>(by) * L.block_size, oz0 = static_cast<float>(bz) * L.block_size;
  float* out = verts + 3ull * block_offset[i];
#pragma unroll 1
  for (u32 k = 0; k < 16; ++k) {
    int x, y, z;
    const u32 cfg = cube_config(threadIdx.x * 16u + k, &x, &y, &z);
    if (cfg == 0 || cfg == 255u) continue;
    const F3 base{ox0 + center_coord(x, L.voxel_size), oy0 + center_coord(y, L.voxel_size), oz0 + center_coord(z, L.voxel_size)};
    const signed char* row = kMcTriangleTable[cfg];
    for (int c = 0; c < 16 && row[c] != -1; c += 3) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int e = row[c + 2 - j];  // vertices are emitted as (col + 2, col + 1, col)
        const int a = kMcEdgePairs[e][0], b = kMcEdgePairs[e][1];
        const int ax = (a ^ (a >> 1)) & 1, ay = (a >> 1) & 1, az = a >> 2;
        const int bxx = (b ^ (b >> 1)) & 1, byy = (b >> 1) & 1, bzz = b >> 2;
        const F3 va{base.x + static_cast<float>(ax) * L.voxel_size, base.y + static_cast<float>(ay) * L.voxel_size, base.z + static_cast<float>(az) * L.voxel_size};
        const F3 vb{base.x + static_cast<float>(bxx) * L.voxel_size, base.y + static_cast<float>(byy) * L.voxel_size, base.z + static_cast<float>(bzz) * L.voxel_size};
        const float sa = sdf[(x + ax) + kCorner * ((y + ay) + kCorner * (z + az))];
        const float sb = sdf[(x + bxx) + kCorner * ((y + byy) + kCorner * (z + bzz))];
        const F3 v = mc_interpolate_vertex(va, vb, sa, sb);
        out[3 * off] = v.x;
        out[3 * off + 1] = v.y;
        out[3 * off + 2] = v.z;
        ++off;
      }
    }
  }
}

// Interpolator<TsdfVoxel>::getVoxel(pos, &voxel, interpolate = true): interpolated distance and weight, false when one of
// the 8 neighbours is missing or unobserved
__constant__ float c_iso_interp_table[8][8] = {{1, 0, 0, 0, 0, 0, 0, 0},   {-1, 0, 0, 0, 1, 0, 0, 0},   {-1, 0, 1, 0, 0, 0, 0, 0},
                                               {-1, 1, 0, 0, 0, 0, 0, 0},  {1, 0, -1, 0, -1, 0, 1, 0},  {1, -1, -1, 1, 0, 0, 0, 0},
                                               {1, -1, 0, 0, -1, 1, 0, 0}, {-1, 1, 1, -1, 1, -1, -1, 1}};
__device__ __forceinline__ float iso_interp_member(const float q[8], const float data[8]) {
  float md[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c) s += c_iso_interp_table[r][c] * data[c];
    md[r] = s;
  }
  float v = 0.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i) v += q[i] * md[i];
  return v;
}
__device__ __forceinline__ bool interp_distance_weight(const TsdfView& L, const float pos[3], float* d_out, float* w_out) {
  float sc[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) sc[k] = pos[k] * L.block_size_inv;
  if (!(index_in_range(sc[0]) && index_in_range(sc[1]) && index_in_range(sc[2]))) return false;
  int b[3], vi[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) b[k] = grid_index(sc[k]);
  if (find_pool(L, b[0], b[1], b[2]) == kInvalid) return false;  // the block of the point itself must exist
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float origin = static_cast<float>(b[k]) * L.block_size;
    int v = grid_index((pos[k] - origin) * L.voxel_size_inv);
    v = v > 15 ? 15 : (v < 0 ? 0 : v);
    const float c = origin + center_coord(v, L.voxel_size);
    if (pos[k] - c < 0.0f) {
      v--;
      if (v < 0) {
        b[k]--;
        v += 16;
      }
    }
    vi[k] = v;
  }
  const u32 base_pool = find_pool(L, b[0], b[1], b[2]);
  if (base_pool == kInvalid) return false;
  float d[8], w[8], off[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float c0 = static_cast<float>(b[k]) * L.block_size + center_coord(vi[k], L.voxel_size);
    off[k] = (pos[k] - c0) * L.voxel_size_inv;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int v[3] = {vi[0] + ((i >> 2) & 1), vi[1] + ((i >> 1) & 1), vi[2] + (i & 1)};
    int nb[3] = {b[0], b[1], b[2]};
    bool moved = false;
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (v[k] >= 16) {
        nb[k]++;
        v[k] -= 16;
        moved = true;
      }
    u32 pool = base_pool;
    if (moved) {
      pool = find_pool(L, nb[0], nb[1], nb[2]);
      if (pool == kInvalid) return false;
    }
    const u32* vox = L.voxels + (static_cast<size_t>(pool) * kVoxelsPerBlock + static_cast<u32>(v[0] + 16 * (v[1] + 16 * v[2]))) * kWordsPerVoxel;
    d[i] = __uint_as_float(vox[0]);
    w[i] = __uint_as_float(vox[1]);
    if (!(w[i] > 0.0f)) return false;
  }
  const float dx = off[0], dy = off[1], dz = off[2];
  const float q[8] = {1.0f, dx, dy, dz, dx * dy, dy * dz, dz * dx, dx * dy * dz};
  *d_out = iso_interp_member(q, d);
  *w_out = iso_interp_member(q, w);
  return true;
}

// createConnectedMesh: a vertex whose grid cell (cell size = proximity threshold) already holds an earlier vertex of the mesh
// is merged into it.  "Earlier" = smaller index in mesh order: every vertex inserts its cell and atomicMin's its index.
struct CellOrigin {
  long long x, y, z;  // subtracted from the cell index so that the rest fits 21 bits per axis
};
__global__ void __launch_bounds__(256) k_iso_insert(const float* __restrict__ verts, u64 n, double inv, CellOrigin org, u64* __restrict__ keys, u32* __restrict__ first,
                                                    u32 mask, u32* __restrict__ vslot, u32* d_err) {
  const u64 v = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (v >= n) return;
  const long long cx = static_cast<long long>(round(static_cast<double>(verts[3 * v]) * inv)) - org.x;
  const long long cy = static_cast<long long>(round(static_cast<double>(verts[3 * v + 1]) * inv)) - org.y;
  const long long cz = static_cast<long long>(round(static_cast<double>(verts[3 * v + 2]) * inv)) - org.z;
  vslot[v] = kInvalid;
  if (cx < 0 || cy < 0 || cz < 0 || cx >= (1 << 21) || cy >= (1 << 21) || cz >= (1 << 21)) {
    atomicOr(d_err, kErrRange);
    return;
  }
  const u64 key = static_cast<u64>(cx) | (static_cast<u64>(cy) << 21) | (static_cast<u64>(cz) << 42);
  bool fresh;
  const u32 slot = ht_insert(keys, mask, key, &fresh);
  if (slot == kInvalid) {
    atomicOr(d_err, kErrTable);
    return;
  }
  atomicMin(&first[slot], static_cast<u32>(v));
  vslot[v] = slot;
}
__global__ void __launch_bounds__(256) k_iso_flag(TsdfView L, const float* __restrict__ verts, u64 n, const u32* __restrict__ first, const u32* __restrict__ vslot,
                                                  u32* __restrict__ flag, float* __restrict__ dw, u32* __restrict__ n_connected) {
  const u64 v = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  bool unique = false, keep = false;
  if (v < n) {
    const u32 slot = vslot[v];
    unique = slot != kInvalid && first[slot] == static_cast<u32>(v);
    if (unique) {
      const float pos[3] = {verts[3 * v], verts[3 * v + 1], verts[3 * v + 2]};
      float d, w;
      keep = interp_distance_weight(L, pos, &d, &w);
      if (keep) {
        dw[2 * v] = d;
        dw[2 * v + 1] = w;
      }
    }
    flag[v] = keep ? 1u : 0u;
  }
  const u64 m = __ballot(unique);
  if (lane_id() == 0 && m) atomicAdd(n_connected, static_cast<u32>(__popcll(m)));
}
__global__ void __launch_bounds__(256) k_iso_write(const float* __restrict__ verts, u64 n, const u32* __restrict__ flag_in, const u32* __restrict__ pos,
                                                   const float* __restrict__ dw, float* __restrict__ out) {
  const u64 v = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (v >= n || !flag_in[v]) return;
  float* p = out + 5ull * pos[v];
  p[0] = verts[3 * v];
  p[1] = verts[3 * v + 1];
  p[2] = verts[3 * v + 2];
  p[3] = dw[2 * v];
  p[4] = dw[2 * v + 1];
}

// ---------------------------------------------------------------------------------------------------------------------
// ESDF
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kHalo = 18;
constexpr int kHaloCells = kHalo * kHalo * kHalo;  // 5832
constexpr u32 kEsdfFixed = 1u;                     // colour word of a fixed voxel

__global__ void __launch_bounds__(256) k_esdf_init(u32* __restrict__ voxels, float min_weight, float min_distance, float default_distance) {
  u32* blk = voxels + static_cast<size_t>(blockIdx.x) * kVoxelsPerBlock * kWordsPerVoxel;
  for (u32 v = threadIdx.x; v < kVoxelsPerBlock; v += 256) {
    const float d = __uint_as_float(blk[3 * v]), w = __uint_as_float(blk[3 * v + 1]);
    float ed = 0.0f, ew = 0.0f;
    u32 flags = 0;
    if (!(w < min_weight)) {  // observed
      ew = 1.0f;
      if (fabsf(d) < min_distance) {
        ed = d;
        flags = kEsdfFixed;
      } else {
        ed = (d > 0.0f) ? default_distance : -default_distance;
      }
    }
    blk[3 * v] = __float_as_uint(ed);
    blk[3 * v + 1] = __float_as_uint(ew);
    blk[3 * v + 2] = flags;
  }
}
// pool index of the 27 blocks around every block (kInvalid where there is none)
__global__ void __launch_bounds__(256) k_esdf_neighbors(TsdfView L, const u64* __restrict__ block_keys, u32 nb, u32* __restrict__ nbr) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nb * 27u) return;
  const u32 pool = t / 27u, j = t % 27u;
  int bx, by, bz;
  unpack_key(block_keys[pool], &bx, &by, &bz);
  const int dx = static_cast<int>(j % 3u) - 1, dy = static_cast<int>((j / 3u) % 3u) - 1, dz = static_cast<int>(j / 9u) - 1;
  const int x = bx + dx, y = by + dy, z = bz + dz;
  u32 p = kInvalid;
  if (x >= -kIdxBias && x < kIdxBias && y >= -kIdxBias && y < kIdxBias && z >= -kIdxBias && z < kIdxBias) p = (j == 13u) ? pool : find_pool(L, x, y, z);
  nbr[t] = p;
}
// One workgroup per block: the block and a one-voxel halo in LDS, relaxed in place until nothing in the tile moves (values
// only ever move towards zero, so stale halo values and the in-place races are harmless: the fixed point is the same).
__global__ void __launch_bounds__(256) k_esdf_sweep(u32* __restrict__ voxels, const u32* __restrict__ nbr, float s1, float s2, float s3, float max_distance,
                                                    u32* __restrict__ changed) {
  __shared__ float dist[kHaloCells];
  __shared__ unsigned char st[kHaloCells];  // bit 0 observed, bit 1 fixed
  __shared__ u32 nb27[27];
  __shared__ u32 moved, moved_any;
  const u32 pool = blockIdx.x;
  if (threadIdx.x < 27) nb27[threadIdx.x] = nbr[pool * 27u + threadIdx.x];
  if (threadIdx.x == 0) moved_any = 0;
  __syncthreads();
  for (u32 c = threadIdx.x; c < kHaloCells; c += 256) {
    const int hx = static_cast<int>(c % kHalo) - 1, hy = static_cast<int>((c / kHalo) % kHalo) - 1, hz = static_cast<int>(c / (kHalo * kHalo)) - 1;
    const int jx = hx < 0 ? 0 : (hx > 15 ? 2 : 1), jy = hy < 0 ? 0 : (hy > 15 ? 2 : 1), jz = hz < 0 ? 0 : (hz > 15 ? 2 : 1);
    const u32 p = nb27[jx + 3 * jy + 9 * jz];
    float d = 0.0f;
    unsigned char s = 0;
    if (p != kInvalid) {
      const u32 lin = static_cast<u32>(hx & 15) | (static_cast<u32>(hy & 15) << 4) | (static_cast<u32>(hz & 15) << 8);
      const u32* vw = voxels + (static_cast<size_t>(p) * kVoxelsPerBlock + lin) * kWordsPerVoxel;
      d = __hip_atomic_load(reinterpret_cast<const float*>(vw), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__uint_as_float(vw[1]) > 0.0f) s = 1 | ((vw[2] & kEsdfFixed) ? 2 : 0);
    }
    dist[c] = d;
    st[c] = s;
  }
  __syncthreads();
  for (int iter = 0; iter < 64; ++iter) {
    if (threadIdx.x == 0) moved = 0;
    __syncthreads();
    bool any = false;
    for (u32 v = threadIdx.x; v < kVoxelsPerBlock; v += 256) {
      const int x = static_cast<int>(v & 15u) + 1, y = static_cast<int>((v >> 4) & 15u) + 1, z = static_cast<int>(v >> 8) + 1;
      const int c = x + kHalo * (y + kHalo * z);
      if (st[c] != 1) continue;  // unobserved or fixed
      const float mine = dist[c];
      float best = mine;
#pragma unroll
      for (int dz = -1; dz <= 1; ++dz)
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
          for (int dx = -1; dx <= 1; ++dx) {
            const int m = (dx != 0) + (dy != 0) + (dz != 0);
            if (m == 0) continue;
            const int n = c + dx + kHalo * (dy + kHalo * dz);
            if (!(st[n] & 1)) continue;
            const float dn = dist[n];
            if (!(fabsf(dn) < max_distance)) continue;  // a voxel at or beyond the maximum does not propagate
            const float step = (m == 1) ? s1 : (m == 2) ? s2 : s3;
            if (dn > 0.0f) {
              const float cand = dn + step;
              if (best > cand) best = cand;
            } else {
              const float cand = dn - step;
              if (best < cand) best = cand;
            }
          }
      // a positive source can only lower a positive voxel, a non-positive source only raise a negative one; the two
      // candidates never both apply (best moved from `mine` in one direction only if mine had that sign)
      if (best != mine && ((mine > 0.0f) == (best > 0.0f))) {
        dist[c] = best;
        any = true;
      }
    }
    if (any) moved = 1;
    __syncthreads();
    const bool go = moved != 0;
    if (go && threadIdx.x == 0) moved_any = 1;
    __syncthreads();
    if (!go) break;
  }
  if (moved_any) {
    u32* blk = voxels + static_cast<size_t>(pool) * kVoxelsPerBlock * kWordsPerVoxel;
    for (u32 v = threadIdx.x; v < kVoxelsPerBlock; v += 256) {
      const int x = static_cast<int>(v & 15u) + 1, y = static_cast<int>((v >> 4) & 15u) + 1, z = static_cast<int>(v >> 8) + 1;
      const int c = x + kHalo * (y + kHalo * z);
      if (st[c] == 1) __hip_atomic_store(reinterpret_cast<float*>(&blk[3 * v]), dist[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0) *changed = 1;
  }
}

}  // namespace

extern "C" {

void cox_esdf_config_default(cox_esdf_config* c) {
  // voxblox EsdfIntegrator::Config defaults (coxgraph_client.yaml:68-69 sets max 4 m, min 0.1 m)
  c->max_distance_m = 2.0f;
  c->min_distance_m = 0.2f;
  c->default_distance_m = 2.0f;
  c->min_weight = 1e-6f;
}

int cox_layer_surface_obb(cox_layer_t* L, float min_xyz[3], float max_xyz[3], uint64_t* n_surface_voxels) {
  COX_ENTRY();
  if (!L || !min_xyz || !max_xyz) return COX_ERR_INVALID_ARG;
  u32 nb = 0;
  std::vector<u64> keys;
  DevBuf<u32> order;
  COX_TRY(sorted_blocks(L, &nb, &keys, &order));
  for (int k = 0; k < 3; ++k) {
    min_xyz[k] = INFINITY;
    max_xyz[k] = -INFINITY;
  }
  if (n_surface_voxels) *n_surface_voxels = 0;
  if (nb == 0) return COX_OK;
  DevBuf<float> partial;
  DevBuf<u32> counts;
  COX_TRY(partial.alloc(6ull * nb));
  COX_TRY(counts.alloc(nb));
  hipLaunchKernelGGL(k_surface_obb, dim3(nb), dim3(256), 0, nullptr, L->voxels, L->block_keys, L->voxel_size, L->block_size, partial.p, counts.p);
  std::vector<float> hp(6ull * nb);
  std::vector<u32> hc(nb);
  COX_HIP(hipMemcpy(hp.data(), partial.p, sizeof(float) * hp.size(), hipMemcpyDeviceToHost));
  COX_HIP(hipMemcpy(hc.data(), counts.p, sizeof(u32) * nb, hipMemcpyDeviceToHost));
  uint64_t total = 0;
  for (u32 i = 0; i < nb; ++i) {
    for (int k = 0; k < 3; ++k) {
      min_xyz[k] = std::min(min_xyz[k], hp[6ull * i + k]);
      max_xyz[k] = std::max(max_xyz[k], hp[6ull * i + 3 + k]);
    }
    total += hc[i];
  }
  if (n_surface_voxels) *n_surface_voxels = total;
  return COX_OK;
}

int cox_regpoints_from_isosurface(cox_layer_t* L, float min_weight, float vertex_proximity_threshold, cox_regpoints_t** out, uint64_t* n_mesh_vertices,
                                  uint64_t* n_connected_vertices) {
  COX_ENTRY();
  if (!L || !out || !(vertex_proximity_threshold > 0.0f)) return COX_ERR_INVALID_ARG;
  u32 nb = 0;
  std::vector<u64> keys;
  DevBuf<u32> order;
  COX_TRY(sorted_blocks(L, &nb, &keys, &order));
  if (n_mesh_vertices) *n_mesh_vertices = 0;
  if (n_connected_vertices) *n_connected_vertices = 0;
  cox_regpoints* R = new (std::nothrow) cox_regpoints();
  if (!R) return COX_ERR_OUT_OF_MEMORY;
  R->device = L->device;
  auto fail = [&](int st) {
    cox_regpoints_destroy(R);
    return st;
  };
  if (nb == 0) {
    *out = R;
    return COX_OK;
  }
  const TsdfView V = tsdf_view(L);
  // 1. vertex count per block -> offsets
  DevBuf<u32> bcount;
  DevBuf<u64> boff;
  if (int st = bcount.alloc(nb)) return fail(st);
  if (int st = boff.alloc(nb)) return fail(st);
  hipLaunchKernelGGL(k_mc_block<false>, dim3(nb), dim3(256), 0, nullptr, V, order.p, L->block_keys, min_weight, bcount.p, static_cast<const u64*>(nullptr),
                     static_cast<float*>(nullptr));
  std::vector<u32> hcount(nb);
  if (hipMemcpy(hcount.data(), bcount.p, sizeof(u32) * nb, hipMemcpyDeviceToHost) != hipSuccess) return fail(COX_ERR_NO_DEVICE);
  std::vector<u64> hoff(nb);
  u64 nv = 0;
  for (u32 i = 0; i < nb; ++i) {
    hoff[i] = nv;
    nv += hcount[i];
  }
  if (n_mesh_vertices) *n_mesh_vertices = nv;
  if (nv == 0) {
    *out = R;
    return COX_OK;
  }
  if (nv > 0x7FFFFFF0ull) return fail(COX_ERR_UNSUPPORTED);
  if (hipMemcpy(boff.p, hoff.data(), sizeof(u64) * nb, hipMemcpyHostToDevice) != hipSuccess) return fail(COX_ERR_NO_DEVICE);
  // 2. vertices in mesh order
  DevBuf<float> verts, dw;
  DevBuf<u32> vslot, flag, pos, first, misc;
  DevBuf<u64> hkeys;
  const u32 hcap = next_pow2(2 * nv);
  if (int st = verts.alloc(3 * nv)) return fail(st);
  if (int st = dw.alloc(2 * nv)) return fail(st);
  if (int st = vslot.alloc(nv)) return fail(st);
  if (int st = flag.alloc(nv)) return fail(st);
  if (int st = pos.alloc(nv)) return fail(st);
  if (int st = first.alloc(hcap)) return fail(st);
  if (int st = hkeys.alloc(hcap)) return fail(st);
  if (int st = misc.alloc(4)) return fail(st);  // [0] error bits, [1] connected vertices, [2] kept vertices
  hipLaunchKernelGGL(k_mc_block<true>, dim3(nb), dim3(256), 0, nullptr, V, order.p, L->block_keys, min_weight, static_cast<u32*>(nullptr), boff.p, verts.p);
  // 3. merge vertices that share a cell of the proximity grid
  (void)hipMemsetAsync(hkeys.p, 0xFF, sizeof(u64) * hcap, nullptr);
  (void)hipMemsetAsync(first.p, 0xFF, sizeof(u32) * hcap, nullptr);
  (void)hipMemsetAsync(misc.p, 0, sizeof(u32) * 4, nullptr);
  const double inv = 1.0 / static_cast<double>(vertex_proximity_threshold);
  // cell index of the lowest block corner, minus a margin: every vertex lies at or above it
  int mnx = INT32_MAX, mny = INT32_MAX, mnz = INT32_MAX;
  for (u64 k : keys) {
    int x, y, z;
    unpack_key(k, &x, &y, &z);
    mnx = std::min(mnx, x);
    mny = std::min(mny, y);
    mnz = std::min(mnz, z);
  }
  auto cell0 = [&](int b) { return static_cast<long long>(std::floor(static_cast<double>(b) * static_cast<double>(L->block_size) * inv)) - 4; };
  const CellOrigin org{cell0(mnx), cell0(mny), cell0(mnz)};
  const u32 grid = static_cast<u32>((nv + 255) / 256);
  hipLaunchKernelGGL(k_iso_insert, dim3(grid), dim3(256), 0, nullptr, verts.p, nv, inv, org, hkeys.p, first.p, hcap - 1, vslot.p, misc.p);
  hipLaunchKernelGGL(k_iso_flag, dim3(grid), dim3(256), 0, nullptr, V, verts.p, nv, first.p, vslot.p, flag.p, dw.p, misc.p + 1);
  // 4. compact the survivors in mesh order
  ScanWorkspace ws;
  DevBuf<u32> sums;
  if (int st = sums.alloc(scan_num_blocks(static_cast<u32>(nv)) + 2)) return fail(st);
  ws.block_sums = sums.p;
  exclusive_scan_u32(flag.p, pos.p, nullptr, static_cast<u32>(nv), static_cast<u32>(nv), misc.p + 2, ws, nullptr);
  u32 hm[4] = {0, 0, 0, 0};
  if (hipMemcpy(hm, misc.p, sizeof(hm), hipMemcpyDeviceToHost) != hipSuccess) return fail(COX_ERR_NO_DEVICE);
  if (hm[0]) return fail(err_bits_to_status(hm[0]));
  if (n_connected_vertices) *n_connected_vertices = hm[1];
  R->n = hm[2];
  if (R->n) {
    if (hipMalloc(reinterpret_cast<void**>(&R->pts), sizeof(float) * 5 * R->n) != hipSuccess) return fail(COX_ERR_OUT_OF_MEMORY);
    hipLaunchKernelGGL(k_iso_write, dim3(grid), dim3(256), 0, nullptr, verts.p, nv, flag.p, pos.p, dw.p, R->pts);
  }
  if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) return fail(COX_ERR_NO_DEVICE);
  *out = R;
  return COX_OK;
}

int cox_esdf_from_tsdf(const cox_layer_t* tsdf, const cox_esdf_config* cfg_in, cox_layer_t** esdf_out) {
  COX_ENTRY();
  if (!tsdf || !esdf_out) return COX_ERR_INVALID_ARG;
  cox_esdf_config cfg;
  if (cfg_in)
    cfg = *cfg_in;
  else
    cox_esdf_config_default(&cfg);
  if (!(cfg.max_distance_m > 0.0f) || !(cfg.min_distance_m > 0.0f) || !(cfg.default_distance_m > 0.0f)) return COX_ERR_INVALID_ARG;
  // same blocks, same hash: the ESDF starts as a copy of the TSDF layer and is rewritten in place
  cox_layer* E = nullptr;
  COX_TRY(cox_layer_clone_to_device(tsdf, tsdf->device, 0, &E));
  auto fail = [&](int st) {
    cox_layer_destroy(E);
    return st;
  };
  u32 nb = 0;
  if (hipMemcpy(&nb, E->d_nblocks, sizeof(u32), hipMemcpyDeviceToHost) != hipSuccess) return fail(COX_ERR_NO_DEVICE);
  if (nb) {
    DevBuf<u32> nbr, changed;
    if (int st = nbr.alloc(27ull * nb)) return fail(st);
    if (int st = changed.alloc(1)) return fail(st);
    hipLaunchKernelGGL(k_esdf_init, dim3(nb), dim3(256), 0, nullptr, E->voxels, cfg.min_weight, cfg.min_distance_m, cfg.default_distance_m);
    hipLaunchKernelGGL(k_esdf_neighbors, dim3((27u * nb + 255) / 256), dim3(256), 0, nullptr, tsdf_view(E), E->block_keys, nb, nbr.p);
    const float vs = E->voxel_size;
    const float s1 = 1.0f * vs, s2 = std::sqrt(2.0f) * vs, s3 = std::sqrt(3.0f) * vs;
    // a sweep moves the wavefront at least one block further; the distance field is at most max_distance wide
    const int max_sweeps = 8 + 4 * static_cast<int>(std::ceil(cfg.max_distance_m / E->block_size)) + 4096;
    bool converged = false;
    for (int s = 0; s < max_sweeps && !converged; ++s) {
      (void)hipMemsetAsync(changed.p, 0, sizeof(u32), nullptr);
      hipLaunchKernelGGL(k_esdf_sweep, dim3(nb), dim3(256), 0, nullptr, E->voxels, nbr.p, s1, s2, s3, cfg.max_distance_m, changed.p);
      u32 h = 0;
      if (hipMemcpy(&h, changed.p, sizeof(u32), hipMemcpyDeviceToHost) != hipSuccess) return fail(COX_ERR_NO_DEVICE);
      converged = h == 0;
    }
    if (!converged) return fail(COX_ERR_INTERNAL);
  }
  if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) return fail(COX_ERR_NO_DEVICE);
  *esdf_out = E;
  return COX_OK;
}

}  // extern "C"
