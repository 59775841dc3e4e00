// MI355X (gfx950): voxblox ProjectiveTsdfIntegrator behind the integrator factory (`method: "projective"`,
// coxgraph/config/tsdf_server_default.yaml:6-9, tsdf_server_carla.yaml:6-9).
//
// Unlike the ray-casting integrators this one is a GATHER: the point cloud becomes a range image, and every voxel of
// every block near a ray looks its range up.  That maps onto the GPU without any ordering problem -- a pixel keeps the
// minimum range (atomicMin on the float's bits), a voxel is written by exactly one thread per frame:
//
//   k_proj_points   thread = point: bearing -> pixel (asin / atan2 of include/coxgraph_hip_math.h, so the pixel is the same
//                   on every machine), atomicMin into the range image; points within [min_ray, max_ray] walk the blocks from
//                   (r + truncation) * bearing back to the sensor (RayCaster at block scale) and mark / allocate them
//   k_proj_update   workgroup = marked block, thread = 16 voxels: centre into the camera frame, bearing -> fractional pixel,
//                   adaptive interpolation of the range, sdf, observation weight, running mean; full-block coalesced
//                   read + write (12 B per voxel each way) -- HBM-bound: 96 KB per marked block
// Semantics and the decisions taken where upstream's details are not certain: oracle/cox_oracle_projective.hpp.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>

#include "../../include/coxgraph_hip.h"
#include "../../include/coxgraph_hip_math.h"
#include "cox_device.hpp"
#include "cox_internal.hpp"

using namespace cox;

namespace {

struct ProjParams {
  float qw, qx, qy, qz, tx, ty, tz;         // T_G_C
  float iqw, iqx, iqy, iqz, itx, ity, itz;  // T_C_G
  float voxel_size, block_size, block_size_inv;
  float trunc, max_weight, min_ray, max_ray;
  float fov_rad, adaptive_gap;
  int rows, cols, scheme;
  int use_const_weight, carving, use_dropoff, deintegrate;
  u32 n_points, frame_id;
};
struct ProjCounters {
  u32 n_touched, n_new_blocks, err, pad;
  // one word takes ~88 atomics/us on this chip: the counters that every wave of a large grid adds to are sharded over 64
  // cache lines (index = workgroup id & 63) and summed by the host.  [s][0] valid points, [s][1] rays cast, [s][2] updates
  u32 shard[64][16];
};
struct ProjLayer {
  u32* voxels;
  u64* ht_keys;
  u32* ht_vals;
  u32* ht_stamp;
  u64* block_keys;
  u32* d_nblocks;
  u32 ht_mask, capacity;
};

__device__ __forceinline__ F3 rigid(float qw, float qx, float qy, float qz, float tx, float ty, float tz, F3 v) {
  const F3 qv{qx, qy, qz};
  F3 uv = cross3(qv, v);
  uv = uv + uv;
  const F3 c = cross3(qv, uv);
  return F3{((v.x + qw * uv.x) + c.x) + tx, ((v.y + qw * uv.y) + c.y) + ty, ((v.z + qw * uv.z) + c.z) + tz};
}
__device__ __forceinline__ bool bearing_to_image(const ProjParams& P, float altitude, float azimuth, float* h, float* w) {
  const double H = P.rows, W = P.cols;
  const float hh = static_cast<float>((H - 1.0) * (0.5 - static_cast<double>(altitude) / static_cast<double>(P.fov_rad)));
  if (hh < 0.0f || static_cast<float>(P.rows - 1) < hh) return false;
  float ww = static_cast<float>(W * static_cast<double>(azimuth) / (2.0 * 3.14159265358979323846));
  if (ww < 0.0f) ww += static_cast<float>(P.cols);
  *h = hh;
  *w = ww;
  return 0.0f < ww && ww < static_cast<float>(P.cols - 1);
}

// RayCaster(start_scaled, end_scaled) + nextRayIndex on block-scaled points (same arithmetic as cox_device.hpp's dda_setup)
struct BlockDda {
  int c[3], sgn[3];
  float t_next[3], t_step[3];
  u32 nsteps;
  bool range_error;
};
__device__ __forceinline__ void block_dda_setup(BlockDda& d, F3 s, F3 e) {
  d.nsteps = 0;
  d.range_error = false;
  if (isnan(s.x) || isnan(s.y) || isnan(s.z) || isnan(e.x) || isnan(e.y) || isnan(e.z)) return;
  if (!(index_in_range(s.x) && index_in_range(s.y) && index_in_range(s.z) && index_in_range(e.x) && index_in_range(e.y) && index_in_range(e.z))) {
    d.range_error = true;
    return;
  }
  const float st[3] = {s.x, s.y, s.z}, en[3] = {e.x, e.y, e.z};
  u32 len = 0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    d.c[k] = grid_index(st[k]);
    const int diff = grid_index(en[k]) - d.c[k];
    len += static_cast<u32>(diff < 0 ? -diff : diff);
    const float ray = en[k] - st[k];
    const int sg = signum(ray);
    d.sgn[k] = sg;
    const float dist = static_cast<float>(sg > 0 ? sg : 0) - (st[k] - static_cast<float>(d.c[k]));
    d.t_next[k] = dist / ray;
    d.t_step[k] = static_cast<float>(sg) / ray;
  }
  d.nsteps = len + 1;
}
__device__ __forceinline__ void block_dda_step(BlockDda& d) {
  int k = 0;
  float best = d.t_next[0];
  if (d.t_next[1] < best) {
    best = d.t_next[1];
    k = 1;
  }
  if (d.t_next[2] < best) k = 2;
  d.c[0] += (k == 0) ? d.sgn[0] : 0;
  d.c[1] += (k == 1) ? d.sgn[1] : 0;
  d.c[2] += (k == 2) ? d.sgn[2] : 0;
  d.t_next[0] = (k == 0) ? d.t_next[0] + d.t_step[0] : d.t_next[0];
  d.t_next[1] = (k == 1) ? d.t_next[1] + d.t_step[1] : d.t_next[1];
  d.t_next[2] = (k == 2) ? d.t_next[2] + d.t_step[2] : d.t_next[2];
}

// Marking is idempotent within a frame, and the 256 neighbouring rays of a workgroup cross the same few dozen blocks: a
// direct-mapped LDS table of keys this workgroup has already marked answers ~95 % of the steps without a hash probe and a
// stamp check in global memory (197 -> 60 us at 5 cm).  An entry is written after its block has been marked, so whatever a
// lane reads there -- old or new -- is a marked block; relaxed LDS accesses, no ordering needed.
constexpr u32 kSeenSlots = 256;
__global__ void __launch_bounds__(256) k_proj_points(ProjParams P, const float* __restrict__ xyz, u32* __restrict__ range, ProjLayer L, u32* __restrict__ touched_slots,
                                                     ProjCounters* cnt, u32* layer_err) {
  __shared__ u64 seen[kSeenSlots];
  seen[threadIdx.x] = kEmptyKey;
  __syncthreads();
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  const u32 lane = lane_id();
  bool valid = false, casts = false;
  BlockDda d;
  d.nsteps = 0;
  if (i < P.n_points) {
    const F3 p{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
    const float distance = sqrtf(dot3(p, p));
    if (distance <= 3.0e38f && !(fabsf(distance) < kEps)) {
      float hf, wf;
      if (bearing_to_image(P, cox_asinf(p.z / distance), cox_atan2f(p.y, p.x), &hf, &wf)) {
        valid = true;
        atomicMin(&range[static_cast<u32>(static_cast<int>(hf)) * static_cast<u32>(P.cols) + static_cast<u32>(static_cast<int>(wf))], __float_as_uint(distance));
        if (P.min_ray <= distance && distance <= P.max_ray) {
          casts = true;
          const float scale = (distance + P.trunc) / distance;
          const F3 far_G = rigid(P.qw, P.qx, P.qy, P.qz, P.tx, P.ty, P.tz, F3{p.x * scale, p.y * scale, p.z * scale});
          block_dda_setup(d, far_G * P.block_size_inv, F3{P.tx, P.ty, P.tz} * P.block_size_inv);
          if (d.range_error) atomicOr(layer_err, kErrRange);
        }
      }
    }
  }
  // the walks of a wave in lockstep (neighbouring lanes are neighbouring pixels: step k of their walks is mostly the same
  // block): a lane whose lower neighbour holds the same block at the same step leaves it to that lane (which marks it, has
  // marked it at an earlier step, or leaves it to ITS neighbour in turn)
  u32 max_steps = d.nsteps;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) max_steps = max(max_steps, static_cast<u32>(__shfl_xor(static_cast<int>(max_steps), off, 64)));
  u64 last = kEmptyKey;
  for (u32 s = 0; s < max_steps; ++s) {
    const bool act = s < d.nsteps;
    u64 bkey = kEmptyKey;
    if (act) {
      bkey = pack_key(d.c[0], d.c[1], d.c[2]);
      block_dda_step(d);
    }
    bool need = act && bkey != last;
    if (act) last = bkey;
    const u64 below = __shfl_up(bkey, 1, 64);
    if (lane > 0 && below == bkey) need = false;
    if (!need) continue;
    u64* cell = &seen[static_cast<u32>((bkey * 0x9E3779B97F4A7C15ull) >> 56) & (kSeenSlots - 1u)];
    if (__hip_atomic_load(cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == bkey) continue;
    bool fresh;
    const u32 slot = ht_insert(L.ht_keys, L.ht_mask, bkey, &fresh);
    if (slot == kInvalid) {
      atomicOr(layer_err, kErrTable);
      continue;
    }
    if (fresh) {
      const u32 pool = atomicAdd(L.d_nblocks, 1u);
      if (pool < L.capacity) {
        L.ht_vals[slot] = pool;
        L.block_keys[pool] = bkey;
        atomicAdd(&cnt->n_new_blocks, 1u);
      } else {
        atomicSub(L.d_nblocks, 1u);
        atomicOr(layer_err, kErrPool);
      }
    }
    if (__hip_atomic_load(&L.ht_stamp[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != P.frame_id &&
        atomicExch(&L.ht_stamp[slot], P.frame_id) != P.frame_id)
      touched_slots[atomicAdd(&cnt->n_touched, 1u)] = slot;
    __hip_atomic_store(cell, bkey, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  const u64 mv = __ballot(valid), mc = __ballot(casts);
  if (lane == 0) {
    u32* sh = cnt->shard[blockIdx.x & 63u];
    if (mv) atomicAdd(&sh[0], static_cast<u32>(__popcll(mv)));
    if (mc) atomicAdd(&sh[1], static_cast<u32>(__popcll(mc)));
  }
}

__device__ __forceinline__ float range_px(const ProjParams& P, const u32* __restrict__ range, int h, int w) {
  const float r = __uint_as_float(range[static_cast<u32>(h) * static_cast<u32>(P.cols) + static_cast<u32>(w)]);
  return r > 3.0e38f ? 0.0f : r;  // the image is cleared to 0x7F7F7F7F ("no return") so that atomicMin can fill it
}
__device__ __forceinline__ float proj_interpolate(const ProjParams& P, const u32* __restrict__ range, float h, float w) {
  const int hr = static_cast<int>(roundf(h)), wr = static_cast<int>(roundf(w));
  const int h0 = static_cast<int>(floorf(h)), w0 = static_cast<int>(floorf(w));
  if (P.scheme == 0) return range_px(P, range, min(hr, P.rows - 1), min(wr, P.cols - 1));
  if (h0 + 1 >= P.rows || w0 + 1 >= P.cols) return range_px(P, range, h0, w0);
  const float a = range_px(P, range, h0, w0), b = range_px(P, range, h0, w0 + 1), c = range_px(P, range, h0 + 1, w0), d = range_px(P, range, h0 + 1, w0 + 1);
  const float mn = std_min(std_min(a, b), std_min(c, d)), mx = std_max(std_max(a, b), std_max(c, d));
  if (P.scheme == 1) return mn;
  if (mn < kEps) return range_px(P, range, min(hr, P.rows - 1), min(wr, P.cols - 1));
  if (P.scheme == 3 && mx - mn > P.adaptive_gap) return mn;
  const float dh = h - static_cast<float>(h0), dw = w - static_cast<float>(w0);
  return (a * (1.0f - dh) + c * dh) * (1.0f - dw) + (b * (1.0f - dh) + d * dh) * dw;
}

// one workgroup per EIGHTH of a marked block (two z slabs: 512 voxels, 6 KB of contiguous wire words, two voxels per thread):
// a frame marks 50-700 blocks, and a voxel is a chain of dependent loads (range pixels, then the voxel) -- one workgroup per
// block left the chip empty and took 16 such chains per thread (33 us however few blocks); the block still streams through
// registers, 48 KB read + 48 KB written, coalesced
__global__ void __launch_bounds__(256) k_proj_update(ProjParams P, const u32* __restrict__ range, ProjLayer L, const u32* __restrict__ touched_slots, ProjCounters* cnt,
                                                     u32* layer_err, u32* __restrict__ h_nblocks) {
  constexpr u32 kParts = 8, kPartVoxels = kVoxelsPerBlock / kParts;
  const u32 n_touched = cnt->n_touched;
  u32 updates = 0;
  for (u32 t8 = blockIdx.x; t8 < n_touched * kParts; t8 += gridDim.x) {
    const u32 t = t8 / kParts, part = t8 % kParts;
    const u32 slot = touched_slots[t];
    const u32 pool = L.ht_vals[slot];
    if (pool == kInvalid) {
      if (threadIdx.x == 0) atomicOr(layer_err, kErrPool);
      continue;
    }
    int bx, by, bz;
    unpack_key(L.ht_keys[slot], &bx, &by, &bz);
    const float ox = static_cast<float>(bx) * P.block_size, oy = static_cast<float>(by) * P.block_size, oz = static_cast<float>(bz) * P.block_size;
    u32* blk = L.voxels + static_cast<size_t>(pool) * kVoxelsPerBlock * kWordsPerVoxel;
    for (u32 v = part * kPartVoxels + threadIdx.x; v < (part + 1) * kPartVoxels; v += 256) {
      const int lx = static_cast<int>(v & 15u), ly = static_cast<int>((v >> 4) & 15u), lz = static_cast<int>(v >> 8);
      const F3 centre{ox + center_coord(lx, P.voxel_size), oy + center_coord(ly, P.voxel_size), oz + center_coord(lz, P.voxel_size)};
      const F3 q = rigid(P.iqw, P.iqx, P.iqy, P.iqz, P.itx, P.ity, P.itz, centre);
      const float dv = sqrtf(dot3(q, q));
      if (dv < P.min_ray || dv > P.max_ray) continue;
      float h, w;
      if (!bearing_to_image(P, cox_asinf(q.z / dv), cox_atan2f(q.y, q.x), &h, &w)) continue;
      const float sdf = proj_interpolate(P, range, h, w) - dv;
      if (sdf < -P.trunc) continue;
      if (!P.carving && sdf > P.trunc) continue;
      float obs = P.deintegrate ? -1.0f : 1.0f;
      if (P.use_dropoff && sdf < -P.voxel_size) {
        obs = obs * ((P.trunc + sdf) / (P.trunc - P.voxel_size));
        obs = std_max(obs, 0.0f);
      }
      if (!P.use_const_weight) obs = obs / (dv * dv);
      const float od = __uint_as_float(blk[3 * v]), ow = __uint_as_float(blk[3 * v + 1]);
      const float nw = std_min(ow + obs, P.max_weight);
      if (P.deintegrate && nw < 1.0f) {
        blk[3 * v] = 0u;
        blk[3 * v + 1] = 0u;
        ++updates;
        continue;
      }
      if (nw < kEps) continue;
      blk[3 * v] = __float_as_uint((od * ow + std_min(P.trunc, sdf) * obs) / nw);
      blk[3 * v + 1] = __float_as_uint(nw);
      ++updates;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) updates += __shfl_xor(updates, off, 64);
  if (lane_id() == 0 && updates) atomicAdd(&cnt->shard[blockIdx.x & 63u][2], updates);
  if (blockIdx.x == 0 && threadIdx.x == 0) *h_nblocks = min(*L.d_nblocks, L.capacity);
}

}  // namespace

struct cox_projective {
  cox_layer* layer = nullptr;
  cox_tsdf_config cfg;
  hipStream_t stream = nullptr;
  u32* range = nullptr;
  u32* touched_slots = nullptr;
  u32 touched_cap = 0;
  ProjCounters* cnt = nullptr;
  ProjCounters* h_cnt = nullptr;  // pinned
  float* own_xyz = nullptr;
  u64 xyz_cap = 0;
  u32 layer_generation = 0;
  cox_frame_stats last{};
  bool pending = false;
  hipEvent_t ring[4] = {};  // end of the last four frames: at most four are in flight (bounds how old the host's view of the pool is)
  uint64_t frames = 0;
  u64 blocks_seen = 0, blocks_delta_max = 0;
};

#define COX_TRY(expr)              \
  do {                             \
    int st_ = (expr);              \
    if (st_ != COX_OK) return st_; \
  } while (0)

void cox_proj_destroy(cox_projective* P) {
  if (!P) return;
  (void)hipSetDevice(P->layer->device);
  if (P->stream) (void)hipStreamSynchronize(P->stream);
  for (void* q : {static_cast<void*>(P->range), static_cast<void*>(P->touched_slots), static_cast<void*>(P->cnt), static_cast<void*>(P->own_xyz)})
    if (q) (void)hipFree(q);
  if (P->h_cnt) (void)hipHostFree(P->h_cnt);
  for (hipEvent_t e : P->ring)
    if (e) (void)hipEventDestroy(e);
  if (P->stream) (void)hipStreamDestroy(P->stream);
  delete P;
}

int cox_proj_create(cox_layer* layer, const cox_tsdf_config* cfg, cox_projective** out) {
  if (cfg->sensor_horizontal_resolution <= 1 || cfg->sensor_vertical_resolution <= 1 || !(cfg->sensor_vertical_field_of_view_degrees > 0.0f)) return COX_ERR_INVALID_ARG;
  if (static_cast<uint64_t>(cfg->sensor_horizontal_resolution) * cfg->sensor_vertical_resolution > (1ull << 28)) return COX_ERR_INVALID_ARG;
  if (cfg->projective_interpolation_scheme < 0 || cfg->projective_interpolation_scheme > 3) return COX_ERR_INVALID_ARG;
  cox_projective* P = new (std::nothrow) cox_projective();
  if (!P) return COX_ERR_OUT_OF_MEMORY;
  P->layer = layer;
  P->cfg = *cfg;
  P->layer_generation = layer->generation;
  const size_t px = static_cast<size_t>(cfg->sensor_horizontal_resolution) * cfg->sensor_vertical_resolution;
  bool ok = hipStreamCreateWithFlags(&P->stream, hipStreamNonBlocking) == hipSuccess;
  for (hipEvent_t& e : P->ring) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
  ok = ok && hipMalloc(reinterpret_cast<void**>(&P->range), sizeof(u32) * px) == hipSuccess;
  ok = ok && hipMalloc(reinterpret_cast<void**>(&P->cnt), sizeof(ProjCounters)) == hipSuccess;
  ok = ok && hipHostMalloc(reinterpret_cast<void**>(&P->h_cnt), sizeof(ProjCounters), hipHostMallocDefault) == hipSuccess;
  ok = ok && hipMalloc(reinterpret_cast<void**>(&P->touched_slots), sizeof(u32) * layer->ht_cap) == hipSuccess;
  P->touched_cap = layer->ht_cap;
  if (!ok || hipDeviceSynchronize() != hipSuccess) {
    cox_proj_destroy(P);
    return COX_ERR_NO_DEVICE;
  }
  *out = P;
  return COX_OK;
}

static int proj_finish(cox_projective* P) {
  COX_HIP(hipStreamSynchronize(P->stream));
  if (P->pending) {
    const ProjCounters& c = *P->h_cnt;
    u64 sums[3] = {0, 0, 0};
    for (int sh = 0; sh < 64; ++sh)
      for (int k = 0; k < 3; ++k) sums[k] += c.shard[sh][k];
    P->last.n_valid = sums[0];
    P->last.n_rays = sums[1];
    P->last.n_updates = sums[2];
    P->last.n_touched_voxels = sums[2];
    P->last.n_touched_blocks = c.n_touched;
    P->last.n_new_blocks = c.n_new_blocks;
    P->last.max_voxel_updates = sums[2] ? 1 : 0;
    P->pending = false;
  }
  u32 lerr = 0;
  COX_HIP(hipMemcpy(&lerr, P->layer->d_err, sizeof(u32), hipMemcpyDeviceToHost));
  if (lerr) {
    COX_HIP(hipMemset(P->layer->d_err, 0, sizeof(u32)));
    COX_HIP(hipDeviceSynchronize());
  }
  return err_bits_to_status(lerr);
}

int cox_proj_integrate(cox_projective* P, const float T[7], const float* xyz_dev, uint64_t n, int deintegrate) {
  cox_layer* L = P->layer;
  hipStream_t s = P->stream;
  // Frames queue up on ONE in-order stream: the range image, the counters and the marked-block list are cleared / rewritten
  // by the frame itself, so a frame needs no host-side wait for its predecessor; statistics and errors are those of the last
  // frame / sticky in the layer's error word and are read when somebody asks (sync, last_stats).
  // Layer::allocateBlockPtrByIndex never fails upstream: double the pool once it is half full -- or will be, at the rate blocks
  // have been allocated lately (the host's view of the count is up to four frames old)
  if (P->frames >= 4) COX_HIP(hipEventSynchronize(P->ring[P->frames & 3]));  // frame t-4 is done
  const u64 n_seen = *L->h_nblocks;
  if (n_seen > P->blocks_seen) P->blocks_delta_max = std::max<u64>(P->blocks_delta_max, n_seen - P->blocks_seen);
  P->blocks_seen = n_seen;
  const u64 need = std::max<u64>(2 * n_seen, n_seen + 6 * P->blocks_delta_max);
  if (L->auto_grow && need > L->capacity && L->capacity < (1ull << 26)) {
    COX_HIP(hipStreamSynchronize(s));
    u64 cap = L->capacity;
    while (cap < need && cap < (1ull << 26)) cap *= 2;
    const int st = cox_internal_layer_reserve(L, std::min<u64>(cap, 1ull << 26));
    if (st != COX_OK && st != COX_ERR_OUT_OF_MEMORY) return st;
    if (st == COX_ERR_OUT_OF_MEMORY) L->auto_grow = false;
  }
  if (P->layer_generation != L->generation || P->touched_cap < L->ht_cap) {
    COX_HIP(hipStreamSynchronize(s));
    (void)hipFree(P->touched_slots);
    P->touched_slots = nullptr;
    COX_HIP(hipMalloc(reinterpret_cast<void**>(&P->touched_slots), sizeof(u32) * L->ht_cap));
    P->touched_cap = L->ht_cap;
    P->layer_generation = L->generation;
  }
  if (n == 0 && P->pending) COX_TRY(proj_finish(P));  // an empty frame enqueues nothing: settle the statistics of the one before it first
  P->last = cox_frame_stats{};
  P->last.n_points = n;
  if (n == 0) return COX_OK;
  const cox_tsdf_config& c = P->cfg;
  ProjParams pp;
  memset(&pp, 0, sizeof(pp));
  pp.qw = T[0], pp.qx = T[1], pp.qy = T[2], pp.qz = T[3], pp.tx = T[4], pp.ty = T[5], pp.tz = T[6];
  // minkindr inverse: conjugate rotation, -(R^T t), Eigen's _transformVector order (same as the oracle's inverse())
  {
    const float q[4] = {T[0], -T[1], -T[2], -T[3]};
    const float qv[3] = {q[1], q[2], q[3]}, v[3] = {T[4], T[5], T[6]};
    float uv[3] = {qv[1] * v[2] - qv[2] * v[1], qv[2] * v[0] - qv[0] * v[2], qv[0] * v[1] - qv[1] * v[0]};
    for (int k = 0; k < 3; ++k) uv[k] = uv[k] + uv[k];
    const float cr[3] = {qv[1] * uv[2] - qv[2] * uv[1], qv[2] * uv[0] - qv[0] * uv[2], qv[0] * uv[1] - qv[1] * uv[0]};
    float r[3];
    for (int k = 0; k < 3; ++k) r[k] = (v[k] + q[0] * uv[k]) + cr[k];
    pp.iqw = q[0], pp.iqx = q[1], pp.iqy = q[2], pp.iqz = q[3];
    pp.itx = -r[0], pp.ity = -r[1], pp.itz = -r[2];
  }
  pp.voxel_size = L->voxel_size;
  pp.block_size = L->block_size;
  pp.block_size_inv = L->block_size_inv;
  pp.trunc = c.default_truncation_distance;
  pp.max_weight = c.max_weight;
  pp.min_ray = c.min_ray_length_m;
  pp.max_ray = c.max_ray_length_m;
  pp.fov_rad = static_cast<float>(static_cast<double>(c.sensor_vertical_field_of_view_degrees) * M_PI / 180.0);
  pp.adaptive_gap = c.projective_adaptive_gap_m;
  pp.rows = c.sensor_vertical_resolution;
  pp.cols = c.sensor_horizontal_resolution;
  pp.scheme = c.projective_interpolation_scheme;
  pp.use_const_weight = c.use_const_weight;
  pp.carving = c.voxel_carving_enabled;
  pp.use_dropoff = c.use_weight_dropoff;
  pp.deintegrate = deintegrate;
  pp.n_points = static_cast<u32>(n);
  pp.frame_id = ++L->frame_id;
  const ProjLayer PL{L->voxels, L->ht_keys, L->ht_vals, L->ht_stamp, L->block_keys, L->d_nblocks, L->ht_cap - 1, static_cast<u32>(L->capacity)};
  const size_t px = static_cast<size_t>(pp.rows) * pp.cols;
  if (cox_layer_order_writer(L, P)) cox_layer_wait_writes(L, s);  // another integrator wrote this layer last
  COX_HIP(hipMemsetAsync(P->range, 0x7F, sizeof(u32) * px, s));
  COX_HIP(hipMemsetAsync(P->cnt, 0, sizeof(ProjCounters), s));
  hipLaunchKernelGGL(k_proj_points, dim3(static_cast<u32>((n + 255) / 256)), dim3(256), 0, s, pp, xyz_dev, P->range, PL, P->touched_slots, P->cnt, L->d_err);
  hipLaunchKernelGGL(k_proj_update, dim3(8192), dim3(256), 0, s, pp, P->range, PL, P->touched_slots, P->cnt, L->d_err, L->h_nblocks);
  COX_HIP(hipMemcpyAsync(P->h_cnt, P->cnt, sizeof(ProjCounters), hipMemcpyDeviceToHost, s));
  COX_HIP(hipEventRecord(L->last_write, s));
  COX_HIP(hipEventRecord(P->ring[P->frames & 3], s));
  P->frames += 1;
  L->has_write = true;
  P->pending = true;
  COX_HIP(hipGetLastError());
  return COX_OK;
}

int cox_proj_integrate_host(cox_projective* P, const float T[7], const float* xyz, uint64_t n, int deintegrate) {
  if (P->pending) COX_TRY(proj_finish(P));  // the staging buffer below is single: the frame that used it last is done
  if (n > P->xyz_cap) {
    if (P->own_xyz) (void)hipFree(P->own_xyz);
    P->own_xyz = nullptr;
    P->xyz_cap = 0;
    COX_HIP(hipMalloc(reinterpret_cast<void**>(&P->own_xyz), sizeof(float) * 3 * n));
    P->xyz_cap = n;
  }
  if (n) COX_HIP(hipMemcpyAsync(P->own_xyz, xyz, sizeof(float) * 3 * n, hipMemcpyHostToDevice, P->stream));
  COX_TRY(cox_proj_integrate(P, T, P->own_xyz, n, deintegrate));
  return proj_finish(P);
}

int cox_proj_sync(cox_projective* P) { return proj_finish(P); }
int cox_proj_last_stats(cox_projective* P, cox_frame_stats* out) {
  if (P->pending) {
    const int st = proj_finish(P);
    if (st != COX_OK) return st;
  }
  *out = P->last;
  return COX_OK;
}
