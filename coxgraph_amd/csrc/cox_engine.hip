// MI355X (gfx950) TSDF fusion engine behind include/coxgraph_hip.h.
//
// Replaces the work of voxblox's TsdfIntegratorBase::integratePointCloud as coxgraph calls it
// (coxgraph/include/coxgraph/map_comm/tsdf_recover.h:75) and of Layer<TsdfVoxel> maintenance
// (tsdf_recover.h:62,92,95; utils/msg_converter.h:49,107).  Design: DESIGN.md.
//
// Pipeline of one frame (all on the integrator's stream):
//   rays      simple: one ray per valid point, canonical order = voxblox "mixed" sequence number
//             merged: points bundled by terminal voxel (frame hash + stable radix sort), one ray per
//                     bundle, canonical order = (clearing?, first visit)
//   lengths   every ray knows its step count in O(1) (L1 index distance) -> exclusive scan -> each ray
//             owns a contiguous slice of the record array
//   touch     rays walk their voxels, insert block keys in the layer hash (bump-allocating pool
//             blocks) and give every block touched this frame a dense ordinal
//   emit      rays walk again and write (ordinal<<12 | linear voxel, ray id) records, ray-major
//   sort      stable radix sort by voxel id -> per voxel, records are in canonical ray order
//   apply     per voxel: the running weighted-mean/clamp update in exactly that order
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/coxgraph_hip.h"
#include "cox_device.hpp"
#include "cox_internal.hpp"
#include "cox_sort.hpp"

using namespace cox;

#define COX_HIP(call)                                  \
  do {                                                 \
    hipError_t e_ = (call);                            \
    if (e_ != hipSuccess) {                            \
      fprintf(stderr, "[coxgraph_hip] %s:%d %s -> %s\n", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
      return (e_ == hipErrorOutOfMemory) ? COX_ERR_OUT_OF_MEMORY : COX_ERR_NO_DEVICE; \
    }                                                  \
  } while (0)

// device error bits (layer->d_err / integrator counters.err)
enum : u32 { kErrPool = 1u, kErrRange = 2u, kErrTable = 4u, kErrRecords = 8u };

static int err_bits_to_status(u32 bits) {
  if (bits & (kErrPool | kErrTable)) return COX_ERR_POOL_EXHAUSTED;
  if (bits & kErrRange) return COX_ERR_INDEX_RANGE;
  if (bits & kErrRecords) return COX_ERR_INTERNAL;
  return COX_OK;
}

static u32 next_pow2(u64 v) {
  u32 p = 1;
  while (p < v) p <<= 1;
  return p;
}
static int ceil_log2(u64 v) {
  int b = 0;
  while ((1ull << b) < v) ++b;
  return b;
}

// =================================================================================================
// Layer
// =================================================================================================

// pool blocks are zero (= TsdfVoxel{0,0,Color()}) whenever they are free, so allocation is a bump
static int layer_reset_storage(cox_layer* L, u64 used_blocks, hipStream_t s) {
  COX_HIP(hipMemsetAsync(L->ht_keys, 0xFF, sizeof(u64) * L->ht_cap, s));
  COX_HIP(hipMemsetAsync(L->ht_vals, 0xFF, sizeof(u32) * L->ht_cap, s));
  COX_HIP(hipMemsetAsync(L->ht_stamp, 0, sizeof(u32) * L->ht_cap, s));
  COX_HIP(hipMemsetAsync(L->ht_ord, 0, sizeof(u32) * L->ht_cap, s));
  if (used_blocks) COX_HIP(hipMemsetAsync(L->voxels, 0, used_blocks * kVoxelsPerBlock * kWordsPerVoxel * sizeof(u32), s));
  COX_HIP(hipMemsetAsync(L->d_nblocks, 0, sizeof(u32), s));
  COX_HIP(hipMemsetAsync(L->d_err, 0, sizeof(u32), s));
  L->frame_id = 0;
  return COX_OK;
}

static int layer_read_counters(const cox_layer* L, u32* nblocks, u32* err) {
  COX_HIP(hipSetDevice(L->device));
  COX_HIP(hipDeviceSynchronize());
  u32 nb = 0, e = 0;
  COX_HIP(hipMemcpy(&nb, L->d_nblocks, sizeof(u32), hipMemcpyDeviceToHost));
  COX_HIP(hipMemcpy(&e, L->d_err, sizeof(u32), hipMemcpyDeviceToHost));
  if (nb > L->capacity) nb = static_cast<u32>(L->capacity);
  *nblocks = nb;
  *err = e;
  return COX_OK;
}

extern "C" {

const char* cox_status_string(int s) {
  switch (s) {
    case COX_OK: return "COX_OK";
    case COX_ERR_INVALID_ARG: return "COX_ERR_INVALID_ARG";
    case COX_ERR_NO_DEVICE: return "COX_ERR_NO_DEVICE";
    case COX_ERR_OUT_OF_MEMORY: return "COX_ERR_OUT_OF_MEMORY";
    case COX_ERR_POOL_EXHAUSTED: return "COX_ERR_POOL_EXHAUSTED";
    case COX_ERR_INDEX_RANGE: return "COX_ERR_INDEX_RANGE";
    case COX_ERR_UNSUPPORTED: return "COX_ERR_UNSUPPORTED";
    case COX_ERR_BUFFER_TOO_SMALL: return "COX_ERR_BUFFER_TOO_SMALL";
    case COX_ERR_INTERNAL: return "COX_ERR_INTERNAL";
  }
  return "COX_ERR_?";
}

int cox_device_count(void) {
  COX_ENTRY();
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

void cox_tsdf_config_default(cox_tsdf_config* c) {
  // voxblox TsdfIntegratorBase::Config defaults
  c->default_truncation_distance = 0.1f;
  c->max_weight = 10000.0f;
  c->voxel_carving_enabled = 1;
  c->min_ray_length_m = 0.1f;
  c->max_ray_length_m = 5.0f;
  c->use_const_weight = 0;
  c->allow_clear = 1;
  c->use_weight_dropoff = 1;
  c->use_sparsity_compensation_factor = 0;
  c->sparsity_compensation_factor = 1.0f;
  c->integrator_threads = 1;
  c->integration_order_mode = 0;
  c->enable_anti_grazing = 0;
  c->start_voxel_subsampling_factor = 2.0f;
  c->max_consecutive_ray_collisions = 2;
  c->clear_checks_every_n_frames = 1;
  c->max_integration_time_s = 3.4e38f;
  c->merged_bundle_order = 0;
  c->fast_exact_sets = 0;
}

int cox_layer_create(float voxel_size, int voxels_per_side, int device, uint64_t capacity_blocks, cox_layer_t** out) {
  COX_ENTRY();
  if (!out || !(voxel_size > 0.0f) || voxels_per_side != kVps) return COX_ERR_INVALID_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return COX_ERR_NO_DEVICE;
  COX_HIP(hipSetDevice(device));
  cox_layer* L = new (std::nothrow) cox_layer();
  if (!L) return COX_ERR_OUT_OF_MEMORY;
  L->device = device;
  L->voxel_size = voxel_size;
  L->voxel_size_inv = static_cast<float>(1.0 / voxel_size);
  L->block_size = voxel_size * static_cast<float>(kVps);
  L->block_size_inv = static_cast<float>(1.0 / L->block_size);
  L->capacity = capacity_blocks ? capacity_blocks : 16384;  // 16384 * 48 KiB = 768 MiB default
  if (L->capacity > (1ull << 26)) {
    delete L;
    return COX_ERR_INVALID_ARG;
  }
  L->ht_cap = next_pow2(2 * L->capacity);
  if (L->ht_cap < 1024) L->ht_cap = 1024;
  const size_t vox_bytes = L->capacity * kVoxelsPerBlock * kWordsPerVoxel * sizeof(u32);
  int st = COX_OK;
  auto alloc = [&](void** p, size_t bytes) {
    if (st != COX_OK) return;
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) st = (e == hipErrorOutOfMemory) ? COX_ERR_OUT_OF_MEMORY : COX_ERR_NO_DEVICE;
  };
  alloc(reinterpret_cast<void**>(&L->voxels), vox_bytes);
  alloc(reinterpret_cast<void**>(&L->ht_keys), sizeof(u64) * L->ht_cap);
  alloc(reinterpret_cast<void**>(&L->ht_vals), sizeof(u32) * L->ht_cap);
  alloc(reinterpret_cast<void**>(&L->ht_stamp), sizeof(u32) * L->ht_cap);
  alloc(reinterpret_cast<void**>(&L->ht_ord), sizeof(u32) * L->ht_cap);
  alloc(reinterpret_cast<void**>(&L->block_keys), sizeof(u64) * L->capacity);
  alloc(reinterpret_cast<void**>(&L->d_nblocks), sizeof(u32));
  alloc(reinterpret_cast<void**>(&L->d_err), sizeof(u32));
  if (st == COX_OK) st = layer_reset_storage(L, L->capacity, nullptr);
  if (st == COX_OK && hipDeviceSynchronize() != hipSuccess) st = COX_ERR_NO_DEVICE;
  if (st != COX_OK) {
    cox_layer_destroy(L);
    return st;
  }
  *out = L;
  return COX_OK;
}

void cox_layer_destroy(cox_layer_t* L) {
  if (!L) return;
  (void)hipSetDevice(L->device);
  (void)hipDeviceSynchronize();
  (void)hipFree(L->voxels);
  (void)hipFree(L->ht_keys);
  (void)hipFree(L->ht_vals);
  (void)hipFree(L->ht_stamp);
  (void)hipFree(L->ht_ord);
  (void)hipFree(L->block_keys);
  (void)hipFree(L->d_nblocks);
  (void)hipFree(L->d_err);
  delete L;
}

int cox_layer_clear(cox_layer_t* L) {
  COX_ENTRY();
  if (!L) return COX_ERR_INVALID_ARG;
  u32 nb, err;
  int st = layer_read_counters(L, &nb, &err);
  if (st != COX_OK) return st;
  st = layer_reset_storage(L, nb, nullptr);
  if (st != COX_OK) return st;
  COX_HIP(hipDeviceSynchronize());
  return COX_OK;
}

int cox_layer_stats(cox_layer_t* L, uint64_t* n_blocks, uint64_t* memory_bytes) {
  COX_ENTRY();
  if (!L) return COX_ERR_INVALID_ARG;
  u32 nb, err;
  int st = layer_read_counters(L, &nb, &err);
  if (st != COX_OK) return st;
  if (n_blocks) *n_blocks = nb;
  if (memory_bytes) *memory_bytes = static_cast<uint64_t>(nb) * kVoxelsPerBlock * kWordsPerVoxel * sizeof(u32);
  return err_bits_to_status(err);
}

int cox_layer_download(cox_layer_t* L, int32_t* block_idx_xyz, uint32_t* voxels_3u32, uint64_t cap_blocks, uint64_t* n_blocks) {
  COX_ENTRY();
  if (!L) return COX_ERR_INVALID_ARG;
  u32 nb, err;
  int st = layer_read_counters(L, &nb, &err);
  if (st != COX_OK) return st;
  if (n_blocks) *n_blocks = nb;
  if (cap_blocks == 0 && !block_idx_xyz && !voxels_3u32) return err_bits_to_status(err);
  if (cap_blocks < nb) return COX_ERR_BUFFER_TOO_SMALL;
  if (nb == 0) return err_bits_to_status(err);
  if (!block_idx_xyz || !voxels_3u32) return COX_ERR_INVALID_ARG;
  std::vector<u64> keys(nb);
  COX_HIP(hipMemcpy(keys.data(), L->block_keys, sizeof(u64) * nb, hipMemcpyDeviceToHost));
  // deterministic (z,y,x) block order: the packed key already orders that way
  std::vector<u32> order(nb);
  for (u32 i = 0; i < nb; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](u32 a, u32 b) { return keys[a] < keys[b]; });
  const size_t block_words = static_cast<size_t>(kVoxelsPerBlock) * kWordsPerVoxel;
  for (u32 i = 0; i < nb; ++i) {
    int x, y, z;
    unpack_key(keys[order[i]], &x, &y, &z);
    block_idx_xyz[3 * i + 0] = x;
    block_idx_xyz[3 * i + 1] = y;
    block_idx_xyz[3 * i + 2] = z;
    COX_HIP(hipMemcpy(voxels_3u32 + i * block_words, L->voxels + static_cast<size_t>(order[i]) * block_words, block_words * sizeof(u32),
                      hipMemcpyDeviceToHost));
  }
  return err_bits_to_status(err);
}

}  // extern "C"

// ---- upload (deserializeMsgToLayer) -------------------------------------------------------------
__global__ void k_upload_insert(u64* ht_keys, u32* ht_vals, u32 ht_mask, u64* block_keys, u32* d_nblocks, u32 capacity, u32* d_err,
                                const int32_t* __restrict__ idx, u32 n, u32* __restrict__ pool_of) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int x = idx[3 * i], y = idx[3 * i + 1], z = idx[3 * i + 2];
  pool_of[i] = kInvalid;
  if (x < -kIdxBias + 1 || x >= kIdxBias - 1 || y < -kIdxBias + 1 || y >= kIdxBias - 1 || z < -kIdxBias + 1 || z >= kIdxBias - 1) {
    atomicOr(d_err, kErrRange);
    return;
  }
  const u64 key = pack_key(x, y, z);
  bool fresh;
  const u32 slot = ht_insert(ht_keys, ht_mask, key, &fresh);
  if (slot == kInvalid) {
    atomicOr(d_err, kErrTable);
    return;
  }
  if (fresh) {
    const u32 pool = atomicAdd(d_nblocks, 1u);
    if (pool >= capacity) {
      atomicOr(d_err, kErrPool);
      return;
    }
    ht_vals[slot] = pool;
    block_keys[pool] = key;
    pool_of[i] = pool | 0x80000000u;  // bit 31: fresh (a message never repeats a block)
  } else {
    pool_of[i] = ht_vals[slot];  // block existed before this launch
  }
}
// one workgroup per uploaded block; action 0/2: overwrite, 1: mergeVoxelAIntoVoxelB
__global__ void __launch_bounds__(256) k_upload_copy(u32* __restrict__ voxels, const u32* __restrict__ src, const u32* __restrict__ pool_of, int action) {
  const u32 b = blockIdx.x;
  const u32 p = pool_of[b];
  if (p == kInvalid) return;
  const u32 pool = p & 0x7FFFFFFFu;
  u32* dst = voxels + static_cast<size_t>(pool) * kVoxelsPerBlock * kWordsPerVoxel;
  const u32* s = src + static_cast<size_t>(b) * kVoxelsPerBlock * kWordsPerVoxel;
  if (action != 1) {
    for (u32 i = threadIdx.x; i < kVoxelsPerBlock * kWordsPerVoxel; i += blockDim.x) dst[i] = s[i];
    return;
  }
  for (u32 v = threadIdx.x; v < kVoxelsPerBlock; v += blockDim.x) {
    const float ad = __uint_as_float(s[3 * v]), aw = __uint_as_float(s[3 * v + 1]);
    const u32 ac = s[3 * v + 2];
    const float bd = __uint_as_float(dst[3 * v]), bw = __uint_as_float(dst[3 * v + 1]);
    const u32 bc = dst[3 * v + 2];
    const float cw = aw + bw;
    if (cw > 0.0f) {
      dst[3 * v] = __float_as_uint((ad * aw + bd * bw) / cw);
      dst[3 * v + 2] = blend_colors(ac, aw, bc, bw);
      dst[3 * v + 1] = __float_as_uint(cw);
    }
  }
}

extern "C" int cox_layer_upload(cox_layer_t* L, const int32_t* block_idx_xyz, const uint32_t* voxels_3u32, uint64_t n_blocks, int action) {
  COX_ENTRY();
  if (!L || action < 0 || action > 2 || (n_blocks && (!block_idx_xyz || !voxels_3u32))) return COX_ERR_INVALID_ARG;
  COX_HIP(hipSetDevice(L->device));
  if (action == 2) {
    int st = cox_layer_clear(L);
    if (st != COX_OK) return st;
  }
  if (n_blocks == 0) return COX_OK;
  if (n_blocks > L->capacity) return COX_ERR_POOL_EXHAUSTED;
  const size_t block_words = static_cast<size_t>(kVoxelsPerBlock) * kWordsPerVoxel;
  int32_t* d_idx = nullptr;
  u32 *d_src = nullptr, *d_pool = nullptr;
  COX_HIP(hipMalloc(reinterpret_cast<void**>(&d_idx), sizeof(int32_t) * 3 * n_blocks));
  COX_HIP(hipMalloc(reinterpret_cast<void**>(&d_pool), sizeof(u32) * n_blocks));
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_src), sizeof(u32) * block_words * n_blocks);
  if (e != hipSuccess) {
    (void)hipFree(d_idx);
    (void)hipFree(d_pool);
    return COX_ERR_OUT_OF_MEMORY;
  }
  COX_HIP(hipMemcpy(d_idx, block_idx_xyz, sizeof(int32_t) * 3 * n_blocks, hipMemcpyHostToDevice));
  COX_HIP(hipMemcpy(d_src, voxels_3u32, sizeof(u32) * block_words * n_blocks, hipMemcpyHostToDevice));
  const u32 n = static_cast<u32>(n_blocks);
  hipLaunchKernelGGL(k_upload_insert, dim3((n + 255) / 256), dim3(256), 0, nullptr, L->ht_keys, L->ht_vals, L->ht_cap - 1, L->block_keys,
                     L->d_nblocks, static_cast<u32>(L->capacity), L->d_err, d_idx, n, d_pool);
  hipLaunchKernelGGL(k_upload_copy, dim3(n), dim3(256), 0, nullptr, L->voxels, d_src, d_pool, action);
  COX_HIP(hipDeviceSynchronize());
  (void)hipFree(d_idx);
  (void)hipFree(d_src);
  (void)hipFree(d_pool);
  u32 nb, err;
  int st = layer_read_counters(L, &nb, &err);
  if (st != COX_OK) return st;
  return err_bits_to_status(err);
}

// =================================================================================================
// Integrator
// =================================================================================================
struct Counters {  // per-frame device counters, zeroed at frame start
  u32 n_valid, n_rays, n_records, n_touched, n_short, n_long, n_updates, err, n_new_blocks, n_depth_points, pad0, pad1;
};

struct LayerView {
  u32* voxels;
  u64* ht_keys;
  u32* ht_vals;
  u32* ht_stamp;
  u32* ht_ord;
  u64* block_keys;
  u32* d_nblocks;
  u32 ht_mask;
  u32 capacity;
};

struct RayArrays {
  float *px, *py, *pz, *w;  // point_G and merged weight of each ray
  u32* color;               // wire-packed colour
  u32* flags;               // bit0 valid, bit1 clearing
  u64* key;                 // terminal voxel key (anti-grazing)
  u32* nsteps;              // records this ray emits
  u32* rec_off;             // exclusive scan of nsteps
};

__device__ __forceinline__ u32 pack_rgba_wire(const uint8_t* rgba, u32 i) {
  if (!rgba) return 0u;
  const u32 v = reinterpret_cast<const u32*>(rgba)[i];  // little endian: r | g<<8 | b<<16 | a<<24
  return ((v >> 24) & 255u) | (((v >> 16) & 255u) << 8) | (((v >> 8) & 255u) << 16) | ((v & 255u) << 24);
}
__device__ __forceinline__ bool point_valid(const FrameParams& P, F3 p, bool* clearing) {
  const float r = sqrtf(dot3(p, p));
  if (r < P.min_ray) return false;
  if (r > P.max_ray) {
    if (P.allow_clear || P.freespace) {
      *clearing = true;
      return true;
    }
    return false;
  }
  *clearing = P.freespace != 0;
  return true;
}
__device__ __forceinline__ float voxel_weight(const FrameParams& P, F3 p) {
  if (P.use_const_weight) return 1.0f;
  const float dz = fabsf(p.z);
  if (dz > kEps) return 1.0f / (dz * dz);
  return 0.0f;
}

// ---- simple: one ray per point, ray id = mixed-order sequence number --------------------------
__global__ void __launch_bounds__(256) k_rays_simple(FrameParams P, const float* __restrict__ xyz, const uint8_t* __restrict__ rgba, RayArrays R,
                                                     Counters* cnt) {
  const u32 seq = blockIdx.x * blockDim.x + threadIdx.x;
  if (seq >= P.n_points) return;
  const u32 idx = mixed_index(seq, P.n_points);
  const F3 p{xyz[3 * idx], xyz[3 * idx + 1], xyz[3 * idx + 2]};
  bool clearing = false;
  const bool valid = point_valid(P, p, &clearing);
  u32 nsteps = 0, flags = 0;
  if (valid) {
    const F3 pg = transform_point(P, p);
    Dda d;
    dda_setup(d, P, pg, clearing);
    if (d.range_error) atomicOr(&cnt->err, kErrRange);
    nsteps = d.nsteps;
    flags = 1u | (clearing ? 2u : 0u);
    R.px[seq] = pg.x;
    R.py[seq] = pg.y;
    R.pz[seq] = pg.z;
    R.w[seq] = voxel_weight(P, p);
    R.color[seq] = pack_rgba_wire(rgba, idx);
  }
  R.flags[seq] = flags;
  R.nsteps[seq] = nsteps;
  const u64 m = __ballot(valid);
  if (lane_id() == 0 && m) {
    atomicAdd(&cnt->n_valid, static_cast<u32>(__popcll(m)));
    atomicAdd(&cnt->n_rays, static_cast<u32>(__popcll(m)));
  }
}

// ---- merged: bundle points by terminal voxel ---------------------------------------------------
// thread = sequence number; inserts the terminal-voxel key in the per-frame hash and records the
// first (smallest) sequence number of each bundle
__global__ void __launch_bounds__(256) k_bundle_insert(FrameParams P, const float* __restrict__ xyz, u64* __restrict__ fh_keys, u32* __restrict__ fh_first,
                                                       u32 fh_mask, u32* __restrict__ pslot, Counters* cnt) {
  const u32 seq = blockIdx.x * blockDim.x + threadIdx.x;
  if (seq >= P.n_points) return;
  const u32 idx = mixed_index(seq, P.n_points);
  const F3 p{xyz[3 * idx], xyz[3 * idx + 1], xyz[3 * idx + 2]};
  bool clearing = false;
  bool valid = point_valid(P, p, &clearing);
  u32 slot = kInvalid;
  if (valid) {
    const F3 pg = transform_point(P, p);
    const float sx = pg.x * P.voxel_size_inv, sy = pg.y * P.voxel_size_inv, sz = pg.z * P.voxel_size_inv;
    if (!(index_in_range(sx) && index_in_range(sy) && index_in_range(sz))) {  // also catches NaN
      atomicOr(&cnt->err, kErrRange);
      valid = false;
    } else {
      const u64 key = pack_key(grid_index(sx), grid_index(sy), grid_index(sz)) | (clearing ? (1ull << 63) : 0ull);
      bool fresh;
      slot = ht_insert(fh_keys, fh_mask, key, &fresh);
      if (slot == kInvalid) {
        atomicOr(&cnt->err, kErrTable);
        valid = false;
      } else {
        atomicMin(&fh_first[slot], seq);
      }
    }
  }
  pslot[seq] = valid ? slot : kInvalid;
  const u64 m = __ballot(valid);
  if (lane_id() == 0 && m) atomicAdd(&cnt->n_valid, static_cast<u32>(__popcll(m)));
}
// sort key of a point = (clearing ? np2 : 0) + first sequence number of its bundle; value = seq
__global__ void __launch_bounds__(256) k_bundle_keys(u32 n, u32 np2, const u64* __restrict__ fh_keys, const u32* __restrict__ fh_first,
                                                     const u32* __restrict__ pslot, u32* __restrict__ skey, u32* __restrict__ sval) {
  const u32 seq = blockIdx.x * blockDim.x + threadIdx.x;
  if (seq >= n) return;
  const u32 slot = pslot[seq];
  u32 k = kInvalid;
  if (slot != kInvalid) k = fh_first[slot] + ((fh_keys[slot] >> 63) ? np2 : 0u);
  skey[seq] = k;
  sval[seq] = seq;
}
__global__ void __launch_bounds__(256) k_bundle_heads(u32 n, const u32* __restrict__ skey, u32* __restrict__ head) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u32 k = skey[i];
  head[i] = (k != kInvalid && (i == 0 || skey[i - 1] != k)) ? 1u : 0u;
}
__global__ void __launch_bounds__(256) k_bundle_starts(u32 n, const u32* __restrict__ skey, const u32* __restrict__ head_scan, u32* __restrict__ bstart) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u32 k = skey[i];
  if (k != kInvalid && (i == 0 || skey[i - 1] != k)) bstart[head_scan[i]] = i;
}
// wave = bundle: the sequential weighted mean of its points in visiting order (bit-exact with the
// single-threaded reference loop).  64 points at a time are gathered in parallel; the running mean
// itself is a dependent chain and is evaluated uniformly by the whole wave out of registers.
__device__ __forceinline__ float readlane_f32(float v, u32 lane) { return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), lane)); }

__global__ void __launch_bounds__(256) k_bundle_merge(FrameParams P, u32 np2, const float* __restrict__ xyz, const uint8_t* __restrict__ rgba,
                                                      const u32* __restrict__ skey, const u32* __restrict__ sval, const u32* __restrict__ bstart,
                                                      RayArrays R, Counters* cnt) {
  const u32 n_bundles = cnt->n_rays;  // written by the scan of head flags
  const u32 n_valid = cnt->n_valid;
  const u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
  const u32 nthreads = gridDim.x * blockDim.x;
  // ray slots beyond the bundle count emit nothing
  for (u32 m = n_bundles + tid; m < P.n_points; m += nthreads) {
    R.nsteps[m] = 0;
    R.flags[m] = 0;
  }
  const u32 lane = lane_id();
  for (u32 m = tid >> 6; m < n_bundles; m += nthreads >> 6) {
    const u32 begin = bstart[m];
    const u32 end = (m + 1 < n_bundles) ? bstart[m + 1] : n_valid;
    const bool clearing = skey[begin] >= np2;
    float mx = 0.0f, my = 0.0f, mz = 0.0f, W = 0.0f;
    u32 mcolor = 0;
    u64 key = 0;
    bool done = false;
    for (u32 base = begin; base < end && !done; base += 64) {
      const u32 i = base + lane;
      float px = 0.0f, py = 0.0f, pz = 0.0f, w = 0.0f;
      u32 col = 0;
      if (i < end) {
        const u32 idx = mixed_index(sval[i], P.n_points);
        px = xyz[3 * idx];
        py = xyz[3 * idx + 1];
        pz = xyz[3 * idx + 2];
        w = voxel_weight(P, F3{px, py, pz});
        col = pack_rgba_wire(rgba, idx);
      }
      if (base == begin) {
        const F3 pg = transform_point(P, F3{readlane_f32(px, 0), readlane_f32(py, 0), readlane_f32(pz, 0)});
        key = pack_key(grid_index(pg.x * P.voxel_size_inv), grid_index(pg.y * P.voxel_size_inv), grid_index(pg.z * P.voxel_size_inv));
      }
      const u32 cnt_in = min(64u, end - base);
      for (u32 k = 0; k < cnt_in; ++k) {
        const float w1 = readlane_f32(w, k);
        if (w1 < kEps) continue;
        const float x1 = readlane_f32(px, k), y1 = readlane_f32(py, k), z1 = readlane_f32(pz, k);
        const float den = W + w1;
        mx = (mx * W + x1 * w1) / den;
        my = (my * W + y1 * w1) / den;
        mz = (mz * W + z1 * w1) / den;
        mcolor = blend_colors(mcolor, W, static_cast<u32>(__builtin_amdgcn_readlane(col, k)), w1);
        W += w1;
        if (clearing) {  // only the first point of a clearing bundle is used
          done = true;
          break;
        }
      }
    }
    if (lane == 0) {
      const F3 pg = transform_point(P, F3{mx, my, mz});
      Dda d;
      dda_setup(d, P, pg, clearing);
      if (d.range_error) atomicOr(&cnt->err, kErrRange);
      R.px[m] = pg.x;
      R.py[m] = pg.y;
      R.pz[m] = pg.z;
      R.w[m] = W;
      R.color[m] = mcolor;
      R.flags[m] = 1u | (clearing ? 2u : 0u);
      R.key[m] = key;
      R.nsteps[m] = d.nsteps;
    }
  }
}

// ---- touch: allocate blocks, give every block touched this frame a dense ordinal ---------------
// anti-grazing (merged only): skip voxels that are the terminal voxel of another (non-clearing) bundle
__device__ __forceinline__ bool grazing_skip(const FrameParams& P, const u64* fh_keys, u32 fh_mask, bool clearing, u64 own_key, int x, int y, int z) {
  if (!P.anti_grazing) return false;
  const u64 k = pack_key(x, y, z);
  if (!clearing && k == own_key) return false;
  return ht_find(fh_keys, fh_mask, k) != kInvalid;
}

__global__ void __launch_bounds__(256) k_touch(FrameParams P, RayArrays R, u32 n_rays_max, LayerView L, u32* __restrict__ touched_slots, Counters* cnt,
                                               u32* layer_err, const u64* __restrict__ fh_keys, u32 fh_mask) {
  const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rays_max) return;
  const u32 ns = R.nsteps[r];
  if (ns == 0) return;
  const u32 flags = R.flags[r];
  const bool clearing = (flags & 2u) != 0;
  const F3 pg{R.px[r], R.py[r], R.pz[r]};
  const u64 own_key = P.anti_grazing ? R.key[r] : 0ull;
  Dda d;
  dda_setup(d, P, pg, clearing);
  u64 last_bkey = kEmptyKey;
  for (u32 s = 0; s < ns; ++s) {
    const int x = d.c[0], y = d.c[1], z = d.c[2];
    dda_step(d);
    if (grazing_skip(P, fh_keys, fh_mask, clearing, own_key, x, y, z)) continue;
    const u64 bkey = pack_key(x >> 4, y >> 4, z >> 4);
    if (bkey == last_bkey) continue;
    last_bkey = bkey;
    bool fresh;
    const u32 slot = ht_insert(L.ht_keys, L.ht_mask, bkey, &fresh);
    if (slot == kInvalid) {
      atomicOr(layer_err, kErrTable);
      continue;
    }
    if (fresh) {
      const u32 pool = atomicAdd(L.d_nblocks, 1u);
      if (pool < L.capacity) {
        L.ht_vals[slot] = pool;  // read by later kernels only
        L.block_keys[pool] = bkey;
        atomicAdd(&cnt->n_new_blocks, 1u);
      } else {
        atomicOr(layer_err, kErrPool);  // ht_vals[slot] stays kInvalid: updates to this block are dropped
      }
    }
    if (__hip_atomic_load(&L.ht_stamp[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != P.frame_id &&
        atomicExch(&L.ht_stamp[slot], P.frame_id) != P.frame_id) {
      const u32 ord = atomicAdd(&cnt->n_touched, 1u);
      touched_slots[ord] = slot;
      L.ht_ord[slot] = ord;
    }
  }
}

// ---- emit: (voxel id, ray id) records, ray-major ------------------------------------------------
__global__ void __launch_bounds__(256) k_emit(FrameParams P, RayArrays R, u32 n_rays_max, LayerView L, u32* __restrict__ rec_key, u32* __restrict__ rec_ray,
                                              u32 rec_cap, Counters* cnt, const u64* __restrict__ fh_keys, u32 fh_mask) {
  const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rays_max) return;
  const u32 ns = R.nsteps[r];
  if (ns == 0) return;
  const u32 off = R.rec_off[r];
  if (static_cast<u64>(off) + ns > rec_cap) {
    atomicOr(&cnt->err, kErrRecords);
    return;
  }
  const u32 flags = R.flags[r];
  const bool clearing = (flags & 2u) != 0;
  const F3 pg{R.px[r], R.py[r], R.pz[r]};
  const u64 own_key = P.anti_grazing ? R.key[r] : 0ull;
  Dda d;
  dda_setup(d, P, pg, clearing);
  u64 last_bkey = kEmptyKey;
  u32 last_ord = kInvalid;
  for (u32 s = 0; s < ns; ++s) {
    const int x = d.c[0], y = d.c[1], z = d.c[2];
    dda_step(d);
    u32 vid = kInvalid;
    if (!grazing_skip(P, fh_keys, fh_mask, clearing, own_key, x, y, z)) {
      const u64 bkey = pack_key(x >> 4, y >> 4, z >> 4);
      if (bkey != last_bkey) {
        last_bkey = bkey;
        const u32 slot = ht_find(L.ht_keys, L.ht_mask, bkey);
        last_ord = (slot != kInvalid && L.ht_vals[slot] != kInvalid) ? L.ht_ord[slot] : kInvalid;
      }
      if (last_ord != kInvalid) vid = (last_ord << 12) | static_cast<u32>((x & 15) | ((y & 15) << 4) | ((z & 15) << 8));
    }
    rec_key[off + s] = vid;
    rec_ray[off + s] = r;
  }
}

// ---- apply: per voxel, the running weighted-mean / clamp update in canonical ray order -----------------
// After the stable sort the records of one voxel are contiguous ("segment") and in ray order.
// One wave owns 64 consecutive records:
//   1. every lane evaluates its own record (voxel centre, sdf, update weight, colour) -- the costly,
//      order-independent part -- fully in parallel;
//   2. the head lane of every segment that ends inside the wave replays its records in order out of the
//      neighbours' registers (__shfl), at most 63 dependent steps;
//   3. a segment that runs past the wave ("long": at most one per wave, a few near-camera voxels that
//      every ray crosses) is then folded by the whole wave, 64 records per step; runs of free-space
//      observations over a voxel already sitting at +trunc are collapsed exactly (saturating_update).
// No atomics on the data path; results are bit-reproducible run to run.
struct VoxelRef {
  u32* ptr;  // 3 words
  int gx, gy, gz;
  bool ok;
};
__device__ __forceinline__ VoxelRef locate_voxel(const LayerView& L, const u32* touched_slots, u32 vid) {
  VoxelRef v;
  const u32 ord = vid >> 12, lin = vid & 4095u;
  const u32 slot = touched_slots[ord];
  const u32 pool = L.ht_vals[slot];
  int bx, by, bz;
  unpack_key(L.ht_keys[slot], &bx, &by, &bz);
  v.gx = bx * 16 + static_cast<int>(lin & 15u);
  v.gy = by * 16 + static_cast<int>((lin >> 4) & 15u);
  v.gz = bz * 16 + static_cast<int>(lin >> 8);
  v.ok = pool != kInvalid;
  v.ptr = L.voxels + (static_cast<size_t>(pool) * kVoxelsPerBlock + lin) * kWordsPerVoxel;
  return v;
}

__global__ void __launch_bounds__(256) k_apply(FrameParams P, RayArrays R, LayerView L, const u32* __restrict__ touched_slots,
                                               const u32* __restrict__ rec_key, const u32* __restrict__ rec_ray, const u32* __restrict__ d_nrec,
                                               Counters* cnt) {
  __shared__ u32 blk_updates, blk_voxels;
  if (threadIdx.x == 0) {
    blk_updates = 0;
    blk_voxels = 0;
  }
  __syncthreads();
  const u32 n = *d_nrec;
  const u32 lane = lane_id();
  const u32 wave_base = (blockIdx.x * blockDim.x + threadIdx.x) & ~63u;
  if (wave_base < n) {
    const u32 i = wave_base + lane;
    const bool in = i < n;
    const u32 key = in ? rec_key[i] : kInvalid;
    const u32 prev = (in && i > 0) ? rec_key[i - 1] : ~key;
    const bool valid = in && key != kInvalid;
    const bool boundary = !in || (key != prev) || i == 0;
    const bool head = valid && boundary;
    // ---- 1. per-record evaluation --------------------------------------------------------------------
    VoxelRef vr{nullptr, 0, 0, 0, false};
    float sdf = 0.0f, uw = 0.0f;
    u32 color = 0;
    if (valid) {
      vr = locate_voxel(L, touched_slots, key);
      const u32 r = rec_ray[i];
      const F3 pg{R.px[r], R.py[r], R.pz[r]};
      sdf = compute_sdf(P, pg, vr.gx, vr.gy, vr.gz);
      uw = update_weight(P, sdf, R.w[r]);
      color = R.color[r];
    }
    // ---- segment geometry -----------------------------------------------------------------------------
    const u64 bmask = __ballot(boundary);
    const u64 later = (lane == 63) ? 0ull : (bmask >> (lane + 1));
    u32 len = 0;
    bool is_long = false;
    if (head) {
      if (later) {
        len = static_cast<u32>(__ffsll(static_cast<long long>(later)));
      } else {
        const u32 nxt = wave_base + 64;
        if (nxt >= n || rec_key[nxt] != key)
          len = min(nxt, n) - i;
        else
          is_long = true;
      }
    }
    // ---- 2. short segments: head lanes replay their records in order ----------------------------------
    const bool run_short = head && !is_long && vr.ok;
    Voxel v{0.0f, 0.0f, 0u};
    if (run_short) {
      v.d = __uint_as_float(vr.ptr[0]);
      v.w = __uint_as_float(vr.ptr[1]);
      v.c = vr.ptr[2];
    }
    u32 max_len = run_short ? len : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) max_len = max(max_len, static_cast<u32>(__shfl_xor(static_cast<int>(max_len), off, 64)));
    for (u32 k = 0; k < max_len; ++k) {
      const int src = static_cast<int>((lane + k) & 63u);
      const float s_k = __shfl(sdf, src, 64);
      const float u_k = __shfl(uw, src, 64);
      const u32 c_k = static_cast<u32>(__shfl(static_cast<int>(color), src, 64));
      if (run_short && k < len) update_voxel(P, v, s_k, u_k, c_k);
    }
    if (run_short) {
      vr.ptr[0] = __float_as_uint(v.d);
      vr.ptr[1] = __float_as_uint(v.w);
      vr.ptr[2] = v.c;
    }
    // ---- 3. the long segment of this wave, if any ------------------------------------------------------
    const u64 lmask = __ballot(is_long && vr.ok);
    if (lmask) {
      const u32 ll = static_cast<u32>(__ffsll(static_cast<long long>(lmask))) - 1u;
      const u32 start = wave_base + ll;
      const u32 lkey = static_cast<u32>(__builtin_amdgcn_readlane(key, ll));
      const int gx = __builtin_amdgcn_readlane(vr.gx, ll), gy = __builtin_amdgcn_readlane(vr.gy, ll), gz = __builtin_amdgcn_readlane(vr.gz, ll);
      const u64 pbits = reinterpret_cast<u64>(vr.ptr);
      u32* vptr = reinterpret_cast<u32*>(static_cast<u64>(static_cast<u32>(__builtin_amdgcn_readlane(static_cast<u32>(pbits), ll))) |
                                         (static_cast<u64>(static_cast<u32>(__builtin_amdgcn_readlane(static_cast<u32>(pbits >> 32), ll))) << 32));
      Voxel lv{__uint_as_float(vptr[0]), __uint_as_float(vptr[1]), vptr[2]};
      for (u32 base = start;; base += 64) {
        const u32 j = base + lane;
        const bool inl = (j < n) && (rec_key[j] == lkey);
        float s2 = 0.0f, u2 = 0.0f;
        u32 c2 = 0;
        bool sat = false;
        if (inl) {
          const u32 r = rec_ray[j];
          const F3 pg{R.px[r], R.py[r], R.pz[r]};
          s2 = compute_sdf(P, pg, gx, gy, gz);
          u2 = update_weight(P, s2, R.w[r]);
          c2 = R.color[r];
          sat = saturating_update(P, s2, u2);
        }
        const u64 in_mask = __ballot(inl);
        const u32 cnt_in = static_cast<u32>(__popcll(in_mask));  // contiguous: lanes [0, cnt_in)
        if (cnt_in == 0) break;
        const bool all_sat = __ballot(inl && sat) == in_mask;
        if (all_sat && lv.d == P.trunc) {
          // distance provably stays at +trunc; only the weight moves
          const bool unit = __ballot(inl && u2 == 1.0f) == in_mask;
          if (lv.w >= P.max_weight) {
            // min(max_weight, w + u) == max_weight for every u > 0
          } else if (unit && lv.w == truncf(lv.w) && lv.w + static_cast<float>(cnt_in) < 16777216.0f) {
            lv.w = std_min(P.max_weight, lv.w + static_cast<float>(cnt_in));  // integers add exactly
          } else {
            for (u32 k = 0; k < cnt_in; ++k) {
              const float nw = lv.w + readlane_f32(u2, k);
              if (!(nw < kEps)) lv.w = std_min(P.max_weight, nw);
            }
          }
        } else {
          for (u32 k = 0; k < cnt_in; ++k) update_voxel(P, lv, readlane_f32(s2, k), readlane_f32(u2, k), static_cast<u32>(__builtin_amdgcn_readlane(c2, k)));
        }
        if (cnt_in < 64) break;
      }
      if (lane == 0) {
        vptr[0] = __float_as_uint(lv.d);
        vptr[1] = __float_as_uint(lv.w);
        vptr[2] = lv.c;
      }
    }
    // ---- frame statistics ------------------------------------------------------------------------------
    const u64 vmask = __ballot(valid), hmask = __ballot(head);
    if (lane == 0) {
      atomicAdd(&blk_updates, static_cast<u32>(__popcll(vmask)));
      atomicAdd(&blk_voxels, static_cast<u32>(__popcll(hmask)));
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (blk_updates) atomicAdd(&cnt->n_updates, blk_updates);
    if (blk_voxels) atomicAdd(&cnt->n_short, blk_voxels);  // n_short + n_long = touched voxels
  }
}

// ---- depth front end ----------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_depth_flags(const float* __restrict__ depth, u32 n, u32* __restrict__ flag) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float d = depth[i];
  flag[i] = (isfinite(d) && d > 0.0f) ? 1u : 0u;
}
__global__ void __launch_bounds__(256) k_depth_points(const float* __restrict__ depth, const uint8_t* __restrict__ rgba, int w, int h, float fx, float fy,
                                                      float cx, float cy, const u32* __restrict__ pos, float* __restrict__ xyz,
                                                      uint8_t* __restrict__ rgba_out) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  const u32 n = static_cast<u32>(w) * static_cast<u32>(h);
  if (i >= n) return;
  const float d = depth[i];
  if (!(isfinite(d) && d > 0.0f)) return;
  const u32 o = pos[i];
  const u32 u = i % static_cast<u32>(w), v = i / static_cast<u32>(w);
  const float xn = (static_cast<float>(u) - cx) / fx;
  const float yn = (static_cast<float>(v) - cy) / fy;
  xyz[3 * o] = d * xn;
  xyz[3 * o + 1] = d * yn;
  xyz[3 * o + 2] = d;
  if (rgba_out) reinterpret_cast<u32*>(rgba_out)[o] = rgba ? reinterpret_cast<const u32*>(rgba)[i] : 0u;
}

// =================================================================================================
// host side of the integrator
// =================================================================================================
struct cox_integrator {
  cox_layer* layer = nullptr;
  cox_tsdf_config cfg;
  int method = 0;
  hipStream_t stream = nullptr;
  // point-sized workspace
  u32 pcap = 0;
  RayArrays rays{};
  u32 *pslot = nullptr, *skey[2] = {nullptr, nullptr}, *sval[2] = {nullptr, nullptr}, *head = nullptr, *bstart = nullptr;
  u64* fh_keys = nullptr;
  u32* fh_first = nullptr;
  u32 fh_cap = 0;
  float* own_xyz = nullptr;  // staging for host / depth inputs
  uint8_t* own_rgba = nullptr;
  u32* depth_flag = nullptr;
  // record-sized workspace
  u32 rcap = 0;
  u32 *rec_key[2] = {nullptr, nullptr}, *rec_ray[2] = {nullptr, nullptr};
  u32* touched_slots = nullptr;  // [layer capacity]
  SortWorkspace sortws;
  u32 sort_counts_cap = 0, scan_sums_cap = 0;
  Counters* d_cnt = nullptr;
  Counters* h_cnt = nullptr;  // pinned
  cox_frame_stats last{};
  bool final_counts_pending = false;  // the frame issued its end-of-frame counter copy
  u32 last_err = 0;
  // timing of the apply kernels (bench roofline)
  bool profiling = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> apply_events;
  double apply_ms = 0.0;
  uint64_t apply_launches = 0;
};

template <typename T>
static int dev_realloc(T** p, size_t count) {
  if (*p) (void)hipFree(*p);
  *p = nullptr;
  if (count == 0) return COX_OK;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T));
  if (e != hipSuccess) return (e == hipErrorOutOfMemory) ? COX_ERR_OUT_OF_MEMORY : COX_ERR_NO_DEVICE;
  return COX_OK;
}
#define COX_TRY(expr)          \
  do {                         \
    int st_ = (expr);          \
    if (st_ != COX_OK) return st_; \
  } while (0)

static int ensure_sort_ws(cox_integrator* I, u32 n_max) {
  const u32 nb = sort_num_blocks(n_max);
  const u32 need_counts = 256u * nb;
  if (need_counts > I->sort_counts_cap) {
    COX_TRY(dev_realloc(&I->sortws.counts, need_counts));
    I->sort_counts_cap = need_counts;
  }
  const u32 need_sums = std::max(scan_num_blocks(need_counts), scan_num_blocks(n_max)) + 1;
  if (need_sums > I->scan_sums_cap) {
    COX_TRY(dev_realloc(&I->sortws.scan.block_sums, need_sums));
    I->scan_sums_cap = need_sums;
  }
  return COX_OK;
}

static int ensure_point_capacity(cox_integrator* I, u32 n) {
  if (n <= I->pcap) return COX_OK;
  COX_HIP(hipStreamSynchronize(I->stream));
  const u32 cap = std::max<u32>(n, 1024);
  COX_TRY(dev_realloc(&I->rays.px, cap));
  COX_TRY(dev_realloc(&I->rays.py, cap));
  COX_TRY(dev_realloc(&I->rays.pz, cap));
  COX_TRY(dev_realloc(&I->rays.w, cap));
  COX_TRY(dev_realloc(&I->rays.color, cap));
  COX_TRY(dev_realloc(&I->rays.flags, cap));
  COX_TRY(dev_realloc(&I->rays.key, cap));
  COX_TRY(dev_realloc(&I->rays.nsteps, cap));
  COX_TRY(dev_realloc(&I->rays.rec_off, cap));
  COX_TRY(dev_realloc(&I->pslot, cap));
  for (int k = 0; k < 2; ++k) {
    COX_TRY(dev_realloc(&I->skey[k], cap));
    COX_TRY(dev_realloc(&I->sval[k], cap));
  }
  COX_TRY(dev_realloc(&I->head, cap));
  COX_TRY(dev_realloc(&I->bstart, cap));
  COX_TRY(dev_realloc(&I->own_xyz, static_cast<size_t>(cap) * 3));
  COX_TRY(dev_realloc(&I->own_rgba, static_cast<size_t>(cap) * 4));
  COX_TRY(dev_realloc(&I->depth_flag, cap));
  I->fh_cap = next_pow2(2ull * cap);
  COX_TRY(dev_realloc(&I->fh_keys, I->fh_cap));
  COX_TRY(dev_realloc(&I->fh_first, I->fh_cap));
  I->pcap = cap;
  return ensure_sort_ws(I, std::max(I->pcap, I->rcap));
}

static int ensure_record_capacity(cox_integrator* I, u64 n) {
  if (n <= I->rcap) return COX_OK;
  if (n > 0xFFFFFFF0ull) return COX_ERR_OUT_OF_MEMORY;
  COX_HIP(hipStreamSynchronize(I->stream));
  const u32 cap = static_cast<u32>(std::min<u64>(0xFFFFFFF0ull, n + n / 4 + 4096));
  for (int k = 0; k < 2; ++k) {
    COX_TRY(dev_realloc(&I->rec_key[k], cap));
    COX_TRY(dev_realloc(&I->rec_ray[k], cap));
  }
  I->rcap = cap;
  return ensure_sort_ws(I, std::max(I->pcap, I->rcap));
}

static FrameParams make_params(const cox_integrator* I, const float T[7], u32 n, int freespace) {
  const cox_tsdf_config& c = I->cfg;
  FrameParams P;
  P.qw = T[0];
  P.qx = T[1];
  P.qy = T[2];
  P.qz = T[3];
  P.tx = T[4];
  P.ty = T[5];
  P.tz = T[6];
  P.voxel_size = I->layer->voxel_size;
  P.voxel_size_inv = I->layer->voxel_size_inv;
  P.trunc = c.default_truncation_distance;
  P.max_weight = c.max_weight;
  P.min_ray = c.min_ray_length_m;
  P.max_ray = c.max_ray_length_m;
  P.sparsity_factor = c.sparsity_compensation_factor;
  P.start_subsampling_inv = c.start_voxel_subsampling_factor * I->layer->voxel_size_inv;
  P.n_points = n;
  P.frame_id = 0;
  P.use_const_weight = c.use_const_weight;
  P.allow_clear = c.allow_clear;
  P.carving = c.voxel_carving_enabled;
  P.use_dropoff = c.use_weight_dropoff;
  P.use_sparsity = c.use_sparsity_compensation_factor;
  P.anti_grazing = (I->method == COX_METHOD_MERGED) ? c.enable_anti_grazing : 0;
  P.freespace = freespace;
  P.cast_from_origin = 1;
  return P;
}

static inline dim3 grid_for(u32 n, u32 block = 256) { return dim3(std::max<u32>(1, (n + block - 1) / block)); }

// the whole frame; xyz / rgba are device pointers
static int integrate_device(cox_integrator* I, const float T[7], const float* xyz, const uint8_t* rgba, u32 n, int freespace) {
  cox_layer* Lh = I->layer;
  hipStream_t s = I->stream;
  COX_TRY(ensure_point_capacity(I, n));
  FrameParams P = make_params(I, T, n, freespace);
  P.frame_id = ++Lh->frame_id;
  LayerView L{Lh->voxels, Lh->ht_keys, Lh->ht_vals, Lh->ht_stamp, Lh->ht_ord, Lh->block_keys, Lh->d_nblocks, Lh->ht_cap - 1, static_cast<u32>(Lh->capacity)};
  RayArrays R = I->rays;
  COX_HIP(hipMemsetAsync(I->d_cnt, 0, sizeof(Counters), s));
  I->last = cox_frame_stats{};
  I->last.n_points = n;
  I->final_counts_pending = false;
  if (n == 0) return COX_OK;

  u32 n_rays_max = n;
  const u32 fh_mask = I->fh_cap - 1;
  if (I->method == COX_METHOD_MERGED) {
    const u32 np2 = next_pow2(static_cast<u64>(n) + 1);
    COX_HIP(hipMemsetAsync(I->fh_keys, 0xFF, sizeof(u64) * I->fh_cap, s));
    COX_HIP(hipMemsetAsync(I->fh_first, 0xFF, sizeof(u32) * I->fh_cap, s));
    hipLaunchKernelGGL(k_bundle_insert, grid_for(n), dim3(256), 0, s, P, xyz, I->fh_keys, I->fh_first, fh_mask, I->pslot, I->d_cnt);
    hipLaunchKernelGGL(k_bundle_keys, grid_for(n), dim3(256), 0, s, n, np2, I->fh_keys, I->fh_first, I->pslot, I->skey[0], I->sval[0]);
    const int kbits = ceil_log2(np2) + 1;  // + clearing bit; kInvalid's low bits exceed every valid key
    const int cur = radix_sort_pairs(I->skey[0], I->sval[0], I->skey[1], I->sval[1], nullptr, n, kbits, I->sortws, s);
    const u32* sk = I->skey[cur];
    const u32* sv = I->sval[cur];
    hipLaunchKernelGGL(k_bundle_heads, grid_for(n), dim3(256), 0, s, n, sk, I->head);
    // bundle ordinal of every head = exclusive scan of the head flags; total = number of bundles (rays)
    exclusive_scan_u32(I->head, I->head, nullptr, n, &I->d_cnt->n_rays, I->sortws.scan, s);
    hipLaunchKernelGGL(k_bundle_starts, grid_for(n), dim3(256), 0, s, n, sk, I->head, I->bstart);
    hipLaunchKernelGGL(k_bundle_merge, dim3(std::min<u32>(2048, std::max<u32>(1, (n + 255) / 256))), dim3(256), 0, s, P, np2, xyz, rgba, sk, sv, I->bstart, R, I->d_cnt);
  } else {
    hipLaunchKernelGGL(k_rays_simple, grid_for(n), dim3(256), 0, s, P, xyz, rgba, R, I->d_cnt);
  }
  // record offsets
  exclusive_scan_u32(R.nsteps, R.rec_off, nullptr, n_rays_max, &I->d_cnt->n_records, I->sortws.scan, s);
  // allocate + stamp blocks
  hipLaunchKernelGGL(k_touch, grid_for(n_rays_max), dim3(256), 0, s, P, R, n_rays_max, L, I->touched_slots, I->d_cnt, Lh->d_err, I->fh_keys, fh_mask);
  // the one host round trip of the frame: record count + touched-block count size the sort
  COX_HIP(hipMemcpyAsync(I->h_cnt, I->d_cnt, sizeof(Counters), hipMemcpyDeviceToHost, s));
  COX_HIP(hipStreamSynchronize(s));
  const u32 n_rec = I->h_cnt->n_records;
  const u32 n_touched = I->h_cnt->n_touched;
  I->last.n_valid = I->h_cnt->n_valid;
  I->last.n_rays = I->h_cnt->n_rays;
  I->last.n_touched_blocks = n_touched;
  I->last.n_new_blocks = I->h_cnt->n_new_blocks;
  I->last_err |= I->h_cnt->err;
  if (n_rec == 0) return COX_OK;
  COX_TRY(ensure_record_capacity(I, n_rec));
  hipLaunchKernelGGL(k_emit, grid_for(n_rays_max), dim3(256), 0, s, P, R, n_rays_max, L, I->rec_key[0], I->rec_ray[0], I->rcap, I->d_cnt, I->fh_keys,
                     fh_mask);
  const int vbits = 12 + ceil_log2(static_cast<u64>(n_touched) + 1);
  const int cur = radix_sort_pairs(I->rec_key[0], I->rec_ray[0], I->rec_key[1], I->rec_ray[1], &I->d_cnt->n_records, n_rec, vbits, I->sortws, s);
  const u32* rk = I->rec_key[cur];
  const u32* rr = I->rec_ray[cur];
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (I->profiling) {
    COX_HIP(hipEventCreate(&e0));
    COX_HIP(hipEventCreate(&e1));
    COX_HIP(hipEventRecord(e0, s));
  }
  hipLaunchKernelGGL(k_apply, grid_for(n_rec), dim3(256), 0, s, P, R, L, I->touched_slots, rk, rr, &I->d_cnt->n_records, I->d_cnt);
  if (I->profiling) {
    COX_HIP(hipEventRecord(e1, s));
    I->apply_events.emplace_back(e0, e1);
  }
  COX_HIP(hipMemcpyAsync(I->h_cnt, I->d_cnt, sizeof(Counters), hipMemcpyDeviceToHost, s));
  I->final_counts_pending = true;
  COX_HIP(hipGetLastError());
  return COX_OK;
}

static int integrator_finish_frame(cox_integrator* I) {
  COX_HIP(hipStreamSynchronize(I->stream));
  if (I->final_counts_pending) {
    I->last.n_updates = I->h_cnt->n_updates;
    I->last.n_touched_voxels = static_cast<uint64_t>(I->h_cnt->n_short) + I->h_cnt->n_long;
    I->last_err |= I->h_cnt->err;
  }
  u32 lerr = 0;
  COX_HIP(hipMemcpy(&lerr, I->layer->d_err, sizeof(u32), hipMemcpyDeviceToHost));
  I->last_err |= lerr;
  for (auto& ev : I->apply_events) {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) {
      I->apply_ms += ms;
      I->apply_launches += 1;
    }
    (void)hipEventDestroy(ev.first);
    (void)hipEventDestroy(ev.second);
  }
  I->apply_events.clear();
  const u32 e = I->last_err;
  I->last_err = 0;
  return err_bits_to_status(e);
}

extern "C" {

int cox_integrator_create(cox_layer_t* layer, const cox_tsdf_config* cfg, int method, cox_integrator_t** out) {
  COX_ENTRY();
  if (!layer || !cfg || !out) return COX_ERR_INVALID_ARG;
  if (method != COX_METHOD_SIMPLE && method != COX_METHOD_MERGED) return (method == COX_METHOD_FAST) ? COX_ERR_UNSUPPORTED : COX_ERR_INVALID_ARG;
  if (cfg->integration_order_mode != 0) return COX_ERR_UNSUPPORTED;
  if (!(cfg->default_truncation_distance > 0.0f) || !(cfg->max_weight > 0.0f)) return COX_ERR_INVALID_ARG;
  COX_HIP(hipSetDevice(layer->device));
  cox_integrator* I = new (std::nothrow) cox_integrator();
  if (!I) return COX_ERR_OUT_OF_MEMORY;
  I->layer = layer;
  I->cfg = *cfg;
  I->method = method;
  int st = COX_OK;
  if (hipStreamCreateWithFlags(&I->stream, hipStreamNonBlocking) != hipSuccess) st = COX_ERR_NO_DEVICE;
  if (st == COX_OK && hipMalloc(reinterpret_cast<void**>(&I->d_cnt), sizeof(Counters)) != hipSuccess) st = COX_ERR_OUT_OF_MEMORY;
  if (st == COX_OK && hipHostMalloc(reinterpret_cast<void**>(&I->h_cnt), sizeof(Counters), hipHostMallocDefault) != hipSuccess) st = COX_ERR_OUT_OF_MEMORY;
  if (st == COX_OK) st = dev_realloc(&I->touched_slots, layer->ht_cap);  // one entry per block key the table can hold
  if (st == COX_OK) {
    memset(I->h_cnt, 0, sizeof(Counters));
    st = ensure_point_capacity(I, 640 * 480);
  }
  if (st != COX_OK) {
    cox_integrator_destroy(I);
    return st;
  }
  *out = I;
  return COX_OK;
}

void cox_integrator_destroy(cox_integrator_t* I) {
  if (!I) return;
  (void)hipSetDevice(I->layer->device);
  if (I->stream) (void)hipStreamSynchronize(I->stream);
  for (auto& ev : I->apply_events) {
    (void)hipEventDestroy(ev.first);
    (void)hipEventDestroy(ev.second);
  }
  void* ptrs[] = {I->rays.px, I->rays.py, I->rays.pz, I->rays.w, I->rays.color, I->rays.flags, I->rays.key, I->rays.nsteps, I->rays.rec_off, I->pslot,
                  I->skey[0], I->skey[1], I->sval[0], I->sval[1], I->head, I->bstart, I->fh_keys, I->fh_first, I->own_xyz, I->own_rgba, I->depth_flag,
                  I->rec_key[0], I->rec_key[1], I->rec_ray[0], I->rec_ray[1], I->touched_slots,
                  I->sortws.counts, I->sortws.scan.block_sums, I->d_cnt};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (I->h_cnt) (void)hipHostFree(I->h_cnt);
  if (I->stream) (void)hipStreamDestroy(I->stream);
  delete I;
}

int cox_integrate_points_dev(cox_integrator_t* I, const float T_G_C[7], const float* xyz_dev, const uint8_t* rgba_dev, uint64_t n, int freespace) {
  COX_ENTRY();
  if (!I || !T_G_C || (n && !xyz_dev) || n > 0x7FFFFFFFull) return COX_ERR_INVALID_ARG;
  COX_HIP(hipSetDevice(I->layer->device));
  return integrate_device(I, T_G_C, xyz_dev, rgba_dev, static_cast<u32>(n), freespace);
}

int cox_integrate_points(cox_integrator_t* I, const float T_G_C[7], const float* xyz, const uint8_t* rgba, uint64_t n, int freespace) {
  COX_ENTRY();
  if (!I || !T_G_C || (n && !xyz) || n > 0x7FFFFFFFull) return COX_ERR_INVALID_ARG;
  COX_HIP(hipSetDevice(I->layer->device));
  COX_TRY(ensure_point_capacity(I, static_cast<u32>(n)));
  if (n) {
    COX_HIP(hipMemcpyAsync(I->own_xyz, xyz, sizeof(float) * 3 * n, hipMemcpyHostToDevice, I->stream));
    if (rgba) COX_HIP(hipMemcpyAsync(I->own_rgba, rgba, 4 * n, hipMemcpyHostToDevice, I->stream));
  }
  COX_TRY(integrate_device(I, T_G_C, I->own_xyz, rgba ? I->own_rgba : nullptr, static_cast<u32>(n), freespace));
  return integrator_finish_frame(I);
}

int cox_integrate_depth_dev(cox_integrator_t* I, const float T_G_C[7], const float* depth_dev, const uint8_t* rgba_dev, int w, int h, const float K[4]) {
  COX_ENTRY();
  if (!I || !T_G_C || !depth_dev || !K || w <= 0 || h <= 0 || static_cast<uint64_t>(w) * h > 0x7FFFFFFFull) return COX_ERR_INVALID_ARG;
  COX_HIP(hipSetDevice(I->layer->device));
  const u32 n = static_cast<u32>(w) * static_cast<u32>(h);
  COX_TRY(ensure_point_capacity(I, n));
  hipStream_t s = I->stream;
  hipLaunchKernelGGL(k_depth_flags, grid_for(n), dim3(256), 0, s, depth_dev, n, I->depth_flag);
  exclusive_scan_u32(I->depth_flag, I->depth_flag, nullptr, n, &I->d_cnt->n_depth_points, I->sortws.scan, s);
  hipLaunchKernelGGL(k_depth_points, grid_for(n), dim3(256), 0, s, depth_dev, rgba_dev, w, h, K[0], K[1], K[2], K[3], I->depth_flag, I->own_xyz,
                     I->own_rgba);
  u32 n_pts = 0;
  COX_HIP(hipMemcpyAsync(&I->h_cnt->n_depth_points, &I->d_cnt->n_depth_points, sizeof(u32), hipMemcpyDeviceToHost, s));
  COX_HIP(hipStreamSynchronize(s));
  n_pts = I->h_cnt->n_depth_points;
  return integrate_device(I, T_G_C, I->own_xyz, I->own_rgba, n_pts, 0);
}

int cox_integrator_sync(cox_integrator_t* I) {
  COX_ENTRY();
  if (!I) return COX_ERR_INVALID_ARG;
  COX_HIP(hipSetDevice(I->layer->device));
  return integrator_finish_frame(I);
}

int cox_integrator_last_stats(cox_integrator_t* I, cox_frame_stats* stats) {
  COX_ENTRY();
  if (!I || !stats) return COX_ERR_INVALID_ARG;
  COX_HIP(hipSetDevice(I->layer->device));
  COX_HIP(hipStreamSynchronize(I->stream));
  if (I->final_counts_pending) {
    I->last.n_updates = I->h_cnt->n_updates;
    I->last.n_touched_voxels = static_cast<uint64_t>(I->h_cnt->n_short) + I->h_cnt->n_long;
  }
  *stats = I->last;
  return COX_OK;
}

int cox_integrator_set_profiling(cox_integrator_t* I, int on) {
  COX_ENTRY();
  if (!I) return COX_ERR_INVALID_ARG;
  I->profiling = on != 0;
  return COX_OK;
}

int cox_integrator_kernel_time(cox_integrator_t* I, double* apply_ms, uint64_t* apply_launches, int reset) {
  COX_ENTRY();
  if (!I) return COX_ERR_INVALID_ARG;
  COX_HIP(hipSetDevice(I->layer->device));
  int st = integrator_finish_frame(I);
  if (apply_ms) *apply_ms = I->apply_ms;
  if (apply_launches) *apply_launches = I->apply_launches;
  if (reset) {
    I->apply_ms = 0.0;
    I->apply_launches = 0;
  }
  return st;
}

}  // extern "C"
