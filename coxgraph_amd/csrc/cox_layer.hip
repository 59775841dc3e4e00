// MI355X (gfx950) TSDF layer storage behind include/coxgraph_hip.h (the integrator is in cox_integrator.hip).
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
#include <cstdlib>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/coxgraph_hip.h"
#include "cox_device.hpp"
#include "cox_internal.hpp"

using namespace cox;

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
  if (L->h_nblocks) *L->h_nblocks = 0;
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
    case COX_ERR_COMM: return "COX_ERR_COMM";
  }
  return "COX_ERR_?";
}

// Four stage streams per integrator: the host asks the runtime for more hardware queues than its default of 4 BEFORE the runtime
// starts (stages that share a queue run back to back: 6.1 k instead of 7.5 k frames/s at 5 cm).  Explicit, not a side effect of
// loading the library (a static constructor calling setenv races with getenv in the other threads of a host such as a ROS node
// that dlopens the engine, and silently changes the queue count of every other HIP user in the process).
bool cox_internal_hip_touched = false;  // set by the first entry point of this library that reaches the HIP runtime
int cox_runtime_prepare(void) {
  if (cox_internal_hip_touched) return 0;
  if (std::getenv("GPU_MAX_HW_QUEUES")) return 0;  // the host's own choice stands
  return setenv("GPU_MAX_HW_QUEUES", "16", 0) == 0 ? 1 : 0;
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
  c->sensor_horizontal_resolution = 0;
  c->sensor_vertical_resolution = 0;
  c->sensor_vertical_field_of_view_degrees = 0.0f;
  c->projective_interpolation_scheme = 3;
  c->projective_adaptive_gap_m = 0.5f;
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
  if (st == COX_OK && hipHostMalloc(reinterpret_cast<void**>(&L->h_nblocks), sizeof(u32), hipHostMallocDefault) != hipSuccess) st = COX_ERR_OUT_OF_MEMORY;
  if (st == COX_OK && hipEventCreateWithFlags(&L->last_write, hipEventDisableTiming) != hipSuccess) st = COX_ERR_NO_DEVICE;
  L->auto_grow = std::getenv("COX_NO_GROW") == nullptr;
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
  if (L->h_nblocks) (void)hipHostFree(L->h_nblocks);
  if (L->last_write) (void)hipEventDestroy(L->last_write);
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
      atomicSub(d_nblocks, 1u);  // the counter settles at the capacity; the key stays without storage (ht_vals == kInvalid)
      atomicOr(d_err, kErrPool);
      return;
    }
    ht_vals[slot] = pool;
    block_keys[pool] = key;
    pool_of[i] = pool | 0x80000000u;  // bit 31: fresh (a message never repeats a block)
  } else {
    pool_of[i] = ht_vals[slot];  // block existed before this launch
    if (pool_of[i] == kInvalid) atomicOr(d_err, kErrPool);  // a key an earlier, exhausted call left without storage: still an error
  }
}
// one workgroup per uploaded block; action 0/2: overwrite, 1: mergeVoxelAIntoVoxelB
__global__ void __launch_bounds__(256) k_upload_copy(u32* __restrict__ voxels, const u32* __restrict__ src, const u32* __restrict__ src_index,
                                                     const u32* __restrict__ pool_of, int action) {
  const u32 b = blockIdx.x;
  const u32 p = pool_of[b];
  if (p == kInvalid) return;
  const u32 pool = p & 0x7FFFFFFFu;
  u32* dst = voxels + static_cast<size_t>(pool) * kVoxelsPerBlock * kWordsPerVoxel;
  const u32* s = src + static_cast<size_t>(src_index ? src_index[b] : b) * kVoxelsPerBlock * kWordsPerVoxel;
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

// shared tail of the uploads: n blocks whose indices and wire words already sit on the layer's GPU
static int upload_from_device(cox_layer* L, const int32_t* d_idx, const u32* d_src, u32 n, int action) {
  // voxblox's Layer grows without bound: make room first (a message never repeats a block, so nb + n is an upper bound)
  u32 nb0, err0;
  int st = layer_read_counters(L, &nb0, &err0);
  if (st != COX_OK) return st;
  if (static_cast<u64>(nb0) + n > L->capacity) {
    if (!L->auto_grow) return COX_ERR_POOL_EXHAUSTED;
    st = cox_internal_layer_reserve(L, std::max<u64>(2 * L->capacity, static_cast<u64>(nb0) + n));
    if (st != COX_OK) return st == COX_ERR_OUT_OF_MEMORY ? COX_ERR_POOL_EXHAUSTED : st;
  }
  u32* d_pool = nullptr;
  COX_HIP(hipMalloc(reinterpret_cast<void**>(&d_pool), sizeof(u32) * n));
  hipLaunchKernelGGL(k_upload_insert, dim3((n + 255) / 256), dim3(256), 0, nullptr, L->ht_keys, L->ht_vals, L->ht_cap - 1, L->block_keys,
                     L->d_nblocks, static_cast<u32>(L->capacity), L->d_err, d_idx, n, d_pool);
  hipLaunchKernelGGL(k_upload_copy, dim3(n), dim3(256), 0, nullptr, L->voxels, d_src, static_cast<const u32*>(nullptr), d_pool, action);
  COX_HIP(hipDeviceSynchronize());
  (void)hipFree(d_pool);
  u32 nb, err;
  st = layer_read_counters(L, &nb, &err);
  if (st != COX_OK) return st;
  *L->h_nblocks = nb;
  return err_bits_to_status(err);
}

extern "C" int cox_layer_upload(cox_layer_t* L, const int32_t* block_idx_xyz, const uint32_t* voxels_3u32, uint64_t n_blocks, int action) {
  COX_ENTRY();
  if (!L || action < 0 || action > 2 || (n_blocks && (!block_idx_xyz || !voxels_3u32)) || n_blocks > (1ull << 26)) return COX_ERR_INVALID_ARG;
  COX_HIP(hipSetDevice(L->device));
  if (action == 2) {
    int st = cox_layer_clear(L);
    if (st != COX_OK) return st;
  }
  if (n_blocks == 0) return COX_OK;
  const size_t block_words = static_cast<size_t>(kVoxelsPerBlock) * kWordsPerVoxel;
  int32_t* d_idx = nullptr;
  u32* d_src = nullptr;
  COX_HIP(hipMalloc(reinterpret_cast<void**>(&d_idx), sizeof(int32_t) * 3 * n_blocks));
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_src), sizeof(u32) * block_words * n_blocks);
  if (e != hipSuccess) {
    (void)hipFree(d_idx);
    return COX_ERR_OUT_OF_MEMORY;
  }
  COX_HIP(hipMemcpy(d_idx, block_idx_xyz, sizeof(int32_t) * 3 * n_blocks, hipMemcpyHostToDevice));
  COX_HIP(hipMemcpy(d_src, voxels_3u32, sizeof(u32) * block_words * n_blocks, hipMemcpyHostToDevice));
  const int st = upload_from_device(L, d_idx, d_src, static_cast<u32>(n_blocks), action);
  (void)hipFree(d_idx);
  (void)hipFree(d_src);
  return st;
}

// deserializeMsgToLayer from a message that already sits in HBM (the submap hand-over between GPUs: the wire arrays
// arrive by RCCL all-gather / peer copy and never touch the host)
extern "C" int cox_layer_upload_dev(cox_layer_t* L, const int32_t* block_idx_xyz_dev, const uint32_t* voxels_3u32_dev, uint64_t n_blocks, int action) {
  COX_ENTRY();
  if (!L || action < 0 || action > 2 || (n_blocks && (!block_idx_xyz_dev || !voxels_3u32_dev)) || n_blocks > (1ull << 26)) return COX_ERR_INVALID_ARG;
  COX_HIP(hipSetDevice(L->device));
  COX_HIP(hipDeviceSynchronize());  // the producer of the device buffers (any stream) is done after this
  if (action == 2) {
    int st = cox_layer_clear(L);
    if (st != COX_OK) return st;
  }
  if (n_blocks == 0) return COX_OK;
  return upload_from_device(L, block_idx_xyz_dev, voxels_3u32_dev, static_cast<u32>(n_blocks), action);
}

// ---- growth -------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_rehash(const u64* __restrict__ block_keys, u32 n, u64* __restrict__ ht_keys, u32* __restrict__ ht_vals, u32 ht_mask,
                                                u32* d_err) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bool fresh;
  const u32 slot = ht_insert(ht_keys, ht_mask, block_keys[i], &fresh);
  if (slot == kInvalid) {
    atomicOr(d_err, kErrTable);
    return;
  }
  ht_vals[slot] = i;
}

// Layer::allocateBlockPtrByIndex never fails in voxblox; here the pool and the hash table are reallocated at (at least)
// capacity_blocks, the blocks copied over and the table rebuilt from block_keys.  Keys that an exhausted frame left without
// storage are dropped by the rebuild.  Nothing may be in flight on the layer (the caller has synchronised its streams).
int cox_internal_layer_reserve(cox_layer* L, u64 capacity_blocks) {
  if (capacity_blocks <= L->capacity) return COX_OK;
  if (capacity_blocks > (1ull << 26)) return COX_ERR_INVALID_ARG;
  // every integrator on this layer (not only the one that asks) may have half of a frame waiting in its submission thread: those
  // jobs read the buffers that are about to be freed.  Enqueued they are covered by the device-wide sync below.  
  cox_drain_submitters();
  u32 nb, err;
  int st = layer_read_counters(L, &nb, &err);  // device-wide sync
  if (st != COX_OK) return st;
  const size_t block_bytes = static_cast<size_t>(kVoxelsPerBlock) * kWordsPerVoxel * sizeof(u32);
  const u32 ht_cap = std::max<u32>(1024, next_pow2(2 * capacity_blocks));
  u32 *voxels = nullptr, *ht_vals = nullptr, *ht_stamp = nullptr, *ht_ord = nullptr;
  u64 *ht_keys = nullptr, *block_keys = nullptr;
  bool ok = hipMalloc(reinterpret_cast<void**>(&voxels), capacity_blocks * block_bytes) == hipSuccess;
  ok = ok && hipMalloc(reinterpret_cast<void**>(&ht_keys), sizeof(u64) * ht_cap) == hipSuccess;
  ok = ok && hipMalloc(reinterpret_cast<void**>(&ht_vals), sizeof(u32) * ht_cap) == hipSuccess;
  ok = ok && hipMalloc(reinterpret_cast<void**>(&ht_stamp), sizeof(u32) * ht_cap) == hipSuccess;
  ok = ok && hipMalloc(reinterpret_cast<void**>(&ht_ord), sizeof(u32) * ht_cap) == hipSuccess;
  ok = ok && hipMalloc(reinterpret_cast<void**>(&block_keys), sizeof(u64) * capacity_blocks) == hipSuccess;
  if (!ok) {
    (void)hipGetLastError();
    for (void* q : {static_cast<void*>(voxels), static_cast<void*>(ht_keys), static_cast<void*>(ht_vals), static_cast<void*>(ht_stamp), static_cast<void*>(ht_ord),
                    static_cast<void*>(block_keys)})
      if (q) (void)hipFree(q);
    return COX_ERR_OUT_OF_MEMORY;
  }
  if (nb) COX_HIP(hipMemcpy(voxels, L->voxels, nb * block_bytes, hipMemcpyDeviceToDevice));
  COX_HIP(hipMemset(reinterpret_cast<char*>(voxels) + nb * block_bytes, 0, (capacity_blocks - nb) * block_bytes));
  if (nb) COX_HIP(hipMemcpy(block_keys, L->block_keys, sizeof(u64) * nb, hipMemcpyDeviceToDevice));
  COX_HIP(hipMemset(ht_keys, 0xFF, sizeof(u64) * ht_cap));
  COX_HIP(hipMemset(ht_vals, 0xFF, sizeof(u32) * ht_cap));
  COX_HIP(hipMemset(ht_stamp, 0, sizeof(u32) * ht_cap));
  COX_HIP(hipMemset(ht_ord, 0, sizeof(u32) * ht_cap));
  COX_HIP(hipMemcpy(L->d_nblocks, &nb, sizeof(u32), hipMemcpyHostToDevice));  // also undoes a transient overshoot
  if (nb) hipLaunchKernelGGL(k_rehash, dim3((nb + 255) / 256), dim3(256), 0, nullptr, block_keys, nb, ht_keys, ht_vals, ht_cap - 1, L->d_err);
  COX_HIP(hipDeviceSynchronize());
  for (void* q : {static_cast<void*>(L->voxels), static_cast<void*>(L->ht_keys), static_cast<void*>(L->ht_vals), static_cast<void*>(L->ht_stamp),
                  static_cast<void*>(L->ht_ord), static_cast<void*>(L->block_keys)})
    (void)hipFree(q);
  L->voxels = voxels;
  L->ht_keys = ht_keys;
  L->ht_vals = ht_vals;
  L->ht_stamp = ht_stamp;
  L->ht_ord = ht_ord;
  L->block_keys = block_keys;
  L->capacity = capacity_blocks;
  L->ht_cap = ht_cap;
  L->generation += 1;
  *L->h_nblocks = nb;
  return COX_OK;
}

extern "C" int cox_layer_reserve(cox_layer_t* L, uint64_t capacity_blocks) {
  COX_ENTRY();
  if (!L) return COX_ERR_INVALID_ARG;
  COX_HIP(hipSetDevice(L->device));
  return cox_internal_layer_reserve(L, capacity_blocks);
}
extern "C" int cox_layer_capacity(cox_layer_t* L, uint64_t* capacity_blocks) {
  if (!L || !capacity_blocks) return COX_ERR_INVALID_ARG;
  *capacity_blocks = L->capacity;
  return COX_OK;
}
extern "C" int cox_layer_set_auto_grow(cox_layer_t* L, int on) {
  if (!L) return COX_ERR_INVALID_ARG;
  L->auto_grow = on != 0;
  return COX_OK;
}

// ---- submap hand-over between GPUs -----------------------------------------------
// The reference's server pulls whole submaps from the clients as ROS messages (coxgraph/src/server/client_handler.cpp:82-104,
// coxgraph_server.cpp:253-258).  Here the block array and the block keys go GPU to GPU (hipMemcpyPeer: xGMI between
// the GPUs of a node, a plain device copy on one GPU) and the destination rebuilds its hash table -- device layout ==
// wire layout, so there is nothing to (de)serialise.
extern "C" int cox_layer_clone_to_device(const cox_layer_t* src_, int dst_device, uint64_t capacity_blocks, cox_layer_t** out) {
  COX_ENTRY();
  cox_layer* src = const_cast<cox_layer*>(src_);
  if (!src || !out) return COX_ERR_INVALID_ARG;
  u32 nb, err;
  int st = layer_read_counters(src, &nb, &err);  // waits for everything in flight on the source GPU
  if (st != COX_OK) return st;
  if (err) return err_bits_to_status(err);
  const u64 cap = std::max<u64>(std::max<u64>(capacity_blocks, nb), 64);
  cox_layer* dst = nullptr;
  st = cox_layer_create(src->voxel_size, kVps, dst_device, cap, &dst);
  if (st != COX_OK) return st;
  if (nb) {
    const size_t block_bytes = static_cast<size_t>(kVoxelsPerBlock) * kWordsPerVoxel * sizeof(u32);
    hipError_t e = hipMemcpyPeer(dst->voxels, dst_device, src->voxels, src->device, nb * block_bytes);
    if (e == hipSuccess) e = hipMemcpyPeer(dst->block_keys, dst_device, src->block_keys, src->device, sizeof(u64) * nb);
    if (e == hipSuccess) e = hipSetDevice(dst_device);
    if (e == hipSuccess) e = hipMemcpy(dst->d_nblocks, &nb, sizeof(u32), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      fprintf(stderr, "[coxgraph_hip] cox_layer_clone_to_device: %s\n", hipGetErrorString(e));
      cox_layer_destroy(dst);
      return COX_ERR_NO_DEVICE;
    }
    hipLaunchKernelGGL(k_rehash, dim3((nb + 255) / 256), dim3(256), 0, nullptr, dst->block_keys, nb, dst->ht_keys, dst->ht_vals, dst->ht_cap - 1, dst->d_err);
    if (hipDeviceSynchronize() != hipSuccess) {
      cox_layer_destroy(dst);
      return COX_ERR_NO_DEVICE;
    }
    *dst->h_nblocks = nb;
  }
  (void)hipSetDevice(src->device);
  *out = dst;
  return COX_OK;
}

// one workgroup per block: pool block order[i] -> wire position i
__global__ void __launch_bounds__(256) k_export_blocks(const u32* __restrict__ voxels, const u64* __restrict__ block_keys, const u32* __restrict__ order,
                                                       int32_t* __restrict__ idx_out, u32* __restrict__ vox_out) {
  const u32 i = blockIdx.x;
  const u32 pool = order[i];
  const u32* s = voxels + static_cast<size_t>(pool) * kVoxelsPerBlock * kWordsPerVoxel;
  u32* d = vox_out + static_cast<size_t>(i) * kVoxelsPerBlock * kWordsPerVoxel;
  for (u32 k = threadIdx.x; k < kVoxelsPerBlock * kWordsPerVoxel; k += 256) d[k] = s[k];
  if (threadIdx.x == 0) {
    int x, y, z;
    unpack_key(block_keys[pool], &x, &y, &z);
    idx_out[3 * i] = x;
    idx_out[3 * i + 1] = y;
    idx_out[3 * i + 2] = z;
  }
}
// serializeLayerAsMsg into DEVICE buffers (same (z,y,x) block order as cox_layer_download): what a rank hands to the
// all-gather of the submap exchange
extern "C" int cox_layer_export_dev(cox_layer_t* L, int32_t* block_idx_xyz_dev, uint32_t* voxels_3u32_dev, uint64_t cap_blocks, uint64_t* n_blocks) {
  COX_ENTRY();
  if (!L) return COX_ERR_INVALID_ARG;
  u32 nb, err;
  int st = layer_read_counters(L, &nb, &err);
  if (st != COX_OK) return st;
  if (n_blocks) *n_blocks = nb;
  if (cap_blocks == 0 && !block_idx_xyz_dev && !voxels_3u32_dev) return err_bits_to_status(err);
  if (cap_blocks < nb) return COX_ERR_BUFFER_TOO_SMALL;
  if (nb == 0) return err_bits_to_status(err);
  if (!block_idx_xyz_dev || !voxels_3u32_dev) return COX_ERR_INVALID_ARG;
  std::vector<u64> keys(nb);
  COX_HIP(hipMemcpy(keys.data(), L->block_keys, sizeof(u64) * nb, hipMemcpyDeviceToHost));
  std::vector<u32> order(nb);
  for (u32 i = 0; i < nb; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](u32 a, u32 b) { return keys[a] < keys[b]; });
  u32* d_order = nullptr;
  COX_HIP(hipMalloc(reinterpret_cast<void**>(&d_order), sizeof(u32) * nb));
  COX_HIP(hipMemcpy(d_order, order.data(), sizeof(u32) * nb, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_export_blocks, dim3(nb), dim3(256), 0, nullptr, L->voxels, L->block_keys, d_order, block_idx_xyz_dev, voxels_3u32_dev);
  COX_HIP(hipDeviceSynchronize());
  (void)hipFree(d_order);
  return err_bits_to_status(err);
}

// =================================================================================================
// registration points of a finished submap ("voxels" / implicit_to_implicit set)
// =================================================================================================
// VoxgraphSubmap::finishSubmap() (called when a submap arrives, coxgraph/include/coxgraph/utils/msg_converter.h:113)
// -> findRelevantVoxelIndices: every voxel with weight > min_voxel_weight and |distance| < max_voxel_distance becomes a
// RegistrationPoint{position = voxel centre (Block::computeCoordinatesFromLinearIndex), distance, weight}.
// Order here: blocks by (z, y, x), voxels by linear index -- deterministic (upstream iterates a hash map).
__device__ __forceinline__ bool relevant_voxel(const u32* __restrict__ vox, float min_w, float max_d) {
  const float d = __uint_as_float(vox[0]), w = __uint_as_float(vox[1]);
  return w > min_w && fabsf(d) < max_d;
}
__global__ void __launch_bounds__(256) k_relevant_count(const u32* __restrict__ voxels, float min_w, float max_d, u32* __restrict__ counts) {
  __shared__ u32 total;
  if (threadIdx.x == 0) total = 0;
  __syncthreads();
  const u32* blk = voxels + static_cast<size_t>(blockIdx.x) * kVoxelsPerBlock * kWordsPerVoxel;
  u32 c = 0;
  for (u32 v = threadIdx.x; v < kVoxelsPerBlock; v += 256) c += relevant_voxel(blk + 3 * v, min_w, max_d) ? 1u : 0u;
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(&total, c);
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = total;
}
// one workgroup per block, voxels in linear order: 16 rounds of 256 with an in-order ballot/prefix compaction
__global__ void __launch_bounds__(256) k_relevant_write(const u32* __restrict__ voxels, const u64* __restrict__ block_keys, const u64* __restrict__ offsets,
                                                        float min_w, float max_d, float voxel_size, float block_size, float* __restrict__ out) {
  __shared__ u32 wave_cnt[4];
  __shared__ u32 base;
  const u32 pool = blockIdx.x;
  const u32* blk = voxels + static_cast<size_t>(pool) * kVoxelsPerBlock * kWordsPerVoxel;
  int bx, by, bz;
  unpack_key(block_keys[pool], &bx, &by, &bz);
  const float ox = static_cast<float>(bx) * block_size, oy = static_cast<float>(by) * block_size, oz = static_cast<float>(bz) * block_size;
  const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) base = 0;
  __syncthreads();
  float* dst = out + offsets[pool] * 5ull;
  for (u32 round = 0; round < kVoxelsPerBlock / 256; ++round) {
    const u32 v = round * 256 + threadIdx.x;
    const bool keep = relevant_voxel(blk + 3 * v, min_w, max_d);
    const u64 m = __ballot(keep);
    if (lane == 0) wave_cnt[wave] = static_cast<u32>(__popcll(m));
    __syncthreads();
    u32 pos = base + static_cast<u32>(__popcll(m & ((1ull << lane) - 1ull)));
    for (u32 w = 0; w < wave; ++w) pos += wave_cnt[w];
    if (keep) {
      const int lx = static_cast<int>(v & 15u), ly = static_cast<int>((v >> 4) & 15u), lz = static_cast<int>(v >> 8);
      float* p = dst + static_cast<size_t>(pos) * 5;
      p[0] = ox + center_coord(lx, voxel_size);
      p[1] = oy + center_coord(ly, voxel_size);
      p[2] = oz + center_coord(lz, voxel_size);
      p[3] = __uint_as_float(blk[3 * v]);
      p[4] = __uint_as_float(blk[3 * v + 1]);
    }
    __syncthreads();
    if (threadIdx.x == 0) base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
}

// builds the point list on the device; *d_out (n*5 floats) is hipMalloc'ed for the caller
static int relevant_points_device(cox_layer* L, float min_w, float max_d, float** d_out, u64* n_out) {
  *d_out = nullptr;
  *n_out = 0;
  u32 nb, err;
  int st = layer_read_counters(L, &nb, &err);
  if (st != COX_OK) return st;
  if (nb == 0) return COX_OK;
  u32* d_counts = nullptr;
  u64* d_off = nullptr;
  COX_HIP(hipMalloc(reinterpret_cast<void**>(&d_counts), sizeof(u32) * nb));
  COX_HIP(hipMalloc(reinterpret_cast<void**>(&d_off), sizeof(u64) * nb));
  hipLaunchKernelGGL(k_relevant_count, dim3(nb), dim3(256), 0, nullptr, L->voxels, min_w, max_d, d_counts);
  std::vector<u32> counts(nb);
  std::vector<u64> keys(nb), off(nb);
  COX_HIP(hipMemcpy(counts.data(), d_counts, sizeof(u32) * nb, hipMemcpyDeviceToHost));
  COX_HIP(hipMemcpy(keys.data(), L->block_keys, sizeof(u64) * nb, hipMemcpyDeviceToHost));
  std::vector<u32> order(nb);
  for (u32 i = 0; i < nb; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](u32 a, u32 b) { return keys[a] < keys[b]; });  // packed key orders (z, y, x)
  u64 run = 0;
  for (u32 i = 0; i < nb; ++i) {
    off[order[i]] = run;
    run += counts[order[i]];
  }
  if (run) {
    COX_HIP(hipMemcpy(d_off, off.data(), sizeof(u64) * nb, hipMemcpyHostToDevice));
    float* d_pts = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_pts), sizeof(float) * 5 * run);
    if (e != hipSuccess) {
      (void)hipFree(d_counts);
      (void)hipFree(d_off);
      return COX_ERR_OUT_OF_MEMORY;
    }
    hipLaunchKernelGGL(k_relevant_write, dim3(nb), dim3(256), 0, nullptr, L->voxels, L->block_keys, d_off, min_w, max_d, L->voxel_size, L->block_size, d_pts);
    COX_HIP(hipDeviceSynchronize());
    *d_out = d_pts;
  }
  *n_out = run;
  (void)hipFree(d_counts);
  (void)hipFree(d_off);
  return COX_OK;
}

extern "C" int cox_layer_registration_points(cox_layer_t* L, float min_voxel_weight, float max_voxel_distance, float* out, uint64_t cap, uint64_t* n) {
  COX_ENTRY();
  if (!L || !n) return COX_ERR_INVALID_ARG;
  float* d = nullptr;
  u64 cnt = 0;
  int st = relevant_points_device(L, min_voxel_weight, max_voxel_distance, &d, &cnt);
  if (st != COX_OK) return st;
  *n = cnt;
  if (out && cnt) {
    if (cap < cnt) {
      (void)hipFree(d);
      return COX_ERR_BUFFER_TOO_SMALL;
    }
    COX_HIP(hipMemcpy(out, d, sizeof(float) * 5 * cnt, hipMemcpyDeviceToHost));
  }
  if (d) (void)hipFree(d);
  return COX_OK;
}

extern "C" int cox_regpoints_from_layer(cox_layer_t* L, float min_voxel_weight, float max_voxel_distance, cox_regpoints_t** out) {
  COX_ENTRY();
  if (!L || !out) return COX_ERR_INVALID_ARG;
  float* d = nullptr;
  u64 cnt = 0;
  int st = relevant_points_device(L, min_voxel_weight, max_voxel_distance, &d, &cnt);
  if (st != COX_OK) return st;
  cox_regpoints* R = new (std::nothrow) cox_regpoints();
  if (!R) {
    if (d) (void)hipFree(d);
    return COX_ERR_OUT_OF_MEMORY;
  }
  R->device = L->device;
  R->pts = d;
  R->n = cnt;
  *out = R;
  return COX_OK;
}

// =================================================================================================
// mergeLayerAintoLayerB(A, [T_B_A,] B)   (voxblox merge_integration.h)
// =================================================================================================
// coxgraph call sites: the client's combined map (coxgraph/src/client/map_server.cpp:59-73) and the server's
// mergeToCliMap (coxgraph/src/server/submap_collection.cpp:24-37).  With a transform, layer A is first resampled onto
// B's grid (transformLayer: candidate output blocks from the transformed block centres, every output voxel centre taken
// back into A and interpolated trilinearly, nearest voxel if that fails, blocks without data dropped), then merged
// voxel by voxel (mergeVoxelAIntoVoxelB).
struct LayerConstView {
  const u32* voxels;
  const u64* ht_keys;
  const u32* ht_vals;
  u32 ht_mask;
  float voxel_size, voxel_size_inv, block_size, block_size_inv;
};
struct RigidParams {
  float qw, qx, qy, qz, tx, ty, tz;
};
__device__ __forceinline__ F3 rigid_apply(const RigidParams& T, F3 v) {
  const F3 qv{T.qx, T.qy, T.qz};
  F3 uv = cross3(qv, v);
  uv = uv + uv;
  const F3 c = cross3(qv, uv);
  return F3{((v.x + T.qw * uv.x) + c.x) + T.tx, ((v.y + T.qw * uv.y) + c.y) + T.ty, ((v.z + T.qw * uv.z) + c.z) + T.tz};
}
__constant__ float c_merge_interp_table[8][8] = {{1, 0, 0, 0, 0, 0, 0, 0},   {-1, 0, 0, 0, 1, 0, 0, 0},   {-1, 0, 1, 0, 0, 0, 0, 0},
                                                 {-1, 1, 0, 0, 0, 0, 0, 0},  {1, 0, -1, 0, -1, 0, 1, 0},  {1, -1, -1, 1, 0, 0, 0, 0},
                                                 {1, -1, 0, 0, -1, 1, 0, 0}, {-1, 1, 1, -1, 1, -1, -1, 1}};
__device__ __forceinline__ float interp_member(const float q[8], const float data[8]) {
  float md[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c) s += c_merge_interp_table[r][c] * data[c];
    md[r] = s;
  }
  float v = 0.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i) v += q[i] * md[i];
  return v;
}
__device__ __forceinline__ const u32* voxel_words(const LayerConstView& L, u32 pool, int vx, int vy, int vz) {
  return L.voxels + (static_cast<size_t>(pool) * kVoxelsPerBlock + static_cast<u32>(vx + 16 * (vy + 16 * vz))) * kWordsPerVoxel;
}
// Interpolator::getVoxel(pos, &voxel, true) || getVoxel(pos, &voxel, false); returns false when neither is possible
__device__ __forceinline__ bool resample_voxel(const LayerConstView& L, const float pos[3], u32 out[3]) {
  float sc[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) sc[k] = pos[k] * L.block_size_inv;
  if (!(index_in_range(sc[0]) && index_in_range(sc[1]) && index_in_range(sc[2]))) return false;
  int b0[3], v0[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) b0[k] = grid_index(sc[k]);
  const u32 slot0 = ht_find(L.ht_keys, L.ht_mask, pack_key(b0[0], b0[1], b0[2]));
  if (slot0 == kInvalid) return false;
  const u32 pool0 = L.ht_vals[slot0];
  if (pool0 == kInvalid) return false;
  int b[3], vi[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float origin = static_cast<float>(b0[k]) * L.block_size;
    int v = grid_index((pos[k] - origin) * L.voxel_size_inv);
    v = v > 15 ? 15 : (v < 0 ? 0 : v);
    v0[k] = v;
    b[k] = b0[k];
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
  // ---- trilinear ----
  bool ok = true;
  u32 base_pool = pool0;
  if (b[0] != b0[0] || b[1] != b0[1] || b[2] != b0[2]) {
    const u32 sl = ht_find(L.ht_keys, L.ht_mask, pack_key(b[0], b[1], b[2]));
    base_pool = (sl == kInvalid) ? kInvalid : L.ht_vals[sl];
    ok = base_pool != kInvalid;
  }
  float d[8], w[8], ch[4][8];
  if (ok) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (!ok) break;
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
        const u32 sl = ht_find(L.ht_keys, L.ht_mask, pack_key(nb[0], nb[1], nb[2]));
        pool = (sl == kInvalid) ? kInvalid : L.ht_vals[sl];
        if (pool == kInvalid) {
          ok = false;
          break;
        }
      }
      const u32* vw = voxel_words(L, pool, v[0], v[1], v[2]);
      const float wi = __uint_as_float(vw[1]);
      if (!(wi > 0.0f)) {
        ok = false;
        break;
      }
      d[i] = __uint_as_float(vw[0]);
      w[i] = wi;
      const u32 c = vw[2];
      ch[0][i] = static_cast<float>(static_cast<int>((c >> 24) & 255u));  // r
      ch[1][i] = static_cast<float>(static_cast<int>((c >> 16) & 255u));  // g
      ch[2][i] = static_cast<float>(static_cast<int>((c >> 8) & 255u));   // b
      ch[3][i] = static_cast<float>(static_cast<int>(c & 255u));          // a
    }
  }
  if (ok) {
    float off[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float c0 = static_cast<float>(b[k]) * L.block_size + center_coord(vi[k], L.voxel_size);
      off[k] = (pos[k] - c0) * L.voxel_size_inv;
    }
    const float dx = off[0], dy = off[1], dz = off[2];
    const float q[8] = {1.0f, dx, dy, dz, dx * dy, dy * dz, dz * dx, dx * dy * dz};
    out[0] = __float_as_uint(interp_member(q, d));
    out[1] = __float_as_uint(interp_member(q, w));
    const u32 r = static_cast<u32>(static_cast<int>(interp_member(q, ch[0]))) & 255u, g = static_cast<u32>(static_cast<int>(interp_member(q, ch[1]))) & 255u;
    const u32 bl = static_cast<u32>(static_cast<int>(interp_member(q, ch[2]))) & 255u, a = static_cast<u32>(static_cast<int>(interp_member(q, ch[3]))) & 255u;
    out[2] = a | (bl << 8) | (g << 16) | (r << 24);
    return true;
  }
  // ---- nearest voxel of the block that contains pos ----
  const u32* vw = voxel_words(L, pool0, v0[0], v0[1], v0[2]);
  out[0] = vw[0];
  out[1] = vw[1];
  out[2] = vw[2];
  return true;
}
// one workgroup per candidate output block
__global__ void __launch_bounds__(256) k_resample_blocks(LayerConstView A, RigidParams T_in_out, const int32_t* __restrict__ cand_idx, float voxel_size_out,
                                                         float block_size_out, u32* __restrict__ tmp, u32* __restrict__ has_data) {
  const u32 c = blockIdx.x;
  const float ox = static_cast<float>(cand_idx[3 * c]) * block_size_out, oy = static_cast<float>(cand_idx[3 * c + 1]) * block_size_out,
              oz = static_cast<float>(cand_idx[3 * c + 2]) * block_size_out;
  u32* dst = tmp + static_cast<size_t>(c) * kVoxelsPerBlock * kWordsPerVoxel;
  bool any = false;
  for (u32 v = threadIdx.x; v < kVoxelsPerBlock; v += 256) {
    const int lx = static_cast<int>(v & 15u), ly = static_cast<int>((v >> 4) & 15u), lz = static_cast<int>(v >> 8);
    const F3 centre{ox + center_coord(lx, voxel_size_out), oy + center_coord(ly, voxel_size_out), oz + center_coord(lz, voxel_size_out)};
    const F3 p = rigid_apply(T_in_out, centre);
    const float pos[3] = {p.x, p.y, p.z};
    u32 w[3];
    if (resample_voxel(A, pos, w)) {
      dst[3 * v] = w[0];
      dst[3 * v + 1] = w[1];
      dst[3 * v + 2] = w[2];
      any = true;
    }
  }
  if (__ballot(any) && (threadIdx.x & 63u) == 0) atomicOr(&has_data[c], 1u);
}

// merge n device-resident blocks (indices d_idx, words src[src_index[i]]) into L, voxel by voxel
static int merge_device_blocks(cox_layer* L, const int32_t* d_idx, const u32* d_src, const u32* d_src_index, u32 n) {
  if (n == 0) return COX_OK;
  {
    u32 nb0, err0;
    int st0 = layer_read_counters(L, &nb0, &err0);
    if (st0 != COX_OK) return st0;
    if (static_cast<u64>(nb0) + n > L->capacity && L->auto_grow) {
      st0 = cox_internal_layer_reserve(L, std::max<u64>(2 * L->capacity, static_cast<u64>(nb0) + n));
      if (st0 != COX_OK && st0 != COX_ERR_OUT_OF_MEMORY) return st0;
    }
  }
  u32* d_pool = nullptr;
  COX_HIP(hipMalloc(reinterpret_cast<void**>(&d_pool), sizeof(u32) * n));
  hipLaunchKernelGGL(k_upload_insert, dim3((n + 255) / 256), dim3(256), 0, nullptr, L->ht_keys, L->ht_vals, L->ht_cap - 1, L->block_keys, L->d_nblocks,
                     static_cast<u32>(L->capacity), L->d_err, d_idx, n, d_pool);
  hipLaunchKernelGGL(k_upload_copy, dim3(n), dim3(256), 0, nullptr, L->voxels, d_src, d_src_index, d_pool, 1);
  COX_HIP(hipDeviceSynchronize());
  (void)hipFree(d_pool);
  return COX_OK;
}

static void host_rotate(const float q[4], const float v[3], float out[3]) {  // Eigen _transformVector, same op order as the kernels
  const float qv[3] = {q[1], q[2], q[3]};
  float uv[3] = {qv[1] * v[2] - qv[2] * v[1], qv[2] * v[0] - qv[0] * v[2], qv[0] * v[1] - qv[1] * v[0]};
  for (int k = 0; k < 3; ++k) uv[k] = uv[k] + uv[k];
  const float c[3] = {qv[1] * uv[2] - qv[2] * uv[1], qv[2] * uv[0] - qv[0] * uv[2], qv[0] * uv[1] - qv[1] * uv[0]};
  for (int k = 0; k < 3; ++k) out[k] = (v[k] + q[0] * uv[k]) + c[k];
}
static float host_center_coord(int idx, float size) { return static_cast<float>((static_cast<float>(idx) + 0.5) * size); }

extern "C" int cox_layer_merge(const cox_layer_t* A_, const float T_B_A[7], cox_layer_t* B) {
  COX_ENTRY();
  cox_layer* A = const_cast<cox_layer*>(A_);
  if (!A || !B || A == B || A->device != B->device) return COX_ERR_INVALID_ARG;
  u32 na, ea;
  int st = layer_read_counters(A, &na, &ea);
  if (st != COX_OK) return st;
  if (na == 0) return COX_OK;
  std::vector<u64> keys(na);
  COX_HIP(hipMemcpy(keys.data(), A->block_keys, sizeof(u64) * na, hipMemcpyDeviceToHost));
  if (!T_B_A) {
    // same grid: block i of A's pool goes to the block with the same index in B
    if (A->voxel_size != B->voxel_size) return COX_ERR_INVALID_ARG;
    std::vector<int32_t> idx(3 * static_cast<size_t>(na));
    for (u32 i = 0; i < na; ++i) unpack_key(keys[i], &idx[3 * i], &idx[3 * i + 1], &idx[3 * i + 2]);
    int32_t* d_idx = nullptr;
    COX_HIP(hipMalloc(reinterpret_cast<void**>(&d_idx), sizeof(int32_t) * idx.size()));
    COX_HIP(hipMemcpy(d_idx, idx.data(), sizeof(int32_t) * idx.size(), hipMemcpyHostToDevice));
    st = merge_device_blocks(B, d_idx, A->voxels, nullptr, na);
    (void)hipFree(d_idx);
  } else {
    // candidate output blocks (transformLayer): input blocks in (z,y,x) order, a float-stepped box around each transformed centre
    std::sort(keys.begin(), keys.end());
    const float q[4] = {T_B_A[0], T_B_A[1], T_B_A[2], T_B_A[3]};
    const float kDiag = static_cast<float>(1.7320508075688772);
    const float inv_out = 1.0f / B->block_size;
    const float offset = static_cast<float>(kDiag * A->block_size * 0.5);
    std::vector<int32_t> cand;
    std::vector<u64> seen;  // sorted-unique check through a hash set would do; candidate counts are small (a few thousand)
    {
      std::vector<u64> set_keys;
      for (u32 i = 0; i < na; ++i) {
        int bx, by, bz;
        unpack_key(keys[i], &bx, &by, &bz);
        const float c_in[3] = {host_center_coord(bx, A->block_size), host_center_coord(by, A->block_size), host_center_coord(bz, A->block_size)};
        float r[3];
        host_rotate(q, c_in, r);
        const float c_out[3] = {r[0] + T_B_A[4], r[1] + T_B_A[5], r[2] + T_B_A[6]};
        for (float x = c_out[0] - offset; x < c_out[0] + offset; x += B->block_size)
          for (float y = c_out[1] - offset; y < c_out[1] + offset; y += B->block_size)
            for (float z = c_out[2] - offset; z < c_out[2] + offset; z += B->block_size) {
              const int ix = static_cast<int>(std::floor(x * inv_out + 1e-6f)), iy = static_cast<int>(std::floor(y * inv_out + 1e-6f)),
                        iz = static_cast<int>(std::floor(z * inv_out + 1e-6f));
              if (ix < -kIdxBias + 1 || ix >= kIdxBias - 1 || iy < -kIdxBias + 1 || iy >= kIdxBias - 1 || iz < -kIdxBias + 1 || iz >= kIdxBias - 1)
                return COX_ERR_INDEX_RANGE;
              const u64 k = static_cast<u64>(static_cast<u32>(ix + kIdxBias)) | (static_cast<u64>(static_cast<u32>(iy + kIdxBias)) << 21) |
                            (static_cast<u64>(static_cast<u32>(iz + kIdxBias)) << 42);
              set_keys.push_back(k);
            }
      }
      std::sort(set_keys.begin(), set_keys.end());
      set_keys.erase(std::unique(set_keys.begin(), set_keys.end()), set_keys.end());
      cand.resize(3 * set_keys.size());
      for (size_t i = 0; i < set_keys.size(); ++i) unpack_key(set_keys[i], &cand[3 * i], &cand[3 * i + 1], &cand[3 * i + 2]);
    }
    const u32 nc = static_cast<u32>(cand.size() / 3);
    if (nc == 0) return COX_OK;
    // inverse transform (minkindr: conjugate rotation, -R^T t)
    const float qi[4] = {q[0], -q[1], -q[2], -q[3]};
    const float tvec[3] = {T_B_A[4], T_B_A[5], T_B_A[6]};
    float rt[3];
    host_rotate(qi, tvec, rt);
    const RigidParams Tinv{qi[0], qi[1], qi[2], qi[3], -rt[0], -rt[1], -rt[2]};
    const size_t block_words = static_cast<size_t>(kVoxelsPerBlock) * kWordsPerVoxel;
    int32_t* d_cand = nullptr;
    u32 *d_tmp = nullptr, *d_has = nullptr;
    COX_HIP(hipMalloc(reinterpret_cast<void**>(&d_cand), sizeof(int32_t) * cand.size()));
    COX_HIP(hipMalloc(reinterpret_cast<void**>(&d_has), sizeof(u32) * nc));
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_tmp), sizeof(u32) * block_words * nc);
    if (e != hipSuccess) {
      (void)hipFree(d_cand);
      (void)hipFree(d_has);
      return COX_ERR_OUT_OF_MEMORY;
    }
    COX_HIP(hipMemcpy(d_cand, cand.data(), sizeof(int32_t) * cand.size(), hipMemcpyHostToDevice));
    COX_HIP(hipMemset(d_has, 0, sizeof(u32) * nc));
    COX_HIP(hipMemset(d_tmp, 0, sizeof(u32) * block_words * nc));
    const LayerConstView AV{A->voxels, A->ht_keys, A->ht_vals, A->ht_cap - 1, A->voxel_size, A->voxel_size_inv, A->block_size, A->block_size_inv};
    hipLaunchKernelGGL(k_resample_blocks, dim3(nc), dim3(256), 0, nullptr, AV, Tinv, d_cand, B->voxel_size, B->block_size, d_tmp, d_has);
    std::vector<u32> has(nc);
    COX_HIP(hipMemcpy(has.data(), d_has, sizeof(u32) * nc, hipMemcpyDeviceToHost));
    std::vector<int32_t> keep_idx;
    std::vector<u32> keep_src;
    for (u32 i = 0; i < nc; ++i)
      if (has[i]) {
        keep_idx.insert(keep_idx.end(), {cand[3 * i], cand[3 * i + 1], cand[3 * i + 2]});
        keep_src.push_back(i);
      }
    const u32 nk = static_cast<u32>(keep_src.size());
    if (nk) {
      int32_t* d_idx = nullptr;
      u32* d_srcidx = nullptr;
      COX_HIP(hipMalloc(reinterpret_cast<void**>(&d_idx), sizeof(int32_t) * keep_idx.size()));
      COX_HIP(hipMalloc(reinterpret_cast<void**>(&d_srcidx), sizeof(u32) * nk));
      COX_HIP(hipMemcpy(d_idx, keep_idx.data(), sizeof(int32_t) * keep_idx.size(), hipMemcpyHostToDevice));
      COX_HIP(hipMemcpy(d_srcidx, keep_src.data(), sizeof(u32) * nk, hipMemcpyHostToDevice));
      st = merge_device_blocks(B, d_idx, d_tmp, d_srcidx, nk);
      (void)hipFree(d_idx);
      (void)hipFree(d_srcidx);
    }
    (void)hipFree(d_cand);
    (void)hipFree(d_has);
    (void)hipFree(d_tmp);
  }
  if (st != COX_OK) return st;
  u32 nb, err;
  st = layer_read_counters(B, &nb, &err);
  if (st != COX_OK) return st;
  *B->h_nblocks = nb;
  return err_bits_to_status(err);
}
