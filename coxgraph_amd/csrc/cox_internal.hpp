// Private host-side definitions shared by the translation units of libcoxgraph_hip.so.
#pragma once
#include <cstdio>
#include <cstdlib>

#include "../../include/coxgraph_hip.h"
#include "cox_device.hpp"

using cox::u32;
using cox::u64;

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

static inline int err_bits_to_status(u32 bits) {
  if (bits & (kErrPool | kErrTable)) return COX_ERR_POOL_EXHAUSTED;
  if (bits & kErrRange) return COX_ERR_INDEX_RANGE;
  if (bits & kErrRecords) return COX_ERR_INTERNAL;
  return COX_OK;
}

static inline u32 next_pow2(u64 v) {
  u32 p = 1;
  while (p < v) p <<= 1;
  return p;
}
static inline int ceil_log2(u64 v) {
  int b = 0;
  while ((1ull << b) < v) ++b;
  return b;
}


struct cox_layer {
  int device = 0;
  float voxel_size = 0, voxel_size_inv = 0, block_size = 0, block_size_inv = 0;
  u64 capacity = 0;        // pool blocks
  u32 ht_cap = 0;          // hash slots (power of two)
  u32* voxels = nullptr;   // [capacity][4096][3] wire words
  u64* ht_keys = nullptr;  // [ht_cap] packed block index, kEmptyKey when free
  u32* ht_vals = nullptr;  // [ht_cap] pool index
  u32* ht_stamp = nullptr; // [ht_cap] last frame id that touched the block
  u32* ht_ord = nullptr;   // [ht_cap] dense ordinal of the block within that frame
  u64* block_keys = nullptr;  // [capacity] key of pool block i
  u32* d_nblocks = nullptr;   // device counter
  u32* d_err = nullptr;       // sticky device error bits
  u32 frame_id = 0;           // shared by every integrator on this layer
  // growth (voxblox's Layer grows without bound): the last kernel of every frame leaves the block count in this pinned
  // word, so the host can see a pool filling up without a sync and double it before the next frame (cox_layer_reserve)
  u32* h_nblocks = nullptr;
  u32 generation = 0;         // bumped whenever the buffers above are reallocated (integrators re-read them)
  bool auto_grow = true;
  // recorded by an integrator after the last kernel of every frame it enqueues; readers on other streams
  // (cox_reg_*, clones) wait for it instead of for the whole device
  hipEvent_t last_write = nullptr;
  bool has_write = false;
  // Several integrators may drive one layer (a merged one for the live stream, a fast one for recover mode ...), each on streams of
  // its own: a frame whose integrator is not the one that wrote last orders its first layer-touching stage behind last_write
  // (cox_layer_order_writer), so the frames of different integrators reach the layer in call order, as they do in voxblox, and the
  // last recorded event stands for every frame before it.
  const void* last_writer = nullptr;
};

// make stream s wait for every frame enqueued so far on the layer (no-op when nothing was enqueued)
static inline void cox_layer_wait_writes(const cox_layer* L, hipStream_t s) {
  if (L->has_write && L->last_write) (void)hipStreamWaitEvent(s, L->last_write, 0);
}
int cox_internal_layer_reserve(cox_layer* L, u64 capacity_blocks);
void cox_drain_submitters();
// Called by an integrator's entry point before it enqueues a frame: true when another integrator wrote the layer last -- its
// submission thread has then enqueued (and recorded last_write behind) everything it was handed, and the caller makes the
// stream of its first layer-touching stage wait for last_write.
static inline bool cox_layer_order_writer(cox_layer* L, const void* writer) {
  const bool foreign = L->last_writer != nullptr && L->last_writer != writer && L->has_write;
  if (foreign) cox_drain_submitters();
  L->last_writer = writer;
  return foreign;
}

// voxgraph registration point set of a submap, resident on one GPU
struct cox_regpoints {
  int device = 0;
  float* pts = nullptr;  // n * {x, y, z, distance, weight}
  u64 n = 0;
  u64* cum = nullptr;    // inclusive prefix sums of the fixed-point weights (built on first use by the weighted sampler)
  u64 cum_total = 0;
};

// hipGetLastError() is a per-thread sticky slot shared with every other HIP user in the process
// (PyTorch probes peers / devices during its lazy init and may leave a benign error behind).  Every
// entry point clears it first so that the check after our own launches only sees our own errors.
extern "C" bool cox_internal_hip_touched;  // cox_layer.hip: the library has reached the HIP runtime (cox_runtime_prepare comes too late after that)
static inline void cox_clear_stale_hip_error(const char* where) {
  cox_internal_hip_touched = true;
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    static const bool verbose = std::getenv("COX_DEBUG") != nullptr;
    if (verbose) fprintf(stderr, "[coxgraph_hip] %s: cleared stale HIP error left by an earlier call: %s\n", where, hipGetErrorString(e));
  }
}
// Integrators hand the second half of every frame (touch .. apply) to a submission thread of their own (cox_integrator.hip):
// a frame that cox_integrate_* has returned from may not be fully ENQUEUED yet.  Every other entry point therefore first
// waits until all such threads have caught up -- then stream / event / device synchronisation sees all the work, as before.
void cox_drain_submitters();
#define COX_ENTRY()                         \
  do {                                      \
    cox_clear_stale_hip_error(__func__);    \
    cox_drain_submitters();                 \
  } while (0)
#define COX_ENTRY_NO_DRAIN() cox_clear_stale_hip_error(__func__)

// the layer an integrator writes to (cox_integrator is private to cox_integrator.hip)
cox_layer* cox_internal_integrator_layer(cox_integrator_t* integ);

// voxblox ProjectiveTsdfIntegrator (cox_projective.hip), reached through the integrator handle when method == COX_METHOD_PROJECTIVE
struct cox_projective;
int cox_proj_create(cox_layer* layer, const cox_tsdf_config* cfg, cox_projective** out);
void cox_proj_destroy(cox_projective* P);
int cox_proj_integrate(cox_projective* P, const float T[7], const float* xyz_dev, uint64_t n, int deintegrate);
int cox_proj_integrate_host(cox_projective* P, const float T[7], const float* xyz, uint64_t n, int deintegrate);
int cox_proj_sync(cox_projective* P);
int cox_proj_last_stats(cox_projective* P, cox_frame_stats* out);
