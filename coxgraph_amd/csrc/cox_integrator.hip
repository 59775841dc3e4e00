// MI355X (gfx950) TSDF integrator behind include/coxgraph_hip.h.
//
// Replaces the work of voxblox's TsdfIntegratorBase::integratePointCloud as coxgraph calls it
// (coxgraph/include/coxgraph/map_comm/tsdf_recover.h:75).  Design: DESIGN.md.
//
// One frame = ~30 kernel launches on the integrator's stream and NO host round trip: every count a
// later kernel needs (rays, records, touched blocks, sort key width) stays in device memory, the host
// only supplies grid-size hints taken from earlier frames.
//
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
#include <cstring>
#include <new>
#include <string>
#include <vector>
#include <atomic>
#include <limits>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>

#include "cox_internal.hpp"
#include "cox_sort.hpp"

using namespace cox;

// =================================================================================================
// device side
// =================================================================================================
// per-frame control words of the observed-set solve of the fast integrator (cox_fast.hpp; part of Counters, zeroed with it at frame start)
constexpr int kFastMaxRounds = 8;
struct FastCtl {
  u32 n_visits[kFastMaxRounds];   // candidate visits of round 0 (cap0 per ray) / of a later round (sum of the grown lists; 0 = the round does not run)
  u32 settled[kFastMaxRounds];    // [round]: the relaxation reached a pass that moved nothing
  u32 want_more[kFastMaxRounds];  // [round]: at that fixed point some ray is at the end of a list shorter than its walk, unstopped
  u32 passes[kFastMaxRounds];     // [round]: passes the relaxation took
  u32 ticks_work[kFastMaxRounds], ticks_wait[kFastMaxRounds];  // [round]: 100 MHz ticks workgroup 0 spent working / waiting at the barrier
  u32 grew[kFastMaxRounds];       // [round >= 1]: the round runs (k_fast_grow gave some ray its whole walk)
  u32 scan_n[kFastMaxRounds];     // [round >= 1]: rays the cap scan of the round covers (0 = the round does not run)
  u32 n_long;        // rays with a list longer than cap1 (one wave each), over all rounds
  u32 overflow;      // lists that do not fit their buffers
  u32 sequential;    // the sequential kernel produced this frame's result (k_fast_sequential)
  u32 pad[5];
  struct Bar {
    u32 arrived, pad0[15], epoch, pad1[15], moved[3], want[3], abort, pad2[9];
  } bar[kFastMaxRounds];
};
struct Counters {  // per-frame device counters, zeroed at frame start
  u32 n_valid;      // points that passed isPointValid
  u32 n_rays;       // rays cast (simple: valid points, merged: bundles)
  u32 n_ray_slots;  // ray ids in use, [0, n_ray_slots) (simple: n_points, merged: bundles)
  u32 n_records;    // sum of ray step counts
  u32 n_touched;    // blocks touched this frame
  u32 n_voxels;     // distinct voxels updated
  u32 n_updates;    // (ray, voxel) updates
  u32 n_long;       // voxels whose record run spans more than one wave
  u32 n_new_blocks;
  u32 err;
  u32 n_depth_points;
  u32 n_sorted_valid;  // valid points as seen in the sorted bundling keys (merged)
  u32 n_piece_slots;   // piece path: sum of the rays' piece bounds = slots of the piece arrays in use
  u32 n_expanded;      // piece partition: records written by k_piece_expand (what k_apply_block reads)
  u32 n_big_tiles, n_big_chunks;  // tiles whose phase 1 is split over the chip (k_big_tiles), and their chunks
  u32 n_block_tiles;              // tiles too large for k_apply_wave: k_apply_block's list
  FastCtl fast;
  // One word takes ~88 atomics/us on this chip, so counters that every wave or workgroup of a large grid adds to
  // are sharded over 64 cache lines (index = workgroup or wave id & 63) and summed by the host.
  // [s][0] valid points, [s][1] updates, [s][2] voxels, [s][3] long runs, [s][4] rays
  u32 shard[64][16];
};
enum : u32 { kShValid = 0, kShUpdates = 1, kShVoxels = 2, kShLong = 3, kShRays = 4, kShMaxBundle = 5, kShMaxRun = 6 };  // 5, 6: maxima, not sums

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
  float *px, *py, *pz, *w;  // point_G and (merged) weight of each ray
  u32* color;               // wire-packed colour
  u32* flags;               // bit0 valid, bit1 clearing
  u64* key;                 // terminal voxel key (anti-grazing)
  u32* nsteps;              // records this ray emits
  u32* rec_off;             // exclusive scan of nsteps
  float* q;                 // merged: 8 words per ray in one 32-B line: point_G - origin (x, y, z), its length, the ray's weight, its colour, 2 unused
  u32* pbound;              // piece path: upper bound of the ray's piece count (piece_bound)
  u32* piece_off;           // exclusive scan of pbound
};

// both ping-pong buffers of the record sort + where the result ended up
struct RecordView {
  const u32* key[2];
  const u32* ray[2];
  const SortInfo* info;
  const u32* d_n;
};

__device__ __forceinline__ u32 pack_rgba_wire(const uint8_t* rgba, u32 i) {
  if (!rgba) return 0u;
  const u32 v = reinterpret_cast<const u32*>(rgba)[i];  // little endian: r | g<<8 | b<<16 | a<<24
  return ((v >> 24) & 255u) | (((v >> 16) & 255u) << 8) | (((v >> 8) & 255u) << 16) | ((v & 255u) << 24);
}
// isPointValid.  Non-finite points (which voxblox_ros filters out before the integrator, and on which upstream's
// float -> int64 index casts are undefined) are defined as invalid here and in the oracle.
__device__ __forceinline__ bool point_valid(const FrameParams& P, F3 p, bool* clearing) {
  const float r = sqrtf(dot3(p, p));
  if (!(r <= 3.0e38f)) return false;  // NaN or inf in any coordinate
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
// wave-uniform values must be made visibly scalar (SGPR): hipcc's divergence analysis treats threadIdx.x >> 6 as
// per-lane, which turns every wave-cooperative loop below into a predicated / waterfall loop
__device__ __forceinline__ u32 uniform_u32(u32 v) { return static_cast<u32>(__builtin_amdgcn_readfirstlane(static_cast<int>(v))); }
__device__ __forceinline__ float readlane_f32(float v, u32 lane) { return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), lane)); }

// ---- simple: one ray per point, ray id = mixed-order sequence number --------------------------
__global__ void __launch_bounds__(256) k_rays_simple(const FrameParams* __restrict__ Pp, RayArrays R,
                                                     Counters* cnt) {
  const FrameParams P = *Pp;
  const u32 seq = blockIdx.x * blockDim.x + threadIdx.x;
  if (seq == 0) cnt->n_ray_slots = P.n_points;
  if (seq >= P.n_points) return;
  const u32 idx = mixed_index(seq, P.n_points);
  const F3 p{P.xyz[3 * idx], P.xyz[3 * idx + 1], P.xyz[3 * idx + 2]};
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
    R.color[seq] = pack_rgba_wire(P.rgba, idx);
  }
  R.flags[seq] = flags;
  R.nsteps[seq] = nsteps;
  const u64 m = __ballot(valid);
  if (lane_id() == 0 && m) {
    u32* sh = cnt->shard[(seq >> 6) & 63u];
    atomicAdd(&sh[kShValid], static_cast<u32>(__popcll(m)));
    atomicAdd(&sh[kShRays], static_cast<u32>(__popcll(m)));
  }
}

// ---- merged: bundle points by terminal voxel ---------------------------------------------------
// thread = point (so neighbouring lanes are neighbouring pixels and mostly share a terminal voxel): the lanes of a
// wave that hold the same key elect one leader, which inserts the key in the per-frame hash once and records the
// smallest sequence number of the group as a candidate for the bundle's first visit.
// First kernel of a frame: with by_value the frame's parameter block arrives as a kernel argument and workgroup 0
// stores it for the kernels that follow (saves the 4 us H2D blit per frame); captured stage graphs keep the copy.
// n_dev (by_value only): the frame's point count is still on the device (depth front end): every workgroup takes it from there.
__device__ __forceinline__ u32 pow2_above(u32 n) {  // power of two > n (FrameParams::np2)
  u32 p = 1;
  while (p <= n && p < 0x80000000u) p <<= 1;
  return p;
}
__global__ void __launch_bounds__(256) k_bundle_insert(FrameParams* __restrict__ Pp, FrameParams Pv, int by_value, const u32* __restrict__ n_dev, u64* __restrict__ fh_keys,
                                                       u32* __restrict__ fh_first, u32 fh_mask, u32* __restrict__ pslot, Counters* cnt) {
  // (the count is taken into a scalar of its own: writing it into the by-value parameter struct sends the whole struct through
  // scratch memory -- 42 MB of writes and 11 us per launch, seen in the PMC pass)
  const FrameParams P = by_value ? Pv : *Pp;
  const u32 n_points = (by_value && n_dev) ? min(*n_dev, P.n_points) : P.n_points;
  if (by_value && blockIdx.x == 0 && threadIdx.x == 0) {
    *Pp = Pv;
    if (n_dev) {
      Pp->n_points = n_points;
      Pp->np2 = pow2_above(n_points);
    }
  }
  const u32 idx = blockIdx.x * blockDim.x + threadIdx.x;
  const u32 lane = lane_id();
  bool valid = false;
  u64 key = 0;
  u32 seq = kInvalid;
  if (idx < n_points) {
    seq = mixed_sequence(idx, n_points);
    const F3 p{P.xyz[3 * idx], P.xyz[3 * idx + 1], P.xyz[3 * idx + 2]};
    bool clearing = false;
    valid = point_valid(P, p, &clearing);
    if (valid) {
      const F3 pg = transform_point(P, p);
      const float sx = pg.x * P.voxel_size_inv, sy = pg.y * P.voxel_size_inv, sz = pg.z * P.voxel_size_inv;
      if (!(index_in_range(sx) && index_in_range(sy) && index_in_range(sz))) {  // also catches NaN
        atomicOr(&cnt->err, kErrRange);
        valid = false;
      } else {
        key = pack_key(grid_index(sx), grid_index(sy), grid_index(sz)) | (clearing ? (1ull << 63) : 0ull);
      }
    }
  }
  // group the lanes by key (ALU only), then let all group leaders touch memory at the same time
  u32 my_leader = lane, group_min = kInvalid;
  bool is_leader = false;
  u64 todo = __ballot(valid);
  while (todo) {
    const u32 leader = static_cast<u32>(__ffsll(static_cast<long long>(todo))) - 1u;
    const u64 k = (static_cast<u64>(static_cast<u32>(__builtin_amdgcn_readlane(static_cast<u32>(key >> 32), leader))) << 32) |
                  static_cast<u64>(static_cast<u32>(__builtin_amdgcn_readlane(static_cast<u32>(key), leader)));
    const bool mine = valid && key == k;
    const u64 peers = __ballot(mine);
    u32 mn = mine ? seq : kInvalid;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mn = min(mn, static_cast<u32>(__shfl_xor(static_cast<int>(mn), off, 64)));
    if (mine) my_leader = leader;
    if (lane == leader) {
      is_leader = true;
      group_min = mn;
    }
    todo &= ~peers;
  }
  u32 sl = kInvalid;
  if (is_leader) {
    bool fresh;
    sl = ht_insert(fh_keys, fh_mask, key, &fresh);
    if (sl == kInvalid)
      atomicOr(&cnt->err, kErrTable);
    else if (__hip_atomic_load(&fh_first[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > group_min)
      atomicMin(&fh_first[sl], group_min);  // a stale (larger) value read above only costs this atomic
  }
  const u32 got = static_cast<u32>(__shfl(static_cast<int>(sl), static_cast<int>(my_leader), 64));
  const u32 slot = valid ? got : kInvalid;
  if (idx < n_points) pslot[idx] = slot;  // kInvalid for points that are not integrated; indexed by point (coalesced)
  const u64 m = __ballot(valid && slot != kInvalid);
  if (lane == 0 && m) atomicAdd(&cnt->shard[(idx >> 6) & 63u][kShValid], static_cast<u32>(__popcll(m)));
}
// the same for the frames whose parameter block is uploaded by a copy (simple, fast, captured stage graphs): one thread patches it
__global__ void k_params_count(FrameParams* __restrict__ Pp, const u32* __restrict__ n_dev) {
  const u32 n = min(*n_dev, Pp->n_points);
  Pp->n_points = n;
  Pp->np2 = pow2_above(n);
}
// sort key of a point = (clearing ? np2 : 0) + first sequence number of its bundle; value = seq.  Also publishes the
// key width of the bundling sort.
__global__ void __launch_bounds__(256) k_bundle_keys(const FrameParams* __restrict__ Pp, const u64* __restrict__ fh_keys, const u32* __restrict__ fh_first,
                                                     const u32* __restrict__ pslot, u32* __restrict__ skey, u32* __restrict__ sval, SortInfo* sort_info) {
  const u32 n = Pp->n_points, np2 = Pp->np2;
  const u32 seq = blockIdx.x * blockDim.x + threadIdx.x;
  if (seq == 0) {
    u32 bits = 1;  // clearing bit + log2(np2); kInvalid's low bits exceed every valid key
    while ((1u << (bits - 1)) < np2) ++bits;
    sort_info->nbits = bits;
    sort_info->parity = 0;
    sort_info->base = 0;
  }
  if (seq >= n) return;
  const u32 slot = pslot[mixed_index(seq, n)];  // the bundling sort starts from visiting order: gather on the read side
  u32 k = kInvalid;
  if (slot != kInvalid) k = fh_first[slot] + ((fh_keys[slot] >> 63) ? np2 : 0u);
  skey[seq] = k;
  sval[seq] = seq;
}
// the two ping-pong buffers of the bundling sort + where its result ended up
struct BundleView {
  const u32* key[2];
  const u32* val[2];
  const SortInfo* info;
};
// Bundle boundaries in two launches (round 1 used five: head flags, a three-kernel scan, starts).  A head is a sorted position
// whose key differs from its predecessor's; a bundle's ordinal is the number of heads before it.
//   k_bundle_count   per tile of 2048 positions: number of heads
//   k_bundle_starts  per tile: its base = sum of the counts of the tiles before it (every workgroup adds them up itself --
//                    a frame has ~150 tiles; a ticketed "last workgroup scans" tail took 12 us, this takes none), heads
//                    again, in-tile exclusive scan, bstart[base + rank] = position; the last tile publishes the bundle count
constexpr u32 kBoundTile = 2048;
__device__ __forceinline__ bool bundle_head(const u32* __restrict__ skey, u32 i, u32 n) {
  if (i >= n) return false;
  const u32 k = skey[i];
  return k != kInvalid && (i == 0 || skey[i - 1] != k);
}
// The frame hash is sized for "every point its own bundle" (6 MB) but a frame fills a few thousand slots: instead of a memset
// per frame, the slots the frame used are put back to empty once its keys have been read (duplicates write the same words) --
// here, behind the bundling sort (round 2 had a launch of its own for it right behind k_bundle_keys).
__global__ void __launch_bounds__(256) k_bundle_count(const FrameParams* __restrict__ Pp, BundleView V, u32* __restrict__ tile_sums, const u32* __restrict__ pslot,
                                                      u64* __restrict__ fh_keys, u32* __restrict__ fh_first, int self_clean) {
  __shared__ u32 lds[4];
  const u32 n = Pp->n_points;
  const u32* __restrict__ skey = V.key[V.info->parity & 1u];
  const u32 n_tiles = (n + kBoundTile - 1) / kBoundTile;
  for (u32 tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    u32 c = 0;
#pragma unroll
    for (u32 q = 0; q < kBoundTile / 256; ++q) {
      const u32 i = tile * kBoundTile + q * 256 + threadIdx.x;
      c += bundle_head(skey, i, n) ? 1u : 0u;
      if (self_clean && i < n) {
        const u32 slot = pslot[i];
        if (slot != kInvalid) {
          fh_keys[slot] = kEmptyKey;
          fh_first[slot] = 0xFFFFFFFFu;
        }
      }
    }
    u32 tot;
    (void)block_exclusive_scan<4>(c, &tot, lds);
    if (threadIdx.x == 0) tile_sums[tile] = tot;
  }
}
__global__ void __launch_bounds__(256) k_bundle_starts(const FrameParams* __restrict__ Pp, BundleView V, const u32* __restrict__ tile_sums, u32* __restrict__ bstart,
                                                       Counters* cnt) {
  __shared__ u32 lds[4], lds2[4];
  const u32 n = Pp->n_points;
  const u32* __restrict__ skey = V.key[V.info->parity & 1u];
  const u32 n_tiles = (n + kBoundTile - 1) / kBoundTile;
  if (n_tiles == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
    cnt->n_rays = 0;
    cnt->n_ray_slots = 0;
  }
  for (u32 tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    u32 below = 0;
    for (u32 t = threadIdx.x; t < tile; t += 256) below += tile_sums[t];
    u32 base;
    (void)block_exclusive_scan<4>(below, &base, lds2);
    // thread t owns the 8 consecutive positions [tile * 2048 + 8 t, + 8)
    const u32 i0 = tile * kBoundTile + threadIdx.x * 8;
    bool h[8];
    u32 c = 0;
#pragma unroll
    for (u32 q = 0; q < 8; ++q) {
      h[q] = bundle_head(skey, i0 + q, n);
      c += h[q] ? 1u : 0u;
    }
    u32 tot;
    u32 rank = base + block_exclusive_scan<4>(c, &tot, lds);
#pragma unroll
    for (u32 q = 0; q < 8; ++q) {
      const u32 i = i0 + q;
      if (h[q]) bstart[rank++] = i;
      if (i < n && skey[i] != kInvalid && (i + 1 == n || skey[i + 1] == kInvalid)) cnt->n_sorted_valid = i + 1;  // invalid keys sort last: one writer
    }
    if (tile + 1 == n_tiles && threadIdx.x == 0) {
      cnt->n_rays = base + tot;  // number of bundles
      cnt->n_ray_slots = base + tot;
    }
  }
}

// two waves per bundle: the sequential weighted mean of its points in visiting order, bit-exact with the
// single-threaded reference loop
//     merged = (merged * W + p * w) / (W + w);  colour = blend(colour, W, c, w);  W += w
// Each component is a recurrence  val <- f_k(val)  whose operands do not depend on the running value:
//     x, y, z (wave 0):     val <- (val * W_k + p * w) / (W_k + w)                   one IEEE divide per step
//     a, b, g, r (wave 1):  val <- round(val * (W_k/(W_k+w)) + c * (w/(W_k+w)))      multiply-add-round per step
// 64 points at a time are gathered in parallel and their operands are computed lane-parallel into an LDS table
// [point][component]; lanes 0-3 then carry one component each through the dependent chain with one (prefetched)
// LDS read per step and no branches.  The chain length is the bundle size, so the two chains of a big bundle run
// side by side on different SIMDs instead of back to back.  Skipped points (w < eps, or anything after the first
// point of a clearing bundle) are the identity step: (M, A, D) = (1, 0, 1).
typedef float MergeOp __attribute__((ext_vector_type(4)));  // (M, A, D, 1/D)

// Piece path (k_touch_pieces): a PIECE is a maximal run of consecutive steps of one ray inside one tile (block, z slab:
// 16 x 16 x 1 voxels) -- the voxel coordinates of a walk are monotone, so a ray meets a tile in one run of at most 31
// steps.  Pieces are written at fixed slots (exclusive scan of this bound over the rays), which keeps them in ray order
// without a sort key for it.  The wave walk starts a piece at every round of 64 steps, at every z step and at every
// change of the x / y block; its per-axis step counts are at most n_axis + 1 (wave_ray_path generates n_axis + 2 crossing
// times and rejects a walk that uses the last one).  Rays that are known here to take the sequential walk get the
// trivial bound (one piece per step); the + 4 covers a ray whose walk is only found to need the fallback later (and the
// sequential walk makes no round pieces) -- if even that is exceeded the frame is dropped and reported, never wrong.
__device__ __forceinline__ u32 piece_bound(const Dda& d, u32 axis_cap) {
  const u32 ns = d.nsteps;
  if (ns == 0) return 0;
  const u32 gen = ns + 1;
  const u32 g0 = min(d.n_axis[0] + 2, gen), g1 = min(d.n_axis[1] + 2, gen), g2 = min(d.n_axis[2] + 2, gen);
  if (d.sgn[0] == 0 || d.sgn[1] == 0 || d.sgn[2] == 0 || g0 > axis_cap || g1 > axis_cap || g2 > axis_cap || ns > 3 * axis_cap) return ns;
  const u32 b = (ns + 63) / 64 + (d.n_axis[2] + 1) + ((d.n_axis[0] + 1) / 16 + 1) + ((d.n_axis[1] + 1) / 16 + 1) + 4;
  return min(b, ns);
}

__global__ void __launch_bounds__(256) k_bundle_merge(const FrameParams* __restrict__ Pp, BundleView V, const u32* __restrict__ bstart, RayArrays R, Counters* cnt,
                                                      u32 piece_axis_cap) {
  const FrameParams P = *Pp;
  const u32 np2 = P.np2;
  const float* __restrict__ xyz = P.xyz;
  const uint8_t* __restrict__ rgba = P.rgba;
  const u32 spar = uniform_u32(V.info->parity & 1u);
  const u32* __restrict__ skey = V.key[spar];
  const u32* __restrict__ sval = V.val[spar];
  __shared__ MergeOp ops[4][64][4];
  const u32 n_bundles = uniform_u32(cnt->n_rays);
  const u32 n_valid = uniform_u32(cnt->n_sorted_valid);
  const u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
  const u32 nthreads = gridDim.x * blockDim.x;
  const u32 lane = lane_id();
  const u32 role = lane & 3u;
  MergeOp(*tbl)[4] = ops[threadIdx.x >> 6];
  for (u32 task = uniform_u32(tid >> 6); task < 2 * n_bundles; task += nthreads >> 6) {
    const u32 m = task >> 1;
    const bool colour_wave = (task & 1u) != 0;
    const u32 begin = uniform_u32(bstart[m]);
    const u32 end = uniform_u32((m + 1 < n_bundles) ? bstart[m + 1] : n_valid);
    const bool clearing = uniform_u32(skey[begin]) >= np2;
    if (!colour_wave && lane == 0) atomicMax(&cnt->shard[m & 63u][kShMaxBundle], end - begin);
    if (colour_wave && rgba == nullptr) {  // no colours: Color() stays (0,0,0,0)
      if (lane == 0) {
        R.color[m] = 0u;
        reinterpret_cast<u32*>(R.q)[static_cast<size_t>(m) * 8u + 5u] = 0u;
      }
      continue;
    }
    float val = 0.0f;  // this lane's component of the running mean / colour channel
    float W = 0.0f;    // uniform
    u64 key = 0;
    bool done = false;
    for (u32 base = begin; base < end && !done; base += 64) {
      const u32 i = base + lane;
      const u32 cnt_in = min(64u, end - base);
      float px = 0.0f, py = 0.0f, pz = 0.0f, w = 0.0f;
      u32 col = 0;
      if (i < end) {
        const u32 idx = mixed_index(sval[i], P.n_points);
        px = xyz[3 * idx];
        py = xyz[3 * idx + 1];
        pz = xyz[3 * idx + 2];
        w = voxel_weight(P, F3{px, py, pz});
        if (colour_wave) col = pack_rgba_wire(rgba, idx);
      }
      if (base == begin && !colour_wave) {
        const F3 pg = transform_point(P, F3{readlane_f32(px, 0), readlane_f32(py, 0), readlane_f32(pz, 0)});
        key = pack_key(grid_index(pg.x * P.voxel_size_inv), grid_index(pg.y * P.voxel_size_inv), grid_index(pg.z * P.voxel_size_inv));
      }
      // which points take part
      bool used = (lane < cnt_in) && !(w < kEps);
      if (clearing) {  // only the first point of a clearing bundle is used
        const u64 um = __ballot(used);
        used = used && (lane == static_cast<u32>(__ffsll(static_cast<long long>(um))) - 1u);
        if (um) done = true;
      }
      // W before each point of the chunk (skipped points leave W unchanged)
      float Wpre;
      const u64 in_mask = (cnt_in == 64) ? ~0ull : ((1ull << cnt_in) - 1ull);
      if (!clearing && (__ballot(w == 1.0f) & in_mask) == in_mask && W == truncf(W) && W < 8388608.0f) {
        Wpre = W + static_cast<float>(lane);  // integers: every partial sum is exact
      } else {
        float run = W;
        Wpre = W;
        const u64 used_mask = __ballot(used);
        for (u32 k = 0; k < cnt_in; ++k) {
          if (lane == k) Wpre = run;
          const float wk = readlane_f32(w, k);
          if ((used_mask >> k) & 1ull) run += wk;
        }
      }
      const float den = Wpre + w;
      const float Wnext = used ? den : Wpre;
      const bool d_ok = !used || (den >= 9.094947e-13f && den <= 1.0995116e12f);  // divisor range of the fast division
      // operand table of this chunk; rows beyond the chunk are the identity so the chain can run in groups of 4
      {
        MergeOp* row = tbl[lane];
        const MergeOp ident{1.0f, 0.0f, 1.0f, 1.0f};
        if (!used) {
#pragma unroll
          for (u32 c = 0; c < 4; ++c) row[c] = ident;
        } else if (!colour_wave) {
          const float r0 = __builtin_amdgcn_rcpf(den);
          const float r = __builtin_fmaf(__builtin_fmaf(-den, r0, 1.0f), r0, r0);  // one Newton step: < 1 ulp from 1/den
          row[0] = MergeOp{Wpre, px * w, den, r};
          row[1] = MergeOp{Wpre, py * w, den, r};
          row[2] = MergeOp{Wpre, pz * w, den, r};
          row[3] = ident;
        } else {
          const float fa = Wpre / den, fb = w / den;  // colour blend factors of this point
#pragma unroll
          for (u32 c = 0; c < 4; ++c) row[c] = MergeOp{fa, static_cast<float>(static_cast<int>((col >> (8u * c)) & 255u)) * fb, 1.0f, 0.0f};
        }
      }
      wave_lds_handover();
      // the dependent chain
      MergeOp nxt = tbl[0][role];
      if (!colour_wave) {
        // IEEE division with everything that depends only on the divisor hoisted off the chain: r = refined 1/D sits in
        // the table; q0 = N r, then two residual corrections -- the same Newton sequence the compiler emits for '/',
        // minus its range scaling.  A correctly rounded quotient is unique, so the bits equal the oracle's '/' whenever
        // no intermediate can leave the normal range; |N| and |D| are tracked off the critical path and the chunk is
        // redone with plain '/' if they ever left [2^-40, 2^40] (or N was 0, where the sign of zero would differ).
        const float val_in = val;
        float n_lo = 1.0f, n_hi = 1.0f;
        // groups of 4 steps with the next group's operands already in flight: the LDS latency (~100 cycles) would
        // otherwise bound every step of the chain
        // Two register sets in turn (eight steps per round): the operands of one group of four are read from LDS while the other
        // group's steps run, with scheduling barriers so that the reads are issued where they are written -- left alone, the compiler
        // moves each group's first read to the top of its own steps and waits for it there (~100 cycles per four steps on the chain).
        // Rows beyond the chunk are the identity, so a trailing group of four is harmless.
        MergeOp a0 = tbl[0][role], a1 = tbl[1][role], a2 = tbl[2][role], a3 = tbl[3][role];
        // (Going on with q1 and checking off the chain that the second correction would not have changed it does not pay: one step
        // in a hundred needs that correction, so nearly every chunk of 3 x 64 steps had to be redone.)
#define COX_MERGE_STEP(op)                                  \
  {                                                         \
    const float N = val * (op).x + (op).y;                  \
    n_lo = fminf(n_lo, fabsf(N));                           \
    n_hi = fmaxf(n_hi, fabsf(N));                           \
    const float q0 = N * (op).w;                            \
    const float e0 = __builtin_fmaf(-(op).z, q0, N);        \
    const float q1 = __builtin_fmaf(e0, (op).w, q0);        \
    const float e1 = __builtin_fmaf(-(op).z, q1, N);        \
    val = __builtin_fmaf(e1, (op).w, q1);                   \
  }
        for (u32 k = 0; k < cnt_in; k += 8) {
          const u32 kb = (k + 4) & 63u;
          const MergeOp b0 = tbl[kb][role], b1 = tbl[kb + 1][role], b2 = tbl[kb + 2][role], b3 = tbl[kb + 3][role];
          __builtin_amdgcn_sched_barrier(0);
          COX_MERGE_STEP(a0)
          COX_MERGE_STEP(a1)
          COX_MERGE_STEP(a2)
          COX_MERGE_STEP(a3)
          __builtin_amdgcn_sched_barrier(0);
          const u32 ka = (k + 8) & 63u;
          a0 = tbl[ka][role];
          a1 = tbl[ka + 1][role];
          a2 = tbl[ka + 2][role];
          a3 = tbl[ka + 3][role];
          __builtin_amdgcn_sched_barrier(0);
          COX_MERGE_STEP(b0)
          COX_MERGE_STEP(b1)
          COX_MERGE_STEP(b2)
          COX_MERGE_STEP(b3)
          __builtin_amdgcn_sched_barrier(0);
        }
#undef COX_MERGE_STEP
        const bool bad = (lane < 3) && !(n_lo >= 9.094947e-13f && n_hi <= 1.0995116e12f);  // NaN fails too
        if (__ballot(bad || !d_ok)) {
          val = val_in;
          nxt = tbl[0][role];
          for (u32 k = 0; k < cnt_in; ++k) {
            const MergeOp op = nxt;
            nxt = tbl[(k + 1) & 63u][role];
            val = (val * op.x + op.y) / op.z;
          }
        }
      } else {
        // (the same two register sets in turn; the identity rows leave an integer as it is)
        // roundf(x) for x >= 0 (a blend of non-negative values with non-negative factors): trunc(x), plus one where the -- exactly
        // computed -- fraction reaches a half; two dependent operations fewer than the sign-preserving general form.
#define COX_COLOUR_STEP(op)                                 \
  {                                                         \
    const float x = val * (op).x + (op).y;                  \
    const float t = truncf(x);                              \
    val = (x - t >= 0.5f) ? t + 1.0f : t;                   \
  }
        MergeOp a0 = tbl[0][role], a1 = tbl[1][role], a2 = tbl[2][role], a3 = tbl[3][role];
        for (u32 k = 0; k < cnt_in; k += 8) {
          const u32 kb = (k + 4) & 63u;
          const MergeOp b0 = tbl[kb][role], b1 = tbl[kb + 1][role], b2 = tbl[kb + 2][role], b3 = tbl[kb + 3][role];
          __builtin_amdgcn_sched_barrier(0);
          COX_COLOUR_STEP(a0)
          COX_COLOUR_STEP(a1)
          COX_COLOUR_STEP(a2)
          COX_COLOUR_STEP(a3)
          __builtin_amdgcn_sched_barrier(0);
          const u32 ka = (k + 8) & 63u;
          a0 = tbl[ka][role];
          a1 = tbl[ka + 1][role];
          a2 = tbl[ka + 2][role];
          a3 = tbl[ka + 3][role];
          __builtin_amdgcn_sched_barrier(0);
          COX_COLOUR_STEP(b0)
          COX_COLOUR_STEP(b1)
          COX_COLOUR_STEP(b2)
          COX_COLOUR_STEP(b3)
          __builtin_amdgcn_sched_barrier(0);
        }
#undef COX_COLOUR_STEP
      }
      wave_lds_handover();
      W = readlane_f32(Wnext, cnt_in - 1);
    }
    if (colour_wave) {
      u32 mcolor = 0;
#pragma unroll
      for (u32 c = 0; c < 4; ++c) mcolor |= (static_cast<u32>(static_cast<int>(readlane_f32(val, c))) & 255u) << (8u * c);
      if (lane == 0) {
        R.color[m] = mcolor;
        reinterpret_cast<u32*>(R.q)[static_cast<size_t>(m) * 8u + 5u] = mcolor;  // (the ray's line: the wave apply takes the colour from there)
      }
    } else {
      const float mx = readlane_f32(val, 0), my = readlane_f32(val, 1), mz = readlane_f32(val, 2);
      if (lane == 0) {
        const F3 pg = transform_point(P, F3{mx, my, mz});
        Dda d;
        dda_setup(d, P, pg, clearing);
        if (d.range_error) atomicOr(&cnt->err, kErrRange);
        R.px[m] = pg.x;
        R.py[m] = pg.y;
        R.pz[m] = pg.z;
        R.w[m] = W;
        R.flags[m] = 1u | (clearing ? 2u : 0u);
        R.key[m] = key;
        R.nsteps[m] = d.nsteps;
        if (piece_axis_cap) R.pbound[m] = piece_bound(d, piece_axis_cap);
        {
          // what compute_sdf derives from the ray alone, with its own operations, once per ray instead of once per step
          const F3 dv = pg - F3{P.tx, P.ty, P.tz};
          typedef float F4 __attribute__((ext_vector_type(4)));
          F4* q = reinterpret_cast<F4*>(R.q + static_cast<size_t>(m) * 8u);
          q[0] = F4{dv.x, dv.y, dv.z, sqrtf(dot3(dv, dv))};
          R.q[static_cast<size_t>(m) * 8u + 4u] = W;  // (word 5 is the colour, written by the bundle's colour wave)
        }
      }
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

__global__ void __launch_bounds__(256) k_touch(const FrameParams* __restrict__ Pp, RayArrays R, LayerView L, u32* __restrict__ touched_slots, Counters* cnt, u32* layer_err,
                                               const u64* __restrict__ fh_keys, u32 fh_mask) {
  const FrameParams P = *Pp;
  const u32 n_slots = cnt->n_ray_slots;
  for (u32 r = blockIdx.x * blockDim.x + threadIdx.x; r < n_slots; r += gridDim.x * blockDim.x) {
    const u32 ns = R.nsteps[r];
    if (ns == 0) continue;
    const bool clearing = (R.flags[r] & 2u) != 0;
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
          atomicSub(L.d_nblocks, 1u);     // the counter settles at the capacity
          atomicOr(layer_err, kErrPool);  // ht_vals[slot] stays kInvalid: updates to this block are dropped, and every later
                                          // frame that meets the key reports the error again (emit kernels)
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
}

// every emit kernel starts by publishing, per block touched this frame, what the update kernels need of it
__device__ __forceinline__ void fill_ord_info(const LayerView& L, const u32* __restrict__ touched_slots, int4* __restrict__ ord_info, u32 n_touched) {
  for (u32 ord = blockIdx.x * blockDim.x + threadIdx.x; ord < n_touched; ord += gridDim.x * blockDim.x) {
    const u32 slot = touched_slots[ord];
    int bx, by, bz;
    unpack_key(L.ht_keys[slot], &bx, &by, &bz);
    ord_info[ord] = make_int4(bx * 16, by * 16, bz * 16, static_cast<int>(L.ht_vals[slot]));
  }
}

// ---- emit: (voxel id, ray id) records, ray-major ------------------------------------------------
// voxel id = ordinal of the block within this frame << 12 | linear voxel index.  Also publishes the key
// width the record sort needs.
__global__ void __launch_bounds__(256) k_emit(const FrameParams* __restrict__ Pp, RayArrays R, LayerView L, u32* __restrict__ rec_key, u32* __restrict__ rec_ray, u32 rec_cap,
                                              Counters* cnt, SortInfo* sort_info, const u64* __restrict__ fh_keys, u32 fh_mask, const u32* __restrict__ touched_slots,
                                              int4* __restrict__ ord_info, int by_block) {
  const FrameParams P = *Pp;
  const u32 n_slots = cnt->n_ray_slots;
  fill_ord_info(L, touched_slots, ord_info, cnt->n_touched);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    // ordinals are < n_touched; kInvalid's low bits (all ones) must sort after every valid id
    u32 bits = 12;
    while ((1ull << (bits - 12)) < static_cast<u64>(cnt->n_touched) + 1ull) ++bits;
    const bool overflow = cnt->n_records > rec_cap;
    // block apply (by_block & 255 = its tile shift): a stable partition by tile is all the global order it needs; bit 8 of
    // by_block: one pass on the low 12 bits of the tile id (buckets) is enough
    const u32 shift = static_cast<u32>(by_block) & 255u;
    sort_info->nbits = overflow ? 0u : ((by_block & 256) ? min(bits - shift, 12u) : bits - shift);  // 0 bits: every sort pass exits at once
    sort_info->parity = 0;
    sort_info->base = shift;
    if (overflow) atomicOr(&cnt->err, kErrRecords);
  }
  if (cnt->n_records > rec_cap) return;  // frame dropped as a whole (reported at sync); never a partial update
  for (u32 r = blockIdx.x * blockDim.x + threadIdx.x; r < n_slots; r += gridDim.x * blockDim.x) {
    const u32 ns = R.nsteps[r];
    if (ns == 0) continue;
    const u32 off = R.rec_off[r];
    const bool clearing = (R.flags[r] & 2u) != 0;
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
          if (last_ord == kInvalid) atomicOr(&cnt->err, kErrPool);  // block without storage: this update is lost
        }
        if (last_ord != kInvalid) vid = (last_ord << 12) | static_cast<u32>((x & 15) | ((y & 15) << 4) | ((z & 15) << 8));
      }
      rec_key[off + s] = vid;
      rec_ray[off + s] = r;
    }
  }
}

// ---- wave-per-ray traversal (few, long rays: the merged integrator's bundles) -----------------------------
// A DDA is a dependent chain, so lane-per-ray leaves the chip empty when a frame has only a few thousand rays.
// Here one wave walks one ray in parallel and reproduces the sequential argmin walk bit for bit:
//   - per axis, the plane-crossing times are the reference's own repeated float additions
//     T_k(j+1) = fl(T_k(j) + t_step_k) (three short chains, one lane each, into LDS);
//   - "pick the smallest t, first axis wins ties" is a 3-way stable merge of those sorted sequences, so the
//     position of crossing (k, j) in the walk is j + #{crossings of the other axes that precede it}, found by
//     binary search; the same counts are the voxel's offset from the start voxel;
//   - rays the argument does not cover (a zero ray component gives -inf / NaN times, very long rays, or a walk
//     that would need more crossings of one axis than were generated) fall back to the sequential walk on lane 0.
// per-axis crossing capacity of the parallel DDA: two instantiations, the small one (12 KB of LDS per workgroup instead
// of 48 KB, so every ray of a frame is resident at once) whenever no ray of the configuration can cross more planes
constexpr u32 kAxisCapSmall = 128, kAxisCapLarge = 512;
constexpr u32 kRayFallback = 4u;  // ray flag

__device__ __forceinline__ u32 pack_path(u32 jx, u32 jy, u32 jz) { return jx | (jy << 10) | (jz << 20); }
// entries of the sorted array a[0, n) that precede t: a[i] < t, or a[i] <= t when inclusive
__device__ __forceinline__ u32 count_before(const float* a, u32 n, float t, bool inclusive) {
  u32 lo = 0, hi = n;
  while (lo < hi) {
    const u32 mid = (lo + hi) >> 1;
    const float v = a[mid];
    const bool before = inclusive ? (v <= t) : (v < t);
    if (before)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}
// The same count, found from an estimate: the sequence is a[i] = fl(a[i-1] + step), i.e. a[0] + i * step up to rounding, so
// (t - a[0]) / step lands within an entry or two of the partition point; the table decides (a sorted array has ONE partition
// point of "before", whichever way it is approached), and anything odd (infinite steps, an estimate that is far off) goes
// to the binary search.  Two or three LDS reads instead of nine dependent ones: the ranking was 2/3 of the wave walk.
__device__ __forceinline__ u32 count_before_guided(const float* a, u32 n, float t, bool inclusive, float a0, float inv_step) {
  if (n == 0) return 0;
  const float est = (t - a0) * inv_step;
  if (!(est > -4.0f && est < 1.0e6f)) return count_before(a, n, t, inclusive);  // NaN / inf / far outside
  u32 c = min(n, static_cast<u32>(max(0.0f, est)) + 1u);  // candidate count
#pragma unroll 1
  for (int guard = 0; guard < 6; ++guard) {
    if (c < n) {
      const float v = a[c];
      if (inclusive ? (v <= t) : (v < t)) {
        ++c;
        continue;
      }
    }
    if (c > 0) {
      const float v = a[c - 1];
      if (!(inclusive ? (v <= t) : (v < t))) {
        --c;
        continue;
      }
    }
    return c;
  }
  return count_before(a, n, t, inclusive);
}
// sequential argmin step that also reports the chosen axis
__device__ __forceinline__ int dda_step_axis(Dda& d) {
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
  return k;
}
// Fills path[0, ns) (LDS, this wave's) with the packed per-axis crossing counts of every step.  Returns false
// when the ray needs the sequential fallback (nothing usable was written).
// `limit`: only the first min(ns, limit) steps are wanted (the fast integrator's capped candidate lists).  Each axis then
// needs its first limit + 1 crossings only: a crossing whose rank is below the limit is preceded by fewer than `limit`
// crossings of any other axis, so the truncated sequences still count them exactly; everything else is discarded.
template <u32 kAxisCap>
__device__ __forceinline__ bool wave_ray_path(const Dda& d0, u32 ns, float* tl /*[3][kAxisCap]*/, u32* path, u32 lane, u32 limit = 0xFFFFFFFFu) {
  if (d0.sgn[0] == 0 || d0.sgn[1] == 0 || d0.sgn[2] == 0) return false;
  const u32 want = min(ns, limit);
  const u32 gen = (want < 0xFFFFFFFEu) ? want + 1u : want;
  const u32 f0 = d0.n_axis[0] + 2, f1 = d0.n_axis[1] + 2, f2 = d0.n_axis[2] + 2;  // whole sequences (two entries past the last crossing)
  const u32 g0 = min(f0, gen), g1 = min(f1, gen), g2 = min(f2, gen);              // generated
  if (g0 > kAxisCap || g1 > kAxisCap || g2 > kAxisCap || want > 3 * kAxisCap) return false;
  const u32 L = want - 1;
  if (lane < 3) {
    // the per-axis values are picked with selects on opaque copies: left alone, the compiler turns "lane == 0 ? a[0] :
    // lane == 1 ? a[1] : a[2]" into a[lane] and moves the whole Dda into scratch memory (72 B per lane written per ray)
    float tn0 = d0.t_next[0], tn1 = d0.t_next[1], tn2 = d0.t_next[2], ts0 = d0.t_step[0], ts1 = d0.t_step[1], ts2 = d0.t_step[2];
    asm volatile("" : "+v"(tn0), "+v"(tn1), "+v"(tn2), "+v"(ts0), "+v"(ts1), "+v"(ts2));
    float T = (lane == 0) ? tn0 : (lane == 1) ? tn1 : tn2;
    const float st = (lane == 0) ? ts0 : (lane == 1) ? ts1 : ts2;
    const u32 g = (lane == 0) ? g0 : (lane == 1) ? g1 : g2;
    float* row = tl + lane * kAxisCap;
    for (u32 j = 0; j < g; ++j) {
      row[j] = T;
      T += st;
    }
  }
  wave_lds_handover();
  const u32 E = g0 + g1 + g2;
  const float inv_step[3] = {1.0f / d0.t_step[0], 1.0f / d0.t_step[1], 1.0f / d0.t_step[2]};  // (estimates only: the tables decide)
  bool bad = false;
  for (u32 eb = 0; eb < E; eb += 64) {
    const u32 e = eb + lane;
    if (e < E) {
      const u32 k = (e < g0) ? 0u : (e < g0 + g1) ? 1u : 2u;
      const u32 j = (k == 0) ? e : (k == 1) ? e - g0 : e - g0 - g1;
      const float t = tl[k * kAxisCap + j];
      if (isnan(t)) bad = true;
      // crossings that precede (k, j): own axis j, lower axes on <=, higher axes on <
      const u32 c0 = (k == 0) ? j : count_before_guided(tl, g0, t, true, d0.t_next[0], inv_step[0]);
      const u32 c1 = (k == 1) ? j : count_before_guided(tl + kAxisCap, g1, t, k > 1, d0.t_next[1], inv_step[1]);
      const u32 c2 = (k == 2) ? j : count_before_guided(tl + 2 * kAxisCap, g2, t, false, d0.t_next[2], inv_step[2]);
      const u32 rank = c0 + c1 + c2;
      if (rank < L) {
        path[rank + 1] = pack_path(c0 + (k == 0 ? 1u : 0u), c1 + (k == 1 ? 1u : 0u), c2 + (k == 2 ? 1u : 0u));
        const u32 f = (k == 0) ? f0 : (k == 1) ? f1 : f2;
        if (j == f - 1) bad = true;  // the walk would go on to a crossing past the ones the ray has
      }
    }
  }
  if (lane == 0) path[0] = 0;
  wave_lds_handover();
  return __ballot(bad) == 0ull;
}

// returns the block's hash slot (kInvalid: table full)
__device__ __forceinline__ u32 touch_block(const FrameParams& P, const LayerView& L, u64 bkey, u32* touched_slots, Counters* cnt, u32* layer_err) {
  bool fresh;
  const u32 slot = ht_insert(L.ht_keys, L.ht_mask, bkey, &fresh);
  if (slot == kInvalid) {
    atomicOr(layer_err, kErrTable);
    return kInvalid;
  }
  if (fresh) {
    const u32 pool = atomicAdd(L.d_nblocks, 1u);
    if (pool < L.capacity) {
      L.ht_vals[slot] = pool;  // read by later kernels only
      L.block_keys[pool] = bkey;
      atomicAdd(&cnt->n_new_blocks, 1u);
    } else {
      atomicSub(L.d_nblocks, 1u);     // the counter settles at the capacity
      atomicOr(layer_err, kErrPool);  // ht_vals[slot] stays kInvalid: updates to this block are dropped, and every later
                                      // frame that meets the key reports the error again (emit kernels)
    }
  }
  if (__hip_atomic_load(&L.ht_stamp[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != P.frame_id &&
      atomicExch(&L.ht_stamp[slot], P.frame_id) != P.frame_id) {
    const u32 ord = atomicAdd(&cnt->n_touched, 1u);
    touched_slots[ord] = slot;
    L.ht_ord[slot] = ord;
  }
  return slot;
}

template <u32 kAxisCap>
__global__ void __launch_bounds__(256) k_touch_wave(const FrameParams* __restrict__ Pp, RayArrays R, LayerView L, u32* __restrict__ touched_slots, u32* __restrict__ path_out,
                                                    u32 rec_cap, Counters* cnt, u32* layer_err, const u64* __restrict__ fh_keys, u32 fh_mask) {
  const FrameParams P = *Pp;
  __shared__ float lds_t[4][3 * kAxisCap];
  __shared__ u32 lds_path[4][3 * kAxisCap];
  const u32 n_slots = uniform_u32(cnt->n_ray_slots);
  const bool overflow = uniform_u32(cnt->n_records) > rec_cap;
  const u32 lane = lane_id();
  const u32 wave = threadIdx.x >> 6;
  float* tl = lds_t[wave];
  u32* path = lds_path[wave];
  const u32 waves_total = (gridDim.x * blockDim.x) >> 6;
  for (u32 r = uniform_u32((blockIdx.x * blockDim.x + threadIdx.x) >> 6); r < n_slots; r += waves_total) {
    const u32 ns = uniform_u32(R.nsteps[r]);
    if (ns == 0) continue;
    const u32 flags = uniform_u32(R.flags[r]);
    const bool clearing = (flags & 2u) != 0;
    const F3 pg{readlane_f32(R.px[r], 0), readlane_f32(R.py[r], 0), readlane_f32(R.pz[r], 0)};
    const u64 own_key = P.anti_grazing ? R.key[r] : 0ull;
    Dda d;
    dda_setup(d, P, pg, clearing);
    const bool par = wave_ray_path<kAxisCap>(d, ns, tl, path, lane);
    if (lane == 0) R.flags[r] = par ? (flags & ~kRayFallback) : (flags | kRayFallback);
    if (par) {
      const u32 off = uniform_u32(R.rec_off[r]);
      u64 carry = kEmptyKey;
      for (u32 base = 0; base < ns; base += 64) {
        const u32 s = base + lane;
        const bool act = s < ns;
        u64 bkey = kEmptyKey;
        bool skip = false;
        if (act) {
          const u32 p = path[s];
          if (!overflow) path_out[off + s] = p;  // emit reads the walk instead of redoing it
          const int x = d.c[0] + static_cast<int>(p & 1023u) * d.sgn[0];
          const int y = d.c[1] + static_cast<int>((p >> 10) & 1023u) * d.sgn[1];
          const int z = d.c[2] + static_cast<int>(p >> 20) * d.sgn[2];
          skip = grazing_skip(P, fh_keys, fh_mask, clearing, own_key, x, y, z);
          bkey = pack_key(x >> 4, y >> 4, z >> 4);
        }
        // a voxel touches its block unless the previous voxel THAT WAS NOT SKIPPED lies in the same block (a block whose
        // first voxel on the ray is skipped by anti-grazing must still be allocated by the next one; found by the fuzzer)
        const u64 kept = __ballot(act && !skip);
        const u64 kept_below = kept & ((1ull << lane) - 1ull);
        const int src = kept_below ? (63 - __clzll(static_cast<long long>(kept_below))) : 0;
        const u64 prev_kept = __shfl(bkey, src, 64);
        const u64 prev = kept_below ? prev_kept : carry;
        if (act && !skip && bkey != prev) touch_block(P, L, bkey, touched_slots, cnt, layer_err);
        if (kept) carry = __shfl(bkey, 63 - __clzll(static_cast<long long>(kept)), 64);
      }
      wave_lds_handover();  // the next ray of this wave reuses the LDS scratch
    } else if (lane == 0) {
      // sequential fallback
      u64 last_bkey = kEmptyKey;
      for (u32 s = 0; s < ns; ++s) {
        const int x = d.c[0], y = d.c[1], z = d.c[2];
        dda_step(d);
        if (grazing_skip(P, fh_keys, fh_mask, clearing, own_key, x, y, z)) continue;
        const u64 bkey = pack_key(x >> 4, y >> 4, z >> 4);
        if (bkey == last_bkey) continue;
        last_bkey = bkey;
        touch_block(P, L, bkey, touched_slots, cnt, layer_err);
      }
    }
  }
}

__global__ void __launch_bounds__(256) k_emit_wave(const FrameParams* __restrict__ Pp, RayArrays R, LayerView L, const u32* __restrict__ path_in, u32* __restrict__ rec_key,
                                                   u32* __restrict__ rec_ray, u32 rec_cap, Counters* cnt, SortInfo* sort_info,
                                                   const u64* __restrict__ fh_keys, u32 fh_mask, const u32* __restrict__ touched_slots, int4* __restrict__ ord_info,
                                                   int by_block) {
  const FrameParams P = *Pp;
  const u32 n_slots = uniform_u32(cnt->n_ray_slots);
  fill_ord_info(L, touched_slots, ord_info, cnt->n_touched);
  const bool overflow = uniform_u32(cnt->n_records) > rec_cap;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    // ordinals are < n_touched; kInvalid's low bits (all ones) must sort after every valid id
    u32 bits = 12;
    while ((1ull << (bits - 12)) < static_cast<u64>(cnt->n_touched) + 1ull) ++bits;
    const u32 shift = static_cast<u32>(by_block) & 255u;
    sort_info->nbits = overflow ? 0u : ((by_block & 256) ? min(bits - shift, 12u) : bits - shift);  // 0 bits: every sort pass exits at once
    sort_info->parity = 0;
    sort_info->base = shift;
    if (overflow) atomicOr(&cnt->err, kErrRecords);
  }
  if (overflow) return;  // frame dropped as a whole (reported at sync); never a partial update
  const u32 lane = lane_id();
  const u32 waves_total = (gridDim.x * blockDim.x) >> 6;
  for (u32 r = uniform_u32((blockIdx.x * blockDim.x + threadIdx.x) >> 6); r < n_slots; r += waves_total) {
    const u32 ns = uniform_u32(R.nsteps[r]);
    if (ns == 0) continue;
    const u32 flags = uniform_u32(R.flags[r]);
    const bool clearing = (flags & 2u) != 0;
    const u32 off = uniform_u32(R.rec_off[r]);
    const F3 pg{readlane_f32(R.px[r], 0), readlane_f32(R.py[r], 0), readlane_f32(R.pz[r], 0)};
    const u64 own_key = P.anti_grazing ? R.key[r] : 0ull;
    Dda d;
    dda_setup(d, P, pg, clearing);
    if (!(flags & kRayFallback)) {
      u64 carry_key = kEmptyKey;
      u32 carry_ord = kInvalid;
      for (u32 base = 0; base < ns; base += 64) {
        const u32 s = base + lane;
        const bool act = s < ns;
        u64 bkey = kEmptyKey;
        bool skip = false;
        u32 lin = 0;
        if (act) {
          const u32 p = path_in[off + s];
          const int x = d.c[0] + static_cast<int>(p & 1023u) * d.sgn[0];
          const int y = d.c[1] + static_cast<int>((p >> 10) & 1023u) * d.sgn[1];
          const int z = d.c[2] + static_cast<int>(p >> 20) * d.sgn[2];
          skip = grazing_skip(P, fh_keys, fh_mask, clearing, own_key, x, y, z);
          bkey = pack_key(x >> 4, y >> 4, z >> 4);
          lin = static_cast<u32>((x & 15) | ((y & 15) << 4) | ((z & 15) << 8));
        }
        u64 prev = __shfl_up(bkey, 1, 64);
        if (lane == 0) prev = carry_key;
        const bool is_head = act && bkey != prev;
        u32 ord = kInvalid;
        if (is_head) {
          const u32 slot = ht_find(L.ht_keys, L.ht_mask, bkey);
          ord = (slot != kInvalid && L.ht_vals[slot] != kInvalid) ? L.ht_ord[slot] : kInvalid;
          if (ord == kInvalid) atomicOr(&cnt->err, kErrPool);  // block without storage: this update is lost
        }
        // every lane takes the ordinal of the nearest head at or below it, or the carry of the previous round
        const u64 heads = __ballot(is_head);
        const u64 below = heads & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
        const int src = below ? (63 - __clzll(static_cast<long long>(below))) : 0;
        const u32 head_ord = static_cast<u32>(__shfl(static_cast<int>(ord), src, 64));
        const u32 my_ord = below ? head_ord : carry_ord;
        if (act) {
          rec_key[off + s] = (skip || my_ord == kInvalid) ? kInvalid : ((my_ord << 12) | lin);
          rec_ray[off + s] = r;
        }
        carry_key = __shfl(bkey, 63, 64);
        carry_ord = static_cast<u32>(__shfl(static_cast<int>(my_ord), 63, 64));
      }
    } else if (lane == 0) {
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
            if (last_ord == kInvalid) atomicOr(&cnt->err, kErrPool);  // block without storage: this update is lost
          }
          if (last_ord != kInvalid) vid = (last_ord << 12) | static_cast<u32>((x & 15) | ((y & 15) << 4) | ((z & 15) << 8));
        }
        rec_key[off + s] = vid;
        rec_ray[off + s] = r;
      }
    }
  }
}

#include "cox_pieces.hpp"

#include "cox_fast.hpp"

#include "cox_apply_records.hpp"
#include "cox_apply_tile.hpp"

// ---- depth front end ----------------------------------------------------------------------------
// depth image -> point list in row-major pixel order (the order depth_image_proc produces), in two launches:
//   k_depth_count   valid pixels per tile of 2048
//   k_depth_points  every workgroup adds up the counts of the tiles before its own (a 640 x 480 image has 150), scans its tile and
//                   writes its points; the last tile leaves the frame's point count on the device -- it never visits the host
constexpr u32 kDepthTile = 2048;
__device__ __forceinline__ bool depth_valid(float d) { return isfinite(d) && d > 0.0f; }
__global__ void __launch_bounds__(256) k_depth_count(const float* __restrict__ depth, u32 n, u32* __restrict__ tile_sums) {
  __shared__ u32 lds[4];
  const u32 n_tiles = (n + kDepthTile - 1) / kDepthTile;
  for (u32 tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    u32 c = 0;
#pragma unroll
    for (u32 q = 0; q < kDepthTile / 256; ++q) {
      const u32 i = tile * kDepthTile + q * 256 + threadIdx.x;
      c += (i < n && depth_valid(depth[i])) ? 1u : 0u;
    }
    u32 tot;
    (void)block_exclusive_scan<4>(c, &tot, lds);
    if (threadIdx.x == 0) tile_sums[tile] = tot;
  }
}
__global__ void __launch_bounds__(256) k_depth_points(const float* __restrict__ depth, const uint8_t* __restrict__ rgba, int w, int h, float fx, float fy,
                                                      float cx, float cy, const u32* __restrict__ tile_sums, float* __restrict__ xyz,
                                                      uint8_t* __restrict__ rgba_out, u32* __restrict__ n_out) {
  __shared__ u32 lds[4], lds2[4];
  const u32 n = static_cast<u32>(w) * static_cast<u32>(h);
  const u32 n_tiles = (n + kDepthTile - 1) / kDepthTile;
  for (u32 tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    u32 below = 0;
    for (u32 t = threadIdx.x; t < tile; t += 256) below += tile_sums[t];
    u32 base;
    (void)block_exclusive_scan<4>(below, &base, lds2);
    // thread t owns the 8 consecutive pixels [tile * 2048 + 8 t, + 8)
    const u32 i0 = tile * kDepthTile + threadIdx.x * 8;
    float d[8];
    u32 c = 0;
#pragma unroll
    for (u32 q = 0; q < 8; ++q) {
      d[q] = (i0 + q < n) ? depth[i0 + q] : 0.0f;
      c += depth_valid(d[q]) ? 1u : 0u;
    }
    u32 tot;
    u32 o = base + block_exclusive_scan<4>(c, &tot, lds);
#pragma unroll
    for (u32 q = 0; q < 8; ++q) {
      if (!depth_valid(d[q])) continue;
      const u32 i = i0 + q;
      const u32 u = i % static_cast<u32>(w), v = i / static_cast<u32>(w);
      const float xn = (static_cast<float>(u) - cx) / fx;
      const float yn = (static_cast<float>(v) - cy) / fy;
      xyz[3 * o] = d[q] * xn;
      xyz[3 * o + 1] = d[q] * yn;
      xyz[3 * o + 2] = d[q];
      if (rgba_out) reinterpret_cast<u32*>(rgba_out)[o] = rgba ? reinterpret_cast<const u32*>(rgba)[i] : 0u;
      ++o;
    }
    if (tile + 1 == n_tiles && threadIdx.x == 0) *n_out = base + tot;
  }
}

// ---- host inputs: pinned host memory read by a kernel ------------------------------------------------------------------------
// hipMemcpyAsync hands a pinned-to-device copy to the SDMA engine and pays two engine hand-overs per copy on the frame's stream;
// pinned host memory is mapped into the device's address space, so an ordinary kernel can read it instead.  The grid is SMALL on
// purpose: every lane of a copy kernel sits on a PCIe read (microseconds), and a chip-filling grid of them holds the vector-memory
// queues of every CU — kernels of neighbouring frames ran 5-8 x longer beside it (k_rs_scatter 14 -> 117 us, k_bundle_count 10 ->
// 86 us in the kernel trace).  kCopyGroups workgroups with four 16-B loads in flight per lane (COX_H2D_GROUPS x 256 x 64 B) cover
// the link's bandwidth-delay product (~55 GB/s x ~2 us) and leave the other CUs alone.  Two segments (points, colours) per launch.
typedef u32 U32x4 __attribute__((ext_vector_type(4)));
constexpr int kCopyGroups = 8;
struct HostSegment {
  U32x4* dst;
  const U32x4* src;
  size_t n16;
  u32 tail_words;  // 4-byte words behind the last whole 16 bytes
};
__device__ __forceinline__ void copy_segment_from_host(const HostSegment& g) {
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
  for (; i + 3 * stride < g.n16; i += 4 * stride) {
    const U32x4 a = __builtin_nontemporal_load(&g.src[i]);
    const U32x4 b = __builtin_nontemporal_load(&g.src[i + stride]);
    const U32x4 c = __builtin_nontemporal_load(&g.src[i + 2 * stride]);
    const U32x4 d = __builtin_nontemporal_load(&g.src[i + 3 * stride]);
    g.dst[i] = a;
    g.dst[i + stride] = b;
    g.dst[i + 2 * stride] = c;
    g.dst[i + 3 * stride] = d;
  }
  for (; i < g.n16; i += stride) g.dst[i] = __builtin_nontemporal_load(&g.src[i]);
  if (blockIdx.x == 0 && threadIdx.x < g.tail_words)
    reinterpret_cast<u32*>(g.dst + g.n16)[threadIdx.x] = reinterpret_cast<const u32*>(g.src + g.n16)[threadIdx.x];
}
__global__ void __launch_bounds__(256) k_copy_from_host(const HostSegment a, const HostSegment b) {
  copy_segment_from_host(a);
  if (b.dst) copy_segment_from_host(b);
}

// ---- self-test: the hoisted-reciprocal division of k_bundle_merge against the compiler's IEEE '/' -----------------
__global__ void __launch_bounds__(256) k_selftest_division(u64 n, u64 seed, u32* __restrict__ mismatches) {
  u64 bad = 0;
  for (u64 i = blockIdx.x * static_cast<u64>(blockDim.x) + threadIdx.x; i < n; i += static_cast<u64>(gridDim.x) * blockDim.x) {
    // splitmix64 -> two floats: divisor like the merge's (small integers and arbitrary values in 2^-40..2^40), arbitrary numerator
    u64 z = seed + 0x9E3779B97F4A7C15ull * (i + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const u32 a = static_cast<u32>(z), b = static_cast<u32>(z >> 32);
    float D, N;
    if ((i & 3) == 0) {
      D = static_cast<float>((b % 200000u) + 1u);  // W + w with unit weights
    } else {
      D = __uint_as_float(((b & 0x007FFFFFu) | (((b >> 23) % 80u + 87u) << 23)));  // exponent -40..39
    }
    N = __uint_as_float((a & 0x807FFFFFu) | ((((a >> 23) & 0xFFu) % 80u + 87u) << 23));
    const float r0 = __builtin_amdgcn_rcpf(D);
    const float r = __builtin_fmaf(__builtin_fmaf(-D, r0, 1.0f), r0, r0);
    const float q0 = N * r;
    const float e0 = __builtin_fmaf(-D, q0, N);
    const float q1 = __builtin_fmaf(e0, r, q0);
    const float e1 = __builtin_fmaf(-D, q1, N);
    const float q = __builtin_fmaf(e1, r, q1);
    const float ref = N / D;
    if (__float_as_uint(q) != __float_as_uint(ref)) ++bad;
  }
  if (bad) atomicAdd(mismatches, static_cast<u32>(bad > 0xFFFFFFFFull ? 0xFFFFFFFFull : bad));
}

// =================================================================================================
// host side
// =================================================================================================
// A frame is four pipeline stages, each on its own HIP stream, so that up to four frames are in flight:
//   A1  bundle hash + bundling sort           (input only)            stream[0]
//   A2  bundle boundaries + sequential means  (-> ray arrays)         stream[1]
//   B1  record offsets, touch, emit, sort     (allocates blocks)      stream[2]
//   B2  apply                                 (writes voxels)         stream[3]
// (simple integrator: A1 only uploads the parameter block, A2 is k_rays_simple.)  B1 of frame t+1 beside B2 of
// frame t is safe: B1 only inserts new hash entries / bumps the pool and restamps ordinals that B2 never reads (it
// goes through its own frame's touched_slots), B2 only writes voxels.  Buffers are replicated by lifetime: what
// lives from A1 to B2 (parameter block, counters, ray arrays, bundle hash) x4, A1->A2 (bundling sort buffers) x2,
// B1->B2 (records, touched blocks, piece summaries, sort info) x2.  Events order producer -> consumer and
// consumer -> next writer of the same copy.
//
// Every per-frame quantity (pose, point count, input pointers, frame id) lives in a device-side parameter block and
// every count in device counters, so the kernel arguments and grids of a stage never change: each stage is captured
// once per buffer combination into a HIP graph (4 stages x 4 combinations) and replayed -- one graph launch instead
// of ~12 kernel launches per stage, which lifts the host-side launch-rate limit (~3 000 frames/s eager).
constexpr int kStatRing = 8;
constexpr int kFrameSets = 6;  // frames in flight: one per stage
constexpr int kStageSets = 3;  // bundle sets (live H..M) and record sets (live T..U); kFrameSets is a multiple, so a frame slot fixes both
constexpr int kInputSets = kFrameSets;  // staging sets for host / depth inputs: as many as frames in flight, so that filling one never has to wait on the device
constexpr int kNumStages = 6;  // H bundle hash | P bundling sort | M bundle boundaries + means | T offsets, touch, emit | R record partition | U apply

struct FrameSet {  // lives A1 .. B2
  FrameParams* d_params = nullptr;
  Counters* cnt = nullptr;
  RayArrays rays{};
  u64* fh_keys = nullptr;  // [fh_cap] keys followed by [fh_cap] first-sequence numbers (one memset)
  u32* fh_first = nullptr;
  hipEvent_t done = nullptr;           // stage U of the frame that used this set
  hipEvent_t params_copied = nullptr;  // the H2D copy of the parameter block has executed (host may rewrite the pinned slot)
  hipEvent_t hand[kNumStages - 1] = {};  // hand[k]: stage k of the frame is enqueued complete (stage k + 1 on another stream waits for it)
  bool used = false;
};
struct BundleSet {  // lives A1 .. A2
  u32 *pslot = nullptr, *skey[2] = {nullptr, nullptr}, *sval[2] = {nullptr, nullptr}, *head = nullptr, *bstart = nullptr;
  SortInfo* sort_info = nullptr;
  hipEvent_t done = nullptr;  // A2 of the frame that used this set
  bool used = false;
};
struct RecordSet {  // lives B1 .. B2
  u32 *rec_key[2] = {nullptr, nullptr}, *rec_ray[2] = {nullptr, nullptr};
  u32 *piece_front = nullptr, *piece_back = nullptr, *piece_wsum = nullptr;
  // piece path (merged): the walk as bytes + the (ray, tile) pieces, ping-pong for their sort; the record arrays stay unallocated
  uint8_t* lin8 = nullptr;
  u32 *pkey[2] = {nullptr, nullptr}, *pstart[2] = {nullptr, nullptr}, *prl[2] = {nullptr, nullptr};
  u64* pbkey = nullptr;                   // piece partition: block key of every piece until k_piece_touch has run
  u32 *plen = nullptr, *pdest = nullptr;  // piece partition: lengths of the sorted pieces, their exclusive scan
  u32* touched_slots = nullptr;  // [layer ht_cap]
  int4* ord_info = nullptr;      // [layer ht_cap] (16 * block index, pool index) per block touched this frame
  u32 *blk_beg = nullptr, *blk_end = nullptr;  // [layer ht_cap * 16] record range of every tile (block apply); zero between frames
  u32 *big_of_tile = nullptr, *big_acc = nullptr;  // large tiles (cox_apply_tile.hpp: BigTiles); zero between frames
  u32* blk_list = nullptr;                         // [layer ht_cap * 16] tiles k_apply_block takes
  uint2* big_chunks = nullptr;
  u32 big_chunk_cap = 0;
  SortInfo* sort_info = nullptr;
  hipEvent_t done = nullptr;  // B2 of the frame that used this set
  bool used = false;
};

// buffers of the fast integrator (method == COX_METHOD_FAST only); see cox_fast.hpp
constexpr u32 kFastCap0Max = 32;  // candidate steps per ray in round 0: FastState::cap0 (COX_FAST_CAP=cap0,cap1)
struct VisitSet {  // the candidate visits of one round: ray-major arrays + their slot-sorted view
  u32 *key[2] = {nullptr, nullptr}, *val[2] = {nullptr, nullptr};  // sort ping-pong: slot, visit id
  u64 *vhash = nullptr, *shash = nullptr;                         // hash of visit v / of the visit at sorted position i
  u32 *vray = nullptr, *pos_of = nullptr, *sinfo = nullptr, *voff = nullptr;
  u32 cap = 0;                 // visits the arrays hold
  int sorted = 0;              // which ping-pong buffer holds the sorted keys
  hipEvent_t done = nullptr;   // the solve of the frame that used the set is complete
  bool used = false;
};
struct FastState {
  u64 *fhash = nullptr, *table_start = nullptr, *table_obs = nullptr;
  u32 *fresh = nullptr, *rank = nullptr;
  u32* cap[kFrameSets] = {};    // candidate-list length per ray (lives from the front to the end of the solve, like the frame set)
  u32* reach[kFrameSets] = {};  // voxels each ray updates: relaxed in place
  VisitSet vs0[2];              // round 0: written by the front of frame t (start-set stream), read by its solve -- two frames' worth
  VisitSet vs1;                 // round 1: lives inside one solve
  u32* long_list = nullptr;     // round 1: the rays with a list longer than cap1
  int fences = 1;                 // COX_FAST_FENCE (experiments): release / acquire fences in the relaxation's barrier
  u32 relax_groups = kFastRelaxGroups;  // COX_FAST_GROUPS (experiments): workgroups of the relaxation (all resident at once: far fewer than the chip holds)
  bool force_sequential = false;  // COX_FAST_SEQUENTIAL=1 (tests): every frame is redone by k_fast_sequential
  int rounds = 2;               // rounds of list growth + relaxation enqueued per frame (2 at coarse voxels, 4 where rays are long in voxels; COX_FAST_ROUNDS)
  u32 cap0 = 8, cap1 = 16;      // candidate steps per ray in round 0; list length of the rays that did not get their whole walk in round 1
  u32* d_stats = nullptr;       // [8] run totals: frames redone by the sequential kernel, frames with a round 1, passes of the relaxation (round 0, round 1), ...
  u64 off_start = 0, off_obs = 0;  // ApproxHashSet::offset_
  int reset_counter = 0;
  uint64_t frames = 0;
};

// ---- submission thread ------------------------------------------------------------------------------------------------
// At 5 cm a frame is 24 launches plus a dozen event operations, and on a box with slow host cores the caller's thread, not
// the GPU, would set the frame rate.  The caller's thread enqueues ray generation (stages H, P, M) and returns; this thread
// enqueues the layer update (T, R, U) of the same frame behind it (at most one frame behind: the per-slot events it records are
// waited for by the caller's thread three and six frames later).
struct Submitter {
  std::thread th;
  std::mutex m;
  std::condition_variable cv_job, cv_done;
  std::deque<std::function<int()>> q;
  uint64_t posted = 0, finished = 0;
  int status = COX_OK;  // first error of a job; reported (and cleared) by the next drain
  bool stop = false;
  bool ready = false;  // the thread has bound its device
  int device = 0;
  void run() {
    (void)hipSetDevice(device);
    (void)hipGetLastError();
    {
      std::lock_guard<std::mutex> lk(m);
      ready = true;
    }
    cv_done.notify_all();
    for (;;) {
      std::function<int()> job;
      {
        std::unique_lock<std::mutex> lk(m);
        cv_job.wait(lk, [&] { return stop || !q.empty(); });
        if (q.empty()) return;
        job = std::move(q.front());
        q.pop_front();
      }
      const int st = job();
      {
        std::lock_guard<std::mutex> lk(m);
        if (st != COX_OK && status == COX_OK) status = st;
        finished += 1;
      }
      cv_done.notify_all();
    }
  }
  void post(std::function<int()> job) {
    {
      std::lock_guard<std::mutex> lk(m);
      q.push_back(std::move(job));
      posted += 1;
    }
    cv_job.notify_one();
  }
  // returns when at most `outstanding` posted jobs have not been enqueued completely
  void wait_outstanding(uint64_t outstanding) {
    std::unique_lock<std::mutex> lk(m);
    cv_done.wait(lk, [&] { return posted - finished <= outstanding; });
  }
  int take_status() {
    std::lock_guard<std::mutex> lk(m);
    const int st = status;
    status = COX_OK;
    return st;
  }
};
// Pageable host inputs go through a pinned bounce buffer, and that CPU copy (4.9 MB per 640 x 480 cloud) is the caller's thread's:
// 0.35 ms with one core -- 2 800 frames/s however fast the GPU is.  A few helper threads take a share each (COX_COPY_THREADS, default 3
// beside the caller; 0: the caller alone).  They never touch HIP.
struct CopyPool {
  struct Part {
    void* dst;
    const void* src;
    size_t bytes;
  };
  std::vector<std::thread> th;
  std::mutex m;
  std::condition_variable cv_job, cv_done;
  std::vector<Part> parts;  // one per helper, bytes == 0: nothing to do
  uint64_t generation = 0;
  u32 pending = 0;
  bool stop = false;
  explicit CopyPool(int n) {
    parts.resize(static_cast<size_t>(n));
    for (int k = 0; k < n; ++k) th.emplace_back([this, k] { run(k); });
  }
  ~CopyPool() {
    {
      std::lock_guard<std::mutex> lk(m);
      stop = true;
    }
    cv_job.notify_all();
    for (std::thread& t : th) t.join();
  }
  void run(int k) {
    uint64_t seen = 0;
    for (;;) {
      Part p;
      {
        std::unique_lock<std::mutex> lk(m);
        cv_job.wait(lk, [&] { return stop || generation != seen; });
        if (stop) return;
        seen = generation;
        p = parts[static_cast<size_t>(k)];
      }
      if (p.bytes) memcpy(p.dst, p.src, p.bytes);
      {
        std::lock_guard<std::mutex> lk(m);
        pending -= 1;
      }
      cv_done.notify_one();
    }
  }
  // dst <- src, split between the helpers and the calling thread; returns when all of it is there
  void copy(void* dst, const void* src, size_t bytes) {
    const size_t n = th.size() + 1;
    const size_t share = ((bytes / n) + 63) & ~static_cast<size_t>(63);
    if (th.empty() || bytes < (1u << 18)) {
      memcpy(dst, src, bytes);
      return;
    }
    size_t off = 0;
    {
      std::lock_guard<std::mutex> lk(m);
      for (size_t k = 0; k < th.size(); ++k) {
        const size_t b = std::min(share, bytes - off);
        parts[k] = Part{static_cast<char*>(dst) + off, static_cast<const char*>(src) + off, b};
        off += b;
      }
      pending = static_cast<u32>(th.size());
      generation += 1;
    }
    cv_job.notify_all();
    if (off < bytes) memcpy(static_cast<char*>(dst) + off, static_cast<const char*>(src) + off, bytes - off);
    std::unique_lock<std::mutex> lk(m);
    cv_done.wait(lk, [&] { return pending == 0; });
  }
};
static std::mutex g_submitters_mutex;
static std::vector<Submitter*> g_submitters;
void cox_drain_submitters() {
  std::lock_guard<std::mutex> lk(g_submitters_mutex);
  for (Submitter* s : g_submitters) s->wait_outstanding(0);
}

struct cox_integrator {
  Submitter* submitter = nullptr;  // COX_SUBMIT_THREAD=0 turns it off
  CopyPool* copy_pool = nullptr;   // helpers for the bounce copy of pageable inputs: created with the first such frame
  bool h2d_kernel = false;         // COX_H2D=kernel: pinned inputs read by a copy kernel of h2d_groups workgroups instead of the copy engine
  int h2d_groups = 8;
  uint64_t last_big_tiles = 0, last_big_chunks = 0;  // of the last frame whose counters were folded (cox_integrator_update_stats)
  uint64_t host_ns = 0, host_wait_ns = 0, host_frames = 0;  // time the caller's thread spends inside the integrate call (enqueueing, waiting for a free slot)
  cox_projective* proj = nullptr;  // method == COX_METHOD_PROJECTIVE: everything else below stays empty
  cox_layer* layer = nullptr;
  FastState fast;
  cox_tsdf_config cfg;
  int method = 0;
  hipStream_t st[kNumStages] = {};  // stream of stage H, P, M, T, R, U (2 / 4 / 6 distinct ones, equal streams adjacent: see create)
  hipStream_t st_alt = nullptr;     // default map: ray generation (H, P, M) of the odd frame slots runs here, beside the even slots' on st[0]
  hipStream_t st_in = nullptr;      // host inputs (copies, depth conversion) on a stream of their own: where that leaves at most four active streams
  int n_streams = 4;
  hipEvent_t ev_a2 = nullptr;  // fast: front -> record stage
  FrameSet fs[kFrameSets];
  BundleSet bs[kStageSets];
  RecordSet rs[kStageSets];
  FrameParams* h_params = nullptr;  // pinned, kFrameSets entries
  u32 pcap = 0, rcap = 0, fh_cap = 0;
  u32 steps_max = 0;  // upper bound of a ray's step count for this configuration
  bool small_axis_cap = false;  // no ray can cross more than kAxisCapSmall - 2 planes of one axis
  u32 layer_generation = 0;     // cox_layer::generation the layer-sized buffers (touched_slots, ord_info, graphs) belong to
  hipEvent_t timeline_ref = nullptr;
  FILE* timeline = nullptr;
  u32 grid_apply = 8192, grid_merge = 4096, grid_touch = 2048;  // grid-stride kernels: any size is correct (COX_GRID_* for experiments)
  bool piece_path = false;      // COX_APPLY=pieces (merged without anti-grazing): pieces instead of records (k_touch_pieces / k_apply_pieces)
  u64 blocks_seen = 0, blocks_delta_max = 0;  // pool growth: last block count seen by the host, largest increase between two looks
  bool bucket_partition = true;  // records partitioned in ONE pass by tile id & 4095 (coarse voxels: few touched blocks); else one or two passes on the whole tile id
  u32 tile_shift = kTileShift;  // log2(voxels per tile) of the tile apply: 8 (one z slab of a block); COX_TILE=9: two
  bool piece_sort = false;      // pieces are walked and sorted, then expanded into records for k_apply_block (fine voxels; COX_PARTITION=pieces|records)
  SortWorkspace sort_pts_alt;   // bundling sort of the odd frame slots (runs beside the even slots' on st_alt)
  ScanWorkspace scanws_p;       // scan of the piece lengths
  ScanWorkspace scanws_h;       // piece partition: scan over the layer's hash slots (ordinals of the stamped blocks)
  u32 piece_cap = 0;
  u32 big_chunk = kBigChunk;
  bool split_big_tiles = true;  // ... and the tiles of more than kBigTileMin records classified chunk by chunk (k_big_classify); COX_SPLIT_TILES=0
  bool wave_apply = true;       // piece partition: k_apply_wave (a wave per tile) instead of k_apply_block (a workgroup per tile); COX_APPLY_WAVE=0
  u32 grid_apply_wave = 2048, wave_tile_max = 512;  // (COX_WAVE_TILE_MAX=256|512: tiles with more records go to k_apply_block)
  bool block_apply = true;      // records partitioned by block + k_apply_block; COX_APPLY=records selects the per-record kernels (full sort)
  // ordering against the caller's stream (cox_integrator_set_input_stream): the first stage waits for what the producer has
  // enqueued, and the producer's stream waits until the engine has read the inputs (stream-ordered allocators may then
  // recycle them)
  bool has_producer = false;
  hipStream_t producer = nullptr;
  hipEvent_t ev_producer = nullptr, ev_inputs_read = nullptr;
  float* own_xyz[kInputSets] = {};  // staging for host / depth inputs: one per bundle set (read by stages H .. M of the frame)
  uint8_t* own_rgba[kInputSets] = {};
  u32* depth_flag = nullptr;  // depth front end: valid pixels per tile
  u32* d_depth_n = nullptr;   // [kInputSets] point count of the depth image converted into each staging set: it stays on the device
  // Host inputs (cox_integrate_points_async, cox_integrate_depth_async) are copied on the ray-generation stream of the frame they
  // belong to -- with the default stream map that is the OTHER ray-generation stream than the previous frame's, so the copy of
  // frame t + 1 still runs beside the kernels of frame t.  A stream of their own was tried first and fell off the cliff every fifth
  // stream falls off (merged 6 500 -> 1 640 frames/s with the extra stream).
  float* own_depth[kInputSets] = {};       // staging for host depth images (cox_integrate_depth_async)
  uint8_t* own_depth_rgba[kInputSets] = {};
  hipEvent_t in_ready[kInputSets] = {};  // staging set k has been filled (the frame's first stage waits for it)
  hipEvent_t in_free[kInputSets] = {};   // the frame that read staging set k last has read it for the last time
  bool in_used[kInputSets] = {};
  float* pin_xyz[kInputSets] = {};       // pinned bounce buffers for pageable host inputs (a pinned caller buffer is copied from directly)
  uint8_t* pin_rgba[kInputSets] = {};
  u32 pin_cap = 0;
  SortWorkspace sort_pts, sort_rec, sort_vis, sort_vis1;  // sort_vis / sort_vis1: the fast integrator's visit sorts (round 0 on the start-set stream, round 1 on the solve's)
  ScanWorkspace scanws_a, scanws_b, scanws_d, scanws_f;
  u32 scan_cap = 0;
  Counters* h_ring = nullptr;  // pinned, kStatRing entries
  uint64_t frame_no = 0;       // frames enqueued
  cox_frame_stats last{};      // host-known part of the last frame's stats
  bool last_has_counts = false;
  bool last_count_on_device = false;  // the last frame's point count was produced on the device (fold_counters fetches it)
  // stage graphs: [stage][frame set index] (the bundle / record set index is the frame set index & 1)
  bool use_graphs = true;
  hipGraphExec_t graphs[kNumStages][kFrameSets] = {};
  // timing of individual kernels (bench roofline); forces eager launches
  bool profiling = false;
  u32 profile_every = 1;  // time the kernels of every n-th frame
  // one (begin, end) event pair per timed region of a frame, by kernel class (cox_kernel_class in coxgraph_hip.h)
  std::vector<std::pair<hipEvent_t, hipEvent_t>> class_events[COX_KERNEL_CLASSES];
  std::vector<hipEvent_t> event_pool[2];  // [1]: the submission thread's
  double class_ms[COX_KERNEL_CLASSES] = {};
  uint64_t class_regions[COX_KERNEL_CLASSES] = {};
};

template <typename T>
static int dev_realloc(T** p, size_t count) {
  if (*p) (void)hipFree(*p);
  *p = nullptr;
  if (count == 0) return COX_OK;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T));
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return (e == hipErrorOutOfMemory) ? COX_ERR_OUT_OF_MEMORY : COX_ERR_NO_DEVICE;
  }
  return COX_OK;
}
#define COX_TRY(expr)              \
  do {                             \
    int st_ = (expr);              \
    if (st_ != COX_OK) return st_; \
  } while (0)

static int alloc_sort_ws(SortWorkspace* ws, u64 capacity) {
  ws->tiles_cap = std::max<u32>(1, sort_num_tiles(capacity));
  COX_TRY(dev_realloc(&ws->counts, sort_counts_words(ws->tiles_cap)));
  if (!ws->totals) {
    COX_TRY(dev_realloc(&ws->totals, sort_totals_words()));
    COX_HIP(hipMemset(ws->totals, 0, sizeof(u32) * sort_totals_words()));  // every sort leaves it zero again
  }
  return COX_OK;
}

// upper bound of ray_length_in_steps + 1 for any ray this configuration can cast
static u32 max_steps_per_ray(const cox_integrator* I) {
  const double reach = static_cast<double>(I->cfg.max_ray_length_m) + static_cast<double>(I->cfg.default_truncation_distance);
  const double per_axis = std::floor(reach / static_cast<double>(I->layer->voxel_size)) + 3.0;
  const double s = 3.0 * per_axis + 1.0;
  return static_cast<u32>(std::min(s, 1.0e6));
}

// stream of a stage of the frame in frame slot `slot`
static inline hipStream_t stage_stream(const cox_integrator* I, int stage, int slot) {
  return (stage <= 2 && I->st_alt && (slot & 1)) ? I->st_alt : I->st[stage];
}

static int sync_all(cox_integrator* I) {
  if (I->submitter) {
    I->submitter->wait_outstanding(0);
    const int st = I->submitter->take_status();
    if (st != COX_OK) return st;
  }
  for (int k = 0; k < kNumStages; ++k)
    if (I->st[k] && (k == 0 || I->st[k] != I->st[k - 1])) COX_HIP(hipStreamSynchronize(I->st[k]));
  if (I->st_alt) COX_HIP(hipStreamSynchronize(I->st_alt));
  if (I->st_in) COX_HIP(hipStreamSynchronize(I->st_in));
  return COX_OK;
}

static void drop_graphs(cox_integrator* I) {
  for (auto& row : I->graphs)
    for (hipGraphExec_t& g : row) {
      if (g) (void)hipGraphExecDestroy(g);
      g = nullptr;
    }
}

static int ensure_capacity(cox_integrator* I, u32 n) {
  if (n <= I->pcap) return COX_OK;
  COX_TRY(sync_all(I));
  drop_graphs(I);  // they hold the old pointers and grids
  const u32 cap = std::max<u32>(n, 1024);
  I->fh_cap = next_pow2(static_cast<u64>(cap) + cap / 2);  // load factor <= 2/3 even if every point is its own bundle
  for (FrameSet& F : I->fs) {
    RayArrays& R = F.rays;
    COX_TRY(dev_realloc(&R.px, cap));
    COX_TRY(dev_realloc(&R.py, cap));
    COX_TRY(dev_realloc(&R.pz, cap));
    COX_TRY(dev_realloc(&R.w, cap));
    COX_TRY(dev_realloc(&R.color, cap));
    COX_TRY(dev_realloc(&R.flags, cap));
    COX_TRY(dev_realloc(&R.key, cap));
    COX_TRY(dev_realloc(&R.nsteps, cap));
    COX_TRY(dev_realloc(&R.rec_off, cap));
    COX_TRY(dev_realloc(&R.pbound, cap));
    COX_TRY(dev_realloc(&R.q, static_cast<size_t>(cap) * 8));
    COX_TRY(dev_realloc(&R.piece_off, cap));
    COX_TRY(dev_realloc(&F.fh_keys, static_cast<size_t>(I->fh_cap) + I->fh_cap / 2 + 1));  // u64 keys + u32 first-seq behind them
    F.fh_first = reinterpret_cast<u32*>(F.fh_keys + I->fh_cap);
    COX_HIP(hipMemset(F.fh_keys, 0xFF, sizeof(u64) * I->fh_cap + sizeof(u32) * I->fh_cap));  // empty; every frame leaves it empty again
  }
  for (BundleSet& B : I->bs) {
    COX_TRY(dev_realloc(&B.pslot, cap));
    for (int k = 0; k < 2; ++k) {
      COX_TRY(dev_realloc(&B.skey[k], cap));
      COX_TRY(dev_realloc(&B.sval[k], cap));
    }
    COX_TRY(dev_realloc(&B.head, cap));
    COX_TRY(dev_realloc(&B.bstart, cap));
  }
  for (int k = 0; k < kInputSets; ++k) {
    COX_TRY(dev_realloc(&I->own_xyz[k], static_cast<size_t>(cap) * 3));
    COX_TRY(dev_realloc(&I->own_rgba[k], static_cast<size_t>(cap) * 4));
    COX_TRY(dev_realloc(&I->own_depth[k], cap));
    COX_TRY(dev_realloc(&I->own_depth_rgba[k], static_cast<size_t>(cap) * 4));
  }
  COX_TRY(dev_realloc(&I->depth_flag, cap));
  COX_TRY(alloc_sort_ws(&I->sort_pts, cap));
  COX_TRY(alloc_sort_ws(&I->sort_pts_alt, cap));
  // records: the worst case (every ray at maximum length) always fits, so a frame can never overflow
  // unless that bound exceeds the 2^31 record limit of the 32-bit offsets
  I->steps_max = max_steps_per_ray(I);
  I->small_axis_cap = (I->steps_max - 1) / 3 + 2 <= kAxisCapSmall;  // steps_max = 3 * (planes per axis bound) + 1
  const u64 want = static_cast<u64>(cap) * I->steps_max;
  const u64 limit = 0x7FFFFFF0ull;
  const u32 rcap = static_cast<u32>(std::min(want, limit));
  const u32 wave_cap = rcap / 64 + 2;
  // pieces: every ray at the bound of the longest wave-walked ray (piece_bound); a frame of sequentially walked rays may need
  // more (one piece per step at worst) -- that frame is dropped and reported like a record overflow
  const u32 planes = (I->steps_max - 1) / 3;
  const u64 per_ray = static_cast<u64>(I->steps_max + 63) / 64 + (planes + 2) + 2 * ((planes + 2) / 16 + 1) + 4;
  const u32 piece_cap = static_cast<u32>(std::min<u64>(rcap, static_cast<u64>(cap) * per_ray));
  for (RecordSet& S : I->rs) {
    if (I->piece_path || I->piece_sort) {
      COX_TRY(dev_realloc(&S.lin8, static_cast<size_t>(piece_cap) * 32));
      for (int k = 0; k < 2; ++k) {
        COX_TRY(dev_realloc(&S.pkey[k], piece_cap));
        COX_TRY(dev_realloc(&S.pstart[k], piece_cap));
        COX_TRY(dev_realloc(&S.prl[k], piece_cap));
      }
      if (I->piece_sort) {
        COX_TRY(dev_realloc(&S.pbkey, piece_cap));
        COX_TRY(dev_realloc(&S.plen, piece_cap));
        COX_TRY(dev_realloc(&S.pdest, piece_cap));
        COX_TRY(dev_realloc(&S.rec_key[0], rcap));
        COX_TRY(dev_realloc(&S.rec_ray[0], rcap));
        S.big_chunk_cap = static_cast<u32>(rcap / kBigChunkMin) + kBigCap + 1u;
        COX_TRY(dev_realloc(&S.big_chunks, S.big_chunk_cap));
      }
      continue;
    }
    for (int k = 0; k < 2; ++k) {
      COX_TRY(dev_realloc(&S.rec_key[k], rcap));
      COX_TRY(dev_realloc(&S.rec_ray[k], rcap));
    }
    COX_TRY(dev_realloc(&S.piece_front, wave_cap));
    COX_TRY(dev_realloc(&S.piece_back, wave_cap));
    COX_TRY(dev_realloc(&S.piece_wsum, static_cast<size_t>(wave_cap) * 2));
  }
  I->piece_cap = piece_cap;
  COX_TRY(alloc_sort_ws(&I->sort_rec, (I->piece_path || I->piece_sort) ? piece_cap : rcap));
  if (I->piece_sort) COX_TRY(dev_realloc(&I->scanws_p.block_sums, scan_num_blocks(piece_cap) + 2));
  if (I->method == COX_METHOD_FAST) {
    FastState& X = I->fast;
    COX_TRY(dev_realloc(&X.fhash, cap));
    COX_TRY(dev_realloc(&X.fresh, cap));
    COX_TRY(dev_realloc(&X.rank, cap));
    COX_TRY(dev_realloc(&X.long_list, cap));
    for (int k = 0; k < kFrameSets; ++k) {
      COX_TRY(dev_realloc(&X.cap[k], cap));
      COX_TRY(dev_realloc(&X.reach[k], cap));
    }
    // round 0: kFastCap0 list slots per point; round 1: the grown lists -- bounded (a frame whose lists do not fit is redone by
    // the sequential kernel), not sized for "every ray walks to the sensor" as the uncapped lists of round 2 were (5 GB at 5 cm)
    const u32 vcap0 = static_cast<u32>(std::min<u64>(static_cast<u64>(cap) * kFastCap0Max, 0xFFFFFFF0ull));
    const u32 vcap1 = static_cast<u32>(std::min<u64>(rcap, std::max<u64>(static_cast<u64>(cap) * 64, 1ull << 24)));
    auto alloc_vs = [&](VisitSet& V, u32 vcap) -> int {
      for (int k = 0; k < 2; ++k) {
        COX_TRY(dev_realloc(&V.key[k], vcap));
        COX_TRY(dev_realloc(&V.val[k], vcap));
      }
      COX_TRY(dev_realloc(&V.vhash, vcap));
      COX_TRY(dev_realloc(&V.shash, vcap));
      COX_TRY(dev_realloc(&V.vray, vcap));
      COX_TRY(dev_realloc(&V.pos_of, vcap));
      COX_TRY(dev_realloc(&V.sinfo, vcap));
      COX_TRY(dev_realloc(&V.voff, cap));
      V.cap = vcap;
      return COX_OK;
    };
    COX_TRY(alloc_vs(X.vs0[0], vcap0));
    COX_TRY(alloc_vs(X.vs0[1], vcap0));
    COX_TRY(alloc_vs(X.vs1, vcap1));
    COX_TRY(alloc_sort_ws(&I->sort_vis, vcap0));
    COX_TRY(alloc_sort_ws(&I->sort_vis1, vcap1));
  }
  const u32 need_scan = scan_num_blocks(cap) + 2;
  if (need_scan > I->scan_cap) {
    COX_TRY(dev_realloc(&I->scanws_a.block_sums, need_scan));
    COX_TRY(dev_realloc(&I->scanws_b.block_sums, need_scan));
    COX_TRY(dev_realloc(&I->scanws_d.block_sums, need_scan));
    COX_TRY(dev_realloc(&I->scanws_f.block_sums, need_scan));
    I->scan_cap = need_scan;
  }
  I->rcap = rcap;
  I->pcap = cap;
  // hipMemset runs on the legacy default stream, which the engine's non-blocking streams do not wait for
  COX_HIP(hipDeviceSynchronize());
  return COX_OK;
}

static FrameParams make_params(const cox_integrator* I, const float T[7], u32 n, int freespace, const float* xyz, const uint8_t* rgba) {
  const cox_tsdf_config& c = I->cfg;
  FrameParams P;
  memset(&P, 0, sizeof(P));
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
  {
    const double t1 = static_cast<double>(c.default_truncation_distance) * (1.0 + 4.0 * 5.9604644775390625e-08 * (1.0 + static_cast<double>(c.max_weight)));
    float f = static_cast<float>(t1);
    if (static_cast<double>(f) < t1) f = std::nextafter(f, std::numeric_limits<float>::infinity());
    P.sat1 = f;  // >= saturating_update's threshold for every update weight >= 1
  }
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
  P.cast_from_origin = (I->method == COX_METHOD_FAST) ? 0 : 1;  // the fast integrator walks from the surface towards the sensor
  P.xyz = xyz;
  P.rgba = rgba;
  P.np2 = next_pow2(static_cast<u64>(n) + 1);
  return P;
}

static inline dim3 grid_for(u32 n, u32 block = 256, u32 cap = 0x7FFFFFFFu) { return dim3(std::min<u32>(cap, std::max<u32>(1, (n + block - 1) / block))); }

// HIP-event timing of one kernel class inside a frame (bench.py roofline): begin / end on the stream the kernels run on
// classes whose regions are opened by the submission thread (stages B1 / B2; not for fast, which has none) take their
// events from a pool of their own
static inline int event_pool_of(const cox_integrator* I, int cls) {
  return (I->submitter && (cls == COX_KC_APPLY || cls == COX_KC_TOUCH_EMIT || cls == COX_KC_RECORD_SORT || cls == COX_KC_FAST_SWEEPS || cls == COX_KC_FAST_ROUND1)) ? 1 : 0;
}
static thread_local u64 tl_frame_no = 0;  // the frame whose stages this thread is enqueueing (StageCtx::frame)
struct TimedRegion {
  cox_integrator* I;
  int cls;
  hipStream_t s;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  TimedRegion(cox_integrator* I_, int cls_, hipStream_t s_) : I(I_), cls(cls_), s(s_) {
    if (!(I->profiling && (tl_frame_no % I->profile_every == 0))) return;
    std::vector<hipEvent_t>& pool = I->event_pool[event_pool_of(I, cls)];
    // events come from a pool that the drain refills: creating them costs more than recording them
    auto take = [&]() -> hipEvent_t {
      if (!pool.empty()) {
        hipEvent_t e = pool.back();
        pool.pop_back();
        return e;
      }
      hipEvent_t e = nullptr;
      return hipEventCreate(&e) == hipSuccess ? e : nullptr;
    };
    e0 = take();
    e1 = take();
    if (!e0 || !e1) {
      e0 = e1 = nullptr;
      return;
    }
    (void)hipEventRecord(e0, s);
  }
  ~TimedRegion() {
    if (!e0) return;
    (void)hipEventRecord(e1, s);
    I->class_events[cls].emplace_back(e0, e1);
  }
};

// ---- the six stages; identical whether launched eagerly or captured into a graph: no argument depends on the frame ----
struct StageCtx {
  cox_integrator* I;
  FrameSet* F;
  BundleSet* B;
  RecordSet* S;
  int slot;
  u64 frame;  // frame number (which frames carry timing events)
  const u32* n_dev = nullptr;  // the frame's point count when it is known on the device only (depth front end); n_points of the parameter block is then an upper bound
};

static LayerView layer_view(const cox_layer* Lh) {
  return LayerView{Lh->voxels, Lh->ht_keys, Lh->ht_vals, Lh->ht_stamp, Lh->ht_ord, Lh->block_keys, Lh->d_nblocks, Lh->ht_cap - 1, static_cast<u32>(Lh->capacity)};
}
static int points_sort_passes(const cox_integrator* I) { return (ceil_log2(next_pow2(static_cast<u64>(I->pcap) + 1)) + 1 + 10) / 11; }

// exclusive scan of at most a few thousand ray step counts in ONE launch (the three-launch scan is latency-bound there)
__global__ void __launch_bounds__(1024) k_scan_small(const u32* __restrict__ in, u32* __restrict__ out, const u32* __restrict__ d_n, u32 n_max,
                                                     u32* __restrict__ d_total) {
  __shared__ u32 lds[16];
  const u32 n = min(*d_n, n_max);
  u32 carry = 0;
  for (u32 base = 0; base < n; base += 1024 * 4) {
    const u32 i0 = base + threadIdx.x * 4;
    u32 v[4], sum = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      v[q] = (i0 + q < n) ? in[i0 + q] : 0u;
      sum += v[q];
    }
    u32 total;
    u32 ex = carry + block_exclusive_scan<16>(sum, &total, lds);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (i0 + q < n) out[i0 + q] = ex;
      ex += v[q];
    }
    carry += total;
  }
  if (threadIdx.x == 0) *d_total = carry;
}

// piece path: the step counts and the piece bounds of the rays, both scanned in one launch
__global__ void __launch_bounds__(1024) k_scan_small2(const u32* __restrict__ in_a, u32* __restrict__ out_a, u32* __restrict__ total_a, const u32* __restrict__ in_b,
                                                      u32* __restrict__ out_b, u32* __restrict__ total_b, const u32* __restrict__ d_n, u32 n_max) {
  __shared__ u32 lds[16];
  const u32 n = min(*d_n, n_max);
  for (int which = 0; which < 2; ++which) {
    const u32* __restrict__ in = which ? in_b : in_a;
    u32* __restrict__ out = which ? out_b : out_a;
    u32 carry = 0;
    for (u32 base = 0; base < n; base += 1024 * 4) {
      const u32 i0 = base + threadIdx.x * 4;
      u32 v[4], sum = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[q] = (i0 + q < n) ? in[i0 + q] : 0u;
        sum += v[q];
      }
      u32 total;
      u32 ex = carry + block_exclusive_scan<16>(sum, &total, lds);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (i0 + q < n) out[i0 + q] = ex;
        ex += v[q];
      }
      carry += total;
    }
    if (threadIdx.x == 0) *(which ? total_b : total_a) = carry;
  }
}

static int stage_hash(const StageCtx& c, hipStream_t s) {
  cox_integrator* I = c.I;
  FrameSet& F = *c.F;
  BundleSet& B = *c.B;
  const bool by_value = (I->method == COX_METHOD_MERGED) && !I->use_graphs;
  if (!by_value) {
    COX_HIP(hipMemcpyAsync(F.d_params, &I->h_params[c.slot], sizeof(FrameParams), hipMemcpyHostToDevice, s));
    if (c.n_dev) hipLaunchKernelGGL(k_params_count, dim3(1), dim3(1), 0, s, F.d_params, c.n_dev);
  }
  COX_HIP(hipMemsetAsync(F.cnt, 0, sizeof(Counters), s));
  if (I->method == COX_METHOD_MERGED) {
    const u32 n = I->pcap;  // grids cover the capacity; the kernels stop at the frame's own point count
    {
      TimedRegion t(I, COX_KC_BUNDLE_HASH, s);
      // with anti-grazing the hash is read again by touch / emit (stage B1), after this frame's pslot may have been reused:
      // that (rare) configuration keeps the memset; otherwise the frame cleans up after itself (in k_bundle_count, stage M)
      const bool self_clean = !I->cfg.enable_anti_grazing;
      if (!self_clean) COX_HIP(hipMemsetAsync(F.fh_keys, 0xFF, sizeof(u64) * I->fh_cap + sizeof(u32) * I->fh_cap, s));
      hipLaunchKernelGGL(k_bundle_insert, grid_for(n), dim3(256), 0, s, F.d_params, I->h_params[c.slot], by_value ? 1 : 0, c.n_dev, F.fh_keys, F.fh_first,
                         I->fh_cap - 1, B.pslot, F.cnt);
      hipLaunchKernelGGL(k_bundle_keys, grid_for(n), dim3(256), 0, s, F.d_params, F.fh_keys, F.fh_first, B.pslot, B.skey[0], B.sval[0], B.sort_info);
      (void)self_clean;  // (the frame cleans its hash up in k_bundle_count, stage M)
    }
  }
  return COX_OK;
}
static int stage_point_sort(const StageCtx& c, hipStream_t s) {
  cox_integrator* I = c.I;
  FrameSet& F = *c.F;
  BundleSet& B = *c.B;
  if (I->method == COX_METHOD_MERGED) {
    const u32 n = I->pcap;
    TimedRegion t(I, COX_KC_POINT_SORT, s);
    (void)radix_sort_pairs<11>(B.skey[0], B.sval[0], B.skey[1], B.sval[1], &F.d_params->n_points, n, n, 0, true, points_sort_passes(I),
                               (I->st_alt && (c.slot & 1)) ? I->sort_pts_alt : I->sort_pts, B.sort_info, s);
  }
  return COX_OK;
}
static int stage_merge(const StageCtx& c, hipStream_t s) {
  cox_integrator* I = c.I;
  FrameSet& F = *c.F;
  BundleSet& B = *c.B;
  const u32 n = I->pcap;
  if (I->method == COX_METHOD_MERGED) {
    BundleView V{{B.skey[0], B.skey[1]}, {B.sval[0], B.sval[1]}, B.sort_info};
    // bundle boundaries: heads per tile, then the starts
    const dim3 gt(std::max<u32>(1, (n + kBoundTile - 1) / kBoundTile));
    hipLaunchKernelGGL(k_bundle_count, gt, dim3(256), 0, s, F.d_params, V, B.head, B.pslot, F.fh_keys, F.fh_first, I->cfg.enable_anti_grazing ? 0 : 1);
    hipLaunchKernelGGL(k_bundle_starts, gt, dim3(256), 0, s, F.d_params, V, B.head, B.bstart, F.cnt);
    {
      TimedRegion t(I, COX_KC_MERGE, s);
      hipLaunchKernelGGL(k_bundle_merge, dim3(I->grid_merge), dim3(256), 0, s, F.d_params, V, B.bstart, F.rays, F.cnt,
                         (I->piece_path || I->piece_sort) ? (I->small_axis_cap ? kAxisCapSmall : kAxisCapLarge) : 0u);
    }
    // record offsets (and piece slots) of the rays: scanned here, on the ray-generation stream -- two frames' worth of it run
    // side by side, the layer-update chain is the one that bounds the frame rate
    const bool odd = I->st_alt && (c.slot & 1);
    const ScanWorkspace& ws = odd ? I->scanws_a : I->scanws_b;
    if (!(I->piece_path || I->piece_sort)) {
      hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(1024), 0, s, F.rays.nsteps, F.rays.rec_off, &F.cnt->n_ray_slots, I->pcap, &F.cnt->n_records);
    } else if (I->small_axis_cap) {  // a few thousand bundles: one launch, one workgroup
      hipLaunchKernelGGL(k_scan_small2, dim3(1), dim3(1024), 0, s, F.rays.nsteps, F.rays.rec_off, &F.cnt->n_records, F.rays.pbound, F.rays.piece_off,
                         &F.cnt->n_piece_slots, &F.cnt->n_ray_slots, I->pcap);
    } else {  // fine voxels, 10^4-10^5 bundles: the three-kernel scans (the one-workgroup scan took 75 us at 1 cm)
      exclusive_scan_u32(F.rays.nsteps, F.rays.rec_off, &F.cnt->n_ray_slots, I->pcap, I->pcap, &F.cnt->n_records, ws, s);
      exclusive_scan_u32(F.rays.pbound, F.rays.piece_off, &F.cnt->n_ray_slots, I->pcap, I->pcap, &F.cnt->n_piece_slots, ws, s);
    }
  } else {
    hipLaunchKernelGGL(k_rays_simple, grid_for(n), dim3(256), 0, s, F.d_params, F.rays, F.cnt);
  }
  return COX_OK;
}
static int stage_touch(const StageCtx& c, hipStream_t s) {
  cox_integrator* I = c.I;
  FrameSet& F = *c.F;
  RecordSet& S = *c.S;
  const LayerView L = layer_view(I->layer);
  const u32 fh_mask = I->fh_cap - 1;
  const bool merged = I->method == COX_METHOD_MERGED;
  if (I->piece_path || I->piece_sort) {
    const PieceArrays PA{S.pkey[0], S.pstart[0], S.prl[0], S.pbkey};
    {
      TimedRegion t_walk(I, COX_KC_TOUCH_EMIT, s);
#define COX_LAUNCH_WALK(CAP, DEFER)                                                                                                                       \
  hipLaunchKernelGGL((k_touch_pieces<CAP, DEFER>), dim3(2048), dim3(256), 0, s, F.d_params, F.rays, L, S.touched_slots, S.lin8, PA, I->rcap, I->piece_cap, \
                     F.cnt, I->layer->d_err)
      if (I->piece_sort) {
        if (I->small_axis_cap)
          COX_LAUNCH_WALK(kAxisCapSmall, true);
        else
          COX_LAUNCH_WALK(kAxisCapLarge, true);
        hipLaunchKernelGGL(k_piece_touch, dim3(2048), dim3(256), 0, s, F.d_params, L, PA, I->rcap, I->piece_cap, S.touched_slots, F.cnt, I->layer->d_err);
        const u32 ht_cap = I->layer->ht_cap;
        hipLaunchKernelGGL(k_ord_flags, grid_for(ht_cap, 256, 4096), dim3(256), 0, s, F.d_params, L);
        exclusive_scan_u32(L.ht_ord, L.ht_ord, nullptr, ht_cap, ht_cap, &F.cnt->n_touched, I->scanws_h, s);
        hipLaunchKernelGGL(k_ord_scatter, grid_for(ht_cap, 256, 4096), dim3(256), 0, s, F.d_params, L, S.touched_slots);
      } else if (I->small_axis_cap) {
        COX_LAUNCH_WALK(kAxisCapSmall, false);
      } else {
        COX_LAUNCH_WALK(kAxisCapLarge, false);
      }
#undef COX_LAUNCH_WALK
      hipLaunchKernelGGL(k_piece_keys, dim3(1024), dim3(256), 0, s, L, S.pkey[0], I->rcap, I->piece_cap, F.cnt, S.sort_info, S.touched_slots, S.ord_info,
                         I->piece_sort ? I->tile_shift - kTileShift : 0u);
    }
    return COX_OK;
  }
  TimedRegion t_walk(I, COX_KC_TOUCH_EMIT, s);
  if (!merged)  // (merged: the offsets are scanned at the end of stage M, off the layer-update chain)
    exclusive_scan_u32(F.rays.nsteps, F.rays.rec_off, &F.cnt->n_ray_slots, I->pcap, I->pcap, &F.cnt->n_records, I->scanws_b, s);
  if (merged) {
    // few long rays: one wave per ray (parallel DDA); the walk found by touch is handed to emit through the spare sort buffer
    if (I->small_axis_cap)
      hipLaunchKernelGGL(k_touch_wave<kAxisCapSmall>, dim3(I->grid_touch), dim3(256), 0, s, F.d_params, F.rays, L, S.touched_slots, S.rec_key[1], I->rcap, F.cnt,
                         I->layer->d_err, F.fh_keys, fh_mask);
    else
      hipLaunchKernelGGL(k_touch_wave<kAxisCapLarge>, dim3(2048), dim3(256), 0, s, F.d_params, F.rays, L, S.touched_slots, S.rec_key[1], I->rcap, F.cnt,
                         I->layer->d_err, F.fh_keys, fh_mask);
    hipLaunchKernelGGL(k_emit_wave, dim3(I->grid_touch), dim3(256), 0, s, F.d_params, F.rays, L, S.rec_key[1], S.rec_key[0], S.rec_ray[0], I->rcap, F.cnt, S.sort_info,
                       F.fh_keys, fh_mask, S.touched_slots, S.ord_info, I->block_apply ? static_cast<int>(I->tile_shift | (I->bucket_partition ? 256u : 0u)) : 0);
  } else {
    hipLaunchKernelGGL(k_touch, grid_for(I->pcap, 256, 8192), dim3(256), 0, s, F.d_params, F.rays, L, S.touched_slots, F.cnt, I->layer->d_err, F.fh_keys, fh_mask);
    hipLaunchKernelGGL(k_emit, grid_for(I->pcap, 256, 8192), dim3(256), 0, s, F.d_params, F.rays, L, S.rec_key[0], S.rec_ray[0], I->rcap, F.cnt, S.sort_info,
                       F.fh_keys, fh_mask, S.touched_slots, S.ord_info, I->block_apply ? static_cast<int>(I->tile_shift | (I->bucket_partition ? 256u : 0u)) : 0);
  }
  return COX_OK;
}
static int stage_record_sort(const StageCtx& c, hipStream_t s) {
  cox_integrator* I = c.I;
  FrameSet& F = *c.F;
  RecordSet& S = *c.S;
  TimedRegion t_sort(I, COX_KC_RECORD_SORT, s);
  if (I->piece_path || I->piece_sort) {
    // 4 + ceil(log2(touched blocks + 1)) key bits: one pass up to 255 touched blocks, two beyond
    (void)radix_sort_pairs<12>(S.pkey[0], S.pstart[0], S.pkey[1], S.pstart[1], &F.cnt->n_piece_slots, I->piece_cap, std::min<u32>(I->piece_cap, 1u << 21), 0,
                               true, 2, I->sort_rec, S.sort_info, s, S.prl[0], S.prl[1]);
    if (I->piece_sort) {  // sorted pieces -> records in tile order
      const u32 hint = std::min<u32>(I->piece_cap, 1u << 22);
      hipLaunchKernelGGL(k_piece_lens, grid_for(hint, 256, 8192), dim3(256), 0, s, S.pkey[0], S.pkey[1], S.prl[0], S.prl[1], S.sort_info, S.plen, F.cnt);
      exclusive_scan_u32(S.plen, S.pdest, &F.cnt->n_piece_slots, I->piece_cap, hint, &F.cnt->n_expanded, I->scanws_p, s);
      hipLaunchKernelGGL(k_piece_expand, dim3(8192), dim3(256), 0, s, S.pkey[0], S.pkey[1], S.pstart[0], S.pstart[1], S.prl[0], S.prl[1], S.sort_info, S.pdest,
                         S.lin8, S.rec_key[0], S.rec_ray[0], F.cnt);
      hipLaunchKernelGGL(k_piece_tile_ranges, grid_for(hint, 256, 8192), dim3(256), 0, s, S.pkey[0], S.pkey[1], S.sort_info, S.plen, S.pdest, S.blk_beg, S.blk_end, F.cnt,
                         I->tile_shift - kTileShift);
    }
    return COX_OK;
  }
  // 12 + ceil(log2(touched blocks + 1)) key bits, known on the device only: digits of up to 12 bits, so two passes up to
  // 4095 touched blocks (23 bits = 12 + 12 at 5 cm), three beyond.
  // Grid hint: ~2 M records keep every CU busy; larger frames grid-stride.
  // Block apply: ONE stable pass on the low 12 bits of the tile id (block ordinal, z slab) whatever the number of touched
  // blocks -- a partition into 4096 buckets; beyond 255 touched blocks several tiles share a bucket and the apply takes them
  // in turn (k_block_starts).  No second pass is ever launched (it used to be launched every frame just to exit at 5 cm).
  (void)radix_sort_pairs<12>(S.rec_key[0], S.rec_ray[0], S.rec_key[1], S.rec_ray[1], &F.cnt->n_records, I->rcap, std::min<u32>(I->rcap, 1u << 21), 0, true,
                             I->block_apply ? (I->bucket_partition ? 1 : 2) : 3, I->sort_rec, S.sort_info, s);
  return COX_OK;
}
static int stage_apply(const StageCtx& c, hipStream_t s) {
  cox_integrator* I = c.I;
  FrameSet& F = *c.F;
  RecordSet& S = *c.S;
  const LayerView L = layer_view(I->layer);
  RecordView V{{S.rec_key[0], S.rec_key[1]}, {S.rec_ray[0], S.rec_ray[1]}, S.sort_info, &F.cnt->n_records};
  if (I->piece_sort) V = RecordView{{S.rec_key[0], S.rec_key[0]}, {S.rec_ray[0], S.rec_ray[0]}, S.sort_info, &F.cnt->n_expanded};  // one buffer: the parity is the piece sort's
  TimedRegion t(I, COX_KC_APPLY, s);
  if (I->piece_path) {
    const RecordView KV{{S.pkey[0], S.pkey[1]}, {nullptr, nullptr}, S.sort_info, &F.cnt->n_piece_slots};
    const PieceView PV{{S.pkey[0], S.pkey[1]}, {S.pstart[0], S.pstart[1]}, {S.prl[0], S.prl[1]}, S.sort_info};
    hipLaunchKernelGGL(k_block_starts, dim3(1024), dim3(256), 0, s, KV, S.blk_beg, S.blk_end, F.cnt, 0u, 0xFFFFFFFFu);
    hipLaunchKernelGGL(k_apply_pieces, dim3(16384), dim3(kPT), 0, s, F.d_params, F.rays, L, S.ord_info, PV, S.lin8, S.blk_beg, S.blk_end, F.cnt, I->layer->d_err,
                       I->layer->h_nblocks);
    return COX_OK;
  }
  if (I->block_apply) {
    if (!I->piece_sort)  // (piece partition: the ranges come with the expansion, k_piece_tile_ranges)
      hipLaunchKernelGGL(k_block_starts, dim3(1024), dim3(256), 0, s, V, S.blk_beg, S.blk_end, F.cnt, I->tile_shift, I->bucket_partition ? 4095u : 0xFFFFFFFFu);
    u32 min_records = 0;
    BigTiles big{nullptr, nullptr, nullptr, 0, 0, nullptr, 0};
#define COX_LAUNCH_APPLY(Q, TS, BUCKET)                                                                                                                       \
  hipLaunchKernelGGL((k_apply_block<Q, TS, BUCKET>), dim3(I->grid_apply), dim3(kBT), 0, s, F.d_params, F.rays, L, S.ord_info, V, S.blk_beg, S.blk_end, F.cnt, \
                     I->layer->d_err, I->layer->h_nblocks, min_records, big)
    if (I->method == COX_METHOD_MERGED) {  // the merge leaves RayArrays::q
      if (I->piece_sort) {                 // tile ranges straight from the pieces (k_piece_tile_ranges)
        if (I->tile_shift == 9)
          COX_LAUNCH_APPLY(true, 9, false);
        else if (I->wave_apply) {  // fine voxels (cox_apply_tile.hpp): the largest tiles classified chunk by chunk by the whole chip, a
                                   // workgroup per tile for the large ones, a wave per tile for the rest
          min_records = I->wave_tile_max + 1;
          if (I->split_big_tiles && S.big_chunks) {
            big = BigTiles{S.big_of_tile, S.big_acc, S.big_chunks, S.big_chunk_cap, I->big_chunk, S.blk_list, I->wave_tile_max};
            hipLaunchKernelGGL(k_big_tiles, dim3(512), dim3(256), 0, s, S.blk_beg, S.blk_end, S.ord_info, F.cnt, big);
            hipLaunchKernelGGL((k_big_classify<true>), dim3(I->grid_apply), dim3(kBT), 0, s, F.d_params, F.rays, S.ord_info, V, S.blk_end, F.cnt, big);
          }
          COX_LAUNCH_APPLY(true, 8, false);
          if (I->wave_tile_max == 256)
            hipLaunchKernelGGL((k_apply_wave<256>), dim3(std::max(8u, I->grid_apply_wave & ~7u)), dim3(kWaveTileWaves * 64), 0, s, F.d_params, F.rays, L, S.ord_info, V,
                               S.blk_beg, S.blk_end, F.cnt, I->layer->d_err, I->layer->h_nblocks);
          else
            hipLaunchKernelGGL((k_apply_wave<512>), dim3(std::max(8u, I->grid_apply_wave & ~7u)), dim3(kWaveTileWaves * 64), 0, s, F.d_params, F.rays, L, S.ord_info, V,
                               S.blk_beg, S.blk_end, F.cnt, I->layer->d_err, I->layer->h_nblocks);
        } else
          COX_LAUNCH_APPLY(true, 8, false);
      } else if (!I->bucket_partition) {
        if (I->tile_shift == 9)
          COX_LAUNCH_APPLY(true, 9, false);
        else
          COX_LAUNCH_APPLY(true, 8, false);
      } else if (I->tile_shift == 9) {
        COX_LAUNCH_APPLY(true, 9, true);
      } else {
        COX_LAUNCH_APPLY(true, 8, true);
      }
    } else if (I->bucket_partition) {
      COX_LAUNCH_APPLY(false, 8, true);
    } else {
      COX_LAUNCH_APPLY(false, 8, false);
    }
#undef COX_LAUNCH_APPLY
    return COX_OK;
  }
  hipLaunchKernelGGL(k_apply_eval, dim3(4096), dim3(256), 0, s, F.d_params, F.rays, L, S.ord_info, V, S.piece_front, S.piece_back, S.piece_wsum, F.cnt);
  hipLaunchKernelGGL(k_apply_long, dim3(2048), dim3(256), 0, s, F.d_params, F.rays, L, S.ord_info, V, S.piece_front, S.piece_back, S.piece_wsum, F.cnt,
                     I->layer->d_err, I->layer->h_nblocks);
  return COX_OK;
}
typedef int (*StageFn)(const StageCtx&, hipStream_t);
static const StageFn kStages[kNumStages] = {stage_hash, stage_point_sort, stage_merge, stage_touch, stage_record_sort, stage_apply};

// launch stage k of the frame in ctx on its stream: replay its graph (capturing it first if needed) or go eager
static int run_stage(int k, const StageCtx& c) {
  cox_integrator* I = c.I;
  hipStream_t s = stage_stream(I, k, c.slot);
  tl_frame_no = c.frame;
  if (!I->use_graphs || I->profiling || (k == 0 && c.n_dev)) return kStages[k](c, s);
  hipGraphExec_t& gx = I->graphs[k][c.slot];
  if (!gx) {
    hipGraph_t g = nullptr;
    COX_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    const int st = kStages[k](c, s);
    const hipError_t e = hipStreamEndCapture(s, &g);
    if (st != COX_OK || e != hipSuccess || !g) {
      if (g) (void)hipGraphDestroy(g);
      (void)hipGetLastError();
      I->use_graphs = false;  // capture not possible here: stay eager from now on
      return kStages[k](c, s);
    }
    const hipError_t ei = hipGraphInstantiate(&gx, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (ei != hipSuccess) {
      gx = nullptr;
      (void)hipGetLastError();
      I->use_graphs = false;
      return kStages[k](c, s);
    }
  }
  COX_HIP(hipGraphLaunch(gx, s));
  return COX_OK;
}

// ---- fast: one frame = three chains, each on a stream of its own, each in frame order (cox_fast.hpp):
//   front   (st[0], caller's thread)      start set -> ray list -> round-0 candidate lists (kFastCap0 steps per ray), sorted by slot;
//                                         touches the start table and this frame's own buffers only
//   solve   (st[1], submission thread)    relaxation of the observed set over the capped lists, whole walks for the rays that got
//                                         through them, relaxation again, commit to the observed table; frame t + 1's solve needs
//                                         the table frame t leaves, so this chain sets the frame rate
//   update  (st[3], submission thread)    the record pipeline of `simple` over every ray's first reach[r] voxels
// No host round trip anywhere: convergence, growth and overflow are decided on the device, the sweep launches of a round are
// enqueued speculatively (they return at once behind a launch that moved nothing), and a frame whose relaxation did not settle
// within them is redone by k_fast_sequential, exactly.
struct FastJob {
  FastFrame FF;
  bool wipe_obs = false;  // ApproxHashSet::resetApproxSet reached its full_reset_threshold: the table is zeroed
  int vs = 0;             // which round-0 visit set the front filled
};
static int fast_front(const StageCtx& c, FastJob* job) {
  cox_integrator* I = c.I;
  FrameSet& F = *c.F;
  BundleSet& B = *c.B;
  FastState& X = I->fast;
  const hipStream_t s = I->st[0];
  // FastTsdfIntegrator::integratePointCloud: both sets are reset every clear_checks_every_n_frames frames;
  // ApproxHashSet::resetApproxSet bumps the offset and wipes the table every 10 000 resets
  if (++X.reset_counter >= I->cfg.clear_checks_every_n_frames) {
    X.reset_counter = 0;
    if (++X.off_start >= kFastFullReset) {
      COX_HIP(hipMemsetAsync(X.table_start, 0, sizeof(u64) * kFastSlots, s));
      X.off_start = 0;
    }
    if (++X.off_obs >= kFastFullReset) {
      job->wipe_obs = true;  // (on the solve's stream, between the previous frame's commit and this frame's sweeps)
      X.off_obs = 0;
    }
  }
  const FastFrame FF{X.off_start, X.off_obs, I->cfg.max_consecutive_ray_collisions};
  job->FF = FF;
  job->vs = static_cast<int>(X.frames & 1u);
  X.frames += 1;
  VisitSet& V0 = X.vs0[job->vs];
  if (V0.used) COX_HIP(hipStreamWaitEvent(s, V0.done, 0));  // the solve of frame t-2 is done with this visit set
  const u32 n = I->pcap;
  const dim3 gp = grid_for(n, 256, 4096), gw(4096);  // gw: one wave per ray, grid-stride
  COX_HIP(hipMemcpyAsync(F.d_params, &I->h_params[c.slot], sizeof(FrameParams), hipMemcpyHostToDevice, s));
  if (c.n_dev) hipLaunchKernelGGL(k_params_count, dim3(1), dim3(1), 0, s, F.d_params, c.n_dev);
  COX_HIP(hipMemsetAsync(F.cnt, 0, sizeof(Counters), s));
  {
    // start set: sort the points by slot (21 key bits: slot + "not integrated"), compare neighbours
    TimedRegion t(I, COX_KC_FAST_START, s);
    hipLaunchKernelGGL(k_fast_points, gp, dim3(256), 0, s, F.d_params, FF, X.fhash, B.skey[0], B.sval[0], F.cnt);
    const int pp = radix_sort_pairs<11>(B.skey[0], B.sval[0], B.skey[1], B.sval[1], &F.d_params->n_points, n, n, kFastSlotBits + 1, false, 2, I->sort_pts, nullptr, s);
    hipLaunchKernelGGL(k_fast_start_flags, gp, dim3(256), 0, s, F.d_params, B.skey[pp], B.sval[pp], X.fhash, X.table_start, X.fresh);
    hipLaunchKernelGGL(k_fast_start_commit, gp, dim3(256), 0, s, F.d_params, B.skey[pp], B.sval[pp], X.fhash, X.table_start);
    exclusive_scan_u32(X.fresh, X.rank, &F.d_params->n_points, n, n, &F.cnt->n_rays, I->scanws_a, s);
    hipLaunchKernelGGL(k_fast_rays, gp, dim3(256), 0, s, F.d_params, X.fresh, X.rank, F.rays, X.cap[c.slot], X.reach[c.slot], X.cap0, V0.cap, F.cnt);
  }
  {
    // round 0's candidate visits: the first kFastCap0 voxels of every ray's walk, sorted by slot of the observed set (they depend on
    // this frame's rays only, so they are made here, beside the previous frame's solve)
    TimedRegion t(I, COX_KC_FAST_VISITS, s);
    if (I->small_axis_cap)
      hipLaunchKernelGGL(k_fast_visits<kAxisCapSmall>, gw, dim3(256), 0, s, F.d_params, FF, F.rays, V0.vhash, V0.key[0], V0.val[0], V0.vray, V0.voff, X.cap[c.slot],
                         X.cap0, 0, F.cnt);
    else
      hipLaunchKernelGGL(k_fast_visits<kAxisCapLarge>, gw, dim3(256), 0, s, F.d_params, FF, F.rays, V0.vhash, V0.key[0], V0.val[0], V0.vray, V0.voff, X.cap[c.slot],
                         X.cap0, 0, F.cnt);
    V0.sorted = radix_sort_pairs<11>(V0.key[0], V0.val[0], V0.key[1], V0.val[1], &F.cnt->fast.n_visits[0], V0.cap, std::min<u32>(V0.cap, 1u << 19), kFastSlotBits + 1,
                                     false, 2, I->sort_vis, nullptr, s);
    hipLaunchKernelGGL(k_fast_inverse, dim3(512), dim3(256), 0, s, V0.key[V0.sorted], V0.val[V0.sorted], V0.vray, V0.vhash, V0.voff, V0.pos_of, V0.sinfo, V0.shash,
                       &F.cnt->fast.n_visits[0], V0.cap);
  }
  return COX_OK;
}
static inline FastVisits fast_view(const VisitSet& V) { return FastVisits{V.key[V.sorted], V.sinfo, V.shash, V.voff, V.pos_of}; }
static int fast_solve(const StageCtx& c, const FastJob& job) {
  cox_integrator* I = c.I;
  FrameSet& F = *c.F;
  FastState& X = I->fast;
  const hipStream_t s = I->st[1];
  const FastFrame FF = job.FF;
  VisitSet& V0 = X.vs0[job.vs];
  VisitSet& V1 = X.vs1;
  u32* cap = X.cap[c.slot];
  u32* reach = X.reach[c.slot];
  FastCtl* ctl = &F.cnt->fast;
  const u32* n_rays = &F.cnt->n_rays;
  const int mc = I->cfg.max_consecutive_ray_collisions;
  const dim3 gr = grid_for(I->pcap, 256, 1024), gw(4096);
  if (job.wipe_obs) COX_HIP(hipMemsetAsync(X.table_obs, 0, sizeof(u64) * kFastSlots, s));
  {
    TimedRegion t(I, COX_KC_FAST_SWEEPS, s);
    hipLaunchKernelGGL(k_fast_relax, dim3(X.relax_groups), dim3(kFastRelaxThreads), 0, s, fast_view(V0), mc, cap, F.rays.nsteps, X.table_obs, reach, ctl, 0,
                       X.long_list, n_rays, X.fences);
  }
  for (int round = 1; round < X.rounds; ++round) {
    // round 1 (every kernel returns at once when round 0 left nobody at the end of a capped list): whole walks for those rays, all
    // lists sorted again, relaxation from round 0's fixed point; further rounds (fine voxels: COX_FAST_ROUNDS) the same for the rays
    // that have outgrown their lists since
    TimedRegion t(I, COX_KC_FAST_ROUND1, s);
    hipLaunchKernelGGL(k_fast_grow, gr, dim3(256), 0, s, F.rays.nsteps, cap, reach, ctl, round, X.cap1, X.long_list, n_rays);
    if (I->pcap <= (1u << 15) || X.rounds == 2) {
      // one workgroup, 4 096 rays at a time: a frame at coarse voxels starts a few thousand rays (the three-launch scan and its
      // bookkeeping kernel were four of the solve chain's eighteen launches; a frame in which every point of a large cloud starts a ray
      // pays 0.2 ms here instead)
      hipLaunchKernelGGL(k_fast_scan_caps, dim3(1), dim3(1024), 0, s, cap, V1.voff, ctl, round, I->pcap, V1.cap);
    } else {  // (rays by the ten thousand, fine voxels: the three-launch scan; it covers scan_n rays, none without growth)
      exclusive_scan_u32(cap, V1.voff, &ctl->scan_n[round], I->pcap, I->pcap, &ctl->n_visits[round], I->scanws_f, s);
      hipLaunchKernelGGL(k_fast_scan_caps_done, dim3(1), dim3(1), 0, s, ctl, round, V1.cap);
    }
    if (I->small_axis_cap)
      hipLaunchKernelGGL(k_fast_visits<kAxisCapSmall>, gw, dim3(256), 0, s, F.d_params, FF, F.rays, V1.vhash, V1.key[0], V1.val[0], V1.vray, V1.voff, cap, 0u, round,
                         F.cnt);
    else
      hipLaunchKernelGGL(k_fast_visits<kAxisCapLarge>, gw, dim3(256), 0, s, F.d_params, FF, F.rays, V1.vhash, V1.key[0], V1.val[0], V1.vray, V1.voff, cap, 0u, round,
                         F.cnt);
    V1.sorted = radix_sort_pairs<11>(V1.key[0], V1.val[0], V1.key[1], V1.val[1], &ctl->n_visits[round], V1.cap, std::min<u32>(V1.cap, 1u << 19), kFastSlotBits + 1, false,
                                     2, I->sort_vis1, nullptr, s);
    hipLaunchKernelGGL(k_fast_inverse, dim3(512), dim3(256), 0, s, V1.key[V1.sorted], V1.val[V1.sorted], V1.vray, V1.vhash, V1.voff, V1.pos_of, V1.sinfo, V1.shash,
                       &ctl->n_visits[round], V1.cap);
    hipLaunchKernelGGL(k_fast_relax, dim3(X.relax_groups), dim3(kFastRelaxThreads), 0, s, fast_view(V1), mc, cap, F.rays.nsteps, X.table_obs, reach, ctl, round,
                       X.long_list, n_rays, X.fences);
  }
  // (the relaxation packs (ray, step) into one word: a configuration whose walks or ray counts do not fit takes the sequential kernel)
  const bool packed_ok = I->steps_max <= kFastStepMask && I->pcap <= (1u << (32 - kFastStepBits));
  hipLaunchKernelGGL(k_fast_sequential, dim3(1), dim3(64), 0, s, F.d_params, FF, F.rays, X.table_obs, reach, ctl, (X.force_sequential || !packed_ok) ? 1 : 0, n_rays);
  hipLaunchKernelGGL(k_fast_obs_commit, dim3(512), dim3(256), 0, s, fast_view(V0), fast_view(V1), reach, X.table_obs, ctl, V0.cap, V1.cap, X.d_stats, F.rays, F.cnt);
  COX_HIP(hipEventRecord(V0.done, s));
  V0.used = true;
  return COX_OK;
}

// Layer::allocateBlockPtrByIndex never fails in voxblox.  The last kernel of every frame leaves the block count in a pinned
// word; once the pool is half full (or the recent rate of allocation says it will be) it is doubled before the next frame is
// enqueued (a frame that still runs out reports
// COX_ERR_POOL_EXHAUSTED at sync, and so does every later frame that meets one of its blocks).  The buffers sized by the
// layer's hash capacity follow the layer whenever it has been reallocated (by this or by cox_layer_reserve / upload).
static int follow_layer(cox_integrator* I) {
  cox_layer* Lh = I->layer;
  // The count the host sees is up to a dozen frames old (frames in flight), so "half full" alone reacts too late when frames
  // allocate fast (2 cm: hundreds of blocks per frame): the largest increase seen between two looks, times the frames that may
  // be in flight, has to fit as well (found by the 600-frame soak test with a small pool).
  const u64 n_seen = *Lh->h_nblocks;
  if (n_seen > I->blocks_seen) I->blocks_delta_max = std::max<u64>(I->blocks_delta_max, n_seen - I->blocks_seen);
  I->blocks_seen = n_seen;
  const u64 need = std::max<u64>(2 * n_seen, n_seen + 12 * I->blocks_delta_max);  // (host ahead of stage H by <= 6 frames, H ahead of U by <= 6)
  if (Lh->auto_grow && need > Lh->capacity && Lh->capacity < (1ull << 26)) {
    COX_TRY(sync_all(I));
    u64 cap = Lh->capacity;
    while (cap < need && cap < (1ull << 26)) cap *= 2;
    const int st = cox_internal_layer_reserve(Lh, std::min<u64>(cap, 1ull << 26));
    if (st != COX_OK && st != COX_ERR_OUT_OF_MEMORY) return st;  // out of memory: carry on with what there is
    if (st == COX_ERR_OUT_OF_MEMORY) Lh->auto_grow = false;
  }
  if (I->layer_generation != Lh->generation) {
    COX_TRY(sync_all(I));
    drop_graphs(I);
    for (RecordSet& S : I->rs) {
      COX_TRY(dev_realloc(&S.touched_slots, Lh->ht_cap));
      COX_TRY(dev_realloc(&S.ord_info, Lh->ht_cap));
      COX_TRY(dev_realloc(&S.blk_beg, static_cast<size_t>(Lh->ht_cap) * kTilesPerBlock));
      COX_TRY(dev_realloc(&S.blk_end, static_cast<size_t>(Lh->ht_cap) * kTilesPerBlock));
      COX_HIP(hipMemset(S.blk_beg, 0, sizeof(u32) * Lh->ht_cap * kTilesPerBlock));
      COX_HIP(hipMemset(S.blk_end, 0, sizeof(u32) * Lh->ht_cap * kTilesPerBlock));
      if (I->piece_sort) {
        COX_TRY(dev_realloc(&S.big_of_tile, static_cast<size_t>(Lh->ht_cap) * kTilesPerBlock));
        COX_TRY(dev_realloc(&S.blk_list, static_cast<size_t>(Lh->ht_cap) * kTilesPerBlock));
        COX_HIP(hipMemset(S.big_of_tile, 0, sizeof(u32) * Lh->ht_cap * kTilesPerBlock));
      }
      COX_HIP(hipDeviceSynchronize());
    }
    if (I->piece_sort) COX_TRY(dev_realloc(&I->scanws_h.block_sums, scan_num_blocks(Lh->ht_cap) + 2));
    I->layer_generation = Lh->generation;
  }
  return COX_OK;
}

// Ordering of one frame's inputs against whoever produces them (the engine's own input stream: staged host buffers, converted depth
// images), besides the caller's stream of cox_integrator_set_input_stream:
//   ready      the frame's first stage waits for it                 consumed   recorded once the inputs have been read for the last time
//   n_dev      the point count, where only the device knows it (n is then an upper bound)
struct FrameInput {
  hipEvent_t ready = nullptr, consumed = nullptr;
  const u32* n_dev = nullptr;
};
// enqueue the whole frame; xyz / rgba are device pointers that must stay valid until the frame's stage A2 is done
static int integrate_device(cox_integrator* I, const float T[7], const float* xyz, const uint8_t* rgba, u32 n, int freespace, bool caller_waits = false,
                            const FrameInput& in = FrameInput()) {
  cox_layer* Lh = I->layer;
  struct HostTimer {
    cox_integrator* I;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    ~HostTimer() {
      I->host_ns += static_cast<uint64_t>(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count());
      I->host_frames += 1;
    }
  } host_timer{I};
  // every frame but the previous one has been enqueued completely: the events this frame waits for (F.done of frame t-6,
  // B.done of t-3) are recorded, and the sets frame t-3 used are not touched by the submission thread any more
  {
    const auto w0 = std::chrono::steady_clock::now();
    if (I->submitter) I->submitter->wait_outstanding(1);
    I->host_wait_ns += static_cast<uint64_t>(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - w0).count());
  }
  COX_TRY(ensure_capacity(I, n));
  COX_TRY(follow_layer(I));
  const bool foreign_writer = (n != 0 || I->method == COX_METHOD_FAST) && cox_layer_order_writer(Lh, I);
  I->last = cox_frame_stats{};
  I->last.n_points = n;
  I->last_has_counts = false;
  // (an empty cloud still resets the fast integrator's two sets, as FastTsdfIntegrator::integratePointCloud does before it looks at a point)
  if (n == 0 && I->method != COX_METHOD_FAST) return COX_OK;
  I->frame_no += 1;
  const int slot = static_cast<int>(I->frame_no % kFrameSets);
  FrameSet& F = I->fs[slot];
  BundleSet& B = I->bs[slot % kStageSets];
  RecordSet& S = I->rs[slot % kStageSets];
  const StageCtx ctx{I, &F, &B, &S, slot, I->frame_no, in.n_dev};
  tl_frame_no = I->frame_no;
  I->last_count_on_device = in.n_dev != nullptr;
  // the pinned parameter slot is free once the copy of the frame that used it last (t-4) has run
  if (F.used) {
    const auto w0 = std::chrono::steady_clock::now();
    COX_HIP(hipEventSynchronize(F.params_copied));
    I->host_wait_ns += static_cast<uint64_t>(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - w0).count());
  }
  FrameParams P = make_params(I, T, n, freespace, xyz, rgba);
  P.frame_id = ++Lh->frame_id;
  I->h_params[slot] = P;

  if (I->has_producer) {  // inputs written on the caller's stream: stage H starts after them
    COX_HIP(hipEventRecord(I->ev_producer, I->producer));
    COX_HIP(hipStreamWaitEvent(stage_stream(I, 0, slot), I->ev_producer, 0));
  }
  if (in.ready) COX_HIP(hipStreamWaitEvent(stage_stream(I, 0, slot), in.ready, 0));
  if (I->method == COX_METHOD_FAST) {  // front on st[0] (this thread), solve on st[1] and update on st[3] (submission thread): fast_front / fast_solve
    if (F.used) COX_HIP(hipStreamWaitEvent(I->st[0], F.done, 0));  // frame t-6 is done with this frame set
    FastJob job;
    COX_TRY(fast_front(ctx, &job));
    COX_HIP(hipEventRecord(F.params_copied, I->st[0]));
    COX_HIP(hipEventRecord(F.hand[0], I->st[0]));
    if (I->has_producer) {  // the inputs are read by the front only
      COX_HIP(hipEventRecord(I->ev_inputs_read, I->st[0]));
      COX_HIP(hipStreamWaitEvent(I->producer, I->ev_inputs_read, 0));
    }
    if (in.consumed) COX_HIP(hipEventRecord(in.consumed, I->st[0]));
    COX_HIP(hipGetLastError());
    auto back = [ctx, job, foreign_writer]() -> int {
      cox_integrator* I = ctx.I;
      FrameSet& F = *ctx.F;
      RecordSet& S = *ctx.S;
      cox_layer* Lh = I->layer;
      tl_frame_no = ctx.frame;
      COX_HIP(hipStreamWaitEvent(I->st[1], F.hand[0], 0));
      COX_TRY(fast_solve(ctx, job));
      COX_HIP(hipEventRecord(F.hand[2], I->st[1]));
      COX_HIP(hipStreamWaitEvent(I->st[3], F.hand[2], 0));
      if (foreign_writer) cox_layer_wait_writes(Lh, I->st[3]);  // another integrator wrote this layer last
      COX_TRY(run_stage(3, ctx));  // touch / emit, record partition, apply on st[3] == st[4] == st[5]
      COX_TRY(run_stage(4, ctx));
      COX_TRY(run_stage(5, ctx));
      COX_HIP(hipEventRecord(F.done, I->st[5]));
      COX_HIP(hipEventRecord(S.done, I->st[5]));
      COX_HIP(hipEventRecord(Lh->last_write, I->st[5]));
      Lh->has_write = true;
      F.used = true;
      S.used = true;
      COX_HIP(hipGetLastError());
      return COX_OK;
    };
    if (I->submitter && !caller_waits) {
      I->submitter->post(back);
    } else {
      if (I->submitter) I->submitter->wait_outstanding(0);
      COX_TRY(back());
    }
    I->last_has_counts = true;
    return COX_OK;
  }
  // stage k + 1 of a frame follows stage k: in stream order, or behind the frame slot's hand-over event when on another stream
  auto chain = [](const StageCtx& x, int k) -> int {  // run stage k behind stage k - 1
    cox_integrator* I = x.I;
    const hipStream_t sk = stage_stream(I, k, x.slot);
    if (k > 0 && sk != stage_stream(I, k - 1, x.slot)) COX_HIP(hipStreamWaitEvent(sk, x.F->hand[k - 1], 0));
    COX_TRY(run_stage(k, x));
    if (k + 1 < kNumStages && stage_stream(I, k + 1, x.slot) != sk) COX_HIP(hipEventRecord(x.F->hand[k], sk));
    return COX_OK;
  };
  const hipStream_t s_h = stage_stream(I, 0, slot), s_m = stage_stream(I, 2, slot);
  // H, P, M (ray generation: depends on the frame's input only)
  if (F.used && s_h != I->st[5]) COX_HIP(hipStreamWaitEvent(s_h, F.done, 0));  // frame t-6 is done with this frame set
  if (B.used && (I->st_alt || s_h != s_m)) COX_HIP(hipStreamWaitEvent(s_h, B.done, 0));  // frame t-3's merge (on the other ray-generation stream, or on M's) is done with this bundle set
  COX_TRY(chain(ctx, 0));
  COX_HIP(hipEventRecord(F.params_copied, s_h));
  COX_TRY(chain(ctx, 1));
  COX_TRY(chain(ctx, 2));
  COX_HIP(hipEventRecord(B.done, s_m));
  B.used = true;
  if (I->has_producer) {  // the inputs are not read after the merge: later work on the caller's stream may overwrite / free them
    COX_HIP(hipEventRecord(I->ev_inputs_read, s_m));
    COX_HIP(hipStreamWaitEvent(I->producer, I->ev_inputs_read, 0));
  }
  if (in.consumed) COX_HIP(hipEventRecord(in.consumed, s_m));
  COX_HIP(hipGetLastError());
  // T, R, U (layer update): on the submission thread when there is one
  auto stage_b = [ctx, chain, foreign_writer]() -> int {
    cox_integrator* I = ctx.I;
    FrameSet& F = *ctx.F;
    RecordSet& S = *ctx.S;
    cox_layer* Lh = I->layer;
    if (S.used && I->st[3] != I->st[5]) COX_HIP(hipStreamWaitEvent(I->st[3], S.done, 0));  // frame t-3's apply is done with this record set
    if (foreign_writer) cox_layer_wait_writes(Lh, I->st[3]);  // another integrator wrote this layer last: stage T (block touches) follows its frames
    COX_TRY(chain(ctx, 3));
    COX_TRY(chain(ctx, 4));
    COX_TRY(chain(ctx, 5));
    COX_HIP(hipEventRecord(F.done, I->st[5]));
    COX_HIP(hipEventRecord(S.done, I->st[5]));
    COX_HIP(hipEventRecord(Lh->last_write, I->st[5]));
    Lh->has_write = true;
    F.used = true;
    S.used = true;
    COX_HIP(hipGetLastError());
    return COX_OK;
  };
  if (I->submitter && !caller_waits) {
    I->submitter->post(stage_b);
  } else {  // a caller that waits for the frame anyway (cox_integrate_points) gains nothing from the hand-over
    if (I->submitter) I->submitter->wait_outstanding(0);
    COX_TRY(stage_b());
  }
  I->last_has_counts = true;
  return COX_OK;
}

// the last frame's device counters are fetched when somebody asks (sync / last_stats), not once per frame: its frame
// set is not reused before four more frames have been enqueued.  Call with all streams idle.
static int fold_counters(cox_integrator* I) {
  if (!I->last_has_counts) return COX_OK;
  Counters& c = I->h_ring[I->frame_no % kStatRing];
  COX_HIP(hipMemcpy(&c, I->fs[I->frame_no % kFrameSets].cnt, sizeof(Counters), hipMemcpyDeviceToHost));
  if (I->last_count_on_device) {  // (depth front end: the host never saw the frame's point count)
    u32 np = 0;
    COX_HIP(hipMemcpy(&np, &I->fs[I->frame_no % kFrameSets].d_params->n_points, sizeof(u32), hipMemcpyDeviceToHost));
    I->last.n_points = np;
  }
  u64 sh[5] = {0, 0, 0, 0, 0};
  u32 max_bundle = 0, max_run = 0;
  for (int s = 0; s < 64; ++s) {
    for (int k = 0; k < 5; ++k) sh[k] += c.shard[s][k];
    max_bundle = std::max(max_bundle, c.shard[s][kShMaxBundle]);
    max_run = std::max(max_run, c.shard[s][kShMaxRun]);
  }
  I->last.max_bundle_points = max_bundle;
  I->last.max_voxel_updates = max_run;
  I->last.n_valid = sh[kShValid];
  I->last.n_rays = (I->method == COX_METHOD_SIMPLE) ? sh[kShRays] : c.n_rays;
  I->last.n_updates = sh[kShUpdates];
  I->last.n_touched_voxels = sh[kShVoxels];
  I->last.n_touched_blocks = c.n_touched;
  I->last.n_new_blocks = c.n_new_blocks;
  I->last_big_tiles = c.n_big_tiles;
  I->last_big_chunks = c.n_big_chunks;
  I->last_has_counts = false;  // folded; I->last keeps the numbers
  return COX_OK;
}

// COX_TIMELINE=<file>: every timed region as "class start_ms end_ms" relative to the moment profiling was switched on
// (hipEventElapsedTime works across streams) -- how the stages of neighbouring frames overlap, without a profiler attached
static void drain_events(std::vector<std::pair<hipEvent_t, hipEvent_t>>& evs, double* ms_acc, uint64_t* n_acc, std::vector<hipEvent_t>* pool, int cls = -1,
                         hipEvent_t ref = nullptr, FILE* timeline = nullptr) {
  for (auto& ev : evs) {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) {
      *ms_acc += ms;
      *n_acc += 1;
    }
    if (timeline && ref) {
      float a = 0.0f, b = 0.0f;
      if (hipEventElapsedTime(&a, ref, ev.first) == hipSuccess && hipEventElapsedTime(&b, ref, ev.second) == hipSuccess)
        fprintf(timeline, "%d %.4f %.4f\n", cls, a, b);
    }
    pool->push_back(ev.first);
    pool->push_back(ev.second);
  }
  evs.clear();
}

// wait for all stages, fold the last frame's counters into the stats, return deferred errors
static int integrator_finish(cox_integrator* I) {
  COX_TRY(sync_all(I));
  COX_TRY(fold_counters(I));
  // errors of every frame since the last sync are sticky in the layer's device error word; report them once
  u32 lerr = 0;
  COX_HIP(hipMemcpy(&lerr, I->layer->d_err, sizeof(u32), hipMemcpyDeviceToHost));
  if (lerr) {
    COX_HIP(hipMemset(I->layer->d_err, 0, sizeof(u32)));
    COX_HIP(hipDeviceSynchronize());  // default-stream memset: the engine's streams would not wait for it
  }
  for (int k = 0; k < COX_KERNEL_CLASSES; ++k)
    drain_events(I->class_events[k], &I->class_ms[k], &I->class_regions[k], &I->event_pool[event_pool_of(I, k)], k, I->timeline_ref, I->timeline);
  if (I->timeline) fflush(I->timeline);
  return err_bits_to_status(lerr);
}

cox_layer* cox_internal_integrator_layer(cox_integrator_t* integ) { return integ->layer; }

extern "C" {

int cox_integrator_create(cox_layer_t* layer, const cox_tsdf_config* cfg, int method, cox_integrator_t** out) {
  COX_ENTRY();
  if (!layer || !cfg || !out) return COX_ERR_INVALID_ARG;
  if (method == COX_METHOD_PROJECTIVE) {
    if (!(cfg->default_truncation_distance > 0.0f) || !(cfg->max_weight > 0.0f) || !(cfg->max_ray_length_m > 0.0f)) return COX_ERR_INVALID_ARG;
    COX_HIP(hipSetDevice(layer->device));
    cox_integrator* I = new (std::nothrow) cox_integrator();
    if (!I) return COX_ERR_OUT_OF_MEMORY;
    I->layer = layer;
    I->cfg = *cfg;
    I->method = method;
    const int st = cox_proj_create(layer, cfg, &I->proj);
    if (st != COX_OK) {
      delete I;
      return st;
    }
    *out = I;
    return COX_OK;
  }
  if (method != COX_METHOD_SIMPLE && method != COX_METHOD_MERGED && method != COX_METHOD_FAST) return COX_ERR_INVALID_ARG;
  if (cfg->integration_order_mode != 0) return COX_ERR_UNSUPPORTED;
  if (method == COX_METHOD_FAST) {
    // reproduced: the reference at integrator_threads = 1 (with more threads its two lossy sets race and the result is
    // not reproducible even by itself).  Not reproduced: a wall-clock budget, and the oracle-only exact-set variant.
    if (cfg->max_integration_time_s < 3.0e38f || cfg->fast_exact_sets != 0) return COX_ERR_UNSUPPORTED;
    if (cfg->clear_checks_every_n_frames < 1 || cfg->max_consecutive_ray_collisions < 0 || !(cfg->start_voxel_subsampling_factor > 0.0f)) return COX_ERR_INVALID_ARG;
  }
  if (!(cfg->default_truncation_distance > 0.0f) || !(cfg->max_weight > 0.0f) || !(cfg->max_ray_length_m > 0.0f)) return COX_ERR_INVALID_ARG;
  COX_HIP(hipSetDevice(layer->device));
  cox_integrator* I = new (std::nothrow) cox_integrator();
  if (!I) return COX_ERR_OUT_OF_MEMORY;
  I->layer = layer;
  I->cfg = *cfg;
  I->method = method;
  // stage graphs: measured slightly slower than eager launches here (2 936 vs 3 117 frames/s), so opt-in
  I->use_graphs = std::getenv("COX_GRAPH") != nullptr && std::getenv("COX_NO_GRAPH") == nullptr;
  I->block_apply = !(std::getenv("COX_APPLY") && std::string(std::getenv("COX_APPLY")) == "records");
  // COX_APPLY=pieces: merged without anti-grazing walks, sorts and applies PIECES instead of records (k_touch_pieces / k_apply_pieces).
  // Bit-identical and covered by the parity tests, but not the default: the piece sort is 2.3x cheaper than the record sort at
  // 1 cm, the piece apply 1.75x dearer than the record apply -- even at 1 cm, slower at 2 cm and 5 cm
  if (const char* e = std::getenv("COX_GRID_APPLY")) I->grid_apply = std::max(1, std::atoi(e));
  if (const char* e = std::getenv("COX_GRID_APPLY_WAVE")) I->grid_apply_wave = static_cast<u32>(std::max(8, std::atoi(e)));
  if (const char* e = std::getenv("COX_APPLY_WAVE")) I->wave_apply = std::atoi(e) != 0;
  if (const char* e = std::getenv("COX_SPLIT_TILES")) I->split_big_tiles = std::atoi(e) != 0;
  I->h2d_kernel = std::getenv("COX_H2D") && std::string(std::getenv("COX_H2D")) == "kernel";
  I->h2d_groups = std::getenv("COX_H2D_GROUPS") ? std::max(1, std::atoi(std::getenv("COX_H2D_GROUPS"))) : kCopyGroups;
  if (const char* e = std::getenv("COX_BIG_CHUNK")) I->big_chunk = static_cast<u32>(std::max<int>(kBigChunkMin, std::atoi(e)));
  if (const char* e = std::getenv("COX_WAVE_TILE_MAX")) I->wave_tile_max = std::atoi(e) == 256 ? 256u : 512u;
  if (const char* e = std::getenv("COX_GRID_MERGE")) I->grid_merge = std::max(1, std::atoi(e));
  if (const char* e = std::getenv("COX_GRID_TOUCH")) I->grid_touch = std::max(1, std::atoi(e));
  I->piece_path = method == COX_METHOD_MERGED && !cfg->enable_anti_grazing && std::getenv("COX_APPLY") && std::string(std::getenv("COX_APPLY")) == "pieces";
  // COX_PARTITION=pieces: the walk leaves pieces, the pieces are sorted by tile and expanded into the records of the default
  // tile apply (k_touch_pieces<deferred touch> / k_piece_touch / k_piece_expand); =records: the record partition
  // Record partition: one pass into 4096 buckets where a frame touches few blocks (coarse voxels: a bucket is a tile, or a
  // few); where rays are long in voxels a frame touches 10^3-10^4 blocks and the tiles of a bucket would each re-read the whole
  // bucket (1 cm: 40 of them), so the whole tile id is sorted there, in two passes (COX_BUCKETS=0|1 overrides).
  I->bucket_partition = (max_steps_per_ray(I) - 1) / 3 + 2 <= kAxisCapSmall;
  if (const char* e = std::getenv("COX_BUCKETS")) I->bucket_partition = std::atoi(e) != 0;
  // Default: pieces where rays are long in voxels (the large-LDS walk: 2 cm and finer with the reference's ray lengths) --
  // 1 cm 513 -> 699 frames/s, 2 cm 2 071 -> 2 159; records otherwise (5 cm: 7 129 vs 5 323, the extra launches cost more than
  // the smaller sort saves at 4 * 10^5 records per frame).
  if (method == COX_METHOD_MERGED && !cfg->enable_anti_grazing && !I->piece_path && I->block_apply) {
    I->piece_sort = (max_steps_per_ray(I) - 1) / 3 + 2 > kAxisCapSmall;
    if (const char* e = std::getenv("COX_PARTITION")) I->piece_sort = std::string(e) == "pieces";
    // COX_TILE=9: tiles of two z slabs (512 voxels).  Half as many tiles did not help at fine voxels (1 cm: apply 0.81 -> 1.17 ms
    // with one frame in flight): the tile apply is bound by its per-record work, not by the round trips per tile.
    if (const char* e = std::getenv("COX_TILE")) I->tile_shift = (std::atoi(e) == 9) ? 9u : 8u;
  }
  int st = COX_OK;
  auto ev = [&](hipEvent_t* e) {
    if (st == COX_OK && hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) st = COX_ERR_NO_DEVICE;
  };
  // A frame's stream is a chain of small kernels with a few microseconds between them, and the frame rate at 5 cm is the
  // length of the longest chain (measured with COX_TIMELINE).  Frames/s at 5 / 2 / 1 cm, same box, 16
  // hardware queues: two streams (H P M | T R U) 5 747 / - / -; four (H P | M | T R | U) 7 636 / 2 128 / 506; six (every stage
  // its own) 4 202 / 1 949 / 531 -- past four the hand-overs between queues cost more than the shorter chains give back.
  I->n_streams = 4;
  if (const char* e = std::getenv("COX_STREAMS")) {
    const int v = std::atoi(e);  // ("4s" parses as 4)
    if (v == 2 || v == 4 || v == 6) I->n_streams = v;
  }
  {
    // stage -> stream: 6: one each; 4: H P | M | T R | U; 2: H P M | T R U.  Equal streams are adjacent.
    // Default (four streams): H P M | H P M | T R | U -- ray generation depends on the frame's input only, so two frames run
    // it side by side (even / odd frame slots), and the frame rate no longer hangs on the longest ray-generation chain (the
    // merge of a frame with large bundles: 105 us); the layer update stays one pipeline, in frame order.
    // Where the layer update is several times the ray generation -- the piece partition: 2 cm and finer -- the four streams go to
    // H P M | T | R | U instead: 2 456 -> 2 910 frames/s at 2 cm; at 1 cm that map lost 7 % while the apply was 0.8 ms of one
    // workgroup's time (round 2) and gains 9 % since the large tiles are split (752 -> 823, round 3).
    // fast: front | solve | update (fast_front / fast_solve): st[0] | st[1] == st[2] | st[3] == st[4] == st[5]; COX_FAST_STREAMS=1 puts
    // all three on one stream (round 2's arrangement).
    static const int kMap[6][kNumStages] = {{0, 0, 0, 1, 1, 1}, {0, 0, 1, 2, 2, 3}, {0, 1, 2, 3, 4, 5}, {0, 0, 0, 1, 1, 2}, {0, 0, 0, 1, 2, 3}, {0, 1, 1, 2, 2, 2}};
    const bool chosen = std::getenv("COX_STREAM_MAP") || std::getenv("COX_STREAMS");
    const bool update_heavy = !chosen && method == COX_METHOD_MERGED && I->piece_sort;
    const bool parity = (I->n_streams == 4 || (std::getenv("COX_STREAMS") && std::string(std::getenv("COX_STREAMS")) == "3p")) && method != COX_METHOD_FAST && !std::getenv("COX_STREAM_MAP") && !update_heavy &&
                        !(std::getenv("COX_STREAMS") && std::string(std::getenv("COX_STREAMS")) == "4s");  // COX_STREAMS=4s: the staged map H P | M | T R | U
    const bool three_p = std::getenv("COX_STREAMS") && std::string(std::getenv("COX_STREAMS")) == "3p";  // H P M | H P M | T R U
    const int* map = kMap[method == COX_METHOD_FAST ? 5 : update_heavy ? 4 : three_p ? 0 : parity ? 3 : I->n_streams == 2 ? 0 : I->n_streams == 4 ? 1 : 2];
    // The input stream (host buffers -> staging sets) is created FIRST, directly in front of the stage streams: the hardware queues of
    // a process are spread over the four pipes of the compute micro-engine in creation order, so four stage streams created back to
    // back sit on four different pipes, and the input stream lands on the pipe of the LAST of them -- the apply's (two long kernels
    // per frame), the partner that minds least (sharing with T/R costs 35 %, with a ray-generation stream 45 %).
    // `fast` has three stage streams: its input stream has a pipe to itself.  COX_INPUT_STREAM=0: inputs ride on the frame's own stream.
    if (!(std::getenv("COX_INPUT_STREAM") && std::atoi(std::getenv("COX_INPUT_STREAM")) == 0) && st == COX_OK &&
        hipStreamCreateWithFlags(&I->st_in, hipStreamNonBlocking) != hipSuccess)
      st = COX_ERR_NO_DEVICE;
    if (parity && st == COX_OK && hipStreamCreateWithFlags(&I->st_alt, hipStreamNonBlocking) != hipSuccess) st = COX_ERR_NO_DEVICE;
    int custom[kNumStages];
    if (const char* e = std::getenv("COX_STREAM_MAP")) {  // experiments: six digits, stage -> stream, equal streams adjacent (e.g. 012334)
      bool ok = std::strlen(e) == kNumStages && e[0] == '0';
      for (int k = 0; ok && k < kNumStages; ++k) {
        custom[k] = e[k] - '0';
        ok = custom[k] >= 0 && custom[k] < kNumStages && (k == 0 || custom[k] == custom[k - 1] || custom[k] == custom[k - 1] + 1);
      }
      if (ok && method != COX_METHOD_FAST) map = custom;
    }
    const int fast_mode = std::getenv("COX_FAST_STREAMS") ? std::atoi(std::getenv("COX_FAST_STREAMS")) : 3;  // 1: one stream, otherwise three
    if (method == COX_METHOD_FAST) {
      I->n_streams = fast_mode == 1 ? 1 : 3;
      if (const char* e = std::getenv("COX_FAST_FENCE")) I->fast.fences = std::atoi(e);
      // At 5 cm two rounds settle every frame of the benchmark stream; at 2 cm and 1 cm nine frames in ten have a ray that outgrows its
      // round-1 list (more rays per surface patch, longer second-order chains), and a frame that ends so is redone by ONE lane
      // (14 and 4 frames/s): four rounds there -- the extra launches are nothing next to frames of a millisecond and more.
      I->fast.rounds = ((max_steps_per_ray(I) - 1) / 3 + 2 <= kAxisCapSmall) ? 2 : kFastMaxRounds;
      if (I->fast.rounds > 2) {  // ... and longer lists from the start (measured, frames/s at 2 cm / 1 cm: caps 8,16: 628 / 5; 16,32: 694 / 28; 32,32: - / 104)
        I->fast.cap0 = ((max_steps_per_ray(I) - 1) / 3 > 200) ? 32u : 16u;
        I->fast.cap1 = 32u;
      }
      if (const char* e = std::getenv("COX_FAST_ROUNDS")) I->fast.rounds = std::min(kFastMaxRounds, std::max(2, std::atoi(e)));
      if (const char* e = std::getenv("COX_FAST_GROUPS")) I->fast.relax_groups = static_cast<u32>(std::min(256, std::max(8, std::atoi(e))));
      I->fast.force_sequential = std::getenv("COX_FAST_SEQUENTIAL") && std::atoi(std::getenv("COX_FAST_SEQUENTIAL")) != 0;
      if (const char* e = std::getenv("COX_FAST_CAP")) {  // candidate steps per ray: "cap0" or "cap0,cap1"
        int a = 0, b = 0;
        const int got = std::sscanf(e, "%d,%d", &a, &b);
        if (got >= 1 && a >= 1 && a <= static_cast<int>(kFastCap0Max)) I->fast.cap0 = static_cast<u32>(a);
        I->fast.cap1 = std::max<u32>(I->fast.cap1, I->fast.cap0);
        if (got >= 2 && b <= static_cast<int>(kFastShortMax) && static_cast<u32>(b) >= I->fast.cap0) I->fast.cap1 = static_cast<u32>(b);
      }
    }
    hipStream_t made[kNumStages] = {};
    for (int k = 0; k < kNumStages; ++k) {
      int m = map[k];
      if (method == COX_METHOD_FAST && fast_mode == 1) m = 0;
      if (!made[m] && st == COX_OK && hipStreamCreateWithFlags(&made[m], hipStreamNonBlocking) != hipSuccess) st = COX_ERR_NO_DEVICE;
      I->st[k] = made[m];
    }
  }
  ev(&I->ev_a2);
  for (int k = 0; k < kInputSets; ++k) {
    if (st == COX_OK && hipEventCreateWithFlags(&I->in_ready[k], hipEventDisableTiming) != hipSuccess) st = COX_ERR_NO_DEVICE;
    ev(&I->in_free[k]);
  }
  ev(&I->ev_producer);
  ev(&I->ev_inputs_read);
  I->layer_generation = layer->generation;
  for (FrameSet& F : I->fs) {
    ev(&F.done);
    ev(&F.params_copied);
    for (hipEvent_t& h : F.hand) ev(&h);
    if (st == COX_OK) st = dev_realloc(&F.cnt, 1);
    if (st == COX_OK) st = dev_realloc(&F.d_params, 1);
  }
  auto info = [&](SortInfo** p) {
    if (st == COX_OK) st = dev_realloc(p, 1);
    if (st == COX_OK && hipMemset(*p, 0, sizeof(SortInfo)) != hipSuccess) st = COX_ERR_NO_DEVICE;
  };
  for (BundleSet& B : I->bs) {
    ev(&B.done);
    info(&B.sort_info);
  }
  for (RecordSet& S : I->rs) {
    ev(&S.done);
    if (st == COX_OK) st = dev_realloc(&S.touched_slots, layer->ht_cap);  // one entry per block key the table can hold
    if (st == COX_OK) st = dev_realloc(&S.ord_info, layer->ht_cap);
    if (st == COX_OK) st = dev_realloc(&S.blk_beg, static_cast<size_t>(layer->ht_cap) * kTilesPerBlock);
    if (st == COX_OK) st = dev_realloc(&S.blk_end, static_cast<size_t>(layer->ht_cap) * kTilesPerBlock);
    if (st == COX_OK && (hipMemset(S.blk_beg, 0, sizeof(u32) * layer->ht_cap * kTilesPerBlock) != hipSuccess ||
                         hipMemset(S.blk_end, 0, sizeof(u32) * layer->ht_cap * kTilesPerBlock) != hipSuccess))
      st = COX_ERR_NO_DEVICE;
    if (st == COX_OK && I->piece_sort) {
      st = dev_realloc(&S.big_of_tile, static_cast<size_t>(layer->ht_cap) * kTilesPerBlock);
      if (st == COX_OK) st = dev_realloc(&S.blk_list, static_cast<size_t>(layer->ht_cap) * kTilesPerBlock);
      if (st == COX_OK) st = dev_realloc(&S.big_acc, static_cast<size_t>(kBigCap) * 2 * kTileVox);
      if (st == COX_OK && (hipMemset(S.big_of_tile, 0, sizeof(u32) * layer->ht_cap * kTilesPerBlock) != hipSuccess ||
                           hipMemset(S.big_acc, 0, sizeof(u32) * kBigCap * 2 * kTileVox) != hipSuccess))
        st = COX_ERR_NO_DEVICE;
    }
    info(&S.sort_info);
  }
  if (st == COX_OK && I->piece_sort) st = dev_realloc(&I->scanws_h.block_sums, scan_num_blocks(layer->ht_cap) + 2);
  if (st == COX_OK && hipHostMalloc(reinterpret_cast<void**>(&I->h_ring), sizeof(Counters) * kStatRing, hipHostMallocDefault) != hipSuccess)
    st = COX_ERR_OUT_OF_MEMORY;
  if (st == COX_OK && hipHostMalloc(reinterpret_cast<void**>(&I->h_params), sizeof(FrameParams) * kFrameSets, hipHostMallocDefault) != hipSuccess)
    st = COX_ERR_OUT_OF_MEMORY;
  if (st == COX_OK) st = dev_realloc(&I->d_depth_n, kInputSets);
  if (st == COX_OK && method == COX_METHOD_FAST) {
    FastState& X = I->fast;
    st = dev_realloc(&X.table_start, kFastSlots);
    if (st == COX_OK) st = dev_realloc(&X.table_obs, kFastSlots);
    if (st == COX_OK) st = dev_realloc(&X.d_stats, 16);
    if (st == COX_OK && (hipMemset(X.table_start, 0, sizeof(u64) * kFastSlots) != hipSuccess || hipMemset(X.table_obs, 0, sizeof(u64) * kFastSlots) != hipSuccess ||
                         hipMemset(X.d_stats, 0, sizeof(u32) * 16) != hipSuccess))
      st = COX_ERR_NO_DEVICE;
    ev(&X.vs0[0].done);
    ev(&X.vs0[1].done);
  }
  if (st == COX_OK) {
    memset(I->h_ring, 0, sizeof(Counters) * kStatRing);
    memset(I->h_params, 0, sizeof(FrameParams) * kFrameSets);
    st = ensure_capacity(I, 640 * 480);
  }
  // the initialising hipMemsets above ran on the legacy default stream; the engine's streams are non-blocking and do not
  // wait for it (found by the fuzz campaign: a table read before it had been cleared, once in 1 500 cases)
  if (st == COX_OK && hipDeviceSynchronize() != hipSuccess) st = COX_ERR_NO_DEVICE;
  if (st != COX_OK) {
    cox_integrator_destroy(I);
    return st;
  }
  if (!(std::getenv("COX_SUBMIT_THREAD") && std::atoi(std::getenv("COX_SUBMIT_THREAD")) == 0)) {
    I->submitter = new (std::nothrow) Submitter();
    if (I->submitter) {
      I->submitter->device = layer->device;
      I->submitter->th = std::thread([sub = I->submitter] { sub->run(); });
      {
        // the thread's first HIP call initialises per-thread runtime state: let it finish before the caller goes on making
        // HIP calls of its own (a caller's next call once failed with a stray "invalid device ordinal" right after a create)
        std::unique_lock<std::mutex> lk(I->submitter->m);
        I->submitter->cv_done.wait(lk, [&] { return I->submitter->ready; });
      }
      std::lock_guard<std::mutex> lk(g_submitters_mutex);
      g_submitters.push_back(I->submitter);
    }
  }
  (void)hipGetLastError();  // nothing of ours stays in the thread's sticky error slot (other HIP users in the process check it after their own calls)
  *out = I;
  return COX_OK;
}

void cox_integrator_destroy(cox_integrator_t* I) {
  if (!I) return;
  if (I->proj) {
    cox_proj_destroy(I->proj);
    delete I;
    return;
  }
  (void)hipSetDevice(I->layer->device);
  (void)sync_all(I);
  if (I->submitter) {
    {
      std::lock_guard<std::mutex> lk(g_submitters_mutex);
      g_submitters.erase(std::remove(g_submitters.begin(), g_submitters.end(), I->submitter), g_submitters.end());
    }
    {
      std::lock_guard<std::mutex> lk(I->submitter->m);
      I->submitter->stop = true;
    }
    I->submitter->cv_job.notify_all();
    if (I->submitter->th.joinable()) I->submitter->th.join();
    delete I->submitter;
    I->submitter = nullptr;
  }
  delete I->copy_pool;
  I->copy_pool = nullptr;
  drop_graphs(I);
  for (auto& evs : I->class_events)
    for (auto& e : evs) {
      (void)hipEventDestroy(e.first);
      (void)hipEventDestroy(e.second);
    }
  for (auto& pool : I->event_pool)
    for (hipEvent_t e : pool) (void)hipEventDestroy(e);
  if (I->timeline_ref) (void)hipEventDestroy(I->timeline_ref);
  if (I->timeline) fclose(I->timeline);
  std::vector<void*> ptrs = {I->depth_flag, I->d_depth_n, I->sort_pts.counts, I->sort_pts.totals, I->sort_pts_alt.counts, I->sort_pts_alt.totals, I->sort_rec.counts,
                             I->sort_rec.totals, I->sort_vis.counts, I->sort_vis.totals, I->sort_vis1.counts, I->sort_vis1.totals, I->scanws_a.block_sums, I->scanws_b.block_sums, I->scanws_d.block_sums,
                             I->scanws_f.block_sums, I->scanws_p.block_sums, I->scanws_h.block_sums};
  std::vector<hipEvent_t> events = {I->ev_a2, I->ev_producer, I->ev_inputs_read};
  {
    FastState& X = I->fast;
    for (void* q : {static_cast<void*>(X.fhash), static_cast<void*>(X.table_start), static_cast<void*>(X.table_obs), static_cast<void*>(X.fresh),
                    static_cast<void*>(X.rank), static_cast<void*>(X.d_stats), static_cast<void*>(X.long_list)})
      ptrs.push_back(q);
    for (int k = 0; k < kFrameSets; ++k) {
      ptrs.push_back(X.cap[k]);
      ptrs.push_back(X.reach[k]);
    }
    for (VisitSet* V : {&X.vs0[0], &X.vs0[1], &X.vs1}) {
      for (void* q : {static_cast<void*>(V->key[0]), static_cast<void*>(V->key[1]), static_cast<void*>(V->val[0]), static_cast<void*>(V->val[1]),
                      static_cast<void*>(V->vhash), static_cast<void*>(V->shash), static_cast<void*>(V->vray), static_cast<void*>(V->pos_of),
                      static_cast<void*>(V->sinfo), static_cast<void*>(V->voff)})
        ptrs.push_back(q);
      events.push_back(V->done);
    }
  }
  for (FrameSet& F : I->fs) {
    const RayArrays& R = F.rays;
    for (void* p : {static_cast<void*>(R.px), static_cast<void*>(R.py), static_cast<void*>(R.pz), static_cast<void*>(R.w), static_cast<void*>(R.color),
                    static_cast<void*>(R.flags), static_cast<void*>(R.key), static_cast<void*>(R.nsteps), static_cast<void*>(R.rec_off),
                    static_cast<void*>(R.pbound), static_cast<void*>(R.piece_off), static_cast<void*>(R.q), static_cast<void*>(F.fh_keys), static_cast<void*>(F.cnt), static_cast<void*>(F.d_params)})
      ptrs.push_back(p);
    events.push_back(F.done);
    events.push_back(F.params_copied);
    for (hipEvent_t h : F.hand) events.push_back(h);
  }
  for (BundleSet& B : I->bs) {
    for (void* p : {static_cast<void*>(B.pslot), static_cast<void*>(B.skey[0]), static_cast<void*>(B.skey[1]), static_cast<void*>(B.sval[0]),
                    static_cast<void*>(B.sval[1]), static_cast<void*>(B.head), static_cast<void*>(B.bstart), static_cast<void*>(B.sort_info)})
      ptrs.push_back(p);
    events.push_back(B.done);
  }
  for (RecordSet& S : I->rs) {
    for (void* p : {static_cast<void*>(S.rec_key[0]), static_cast<void*>(S.rec_key[1]), static_cast<void*>(S.rec_ray[0]), static_cast<void*>(S.rec_ray[1]),
                    static_cast<void*>(S.piece_front), static_cast<void*>(S.piece_back), static_cast<void*>(S.piece_wsum), static_cast<void*>(S.touched_slots), static_cast<void*>(S.ord_info),
                    static_cast<void*>(S.blk_beg), static_cast<void*>(S.blk_end), static_cast<void*>(S.big_of_tile), static_cast<void*>(S.big_acc), static_cast<void*>(S.blk_list),
                    static_cast<void*>(S.big_chunks), static_cast<void*>(S.lin8), static_cast<void*>(S.pkey[0]),
                    static_cast<void*>(S.pkey[1]), static_cast<void*>(S.pstart[0]), static_cast<void*>(S.pstart[1]), static_cast<void*>(S.prl[0]),
                    static_cast<void*>(S.prl[1]), static_cast<void*>(S.pbkey), static_cast<void*>(S.plen), static_cast<void*>(S.pdest),
                    static_cast<void*>(S.sort_info)})
      ptrs.push_back(p);
    events.push_back(S.done);
  }
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  for (hipEvent_t e : events)
    if (e) (void)hipEventDestroy(e);
  if (I->h_ring) (void)hipHostFree(I->h_ring);
  for (int k = 0; k < kInputSets; ++k) {
    if (I->pin_xyz[k]) (void)hipHostFree(I->pin_xyz[k]);
    if (I->pin_rgba[k]) (void)hipHostFree(I->pin_rgba[k]);
    if (I->in_ready[k]) (void)hipEventDestroy(I->in_ready[k]);
    if (I->in_free[k]) (void)hipEventDestroy(I->in_free[k]);
  }
  for (int k = 0; k < kInputSets; ++k) {
    if (I->own_xyz[k]) (void)hipFree(I->own_xyz[k]);
    if (I->own_rgba[k]) (void)hipFree(I->own_rgba[k]);
    if (I->own_depth[k]) (void)hipFree(I->own_depth[k]);
    if (I->own_depth_rgba[k]) (void)hipFree(I->own_depth_rgba[k]);
  }
  if (I->h_params) (void)hipHostFree(I->h_params);
  for (int k = 0; k < kNumStages; ++k)
    if (I->st[k] && (k == 0 || I->st[k] != I->st[k - 1])) (void)hipStreamDestroy(I->st[k]);
  if (I->st_alt) (void)hipStreamDestroy(I->st_alt);
  if (I->st_in) (void)hipStreamDestroy(I->st_in);
  delete I;
  (void)hipGetLastError();
}

int cox_integrate_points_dev(cox_integrator_t* I, const float T_G_C[7], const float* xyz_dev, const uint8_t* rgba_dev, uint64_t n, int freespace) {
  COX_ENTRY_NO_DRAIN();
  if (!I || !T_G_C || (n && !xyz_dev) || n > 0x7FFFFFFFull) return COX_ERR_INVALID_ARG;
  COX_HIP(hipSetDevice(I->layer->device));
  if (I->proj) return cox_proj_integrate(I->proj, T_G_C, xyz_dev, n, 0);
  return integrate_device(I, T_G_C, xyz_dev, rgba_dev, static_cast<u32>(n), freespace);
}

int cox_integrate_points_ex(cox_integrator_t* I, const float T_G_C[7], const float* xyz, const uint8_t* rgba, uint64_t n, int freespace, int deintegrate) {
  if (!I) return COX_ERR_INVALID_ARG;
  if (!deintegrate) return cox_integrate_points(I, T_G_C, xyz, rgba, n, freespace);
  COX_ENTRY();
  if (!I->proj) return COX_ERR_UNSUPPORTED;  // only the projective integrator can take a cloud out again
  if (!T_G_C || (n && !xyz) || n > 0x7FFFFFFFull) return COX_ERR_INVALID_ARG;
  COX_HIP(hipSetDevice(I->layer->device));
  return cox_proj_integrate_host(I->proj, T_G_C, xyz, n, 1);
}

// Host buffers -> staging set k of the frame about to be enqueued, on that frame's ray-generation stream, without waiting for
// anything but the staging set itself.  Pageable memory goes through a pinned bounce buffer (one CPU copy; the caller's buffer is
// free again when the call returns), pinned memory is copied from directly.
// -> the address the device reads the buffer at (pinned memory is mapped), nullptr for pageable memory
static const void* host_pointer_device_view(const void* p) {
  hipPointerAttribute_t a;
  const hipError_t e = hipPointerGetAttributes(&a, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();  // pageable memory is "invalid value" to the runtime
    return nullptr;
  }
  if (a.type != hipMemoryTypeHost) return nullptr;
  return a.devicePointer ? a.devicePointer : p;
}
// one or two arrays (bytes a multiple of 4) from pinned host memory, seen by the device at src_dev_*, to dst_* on stream s
static int copy_pinned_to_device(const cox_integrator* I, void* dst_a, const void* src_a, const void* src_dev_a, size_t bytes_a, void* dst_b, const void* src_b,
                                 const void* src_dev_b, size_t bytes_b, hipStream_t s) {
  const bool use_memcpy = !I->h2d_kernel;
  const int groups = I->h2d_groups;
  auto unaligned = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) != 0; };
  const bool two = dst_b != nullptr;
  if (use_memcpy || !src_dev_a || unaligned(src_dev_a) || unaligned(dst_a) || (bytes_a & 3u) ||
      (two && (!src_dev_b || unaligned(src_dev_b) || unaligned(dst_b) || (bytes_b & 3u)))) {
    COX_HIP(hipMemcpyAsync(dst_a, src_a, bytes_a, hipMemcpyHostToDevice, s));
    if (two) COX_HIP(hipMemcpyAsync(dst_b, src_b, bytes_b, hipMemcpyHostToDevice, s));
    return COX_OK;
  }
  HostSegment A{static_cast<U32x4*>(dst_a), static_cast<const U32x4*>(src_dev_a), bytes_a / 16, static_cast<u32>((bytes_a & 15u) / 4)};
  HostSegment B{nullptr, nullptr, 0, 0};
  if (two) B = HostSegment{static_cast<U32x4*>(dst_b), static_cast<const U32x4*>(src_dev_b), bytes_b / 16, static_cast<u32>((bytes_b & 15u) / 4)};
  hipLaunchKernelGGL(k_copy_from_host, dim3(groups), dim3(256), 0, s, A, B);
  return COX_OK;
}
// the stream the next frame's first stage will run on
static inline hipStream_t next_frame_stream(const cox_integrator* I) { return stage_stream(I, 0, static_cast<int>((I->frame_no + 1) % kFrameSets)); }
// ... and the stream host inputs are staged on: the input stream, or (COX_INPUT_STREAM=0) the frame's own
static inline hipStream_t next_input_stream(const cox_integrator* I) { return I->st_in ? I->st_in : next_frame_stream(I); }
// Staging set k was read last by the frame kInputSets frames ago (its stages H .. M, enqueued by the caller's thread: the event is
// recorded by now).  On the frame's own stream that is a device-side wait; on the input stream it is a HOST wait -- the caller can
// not be more than kFrameSets frames ahead of the device anyway (frame slots), and the input stream's queue then carries nothing
// but the copies and one event record per frame: no barrier packet that holds its pipe.
static int wait_staging_set_free(cox_integrator* I, int k, hipStream_t s_in) {
  if (!I->in_used[k]) return COX_OK;
  if (s_in == I->st_in) {
    COX_HIP(hipEventSynchronize(I->in_free[k]));
  } else {
    COX_HIP(hipStreamWaitEvent(s_in, I->in_free[k], 0));
  }
  return COX_OK;
}
// two host arrays (a: a_bytes per element, b: b_bytes per element or absent) -> the device buffers dst_a / dst_b
static int stage_host_inputs(cox_integrator* I, int k, const void* a, size_t a_bytes, void* dst_a, const void* b, size_t b_bytes, void* dst_b, u32 n, FrameInput* in) {
  const hipStream_t s_in = next_input_stream(I);
  // the staging set is free once the frame that used it three frames ago has read it for the last time; that frame's stages
  // H .. M are enqueued by the caller's thread, so the event is recorded by now
  COX_TRY(wait_staging_set_free(I, k, s_in));
  const void* dev_a = host_pointer_device_view(a);
  const void* dev_b = b ? host_pointer_device_view(b) : nullptr;
  const bool pinned = dev_a != nullptr && (!b || dev_b != nullptr);
  const void* src_a = a;
  const void* src_b = b;
  if (!pinned) {
    if (I->pin_cap < I->pcap) {
      COX_TRY(sync_all(I));
      for (int q = 0; q < kInputSets; ++q) {
        if (I->pin_xyz[q]) (void)hipHostFree(I->pin_xyz[q]);
        if (I->pin_rgba[q]) (void)hipHostFree(I->pin_rgba[q]);
        I->pin_xyz[q] = nullptr;
        I->pin_rgba[q] = nullptr;
        if (hipHostMalloc(reinterpret_cast<void**>(&I->pin_xyz[q]), sizeof(float) * 3 * I->pcap, hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc(reinterpret_cast<void**>(&I->pin_rgba[q]), 4ull * I->pcap, hipHostMallocDefault) != hipSuccess) {
          (void)hipGetLastError();
          I->pin_cap = 0;
          return COX_ERR_OUT_OF_MEMORY;
        }
      }
      I->pin_cap = I->pcap;
    }
    // the bounce buffer's previous copy (three frames ago) has left it: in_ready[k] was recorded behind that copy
    if (I->in_used[k]) COX_HIP(hipEventSynchronize(I->in_ready[k]));
    if (!I->copy_pool) {
      const int helpers = std::getenv("COX_COPY_THREADS") ? std::min(15, std::max(0, std::atoi(std::getenv("COX_COPY_THREADS")))) : 3;
      I->copy_pool = new CopyPool(helpers);
    }
    I->copy_pool->copy(I->pin_xyz[k], a, a_bytes * n);
    if (b) I->copy_pool->copy(I->pin_rgba[k], b, b_bytes * n);
    src_a = I->pin_xyz[k];
    src_b = b ? I->pin_rgba[k] : nullptr;
    dev_a = host_pointer_device_view(src_a);
    dev_b = b ? host_pointer_device_view(src_b) : nullptr;
  }
  COX_TRY(copy_pinned_to_device(I, dst_a, src_a, dev_a, a_bytes * n, b ? dst_b : nullptr, src_b, dev_b, b ? b_bytes * n : 0, s_in));
  COX_HIP(hipEventRecord(I->in_ready[k], s_in));
  I->in_used[k] = true;
  in->ready = I->st_in ? I->in_ready[k] : nullptr;  // (without an input stream: same stream as the frame's first stage, stream order)
  in->consumed = I->in_free[k];
  return COX_OK;
}

int cox_integrate_points(cox_integrator_t* I, const float T_G_C[7], const float* xyz, const uint8_t* rgba, uint64_t n, int freespace) {
  COX_ENTRY_NO_DRAIN();
  if (!I || !T_G_C || (n && !xyz) || n > 0x7FFFFFFFull) return COX_ERR_INVALID_ARG;
  COX_HIP(hipSetDevice(I->layer->device));
  if (I->proj) return cox_proj_integrate_host(I->proj, T_G_C, xyz, n, 0);
  COX_TRY(ensure_capacity(I, static_cast<u32>(n)));
  if (I->submitter) I->submitter->wait_outstanding(0);
  const int k = static_cast<int>((I->frame_no + 1) % kInputSets);  // the bundle set of the frame about to be enqueued
  FrameInput in;
  if (n) {  // synchronous call: copied straight from the caller's buffers (the runtime stages pageable memory itself), the call returns after the frame
    const hipStream_t s_in = next_input_stream(I);
    COX_TRY(wait_staging_set_free(I, k, s_in));
    COX_HIP(hipMemcpyAsync(I->own_xyz[k], xyz, sizeof(float) * 3 * n, hipMemcpyHostToDevice, s_in));
    if (rgba) COX_HIP(hipMemcpyAsync(I->own_rgba[k], rgba, 4 * n, hipMemcpyHostToDevice, s_in));
    COX_HIP(hipEventRecord(I->in_ready[k], s_in));
    I->in_used[k] = true;
    in.ready = I->st_in ? I->in_ready[k] : nullptr;
    in.consumed = I->in_free[k];
  }
  COX_TRY(integrate_device(I, T_G_C, I->own_xyz[k], rgba ? I->own_rgba[k] : nullptr, static_cast<u32>(n), freespace, true, in));
  return integrator_finish(I);
}

int cox_integrate_points_async(cox_integrator_t* I, const float T_G_C[7], const float* xyz, const uint8_t* rgba, uint64_t n, int freespace) {
  COX_ENTRY_NO_DRAIN();
  if (!I || !T_G_C || (n && !xyz) || n > 0x7FFFFFFFull) return COX_ERR_INVALID_ARG;
  COX_HIP(hipSetDevice(I->layer->device));
  if (I->proj) return cox_proj_integrate_host(I->proj, T_G_C, xyz, n, 0);
  COX_TRY(ensure_capacity(I, static_cast<u32>(n)));
  if (I->submitter) I->submitter->wait_outstanding(1);
  const int k = static_cast<int>((I->frame_no + 1) % kInputSets);
  FrameInput in;
  if (n) COX_TRY(stage_host_inputs(I, k, xyz, sizeof(float) * 3, I->own_xyz[k], rgba, 4, I->own_rgba[k], static_cast<u32>(n), &in));
  // (the staging buffers are the engine's own: no ordering against the caller's stream for them)
  const bool producer = I->has_producer;
  I->has_producer = false;
  const int st = integrate_device(I, T_G_C, I->own_xyz[k], rgba ? I->own_rgba[k] : nullptr, static_cast<u32>(n), freespace, false, in);
  I->has_producer = producer;
  return st;
}

int cox_integrator_wait_inputs(cox_integrator_t* I) {
  COX_ENTRY_NO_DRAIN();
  if (!I) return COX_ERR_INVALID_ARG;
  if (I->proj) return COX_OK;  // (the projective integrator's host entry point is synchronous)
  COX_HIP(hipSetDevice(I->layer->device));
  for (int k = 0; k < kInputSets; ++k)
    if (I->in_used[k]) COX_HIP(hipEventSynchronize(I->in_ready[k]));
  return COX_OK;
}

// depth image (device) -> point list in staging set k on stream s; the point count lands in d_depth_n[k]
static void convert_depth(cox_integrator* I, int k, const float* depth_dev, const uint8_t* rgba_dev, int w, int h, const float K[4], hipStream_t s) {
  const u32 n = static_cast<u32>(w) * static_cast<u32>(h);
  const dim3 gt(std::max<u32>(1, (n + kDepthTile - 1) / kDepthTile));
  u32* tile_sums = I->depth_flag + static_cast<size_t>(k) * (I->pcap / kDepthTile + 1);  // one set per staging set: consecutive frames convert on different streams
  hipLaunchKernelGGL(k_depth_count, gt, dim3(256), 0, s, depth_dev, n, tile_sums);
  hipLaunchKernelGGL(k_depth_points, gt, dim3(256), 0, s, depth_dev, rgba_dev, w, h, K[0], K[1], K[2], K[3], tile_sums, I->own_xyz[k], I->own_rgba[k],
                     I->d_depth_n + k);
}

int cox_integrate_depth_dev(cox_integrator_t* I, const float T_G_C[7], const float* depth_dev, const uint8_t* rgba_dev, int w, int h, const float K[4]) {
  COX_ENTRY_NO_DRAIN();
  if (!I || !T_G_C || !depth_dev || !K || w <= 0 || h <= 0 || static_cast<uint64_t>(w) * h > 0x7FFFFFFFull) return COX_ERR_INVALID_ARG;
  if (I->proj) return COX_ERR_UNSUPPORTED;  // the projective integrator takes point clouds (it builds its own range image)
  COX_HIP(hipSetDevice(I->layer->device));
  const u32 n = static_cast<u32>(w) * static_cast<u32>(h);
  COX_TRY(ensure_capacity(I, n));
  // Frames stay in flight and nothing comes back to the host: the point list goes to the staging set of the frame's bundle set
  // (read by its stages H .. M), converted on the frame's ray-generation stream, and the point count -- which the "mixed" visiting
  // order is a function of -- stays in device memory: the frame's first kernel takes it from there (k_bundle_insert / k_params_count).
  if (I->submitter) I->submitter->wait_outstanding(1);
  const int k = static_cast<int>((I->frame_no + 1) % kInputSets);
  const hipStream_t s = next_frame_stream(I);  // (not the input stream: a wait for the caller's stream would sit in its queue)
  COX_TRY(wait_staging_set_free(I, k, s));
  if (I->has_producer) {  // the images were written on the caller's stream
    COX_HIP(hipEventRecord(I->ev_producer, I->producer));
    COX_HIP(hipStreamWaitEvent(s, I->ev_producer, 0));
  }
  convert_depth(I, k, depth_dev, rgba_dev, w, h, K, s);
  if (I->has_producer) {  // the images are not read after this: later work on the caller's stream may overwrite / free them
    COX_HIP(hipEventRecord(I->ev_inputs_read, s));
    COX_HIP(hipStreamWaitEvent(I->producer, I->ev_inputs_read, 0));
  }
  COX_HIP(hipEventRecord(I->in_ready[k], s));
  I->in_used[k] = true;
  FrameInput in;
  in.consumed = I->in_free[k];
  in.n_dev = I->d_depth_n + k;
  // (the staging buffers are the engine's own: no producer ordering for them)
  const bool producer = I->has_producer;
  I->has_producer = false;
  const int st = integrate_device(I, T_G_C, I->own_xyz[k], I->own_rgba[k], n, 0, false, in);
  I->has_producer = producer;
  return st;
}

int cox_integrate_depth_async(cox_integrator_t* I, const float T_G_C[7], const float* depth, const uint8_t* rgba, int w, int h, const float K[4]) {
  COX_ENTRY_NO_DRAIN();
  if (!I || !T_G_C || !depth || !K || w <= 0 || h <= 0 || static_cast<uint64_t>(w) * h > 0x7FFFFFFFull) return COX_ERR_INVALID_ARG;
  if (I->proj) return COX_ERR_UNSUPPORTED;
  COX_HIP(hipSetDevice(I->layer->device));
  const u32 n = static_cast<u32>(w) * static_cast<u32>(h);
  COX_TRY(ensure_capacity(I, n));
  if (I->submitter) I->submitter->wait_outstanding(1);
  const int k = static_cast<int>((I->frame_no + 1) % kInputSets);
  FrameInput in;
  COX_TRY(stage_host_inputs(I, k, depth, sizeof(float), I->own_depth[k], rgba, 4, I->own_depth_rgba[k], n, &in));
  // the conversion (two kernels) runs on the frame's own stream behind the copies' event: the input stream shares the apply's pipe for
  // `merged`, and kernels there cost more than 25 us on the ray-generation chain do (640x480 images, frames/s: 6 380 vs 5 170; `fast`,
  // whose input stream has a pipe of its own: 2 780 vs 2 830).  COX_DEPTH_CONVERT=input puts it on the input stream.
  static const bool on_frame_stream = !(std::getenv("COX_DEPTH_CONVERT") && std::string(std::getenv("COX_DEPTH_CONVERT")) == "input");
  if (I->st_in && on_frame_stream) {
    COX_HIP(hipStreamWaitEvent(next_frame_stream(I), I->in_ready[k], 0));
    in.ready = nullptr;
    convert_depth(I, k, I->own_depth[k], rgba ? I->own_depth_rgba[k] : nullptr, w, h, K, next_frame_stream(I));
  } else {
    convert_depth(I, k, I->own_depth[k], rgba ? I->own_depth_rgba[k] : nullptr, w, h, K, next_input_stream(I));
    if (I->st_in) COX_HIP(hipEventRecord(I->in_ready[k], I->st_in));  // (recorded again: behind the conversion)
  }
  in.n_dev = I->d_depth_n + k;
  const bool producer = I->has_producer;
  I->has_producer = false;
  const int st = integrate_device(I, T_G_C, I->own_xyz[k], I->own_rgba[k], n, 0, false, in);
  I->has_producer = producer;
  return st;
}

int cox_integrator_set_input_stream(cox_integrator_t* I, void* hip_stream, int enable) {
  COX_ENTRY();
  if (!I) return COX_ERR_INVALID_ARG;
  I->has_producer = enable != 0;
  I->producer = static_cast<hipStream_t>(hip_stream);
  return COX_OK;
}

int cox_integrator_sync(cox_integrator_t* I) {
  COX_ENTRY();
  if (!I) return COX_ERR_INVALID_ARG;
  COX_HIP(hipSetDevice(I->layer->device));
  if (I->proj) return cox_proj_sync(I->proj);
  return integrator_finish(I);
}

int cox_integrator_last_stats(cox_integrator_t* I, cox_frame_stats* stats) {
  COX_ENTRY();
  if (!I || !stats) return COX_ERR_INVALID_ARG;
  COX_HIP(hipSetDevice(I->layer->device));
  if (I->proj) return cox_proj_last_stats(I->proj, stats);
  COX_TRY(sync_all(I));
  COX_TRY(fold_counters(I));
  *stats = I->last;
  return COX_OK;
}

int cox_integrator_set_profiling(cox_integrator_t* I, int on) {
  COX_ENTRY();
  if (!I) return COX_ERR_INVALID_ARG;
  I->profiling = on > 0;
  I->profile_every = on > 1 ? static_cast<u32>(on) : 1u;
  if (on > 0 && !I->proj && !I->timeline && std::getenv("COX_TIMELINE")) {
    I->timeline = fopen(std::getenv("COX_TIMELINE"), "a");
    if (I->timeline && hipEventCreate(&I->timeline_ref) == hipSuccess) {
      COX_HIP(hipSetDevice(I->layer->device));
      COX_TRY(sync_all(I));
      COX_HIP(hipEventRecord(I->timeline_ref, I->st[0]));
      fprintf(I->timeline, "# integrator %p method %d\n", static_cast<void*>(I), I->method);
    }
  }
  return COX_OK;
}

int cox_integrator_kernel_time(cox_integrator_t* I, double* apply_ms, uint64_t* apply_launches, int reset) {
  COX_ENTRY();
  if (!I) return COX_ERR_INVALID_ARG;
  if (I->proj) return COX_ERR_UNSUPPORTED;
  COX_HIP(hipSetDevice(I->layer->device));
  int st = integrator_finish(I);
  if (apply_ms) *apply_ms = I->class_ms[COX_KC_APPLY];
  if (apply_launches) *apply_launches = I->class_regions[COX_KC_APPLY];
  if (reset) {
    I->class_ms[COX_KC_APPLY] = 0.0;
    I->class_regions[COX_KC_APPLY] = 0;
  }
  return st;
}

int cox_selftest_division(int device, uint64_t n, uint64_t seed, uint64_t* mismatches) {
  COX_ENTRY();
  if (!mismatches) return COX_ERR_INVALID_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return COX_ERR_NO_DEVICE;
  COX_HIP(hipSetDevice(device));
  u32* d = nullptr;
  COX_HIP(hipMalloc(reinterpret_cast<void**>(&d), sizeof(u32)));
  COX_HIP(hipMemset(d, 0, sizeof(u32)));
  hipLaunchKernelGGL(k_selftest_division, dim3(4096), dim3(256), 0, nullptr, n, seed, d);
  u32 h = 0;
  COX_HIP(hipMemcpy(&h, d, sizeof(u32), hipMemcpyDeviceToHost));
  (void)hipFree(d);
  *mismatches = h;
  return COX_OK;
}

int cox_integrator_stage_times(cox_integrator_t* I, double ms[2], uint64_t launches[2], int reset) {
  COX_ENTRY();
  if (!I || !ms || !launches) return COX_ERR_INVALID_ARG;
  if (I->proj) return COX_ERR_UNSUPPORTED;
  COX_HIP(hipSetDevice(I->layer->device));
  int st = integrator_finish(I);
  ms[0] = I->class_ms[COX_KC_MERGE];
  launches[0] = I->class_regions[COX_KC_MERGE];
  ms[1] = I->class_ms[COX_KC_APPLY];
  launches[1] = I->class_regions[COX_KC_APPLY];
  if (reset)
    for (int k : {COX_KC_MERGE, COX_KC_APPLY}) {
      I->class_ms[k] = 0.0;
      I->class_regions[k] = 0;
    }
  return st;
}

int cox_integrator_host_time(cox_integrator_t* I, double* ms_total, uint64_t* frames, int reset) {
  COX_ENTRY();
  if (!I || !ms_total || !frames) return COX_ERR_INVALID_ARG;
  *ms_total = static_cast<double>(I->host_ns - I->host_wait_ns) * 1e-6;  // enqueueing only: waits for a free slot are the GPU's time
  *frames = I->host_frames;
  if (reset) {
    I->host_ns = 0;
    I->host_wait_ns = 0;
    I->host_frames = 0;
  }
  return COX_OK;
}

int cox_integrator_fast_stats(cox_integrator_t* I, uint64_t out[10]) {
  COX_ENTRY();
  if (!I || !out) return COX_ERR_INVALID_ARG;
  if (I->method != COX_METHOD_FAST || I->proj) return COX_ERR_UNSUPPORTED;
  COX_HIP(hipSetDevice(I->layer->device));
  COX_TRY(sync_all(I));
  u32 h[16] = {};
  COX_HIP(hipMemcpy(h, I->fast.d_stats, sizeof(h), hipMemcpyDeviceToHost));
  for (int k = 0; k < 4; ++k) out[k] = h[k];
  out[4] = I->fast.frames;
  out[5] = h[4];
  out[6] = h[5];
  out[7] = h[6];
  out[8] = static_cast<uint64_t>(h[7]) * 10ull;  // ticks of 10 ns
  out[9] = static_cast<uint64_t>(h[8]) * 10ull;
  if (std::getenv("COX_DEBUG")) fprintf(stderr, "[coxgraph_hip] fast: long rays %u, visits of the last round %llu, of round 0 %llu (totals over %llu frames)\n", h[9],
                                        static_cast<unsigned long long>(h[10]) * 16ull, static_cast<unsigned long long>(h[11]) * 16ull, static_cast<unsigned long long>(I->fast.frames));
  return COX_OK;
}

int cox_integrator_update_stats(cox_integrator_t* I, uint64_t out[2]) {
  COX_ENTRY();
  if (!I || !out) return COX_ERR_INVALID_ARG;
  out[0] = out[1] = 0;
  if (I->proj) return COX_OK;
  COX_HIP(hipSetDevice(I->layer->device));
  COX_TRY(integrator_finish(I));
  out[0] = I->last_big_tiles;
  out[1] = I->last_big_chunks;
  return COX_OK;
}

int cox_integrator_class_times(cox_integrator_t* I, double ms[COX_KERNEL_CLASSES], uint64_t regions[COX_KERNEL_CLASSES], int reset) {
  COX_ENTRY();
  if (!I || !ms || !regions) return COX_ERR_INVALID_ARG;
  if (I->proj) return COX_ERR_UNSUPPORTED;
  COX_HIP(hipSetDevice(I->layer->device));
  int st = integrator_finish(I);
  for (int k = 0; k < COX_KERNEL_CLASSES; ++k) {
    ms[k] = I->class_ms[k];
    regions[k] = I->class_regions[k];
    if (reset) {
      I->class_ms[k] = 0.0;
      I->class_regions[k] = 0;
    }
  }
  return st;
}

}  // extern "C"
