// voxblox FastTsdfIntegrator on the GPU, reproducing the reference's result at integrator_threads = 1.
// Included by cox_integrator.hip (uses its RayArrays / Counters / helpers).
//
// The reference walks the points one after the other (mixed order) and keeps two lossy sets
// (ApproxHashSet<20, 10000, GlobalIndex, LongIndexHash>: 2^20 slots, slot = hash & mask, "replaceHash" returns
// whether the slot held another value and stores the new one):
//   * start set: a point whose half-voxel cell's hash is already in its slot is skipped;
//   * observed set: a ray walks from the surface towards the sensor; a voxel whose hash is already in its slot counts
//     as a collision; after more than max_consecutive_ray_collisions collisions in a row the ray stops.
// Both are "last writer wins" tables, so what an operation sees is the hash of the PREVIOUS OPERATION ON ITS SLOT in
// visiting order (or what earlier frames left there).  That makes the sequential algorithm a sort problem:
//   start set     every valid point performs its operation whatever the outcome -> sort the points by slot (stable, from
//                 visiting order), compare each with its predecessor in the slot.  One pass, exact.
//   observed set  which operations happen depends on where earlier rays stopped.  reach[r] (voxels ray r updates) is the
//                 unique solution of a triangular system (a ray only depends on lower-numbered rays); it is found by
//                 Jacobi sweeps: given a guess of every ray's reach, the operations are known; sort all candidate
//                 visits by slot once per frame, and per sweep (a) prefix-max over "is this visit performed" gives every
//                 visit the previous performed operation on its slot, (b) every ray re-walks its visits and recomputes
//                 where it stops.  A sweep that changes nothing has reached the sequential result (15-18 sweeps on the
//                 benchmark frames; each sweep fixes at least the lowest-numbered wrong ray, so it always terminates).
// The updates themselves then go through the `simple` pipeline (records sorted by voxel, replayed in visiting order).
#pragma once

namespace cox {

constexpr u32 kFastSlotBits = 20;
constexpr u32 kFastSlots = 1u << kFastSlotBits, kFastSlotMask = kFastSlots - 1u;
constexpr u32 kFastFullReset = 10000;  // ApproxHashSet full_reset_threshold
constexpr u32 kFastTile = 2048;        // positions per workgroup in the prefix-max

struct FastFrame {  // per-frame constants of the fast integrator
  u64 off_start, off_obs;  // ApproxHashSet::offset_ of the two sets
  int max_collisions;
};

// voxblox LongIndexHash on a GlobalIndex (int64 components), in size_t arithmetic
__device__ __forceinline__ u64 long_index_hash(int x, int y, int z) {
  constexpr u64 sl = 17191ull, sl2 = sl * sl;
  return static_cast<u64>(static_cast<int64_t>(x)) + static_cast<u64>(static_cast<int64_t>(y)) * sl + static_cast<u64>(static_cast<int64_t>(z)) * sl2;
}

// ---- start set ----------------------------------------------------------------------------------------------------
// thread = visiting sequence number: the point's start-set hash (hash of its half-voxel cell + offset) and slot
__global__ void __launch_bounds__(256) k_fast_points(const FrameParams* __restrict__ Pp, FastFrame FF, u64* __restrict__ fhash, u32* __restrict__ skey,
                                                     u32* __restrict__ sval, Counters* cnt) {
  const FrameParams P = *Pp;
  for (u32 seq = blockIdx.x * blockDim.x + threadIdx.x; seq < P.n_points; seq += gridDim.x * blockDim.x) {
    const u32 idx = mixed_index(seq, P.n_points);
    const F3 p{P.xyz[3 * idx], P.xyz[3 * idx + 1], P.xyz[3 * idx + 2]};
    bool clearing = false;
    bool valid = point_valid(P, p, &clearing);
    u32 key = kFastSlots;  // sorts behind every slot
    u64 h = 0;
    if (valid) {
      const F3 pg = transform_point(P, p);
      const float sx = pg.x * P.start_subsampling_inv, sy = pg.y * P.start_subsampling_inv, sz = pg.z * P.start_subsampling_inv;
      if (!(index_in_range(sx) && index_in_range(sy) && index_in_range(sz))) {
        atomicOr(&cnt->err, kErrRange);
        valid = false;
      } else {
        h = long_index_hash(grid_index(sx), grid_index(sy), grid_index(sz)) + FF.off_start;
        key = static_cast<u32>(h) & kFastSlotMask;
      }
    }
    fhash[seq] = h;
    skey[seq] = key;
    sval[seq] = seq;
    const u64 m = __ballot(valid);
    if (lane_id() == 0 && m) atomicAdd(&cnt->shard[(seq >> 6) & 63u][kShValid], static_cast<u32>(__popcll(m)));
  }
}
// thread = position in the slot-sorted order: fresh = the slot did not hold this hash
__global__ void __launch_bounds__(256) k_fast_start_flags(const FrameParams* __restrict__ Pp, const u32* __restrict__ skey, const u32* __restrict__ sval,
                                                          const u64* __restrict__ fhash, const u64* __restrict__ table_start, u32* __restrict__ fresh) {
  const u32 n = Pp->n_points;
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const u32 key = skey[i], seq = sval[i];
    u32 f = 0;
    if (key < kFastSlots) {
      const u64 h = fhash[seq];
      const u64 prev = (i > 0 && skey[i - 1] == key) ? fhash[sval[i - 1]] : table_start[key];
      f = (prev != h) ? 1u : 0u;
    }
    fresh[seq] = f;
  }
}
// the table keeps the hash of the last operation on each slot
__global__ void __launch_bounds__(256) k_fast_start_commit(const FrameParams* __restrict__ Pp, const u32* __restrict__ skey, const u32* __restrict__ sval,
                                                           const u64* __restrict__ fhash, u64* __restrict__ table_start) {
  const u32 n = Pp->n_points;
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const u32 key = skey[i];
    if (key < kFastSlots && (i + 1 == n || skey[i + 1] != key)) table_start[key] = fhash[sval[i]];
  }
}
// thread = sequence number: the points that start a ray, compacted in visiting order (ray id = rank among them)
__global__ void __launch_bounds__(256) k_fast_rays(const FrameParams* __restrict__ Pp, const u32* __restrict__ fresh, const u32* __restrict__ rank, RayArrays R,
                                                   u32* __restrict__ cap, u32 cap0, Counters* cnt) {
  const FrameParams P = *Pp;
  for (u32 seq = blockIdx.x * blockDim.x + threadIdx.x; seq < P.n_points; seq += gridDim.x * blockDim.x) {
    if (!fresh[seq]) continue;
    const u32 r = rank[seq];
    const u32 idx = mixed_index(seq, P.n_points);
    const F3 p{P.xyz[3 * idx], P.xyz[3 * idx + 1], P.xyz[3 * idx + 2]};
    bool clearing = false;
    (void)point_valid(P, p, &clearing);
    const F3 pg = transform_point(P, p);
    Dda d;
    dda_setup(d, P, pg, clearing);  // P.cast_from_origin == 0: from the far end of the ray towards the sensor
    if (d.range_error) atomicOr(&cnt->err, kErrRange);
    R.px[r] = pg.x;
    R.py[r] = pg.y;
    R.pz[r] = pg.z;
    R.w[r] = voxel_weight(P, p);
    R.color[r] = pack_rgba_wire(P.rgba, idx);
    R.flags[r] = 1u | (clearing ? 2u : 0u);
    R.nsteps[r] = d.nsteps;  // the whole walk; replaced by the ray's reach once that is known
    cap[r] = min(d.nsteps, cap0);  // candidate visits are generated for the first cap[r] steps (k_fast_grow_caps)
  }
}

// ---- observed set -------------------------------------------------------------------------------------------------
// wave = ray: every voxel the ray would visit if nothing stopped it, in walking order (parallel DDA of the merged
// integrator's bundles, wave_ray_path; sequential on lane 0 for the rays that one does not cover)
template <u32 kAxisCap>
__global__ void __launch_bounds__(256) k_fast_visits(const FrameParams* __restrict__ Pp, FastFrame FF, RayArrays R, u64* __restrict__ vhash, u32* __restrict__ vkey,
                                                     u32* __restrict__ vval, u32* __restrict__ vray, u32* __restrict__ reach, const u32* __restrict__ cap, int init_reach, u32 vcap,
                                                     Counters* cnt) {
  const FrameParams P = *Pp;
  __shared__ float lds_t[4][3 * kAxisCap];
  __shared__ u32 lds_path[4][3 * kAxisCap];
  const u32 n_rays = uniform_u32(cnt->n_rays);
  const bool overflow = uniform_u32(cnt->n_records) > vcap;  // cannot happen unless the worst-case bound itself exceeds the 2^31 limit
  if (overflow && blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&cnt->err, kErrRecords);
  const u32 lane = lane_id();
  const u32 wave = threadIdx.x >> 6;
  float* tl = lds_t[wave];
  u32* path = lds_path[wave];
  const u32 waves_total = (gridDim.x * blockDim.x) >> 6;
  for (u32 r = uniform_u32((blockIdx.x * blockDim.x + threadIdx.x) >> 6); r < n_rays; r += waves_total) {
    if (overflow) {  // the frame is dropped as a whole
      if (lane == 0) {
        R.nsteps[r] = 0;
        reach[r] = 0;
      }
      continue;
    }
    const u32 nfull = uniform_u32(R.nsteps[r]);
    const u32 ns = uniform_u32(cap[r]);  // <= nfull
    if (lane == 0 && init_reach) reach[r] = ns;
    if (ns == 0) continue;
    const u32 off = uniform_u32(R.rec_off[r]);
    const F3 pg{readlane_f32(R.px[r], 0), readlane_f32(R.py[r], 0), readlane_f32(R.pz[r], 0)};
    Dda d;
    dda_setup(d, P, pg, (uniform_u32(R.flags[r]) & 2u) != 0);
    if (wave_ray_path<kAxisCap>(d, nfull, tl, path, lane, ns)) {
      for (u32 s = lane; s < ns; s += 64) {
        const u32 p = path[s];
        const int x = d.c[0] + static_cast<int>(p & 1023u) * d.sgn[0];
        const int y = d.c[1] + static_cast<int>((p >> 10) & 1023u) * d.sgn[1];
        const int z = d.c[2] + static_cast<int>(p >> 20) * d.sgn[2];
        const u64 h = long_index_hash(x, y, z) + FF.off_obs;
        vhash[off + s] = h;
        vkey[off + s] = static_cast<u32>(h) & kFastSlotMask;
        vval[off + s] = off + s;
        vray[off + s] = r;
      }
      wave_lds_handover();  // the next ray of this wave reuses the LDS scratch
    } else if (lane == 0) {
      for (u32 s = 0; s < ns; ++s) {
        const u64 h = long_index_hash(d.c[0], d.c[1], d.c[2]) + FF.off_obs;
        dda_step(d);
        vhash[off + s] = h;
        vkey[off + s] = static_cast<u32>(h) & kFastSlotMask;
        vval[off + s] = off + s;
        vray[off + s] = r;
      }
    }
  }
}

// visits of the frame (0 when the frame was dropped for overflowing the visit buffers)
__device__ __forceinline__ u32 fast_num_visits(const Counters* cnt, u32 vcap) { return cnt->n_records > vcap ? 0u : cnt->n_records; }
// the visits in slot-sorted order: where each visit went (pos_of), and per sorted position its ray, its step on that
// ray and its hash (so the sweeps read coalesced arrays instead of chasing visit -> ray -> offset)
__global__ void __launch_bounds__(256) k_fast_inverse(const u32* __restrict__ vval_sorted, const u32* __restrict__ vray, const u64* __restrict__ vhash,
                                                      const u32* __restrict__ voff, u32* __restrict__ pos_of, u32* __restrict__ sray, u32* __restrict__ sstep,
                                                      u64* __restrict__ shash, const Counters* cnt, u32 vcap) {
  const u32 n = fast_num_visits(cnt, vcap);
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const u32 v = vval_sorted[i];
    const u32 r = vray[v];
    pos_of[v] = i;
    sray[i] = r;
    sstep[i] = v - voff[r];
    shash[i] = vhash[v];
  }
}

template <int NW>
__device__ __forceinline__ u32 block_exclusive_max(u32 v, u32* total, u32* lds /*[NW]*/) {
  const u32 lane = lane_id();
  const u32 wave = threadIdx.x >> 6;
  u32 inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const u32 o = static_cast<u32>(__shfl_up(static_cast<int>(inc), off, 64));
    if (lane >= static_cast<u32>(off)) inc = max(inc, o);
  }
  u32 ex = static_cast<u32>(__shfl_up(static_cast<int>(inc), 1, 64));
  if (lane == 0) ex = 0;
  __syncthreads();
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  u32 base = 0, tot = 0;
#pragma unroll
  for (u32 w = 0; w < NW; ++w) {
    const u32 s = lds[w];
    if (w < wave) base = max(base, s);
    tot = max(tot, s);
  }
  *total = tot;
  return max(base, ex);
}

struct FastVisits {
  const u32* skey;    // slot of the visit at sorted position i
  const u32* sray;    // its ray
  const u32* sstep;   // its step on that ray
  const u64* shash;   // its hash
  const u32* voff;    // first visit of ray r (ray-major numbering)
  const u32* pos_of;  // sorted position of visit v
};
// is the visit at sorted position i performed under the current guess?  (steps 0 .. reach, the last one being the
// operation that made the ray stop; a ray that never stops has reach == its whole walk)
__device__ __forceinline__ bool fast_active(const FastVisits& V, const u32* __restrict__ reach, u32 i) { return V.sstep[i] <= reach[V.sray[i]]; }
// sweep, part 1: eloc[i] = 1 + the last performed position before i inside i's tile (0 = none), tmax[tile] = same over the tile
// prev_changed (both sweep kernels): the "changed" flag of the sweep before this one.  Zero means the iteration has converged:
// both reach buffers hold the fixed point and eloc / tmax belong to it, so a sweep enqueued speculatively (the host reads the
// flags one batch behind) returns at once instead of reproducing the fixed point.
__global__ void __launch_bounds__(256) k_fast_scan_tiles(FastVisits V, const u32* __restrict__ reach, u32* __restrict__ eloc, u32* __restrict__ tmax,
                                                         const Counters* cnt, u32 vcap, const u32* __restrict__ prev_changed) {
  __shared__ u32 lds[4];
  if (prev_changed && *prev_changed == 0u) return;
  const u32 n = fast_num_visits(cnt, vcap);
  const u32 n_tiles = (n + kFastTile - 1) / kFastTile;
  for (u32 tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const u32 base = tile * kFastTile + threadIdx.x * 8;
    u32 a[8], mx = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const u32 i = base + q;
      a[q] = (i < n && fast_active(V, reach, i)) ? i + 1 : 0u;
      mx = max(mx, a[q]);
    }
    u32 total;
    u32 ex = block_exclusive_max<4>(mx, &total, lds);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (base + q < n) eloc[base + q] = ex;
      ex = max(ex, a[q]);
    }
    if (threadIdx.x == 0) tmax[tile] = total;
  }
}
// 1 + sorted position of the last performed visit of slot `key` before position i (0 = none this frame).
// eloc covers i's own tile; a slot's visits are contiguous, so earlier tiles only matter while the run reaches back
// into them, and there the candidate is that tile's last performed visit (tmax): either it belongs to the run, or --
// keys being sorted -- no performed visit of the run lies in that tile and the run starts behind it.
__device__ __forceinline__ u32 fast_prev_performed(const FastVisits& V, const u32* __restrict__ eloc, const u32* __restrict__ tmax, u32 i, u32 key) {
  const u32 e = eloc[i];
  if (e > 0) return V.skey[e - 1] == key ? e : 0u;
  u32 t = i / kFastTile;
  while (t > 0 && V.skey[t * kFastTile - 1] == key) {
    --t;
    const u32 m = tmax[t];
    if (m > 0) return V.skey[m - 1] == key ? m : 0u;
  }
  return 0u;
}
// was the slot of the visit at sorted position i last written with the same hash?
__device__ __forceinline__ bool fast_collision(const FastVisits& V, const u32* __restrict__ eloc, const u32* __restrict__ tmax,
                                               const u64* __restrict__ table_obs, u32 i) {
  const u32 key = V.skey[i];
  const u32 e = fast_prev_performed(V, eloc, tmax, i, key);
  const u64 prev = e ? V.shash[e - 1] : table_obs[key];
  return prev == V.shash[i];
}
// sweep, part 3: wave = ray.  Up to 256 steps of the walk are tested at once; the reference's "more than max_collisions
// collisions in a row" is the first lane whose run of set bits (continued from the previous 64 steps) is long enough.
__global__ void __launch_bounds__(256) k_fast_sweep(FastVisits V, int max_collisions, const u32* __restrict__ nfull, const u32* __restrict__ eloc,
                                                    const u32* __restrict__ tmax, const u64* __restrict__ table_obs, const u32* __restrict__ reach_in,
                                                    u32* __restrict__ reach_out, u32* __restrict__ changed, const Counters* cnt, u32 vcap,
                                                    const u32* __restrict__ prev_changed) {
  if (cnt->n_records > vcap) return;  // frame dropped (k_fast_visits): there are no visits to walk
  if (prev_changed && *prev_changed == 0u) return;  // converged (see k_fast_scan_tiles); changed[this sweep] stays 0
  const u32 n_rays = cnt->n_rays;
  const u32 lane = lane_id();
  const u32 n_waves = (gridDim.x * blockDim.x) >> 6;
  bool any = false;
  for (u32 r = uniform_u32((blockIdx.x * blockDim.x + threadIdx.x) >> 6); r < n_rays; r += n_waves) {
    const u32 ns = nfull[r], off = V.voff[r];
    u32 carry = 0, stop = ns;
    // Most rays stop within their first few steps and every tested step costs half a dozen gathers, so the walk is
    // tested in growing segments: 16 steps, the rest of the first 64, then 256 at a time (four independent gathers
    // per lane in flight) for the few rays that are still going.
    u32 base = 0, seg = 16;
    while (base < ns && stop == ns) {
      const u32 len = uniform_u32(min(seg, ns - base));
      bool coll[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u32 q = 64u * j + lane;
        coll[j] = (q < len) && fast_collision(V, eloc, tmax, table_obs, V.pos_of[off + base + q]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (stop != ns || 64u * j >= len) break;
        const u32 sub = uniform_u32(min(64u, len - 64u * j));  // lanes of this group that hold a step
        const u64 m = __ballot(coll[j]);
        // length of the run of collisions that ends at this lane
        const u64 zeros_below = ~m & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
        const u32 run = zeros_below ? lane - (63u - static_cast<u32>(__clzll(static_cast<long long>(zeros_below)))) : lane + 1u + carry;
        const u64 hit = __ballot(coll[j] && run > static_cast<u32>(max_collisions));
        if (hit) stop = base + 64u * j + static_cast<u32>(__ffsll(static_cast<long long>(hit))) - 1u;
        carry = static_cast<u32>(__builtin_amdgcn_readlane(static_cast<int>(run), static_cast<int>(sub - 1u)));
      }
      base += len;
      seg = (base < 64u) ? 64u - base : 256u;
    }
    if (lane == 0) {
      reach_out[r] = stop;
      any |= (stop != reach_in[r]);
    }
  }
  // one word for the whole grid: only the first few waves that changed something write it (a same-address atomic per wave of the
  // first sweep, where every ray changes, was 50 us of serialised traffic)
  if (any && __hip_atomic_load(changed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) atomicOr(changed, 1u);
}
// after the last sweep: the table keeps the hash of the last performed operation on each slot
__global__ void __launch_bounds__(256) k_fast_obs_commit(FastVisits V, const u32* __restrict__ reach, const u32* __restrict__ eloc, const u32* __restrict__ tmax,
                                                         u64* __restrict__ table_obs, const Counters* cnt, u32 vcap) {
  const u32 n = fast_num_visits(cnt, vcap);
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const u32 key = V.skey[i];
    if (i + 1 < n && V.skey[i + 1] == key) continue;  // not the end of the slot's run
    const u32 e = fast_active(V, reach, i) ? i + 1 : fast_prev_performed(V, eloc, tmax, i, key);
    if (e) table_obs[key] = V.shash[e - 1];
  }
}
// Candidate visits are generated for the first cap[r] steps of a ray only (most rays stop within a few steps, their walks
// are hundreds of voxels long at fine voxel sizes).  A converged ray that reached its cap without stopping may want to go
// on: it gets its whole walk and the round is repeated with the longer list; rays that stopped are final as far as their own
// list goes, and the fixed point of the last round (nobody at a cap short of its whole walk) contains every performed visit,
// so it is the sequential result.
__global__ void __launch_bounds__(256) k_fast_grow_caps(const u32* __restrict__ nfull, u32* __restrict__ cap, u32* __restrict__ reach_a, u32* __restrict__ reach_b,
                                                        u32* __restrict__ grew, const Counters* cnt) {
  const u32 n_rays = cnt->n_rays;
  bool any = false;
  for (u32 r = blockIdx.x * blockDim.x + threadIdx.x; r < n_rays; r += gridDim.x * blockDim.x) {
    const u32 c = cap[r], nf = nfull[r];
    if (c < nf && reach_a[r] >= c) {
      const u32 nc = nf;  // a ray that got through its first steps unstopped mostly goes all the way: take the whole walk
      cap[r] = nc;
      reach_a[r] = nc;  // guess: it keeps going
      reach_b[r] = nc;
      any = true;
    }
  }
  if (__ballot(any) && lane_id() == 0 && __hip_atomic_load(grew, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) atomicOr(grew, 1u);
}
// hand the rays over to the record pipeline: a ray emits its first reach[r] voxels
__global__ void __launch_bounds__(256) k_fast_finish(RayArrays R, const u32* __restrict__ reach, Counters* cnt) {
  const u32 n_rays = cnt->n_rays;
  if (blockIdx.x == 0 && threadIdx.x == 0) cnt->n_ray_slots = n_rays;
  for (u32 r = blockIdx.x * blockDim.x + threadIdx.x; r < n_rays; r += gridDim.x * blockDim.x) R.nsteps[r] = reach[r];
}

}  // namespace cox
