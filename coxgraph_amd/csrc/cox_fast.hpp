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
                                                   u32* __restrict__ cap, u32* __restrict__ reach, u32 cap0, u32 vcap0, Counters* cnt) {
  const FrameParams P = *Pp;
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // round 0: cap0 list slots per ray at a fixed stride (no scan)
    const u64 nv = static_cast<u64>(cnt->n_rays) * cap0;
    if (nv > vcap0) {  // (more rays than list slots: only beyond 2^32 / cap0 points; the sequential kernel takes the frame)
      cnt->fast.n_visits[0] = 0u;
      cnt->fast.overflow = 2u;
    } else {
      cnt->fast.n_visits[0] = static_cast<u32>(nv);
    }
  }
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
    const u32 c = min(d.nsteps, cap0);
    cap[r] = c;    // candidate visits are generated for the first cap[r] steps (k_fast_grow gives a ray its whole walk)
    reach[r] = c;  // first guess: nobody stops
  }
}

// ---- observed set -------------------------------------------------------------------------------------------------
// wave = ray: the first cap[r] voxels the ray would visit if nothing stopped it, in walking order (parallel DDA of the merged
// integrator's bundles, wave_ray_path; sequential on lane 0 for the rays that one does not cover).
// Round 0 (stride > 0): ray r owns the list slots [r * stride, (r + 1) * stride); slots past the end of a short walk get the
// key kFastSlots, which sorts behind every slot.  Round 1 (stride == 0): lists at voff[r] (scan of the grown caps), only when
// round 0 grew something.
template <u32 kAxisCap>
__global__ void __launch_bounds__(256) k_fast_visits(const FrameParams* __restrict__ Pp, FastFrame FF, RayArrays R, u64* __restrict__ vhash, u32* __restrict__ vkey,
                                                     u32* __restrict__ vval, u32* __restrict__ vray, u32* __restrict__ voff, const u32* __restrict__ cap, u32 stride,
                                                     int round, Counters* cnt) {
  const FrameParams P = *Pp;
  __shared__ float lds_t[4][3 * kAxisCap];
  __shared__ u32 lds_path[4][3 * kAxisCap];
  if (uniform_u32(cnt->fast.overflow) != 0u) return;  // lists that do not fit: the sequential kernel takes the frame
  if (round > 0 && uniform_u32(cnt->fast.grew[round]) == 0u) return;
  const u32 n_rays = uniform_u32(cnt->n_rays);
  const u32 lane = lane_id();
  const u32 wave = threadIdx.x >> 6;
  float* tl = lds_t[wave];
  u32* path = lds_path[wave];
  const u32 waves_total = (gridDim.x * blockDim.x) >> 6;
  for (u32 r = uniform_u32((blockIdx.x * blockDim.x + threadIdx.x) >> 6); r < n_rays; r += waves_total) {
    const u32 nfull = uniform_u32(R.nsteps[r]);
    const u32 ns = uniform_u32(cap[r]);  // <= nfull
    const u32 off = stride ? r * stride : uniform_u32(voff[r]);
    if (stride) {
      if (lane == 0) voff[r] = off;
      for (u32 s = ns + lane; s < stride; s += 64) {
        vkey[off + s] = kFastSlots;
        vval[off + s] = off + s;
      }
    }
    if (ns == 0) continue;
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

struct FastVisits {
  const u32* skey;    // slot of the visit at sorted position i
  const u32* sinfo;   // its ray << kFastStepBits | its step on that ray
  const u64* shash;   // its hash
  const u32* voff;    // first visit of ray r (ray-major numbering)
  const u32* pos_of;  // sorted position of visit v
};
constexpr u32 kFastStepBits = 11, kFastStepMask = (1u << kFastStepBits) - 1u;  // walks of up to 2047 steps, 2^21 rays (else: the sequential kernel)
// reach[] is updated in place by the relaxation: reads go past the non-coherent caches
__device__ __forceinline__ u32 fast_ld(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// is the visit at sorted position j performed under the current guess?  (steps 0 .. reach, the last one being the
// operation that made the ray stop; a ray that never stops has reach == its whole list)
__device__ __forceinline__ bool fast_active(const FastVisits& V, const u32* reach, u32 j) {
  const u32 inf = V.sinfo[j];
  return (inf & kFastStepMask) <= fast_ld(&reach[inf >> kFastStepBits]);
}
// 1 + sorted position of the last performed visit of slot `key` before position i (0 = none this frame).  A slot's visits are
// contiguous and in (ray, step) order (the sort is stable), so this is a walk back through the run -- eight positions at a
// time: their keys and (ray, step) words are one batch of independent loads, the rays' reaches a second one, so a walk that
// ends within a window costs two dependent round trips however many of the eight it looks at (one position at a time it was
// three round trips per position, and a wave waits for its longest walk: 65 us per pass of the relaxation instead of 10).
__device__ __forceinline__ u32 fast_prev_performed(const FastVisits& V, const u32* reach, u32 i, u32 key) {
  constexpr u32 W = 8;
  for (u32 base = i; base > 0;) {
    u32 k[W], inf[W], rc[W];
#pragma unroll
    for (u32 w = 0; w < W; ++w) {
      const bool in = base > w;  // position base - 1 - w exists
      const u32 j = in ? base - 1u - w : 0u;
      k[w] = in ? V.skey[j] : ~key;
      inf[w] = V.sinfo[j];
    }
#pragma unroll
    for (u32 w = 0; w < W; ++w) rc[w] = (k[w] == key) ? fast_ld(&reach[inf[w] >> kFastStepBits]) : 0u;
#pragma unroll
    for (u32 w = 0; w < W; ++w) {
      if (k[w] != key) return 0u;  // the run starts here: no performed visit before i
      if ((inf[w] & kFastStepMask) <= rc[w]) return base - w;
    }
    base = base > W ? base - W : 0u;
  }
  return 0u;
}
// was the slot of the visit at sorted position i last written with the same hash?
__device__ __forceinline__ bool fast_collision(const FastVisits& V, const u32* reach, const u64* __restrict__ table_obs, u32 i) {
  const u32 key = V.skey[i];
  const u64 mine = V.shash[i];
  const u32 e = fast_prev_performed(V, reach, i, key);
  const u64 prev = e ? V.shash[e - 1] : table_obs[key];
  return prev == mine;
}

// The visits in slot-sorted order: where each visit went (pos_of), and per sorted position its (ray, step) and its hash (so the
// relaxation reads coalesced arrays instead of chasing visit -> ray -> offset).  Entries of unused list slots (key >= kFastSlots,
// sorted last) are skipped.
__global__ void __launch_bounds__(256) k_fast_inverse(const u32* __restrict__ skey_sorted, const u32* __restrict__ vval_sorted, const u32* __restrict__ vray,
                                                      const u64* __restrict__ vhash, const u32* __restrict__ voff, u32* __restrict__ pos_of, u32* __restrict__ sinfo,
                                                      u64* __restrict__ shash, const u32* __restrict__ d_n, u32 vcap) {
  const u32 n = min(*d_n, vcap);
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (skey_sorted[i] >= kFastSlots) continue;
    const u32 v = vval_sorted[i];
    const u32 r = vray[v];
    pos_of[v] = i;
    sinfo[i] = (r << kFastStepBits) | (v - voff[r]);
    shash[i] = vhash[v];
  }
}

// ---- the relaxation: ONE persistent launch per round ------------------------------------------------------------------------
// reach[r] <- where ray r stops given everybody's current reach, IN PLACE, pass after pass until a pass moves nothing.  The
// system is triangular (a ray only depends on lower-numbered rays and on itself), so any schedule of re-evaluations converges to
// the one fixed point, which is the sequential result; a pass that moved nothing has seen every ray at it.
// A pass is a few microseconds of dependent gathers over ~10^5 visits, and a frame needs 5 (14-18 while the sphere is in view):
// as launches with a host look at the "changed" flag in between (round 2) that was 0.11 ms per frame plus the round trips.  Here
// a small grid (kFastRelaxGroups workgroups, far fewer than the chip holds at once) stays resident and meets at a barrier
// after every pass.  The barrier cannot hang: a workgroup that waits longer than kFastBarrierPolls polls gives up, raises `abort`,
// everybody leaves, and the frame is redone by k_fast_sequential -- exact either way, and no host round trip.
//   short lists (<= 32 steps: every list of round 0, later the lists of the rays that did not get their whole walk): eight lanes
//     per ray, eight rays per wave, eight steps at a time; the collision flags of a ray are one ballot, its stop a few bit operations
//   long lists (the rays that got their whole walk): one workgroup per ray, 1024 steps tested at once
constexpr u32 kFastRelaxGroups = 128, kFastRelaxThreads = 1024;  // 2048 waves: one pass over 1.6 * 10^4 short lists without a second trip
constexpr u32 kFastBarrierPolls = 2000000;  // x ~0.5 us: about a second
constexpr u32 kFastMaxPasses = 4096;
typedef FastCtl::Bar FastBarrier;  // arrived: workgroups that have arrived, over all passes (never reset: pass p is complete at groups * (p + 1));
                                   // epoch: passes completed; moved / want [pass % 3]: flags of that pass; abort: a workgroup gave up waiting
// The barrier is the "one monotonic counter" form: every wave drains its stores, the workgroup meets, ONE lane releases
// (agent scope), adds to the counter and polls it with relaxed loads and s_sleep -- a word that is only ever written by
// atomic adds, which is what makes a relaxed poll see it across the XCDs' private L2s: polling a separate "epoch" word
// published by a store was seen 15 ms late, and an acquire in the polling loop invalidates caches on every poll (110 us per
// pass) -- then acquires ONCE, and the workgroup meets again.
__device__ __forceinline__ bool fast_grid_barrier(FastBarrier* b, u32 pass, u32* lds_ok, int fences) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave
  __syncthreads();
  if (threadIdx.x == 0) {
    const u32 target = gridDim.x * (pass + 1u);
    bool ok = true;
    if (fences) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    (void)atomicAdd(&b->arrived, 1u);
    u32 polls = 0;
    while (__hip_atomic_load(&b->arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++polls > kFastBarrierPolls || (polls % 64u == 0u && __hip_atomic_load(&b->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
        (void)atomicOr(&b->abort, 1u);
        ok = false;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    if (fences) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (__hip_atomic_load(&b->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) ok = false;
    *lds_ok = ok ? 1u : 0u;
  }
  __syncthreads();
  return *lds_ok != 0u;
}
// first step at which more than max_collisions collisions in a row have happened, from the collision flags of a list of ns <= 32 steps
__device__ __forceinline__ u32 fast_stop_from_bits(u32 m, u32 ns, int max_collisions) {
  u32 x = m;
  for (int k = 1; k <= max_collisions && k < 32 && x; ++k) x &= m >> k;  // bit s: collisions at s, s + 1, ..., s + max_collisions
  if (max_collisions >= 32) x = 0;                                       // (a list of 32 steps cannot hold that many in a row)
  return x ? static_cast<u32>(__ffs(static_cast<int>(x))) - 1u + static_cast<u32>(max_collisions) : ns;
}
constexpr u32 kFastShortMax = 32;  // lists up to this long: eight lanes per ray, eight steps at a time; longer ones: one workgroup per ray
__global__ void __launch_bounds__(kFastRelaxThreads) k_fast_relax(FastVisits V, int max_collisions, const u32* __restrict__ list_len, const u32* __restrict__ nfull,
                                                                  const u64* __restrict__ table_obs, u32* reach, FastCtl* ctl, int round,
                                                                  const u32* __restrict__ long_list, const u32* __restrict__ d_n_rays, int fences) {
  __shared__ u32 lds_ok;
  __shared__ u64 lds_bits[kFastRelaxThreads / 64];
  FastBarrier* bar = &ctl->bar[round];
  if (ctl->overflow || (round > 0 && ctl->grew[round] == 0u)) return;  // (uniform over the grid: nobody waits for anybody)
  const u32 n_rays = *d_n_rays;
  const u32 n_long = round > 0 ? ctl->n_long : 0u;
  const u32 lane = lane_id();
  const u32 n_waves = (gridDim.x * blockDim.x) >> 6;
  const u32 wave_in_wg = uniform_u32(threadIdx.x >> 6);
  const u32 wave0 = uniform_u32((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const u32 sub = lane >> 3, sl = lane & 7u;  // eight rays per wave, eight lanes per ray
  const bool wide_ok = max_collisions < 63;   // (the 512-step chunks of the long lists test runs with 64-bit shifts)
  u64 t_mark = wall_clock64(), t_work = 0, t_wait = 0;  // workgroup 0 keeps the time (100 MHz ticks): a pass's own work / its wait at the barrier
  for (u32 pass = 0; pass < kFastMaxPasses; ++pass) {
    const u32 f = pass % 3u;
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // the flags of the pass after this one: last read behind the barrier before the previous one
      (void)atomicAnd(&bar->moved[(pass + 1u) % 3u], 0u);  // (every access to these words is an atomic: they are seen across the XCDs' L2s)
      (void)atomicAnd(&bar->want[(pass + 1u) % 3u], 0u);
    }
    bool any = false, wants = false;
    // ---- short lists (every list of round 0; later the lists of the rays that did not get their whole walk) --------------------
    // Eight steps at a time, and of the first eight only the steps up to the ray's present stop: a stable ray (most of them, from
    // the second pass on) is confirmed by those alone, and they are the cheap ones -- near the surface every slot's run is full of
    // performed visits, while the voxels in front of it, steps 4 .. 7 of every ray around, have runs of nothing but unperformed
    // visits that a look-up has to walk through (a wave waits for its longest walk).  Only a ray that does NOT stop within what has
    // been looked at goes on to the next steps of its list.
    for (u32 g = wave0; g * 8u < n_rays; g += n_waves) {
      const u32 r = g * 8u + sub;
      u32 ns = 0, off = 0, old = 0;
      bool short_ray = false;
      if (r < n_rays) {
        ns = list_len[r];
        off = V.voff[r];
        old = fast_ld(&reach[r]);
        short_ray = ns <= kFastShortMax;
      }
      const u32 first = min(ns, old + 1u);
      u32 bits = 0, evaluated = 0, stop = ns;
      for (;;) {
        const bool active = short_ray && stop == ns && evaluated < ns;
        if (!__ballot(active)) break;
        const u32 bound = min(evaluated + 8u, evaluated < first ? first : ns);
        const u32 st = evaluated + sl;
        const bool coll = active && st < bound && fast_collision(V, reach, table_obs, V.pos_of[off + st]);
        const u32 m8 = static_cast<u32>((__ballot(coll) >> (sub * 8u)) & 0xFFull);
        if (active) {
          bits |= m8 << evaluated;
          evaluated = bound;
          const u32 found = fast_stop_from_bits(bits, ns, max_collisions);  // (steps not looked at yet are zeros: a run that completes is real)
          if (found != ns) stop = found;
        }
      }
      if (short_ray && sl == 0) {
        if (stop == ns && ns < nfull[r]) wants = true;
        if (stop != old) {
          __hip_atomic_store(&reach[r], stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          any = true;
        }
      }
    }
    // ---- long lists (the few rays -- a few dozen per frame at 5 cm -- that got their whole walk): ONE WORKGROUP per ray, 1024 steps
    // at a time, one step per lane.  (One wave per ray, four steps per lane one after the other, was the pass: 94 us instead of 36.)
    // Where the long lists are many (fine voxels: 10^3-10^4 of them) a workgroup each would take them one after the other: there
    // one WAVE takes a list, up to 256 steps at a time (four independent look-ups per lane in flight), in growing segments for
    // the lists that stopped early last time.
    const bool wave_per_list = n_long > 2u * gridDim.x;
    for (u32 q = wave0; wave_per_list && q < n_long; q += n_waves) {
      const u32 r = uniform_u32(long_list[q]);
      const u32 ns = uniform_u32(list_len[r]);
      if (ns <= kFastShortMax) continue;
      const u32 off = uniform_u32(V.voff[r]);
      const u32 old = fast_ld(&reach[r]);
      u32 carry = 0, stop = ns;
      u32 base = 0, seg = (old >= 48u) ? 256u : 16u;
      while (base < ns && stop == ns) {
        const u32 len = uniform_u32(min(seg, ns - base));
        bool coll[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const u32 t = 64u * j + lane;
          coll[j] = (t < len) && fast_collision(V, reach, table_obs, V.pos_of[off + base + t]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (stop != ns || 64u * j >= len) break;
          const u32 cnt_in = uniform_u32(min(64u, len - 64u * j));  // lanes of this group that hold a step
          const u64 m = __ballot(coll[j]);
          // length of the run of collisions that ends at this lane
          const u64 zeros_below = ~m & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
          const u32 run = zeros_below ? lane - (63u - static_cast<u32>(__clzll(static_cast<long long>(zeros_below)))) : lane + 1u + carry;
          const u64 hit = __ballot(coll[j] && run > static_cast<u32>(max_collisions));
          if (hit) stop = base + 64u * j + static_cast<u32>(__ffsll(static_cast<long long>(hit))) - 1u;
          carry = static_cast<u32>(__builtin_amdgcn_readlane(static_cast<int>(run), static_cast<int>(cnt_in - 1u)));
        }
        base += len;
        seg = (base < 64u) ? 64u - base : 256u;
      }
      if (lane == 0) {
        if (stop == ns && ns < nfull[r]) wants = true;
        if (stop != old) {
          __hip_atomic_store(&reach[r], stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          any = true;
        }
      }
    }
    for (u32 q = blockIdx.x; !wave_per_list && q < n_long; q += gridDim.x) {
      const u32 r = uniform_u32(long_list[q]);
      const u32 ns = uniform_u32(list_len[r]);
      if (ns <= kFastShortMax) continue;  // (uniform) handled with the short ones
      const u32 off = uniform_u32(V.voff[r]);
      u32 stop = ns;
      u64 prev = 0;  // the collision flags of the 64 steps before this wave's (runs continue across words)
      for (u32 base = 0; base < ns && stop == ns; base += kFastRelaxThreads) {
        const u32 st = base + threadIdx.x;
        const bool coll = st < ns && fast_collision(V, reach, table_obs, V.pos_of[off + st]);
        const u64 m = __ballot(coll);
        __syncthreads();  // (the words of the previous chunk have been read)
        if (lane == 0) lds_bits[wave_in_wg] = m;
        __syncthreads();
        if (wide_ok) {
          for (u32 w = 0; w < kFastRelaxThreads / 64 && stop == ns; ++w) {
            const u64 mw = lds_bits[w];
            u64 e = mw;  // bit p: collisions at p, p - 1, ..., p - max_collisions (reaching back into the previous word)
            for (int k = 1; k <= max_collisions && e; ++k) e &= (mw << k) | (prev >> (64 - k));
            if (e) stop = base + 64u * w + static_cast<u32>(__ffsll(static_cast<long long>(e))) - 1u;
            prev = mw;
          }
        } else {  // (max_consecutive_ray_collisions of 63 and more: step by step)
          u32 run = 0;
          for (u32 w = 0; w < kFastRelaxThreads / 64 && stop == ns; ++w) {
            const u64 mw = lds_bits[w];
            for (u32 bit = 0; bit < 64 && stop == ns; ++bit) {
              run = ((mw >> bit) & 1ull) ? run + 1u : 0u;
              if (run > static_cast<u32>(max_collisions)) stop = base + 64u * w + bit;
            }
          }
        }
        stop = min(stop, ns);  // (flags past the end of the list are zero, so this only guards the arithmetic)
      }
      if (threadIdx.x == 0) {
        if (stop == ns && ns < nfull[r]) wants = true;
        if (stop != fast_ld(&reach[r])) {
          __hip_atomic_store(&reach[r], stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          any = true;
        }
      }
    }
    // one word per flag for the whole grid: only the first few waves that have something to say write it
    if (__ballot(any) && lane == 0 && __hip_atomic_load(&bar->moved[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) (void)atomicOr(&bar->moved[f], 1u);
    if (__ballot(wants) && lane == 0 && __hip_atomic_load(&bar->want[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) (void)atomicOr(&bar->want[f], 1u);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      const u64 t = wall_clock64();
      t_work += t - t_mark;
      t_mark = t;
    }
    if (!fast_grid_barrier(bar, pass, &lds_ok, fences)) return;  // gave up: settled[round] stays 0
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      const u64 t = wall_clock64();
      t_wait += t - t_mark;
      t_mark = t;
    }
    if (__hip_atomic_load(&bar->moved[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {  // a pass that moved nothing: every ray is at the fixed point
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        ctl->want_more[round] = __hip_atomic_load(&bar->want[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ctl->passes[round] = pass + 1u;
        ctl->ticks_work[round] = static_cast<u32>(t_work);
        ctl->ticks_wait[round] = static_cast<u32>(t_wait);
        ctl->settled[round] = 1u;
      }
      return;
    }
  }
}

// Candidate visits are generated for the first cap0 steps of a ray only (most rays stop within a few steps, their walks are
// hundreds of voxels long).  A converged ray that reached the end of its list without stopping wants to go on: it gets its whole
// walk and round 1 relaxes again over the longer lists, from the fixed point of round 0.  The fixed point of a round in which
// nobody is at the end of a list shorter than its walk contains every performed visit, so it is the sequential result.
// The longer lists change what the rays behind them see: a ray that stopped at its sixth voxel in round 0 may stop at its third
// now, its later visits are no longer performed, and a ray that collided with those goes on instead -- past the end of its
// capped list, once in a few frames.  So every other ray's list grows as well in round 1, to cap1 steps (a few times the cap of
// round 0): room for such second-order moves.  A ray that still ends up at the end of a list shorter than its walk sends the
// frame to k_fast_sequential.
__global__ void __launch_bounds__(256) k_fast_grow(const u32* __restrict__ nfull, u32* __restrict__ cap, u32* reach, FastCtl* ctl, int round, u32 cap1,
                                                   u32* __restrict__ long_list, const u32* __restrict__ d_n_rays) {
  const u32 n_rays = *d_n_rays;
  const int prev = round - 1;
  // the round before did not run or did not settle (k_fast_sequential takes over) / nobody wants more
  if (ctl->overflow || (prev > 0 && ctl->grew[prev] == 0u) || ctl->settled[prev] == 0u || ctl->want_more[prev] == 0u) return;
  bool any = false;
  for (u32 r = blockIdx.x * blockDim.x + threadIdx.x; r < n_rays; r += gridDim.x * blockDim.x) {
    const u32 c = cap[r], nf = nfull[r];
    if (c < nf && reach[r] >= c) {
      cap[r] = nf;    // a ray that got through its first steps unstopped mostly goes all the way: take the whole walk
      reach[r] = nf;  // guess: it keeps going
      if (nf > cap1) long_list[atomicAdd(&ctl->n_long, 1u)] = r;  // (one wave each from now on; their order does not matter)
      any = true;
    } else if (c < nf) {
      cap[r] = min(nf, cap1);  // (its reach stays: the guess of the previous round's fixed point)
    }
  }
  if (__ballot(any) && lane_id() == 0) {
    if (__hip_atomic_load(&ctl->grew[round], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) atomicOr(&ctl->grew[round], 1u);
    ctl->scan_n[round] = n_rays;  // (every writer stores the same value)
  }
}
// round 1: offsets of the grown lists (exclusive scan of cap over the rays, one workgroup; nothing to do without growth)
__global__ void __launch_bounds__(1024) k_fast_scan_caps(const u32* __restrict__ cap, u32* __restrict__ voff, FastCtl* ctl, int round, u32 n_max, u32 vcap) {
  __shared__ u32 lds[16];
  const u32 n = min(ctl->scan_n[round], n_max);
  if (n == 0) return;
  u32 carry = 0;
  for (u32 base = 0; base < n; base += 1024 * 4) {
    const u32 i0 = base + threadIdx.x * 4;
    u32 v[4], sum = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      v[q] = (i0 + q < n) ? cap[i0 + q] : 0u;
      sum += v[q];
    }
    u32 total;
    u32 ex = carry + block_exclusive_scan<16>(sum, &total, lds);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (i0 + q < n) voff[i0 + q] = ex;
      ex += v[q];
    }
    carry += total;  // (sums beyond 2^32 cannot occur: the lists' total is bounded by the record capacity, < 2^31)
  }
  if (threadIdx.x == 0) {
    ctl->n_visits[round] = carry > vcap ? 0u : carry;
    if (carry > vcap) ctl->overflow = 1u;
  }
}
// the same bookkeeping behind the three-launch scan (many rays: fine voxels)
__global__ void k_fast_scan_caps_done(FastCtl* ctl, int round, u32 vcap) {
  if (ctl->scan_n[round] == 0u) ctl->n_visits[round] = 0u;
  if (ctl->n_visits[round] > vcap) {
    ctl->n_visits[round] = 0u;
    ctl->overflow = 1u;
  }
}

// did the relaxation produce the frame's result?  (round 0 settled; if it grew lists, round 1 settled too, its lists fit, and
// nobody is at the end of a list shorter than its walk)
// the last round that ran
__device__ __forceinline__ int fast_last_round(const FastCtl* ctl) {
  int last = 0;
  for (int r = 1; r < kFastMaxRounds; ++r)
    if (ctl->grew[r]) last = r;
  return last;
}
__device__ __forceinline__ bool fast_solved(const FastCtl* ctl) {
  if (ctl->overflow) return false;
  const int last = fast_last_round(ctl);
  return ctl->settled[last] != 0u && ctl->want_more[last] == 0u;
}
// The relaxation did not produce the result (a ray outgrew its round-1 list -- once in a few hundred frames of the benchmark
// stream with cap1 = cap0, never seen with cap1 = 4 cap0 --, lists that do not fit, a barrier that gave up): ONE lane runs the
// reference's loop as it is written -- rays in visiting order, replaceHash on the table itself.  Slow (tens of milliseconds),
// exact, and with no host round trip either.
__global__ void __launch_bounds__(64) k_fast_sequential(const FrameParams* __restrict__ Pp, FastFrame FF, RayArrays R, u64* __restrict__ table_obs, u32* reach,
                                                        FastCtl* ctl, int force, const u32* __restrict__ d_n_rays) {
  if (fast_solved(ctl) && !force) return;  // (force: COX_FAST_SEQUENTIAL=1, so that the tests can take this path on any frame)
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const FrameParams P = *Pp;
  const u32 n_rays = *d_n_rays;
  for (u32 r = 0; r < n_rays; ++r) {
    const u32 ns = R.nsteps[r];
    Dda d;
    dda_setup(d, P, F3{R.px[r], R.py[r], R.pz[r]}, (R.flags[r] & 2u) != 0);
    int collisions = 0;
    u32 s = 0;
    for (; s < ns; ++s) {
      const u64 h = long_index_hash(d.c[0], d.c[1], d.c[2]) + FF.off_obs;
      dda_step(d);
      const u32 slot = static_cast<u32>(h) & kFastSlotMask;
      if (table_obs[slot] == h) {
        ++collisions;
      } else {
        table_obs[slot] = h;
        collisions = 0;
      }
      if (collisions > FF.max_collisions) break;
    }
    reach[r] = s;
  }
  ctl->sequential = 1u;
}
// after the relaxation: the table keeps the hash of the last performed operation on each slot
__global__ void __launch_bounds__(256) k_fast_obs_commit(FastVisits V0, FastVisits V1, const u32* reach, u64* __restrict__ table_obs, const FastCtl* ctl,
                                                         u32 vcap0, u32 vcap1, u32* __restrict__ stats, RayArrays R, Counters* cnt) {
  {  // hand the rays over to the record pipeline: a ray emits its first reach[r] voxels (was a launch of its own behind this one)
    const u32 n_rays = cnt->n_rays;
    if (blockIdx.x == 0 && threadIdx.x == 0) cnt->n_ray_slots = n_rays;
    for (u32 r = blockIdx.x * blockDim.x + threadIdx.x; r < n_rays; r += gridDim.x * blockDim.x) R.nsteps[r] = reach[r];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // run totals (cox_integrator_fast_stats): how often the relaxation was not enough, and how much of it there was
    const int last = fast_last_round(ctl);
    bool aborted = false;
    u32 later_passes = 0;
    for (int r = 0; r < kFastMaxRounds; ++r) {
      aborted |= ctl->bar[r].abort != 0u;
      if (r > 0) later_passes += ctl->passes[r];
    }
    if (ctl->sequential) atomicAdd(&stats[0], 1u);
    if (ctl->sequential && !ctl->overflow && ctl->settled[last] && ctl->want_more[last]) atomicAdd(&stats[4], 1u);  // ... because a ray outgrew its list in the last round
    if (ctl->sequential && aborted) atomicAdd(&stats[5], 1u);                                                        // ... because a barrier gave up
    if (ctl->grew[1]) atomicAdd(&stats[1], 1u);
    if (last >= 2) atomicAdd(&stats[6], 1u);
    atomicAdd(&stats[2], ctl->passes[0]);
    atomicAdd(&stats[3], later_passes);
    u32 tw = 0, tb = 0;
    for (int r = 0; r < kFastMaxRounds; ++r) {
      tw += ctl->ticks_work[r];
      tb += ctl->ticks_wait[r];
    }
    atomicAdd(&stats[9], ctl->n_long);
    atomicAdd(&stats[10], ctl->n_visits[last] >> 4);  // (units of 16 visits)
    atomicAdd(&stats[11], ctl->n_visits[0] >> 4);
    atomicAdd(&stats[7], tw);   // 100 MHz ticks workgroup 0 spent in the passes' own work ...
    atomicAdd(&stats[8], tb);   // ... and waiting at the barriers (the slowest workgroup's work shows up here)
  }
  if (ctl->sequential) return;  // the sequential kernel wrote the table as it went
  const int last = fast_last_round(ctl);
  const FastVisits& V = last ? V1 : V0;
  const u32 n = last ? min(ctl->n_visits[last], vcap1) : min(ctl->n_visits[0], vcap0);
  // thread = performed visit: it is its slot's last one unless a later visit of the run is performed too (walking FORWARD from the
  // performed visits: the runs of nothing but unperformed visits -- the voxels in front of the surfaces -- cost nothing)
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const u32 key = V.skey[i];
    if (key >= kFastSlots || !fast_active(V, reach, i)) continue;
    bool last = true;
    constexpr u32 W = 8;
    for (u32 base = i + 1; last && base < n; base += W) {
      u32 k[W], inf[W], rc[W];
#pragma unroll
      for (u32 w = 0; w < W; ++w) {
        const bool in = base + w < n;
        const u32 j = in ? base + w : i;
        k[w] = in ? V.skey[j] : ~key;
        inf[w] = V.sinfo[j];
      }
#pragma unroll
      for (u32 w = 0; w < W; ++w) rc[w] = (k[w] == key) ? fast_ld(&reach[inf[w] >> kFastStepBits]) : 0u;
      bool end = false;
#pragma unroll
      for (u32 w = 0; w < W; ++w) {
        if (end || !last) continue;
        if (k[w] != key) end = true;                                 // the run ends here
        else if ((inf[w] & kFastStepMask) <= rc[w]) last = false;  // a later performed visit
      }
      if (end) break;
    }
    if (last) table_obs[key] = V.shash[i];
  }
}
}  // namespace cox
