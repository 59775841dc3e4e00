// The piece path of the merged integrator: wave walk with deferred block touch, ordinals by scan, piece keys, partition and expansion -- part of cox_integrator.hip (included there, in this order: the kernels use what is defined above them in that file).
#pragma once

// ---- piece path (merged, no anti-grazing): the walk is written once, as bytes, and partitioned by PIECES ---------------------
// Instead of one (voxel id, ray) record per step -- 8 B written, sorted twice and read again -- the walk leaves
//   one piece per (ray, tile) run (piece_bound), at slot piece_off[r] + k:  key = hash slot of the block << 4 | z & 15,
//                                                 start = the slot itself,  raylen = ray << 5 | steps (<= 31), and
//   lin8[32 * slot + j]    the voxel's (x & 15) | (y & 15) << 4 inside its tile for step j of the piece: one byte per step,
//                          a piece's bytes in one aligned 32-B segment (eight lanes read a piece with one dword load each).
// Only the pieces (a sixth to a tenth of the records) are sorted by tile; k_apply_pieces gathers each tile's steps through
// them.  Slots of a ray's bound that it does not use carry the invalid key.
struct PieceArrays {
  u32* key;     // [piece slots]
  u32* start;
  u32* raylen;
  u64* bkey;    // deferred touch only: the piece's block key (k_piece_touch turns it into the hash slot)
};
constexpr u32 kPieceLenBits = 5;

// kDefer: the walk touches no block itself (a hash probe and a stamp check are two dependent global round trips per round of
// 64 steps, and the wave waits for the slowest of its lanes: they were 2/3 of the walk's time at 1 cm); it leaves the block
// key with every piece (key word = z slab for now) and k_piece_touch does all the touching at once, one thread per piece.
template <u32 kAxisCap, bool kDefer>
__global__ void __launch_bounds__(256) k_touch_pieces(const FrameParams* __restrict__ Pp, RayArrays R, LayerView L, u32* __restrict__ touched_slots, uint8_t* __restrict__ lin8,
                                                      PieceArrays PA, u32 rec_cap, u32 piece_cap, Counters* cnt, u32* layer_err) {
  const FrameParams P = *Pp;
  __shared__ float lds_t[4][3 * kAxisCap];
  __shared__ u32 lds_path[4][3 * kAxisCap];
  const u32 n_slots = uniform_u32(cnt->n_ray_slots);
  const bool overflow = uniform_u32(cnt->n_records) > rec_cap || uniform_u32(cnt->n_piece_slots) > piece_cap;  // frame dropped (k_piece_keys reports it)
  const u32 lane = lane_id();
  const u32 wave = threadIdx.x >> 6;
  float* tl = lds_t[wave];
  u32* path = lds_path[wave];
  const u32 waves_total = (gridDim.x * blockDim.x) >> 6;
  for (u32 r = uniform_u32((blockIdx.x * blockDim.x + threadIdx.x) >> 6); r < n_slots; r += waves_total) {
    const u32 ns = uniform_u32(R.nsteps[r]);
    if (ns == 0) continue;
    const bool clearing = (uniform_u32(R.flags[r]) & 2u) != 0;
    const F3 pg{readlane_f32(R.px[r], 0), readlane_f32(R.py[r], 0), readlane_f32(R.pz[r], 0)};
    const u32 poff = uniform_u32(R.piece_off[r]), bound = uniform_u32(R.pbound[r]);
    Dda d;
    dda_setup(d, P, pg, clearing);
    u32 pk = 0;  // pieces of this ray so far
    if (wave_ray_path<kAxisCap>(d, ns, tl, path, lane)) {
      u64 carry_key = kEmptyKey;
      u32 carry_slot = kInvalid;
      for (u32 base = 0; base < ns; base += 64) {
        const u32 s = base + lane;
        const bool act = s < ns;
        u64 bkey = kEmptyKey;
        u32 zs = 0, lin = 0;
        if (act) {
          const u32 p = path[s];
          const int x = d.c[0] + static_cast<int>(p & 1023u) * d.sgn[0];
          const int y = d.c[1] + static_cast<int>((p >> 10) & 1023u) * d.sgn[1];
          const int z = d.c[2] + static_cast<int>(p >> 20) * d.sgn[2];
          lin = static_cast<u32>((x & 15) | ((y & 15) << 4));
          bkey = pack_key(x >> 4, y >> 4, z >> 4);
          zs = static_cast<u32>(z & 15);
        }
        u64 prev_key = __shfl_up(bkey, 1, 64);
        if (lane == 0) prev_key = carry_key;
        const bool bhead = act && bkey != prev_key;
        u32 slot = kInvalid;
        if constexpr (!kDefer)
          if (bhead) slot = touch_block(P, L, bkey, touched_slots, cnt, layer_err);
        // every lane takes the slot of the nearest block head at or below it, or the carry of the previous round
        const u64 bheads = __ballot(bhead);
        const u64 upto = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
        const u64 b_below = bheads & upto;
        const int src = b_below ? (63 - __clzll(static_cast<long long>(b_below))) : 0;
        const u32 head_slot = static_cast<u32>(__shfl(static_cast<int>(slot), src, 64));
        const u32 my_slot = b_below ? head_slot : carry_slot;
        const u32 prev_z = static_cast<u32>(__shfl_up(static_cast<int>(zs), 1, 64));
        const bool thead = act && (lane == 0 || bhead || zs != prev_z);
        const u64 theads = __ballot(thead);
        // the step's byte, at its offset inside its piece (lane 0 is a head: every lane has one at or below it)
        if constexpr (kAxisCap * 3 * sizeof(float) >= 64 * 32) {
          // staged in LDS (the crossing times are not needed any more) and written out as whole 32-B slots: the pieces of a
          // round are consecutive slots, so the wave streams np x 32 contiguous bytes instead of 64 scattered bytes
          uint8_t* stage = reinterpret_cast<uint8_t*>(tl);
          if (act) {
            const u64 t_below = theads & upto;
            const u32 kl = static_cast<u32>(__popcll(t_below)) - 1u;
            const u32 hl = 63u - static_cast<u32>(__clzll(static_cast<long long>(t_below)));
            stage[kl * 32u + (lane - hl)] = static_cast<uint8_t>(lin);
          }
          wave_lds_handover();
          const u32 ndw = static_cast<u32>(__popcll(theads)) * 8u;
          const u32* stage32 = reinterpret_cast<const u32*>(tl);
          u32* out32 = reinterpret_cast<u32*>(lin8);
          if (!overflow)
            for (u32 i = lane; i < ndw; i += 64)
              if (pk + (i >> 3) < bound) out32[static_cast<size_t>(poff + pk) * 8u + i] = stage32[i];
          wave_lds_handover();
        } else if (act && !overflow) {
          const u64 t_below = theads & upto;
          const u32 k = pk + static_cast<u32>(__popcll(t_below)) - 1u;
          const u32 hl = 63u - static_cast<u32>(__clzll(static_cast<long long>(t_below)));
          if (k < bound) lin8[static_cast<size_t>(poff + k) * 32u + (lane - hl)] = static_cast<uint8_t>(lin);
        }
        if (thead && !overflow) {
          const u32 k = pk + static_cast<u32>(__popcll(theads & ((1ull << lane) - 1ull)));
          const u64 above = (lane == 63) ? 0ull : (theads >> (lane + 1));
          const u32 round_end = min(64u, ns - base);
          const u32 nxt = above ? (lane + static_cast<u32>(__ffsll(static_cast<long long>(above)))) : round_end;
          if (k < bound) {
            if constexpr (kDefer) {
              PA.key[poff + k] = zs;
              PA.bkey[poff + k] = bkey;
            } else {
              PA.key[poff + k] = (my_slot == kInvalid) ? kInvalid : ((my_slot << 4) | zs);
            }
            PA.start[poff + k] = poff + k;
            PA.raylen[poff + k] = (r << kPieceLenBits) | (nxt - lane);
          } else {
            atomicOr(&cnt->err, kErrRecords);
          }
        }
        pk += static_cast<u32>(__popcll(theads));
        carry_key = __shfl(bkey, 63, 64);
        carry_slot = static_cast<u32>(__shfl(static_cast<int>(my_slot), 63, 64));
      }
      wave_lds_handover();  // the next ray of this wave reuses the LDS scratch
    } else {
      // sequential fallback (lane 0)
      if (lane == 0) {
        u64 last_bkey = kEmptyKey;
        u32 last_slot = kInvalid, last_z = 0, run = 0;
        for (u32 s = 0; s < ns; ++s) {
          const int x = d.c[0], y = d.c[1], z = d.c[2];
          dda_step(d);
          const u64 bkey = pack_key(x >> 4, y >> 4, z >> 4);
          const u32 zs = static_cast<u32>(z & 15);
          bool head = (s == 0) || zs != last_z || run == 31u;
          if (bkey != last_bkey) {
            last_bkey = bkey;
            if constexpr (!kDefer) last_slot = touch_block(P, L, bkey, touched_slots, cnt, layer_err);
            head = true;
          }
          last_z = zs;
          if (overflow) continue;
          if (head) {
            if (pk < bound) {
              if constexpr (kDefer) {
                PA.key[poff + pk] = zs;
                PA.bkey[poff + pk] = bkey;
              } else
                PA.key[poff + pk] = (last_slot == kInvalid) ? kInvalid : ((last_slot << 4) | zs);
              PA.start[poff + pk] = poff + pk;
              PA.raylen[poff + pk] = (r << kPieceLenBits) | 1u;
            } else {
              atomicOr(&cnt->err, kErrRecords);
            }
            pk += 1;
            run = 0;
          } else if (pk <= bound) {
            PA.raylen[poff + pk - 1] += 1u;
          }
          if (pk <= bound) lin8[static_cast<size_t>(poff + pk - 1) * 32u + run] = static_cast<uint8_t>((x & 15) | ((y & 15) << 4));
          run += 1;
        }
      }
      pk = static_cast<u32>(__shfl(static_cast<int>(pk), 0, 64));
    }
    // the slots of the bound this ray did not use
    if (!overflow)
      for (u32 i = pk + lane; i < bound; i += 64) PA.key[poff + i] = kInvalid;
  }
}

// deferred touch: one thread per piece slot.  Consecutive pieces of a ray mostly stay in one block (z slab changes), so only
// the first piece of a run of equal block keys touches the block; the others take its slot through the wave.
// touch_block without the ordinal: insert, give a fresh block its storage, stamp.  The stamp is a plain store (every
// writer of a frame stores the same value); ordinals come from a scan over the stamps afterwards (k_ord_flags / k_ord_scatter)
// instead of one atomicAdd per touched block on ONE word (10^4 blocks per frame at 1 cm = 114 us at 88 atomics / us).
__device__ __forceinline__ u32 touch_block_stamp(const FrameParams& P, const LayerView& L, u64 bkey, Counters* cnt, u32* layer_err) {
  bool fresh;
  const u32 slot = ht_insert(L.ht_keys, L.ht_mask, bkey, &fresh);
  if (slot == kInvalid) {
    atomicOr(layer_err, kErrTable);
    return kInvalid;
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
  if (L.ht_stamp[slot] != P.frame_id) L.ht_stamp[slot] = P.frame_id;
  return slot;
}
// ordinals of the blocks stamped this frame: flags over the hash slots -> exclusive scan (in ht_ord) -> dense slot list
__global__ void __launch_bounds__(256) k_ord_flags(const FrameParams* __restrict__ Pp, LayerView L) {
  const u32 frame = Pp->frame_id, n = L.ht_mask + 1u;
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) L.ht_ord[i] = (L.ht_stamp[i] == frame) ? 1u : 0u;
}
__global__ void __launch_bounds__(256) k_ord_scatter(const FrameParams* __restrict__ Pp, LayerView L, u32* __restrict__ touched_slots) {
  const u32 frame = Pp->frame_id, n = L.ht_mask + 1u;
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    if (L.ht_stamp[i] == frame) touched_slots[L.ht_ord[i]] = i;
}
constexpr u32 kPieceSeen = 512;
__global__ void __launch_bounds__(256) k_piece_touch(const FrameParams* __restrict__ Pp, LayerView L, PieceArrays PA, u32 rec_cap, u32 piece_cap, u32* __restrict__ touched_slots,
                                                     Counters* cnt, u32* layer_err) {
  const FrameParams P = *Pp;
  // The blocks around the sensor are crossed by every ray: tens of thousands of lanes probing the same few hash lines
  // serialise in L2 (230 us at 1 cm).  A per-wave table block key -> hash slot answers the repeats from LDS.  Two words per
  // entry, so an entry is (1) invalidated, (2) given its key, (3) given its slot by the lane whose key is found there on
  // reading back; lanes of one wave execute these steps in order, which is why the table is not shared between waves.
  __shared__ u64 ckey[4][kPieceSeen];
  __shared__ u32 cslot[4][kPieceSeen];
  for (u32 q = threadIdx.x; q < 4 * kPieceSeen; q += 256) {
    (&ckey[0][0])[q] = kEmptyKey;
    (&cslot[0][0])[q] = kInvalid;
  }
  __syncthreads();
  const u32 n = cnt->n_piece_slots;
  if (cnt->n_records > rec_cap || n > piece_cap) return;  // frame dropped (k_piece_keys reports it)
  const u32 lane = lane_id(), wv = threadIdx.x >> 6;
  const u32 n_round = (n + 63u) & ~63u;
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += gridDim.x * blockDim.x) {  // whole waves: shuffles below
    const u32 k = (i < n) ? PA.key[i] : kInvalid;
    const bool valid = k != kInvalid;
    const u64 bkey = valid ? PA.bkey[i] : kEmptyKey;
    const u64 prev = __shfl_up(bkey, 1, 64);
    const bool head = valid && (lane == 0 || prev != bkey);
    u32 slot = kInvalid;
    const u32 ci = static_cast<u32>((bkey * 0x9E3779B97F4A7C15ull) >> 40) & (kPieceSeen - 1u);
    bool miss = head;
    if (head && ckey[wv][ci] == bkey) {
      slot = cslot[wv][ci];
      miss = slot == kInvalid;
    }
    if (miss) slot = touch_block_stamp(P, L, bkey, cnt, layer_err);
    if (__ballot(miss)) {  // (wave-uniform) publish what was looked up
      const bool pub = miss && slot != kInvalid;
      if (pub) cslot[wv][ci] = kInvalid;
      wave_lds_handover();
      if (pub) ckey[wv][ci] = bkey;
      wave_lds_handover();
      if (pub && ckey[wv][ci] == bkey) cslot[wv][ci] = slot;
      wave_lds_handover();
    }
    const u64 heads = __ballot(head);
    const u64 below = heads & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
    const int src = below ? (63 - __clzll(static_cast<long long>(below))) : 0;
    const u32 my_slot = static_cast<u32>(__shfl(static_cast<int>(slot), src, 64));  // (a valid lane always has a head at or below it)
    if (valid) PA.key[i] = (my_slot == kInvalid) ? kInvalid : ((my_slot << 4) | k);
  }
}

// ---- piece partition: pieces sorted by tile, then EXPANDED into the (voxel id, ray) records k_apply_block reads -------------
// The record partition moves 16 B per record and pass; the pieces are 6-10 x fewer and 12 B each.  After the piece sort a
// scan of the piece lengths gives every piece its place in the record array, and the expansion writes the records of 256
// pieces at a time, lane = record (fully coalesced stores), finding each record's piece by binary search in LDS.
__global__ void __launch_bounds__(256) k_piece_lens(const u32* __restrict__ key0, const u32* __restrict__ key1, const u32* __restrict__ rl0, const u32* __restrict__ rl1,
                                                    const SortInfo* __restrict__ info, u32* __restrict__ len, const Counters* cnt) {
  const u32 par = info->parity & 1u;
  const u32* __restrict__ key = par ? key1 : key0;
  const u32* __restrict__ rl = par ? rl1 : rl0;
  const u32 n = (cnt->err & kErrRecords) ? 0u : cnt->n_piece_slots;
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) len[i] = (key[i] != kInvalid) ? (rl[i] & ((1u << kPieceLenBits) - 1u)) : 0u;
}
// record range of every tile from the sorted pieces and their scan (5-10 x fewer elements than the records k_block_starts reads)
__global__ void __launch_bounds__(256) k_piece_tile_ranges(const u32* __restrict__ key0, const u32* __restrict__ key1, const SortInfo* __restrict__ info,
                                                           const u32* __restrict__ len, const u32* __restrict__ dest, u32* __restrict__ tile_beg, u32* __restrict__ tile_end,
                                                           const Counters* cnt, u32 slab_shift) {
  const u32* __restrict__ key = (info->parity & 1u) ? key1 : key0;
  const u32 n = (cnt->err & kErrRecords) ? 0u : cnt->n_piece_slots;
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const u32 k = key[i];
    if (k == kInvalid) continue;  // (invalid keys sort last)
    const u32 t = k >> slab_shift;
    if (i == 0 || (key[i - 1] >> slab_shift) != t) tile_beg[t] = dest[i];
    if (i + 1 == n || (key[i + 1] >> slab_shift) != t) tile_end[t] = dest[i] + len[i];  // (an invalid neighbour shifts to another id)
  }
}
__global__ void __launch_bounds__(256) k_piece_expand(const u32* __restrict__ key0, const u32* __restrict__ key1, const u32* __restrict__ st0, const u32* __restrict__ st1,
                                                      const u32* __restrict__ rl0, const u32* __restrict__ rl1, const SortInfo* __restrict__ info,
                                                      const u32* __restrict__ dest, const uint8_t* __restrict__ lin8, u32* __restrict__ rec_key, u32* __restrict__ rec_ray,
                                                      const Counters* cnt) {
  __shared__ u32 s_off[257], s_key[256], s_ray[256], s_slot[256], lds[4];
  const u32 par = info->parity & 1u;
  const u32* __restrict__ key = par ? key1 : key0;
  const u32* __restrict__ start = par ? st1 : st0;
  const u32* __restrict__ rl = par ? rl1 : rl0;
  const u32 n = (cnt->err & kErrRecords) ? 0u : cnt->n_piece_slots;
  const u32 n_chunks = (n + 255u) / 256u;
  for (u32 c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const u32 i = c * 256u + threadIdx.x;
    u32 k = kInvalid, len = 0;
    if (i < n) {
      k = key[i];
      if (k != kInvalid) {
        const u32 v = rl[i];
        len = v & ((1u << kPieceLenBits) - 1u);
        s_ray[threadIdx.x] = v >> kPieceLenBits;
        s_slot[threadIdx.x] = start[i];
        s_key[threadIdx.x] = k << 8;  // tile id (ordinal << 4 | z slab) -> voxel id without its low byte
      }
    }
    u32 tot;
    const u32 off = block_exclusive_scan<4>(len, &tot, lds);
    s_off[threadIdx.x] = off;
    if (threadIdx.x == 0) s_off[256] = tot;
    __syncthreads();
    const u32 base = dest[c * 256u];  // records before this chunk (exclusive scan of the lengths)
    for (u32 t = threadIdx.x; t < tot; t += 256) {
      u32 lo = 0, hi = 256;  // the piece p with s_off[p] <= t < s_off[p + 1] (pieces of length 0 are never hit)
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const u32 mid = (lo + hi) >> 1;
        if (s_off[mid] <= t)
          lo = mid;
        else
          hi = mid;
      }
      // among pieces with the same offset (empty ones) lo is the last: the one that holds t
      const u32 j = t - s_off[lo];
      rec_key[base + t] = s_key[lo] | lin8[static_cast<size_t>(s_slot[lo]) * 32u + j];
      rec_ray[base + t] = s_ray[lo];
    }
    __syncthreads();
  }
}

// block ordinals of the frame -> piece keys = ordinal << 4 | z slab (the tile id); also what k_emit* publish for the record
// path: the ordinal table and the key width of the sort
__global__ void __launch_bounds__(256) k_piece_keys(LayerView L, u32* __restrict__ pkey, u32 rec_cap, u32 piece_cap, Counters* cnt, SortInfo* sort_info,
                                                    const u32* __restrict__ touched_slots, int4* __restrict__ ord_info, u32 slab_shift) {
  fill_ord_info(L, touched_slots, ord_info, cnt->n_touched);
  const u32 n = cnt->n_piece_slots;
  const bool overflow = cnt->n_records > rec_cap || n > piece_cap;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    u32 bits = 4;  // ordinals are < n_touched; the invalid key's bits (all ones) must sort after every valid tile id
    while ((1ull << (bits - 4)) < static_cast<u64>(cnt->n_touched) + 1ull) ++bits;
    sort_info->nbits = overflow ? 0u : bits - slab_shift;  // 0 bits: every sort pass exits at once
    sort_info->parity = 0;
    sort_info->base = slab_shift;  // tiles of two z slabs (slab_shift = 1): the lowest slab bit is not part of the tile id
    if (overflow) atomicOr(&cnt->err, kErrRecords);
  }
  if (overflow) return;
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const u32 k = pkey[i];
    if (k == kInvalid) continue;
    const u32 slot = k >> 4;
    if (L.ht_vals[slot] == kInvalid) {
      atomicOr(&cnt->err, kErrPool);  // block without storage: these updates are lost
      pkey[i] = kInvalid;
    } else {
      pkey[i] = (L.ht_ord[slot] << 4) | (k & 15u);
    }
  }
}
