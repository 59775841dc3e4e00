// The tile apply: k_block_starts, tile_flush, k_apply_block, and the piece apply k_apply_pieces (DESIGN.md sections 5, 5e) -- part of cox_integrator.hip (included there, in this order: the kernels use what is defined above them in that file).
#pragma once

// ---- block apply: the TSDF update with the voxel tiles staged in LDS -----------------------------------------------------
// The records arrive PARTITIONED by tile = (block ordinal, z-slab of the block: 16 x 16 x 1 voxels = 256 voxels, 3 KB of
// contiguous wire words) -- one or two stable radix passes on those bits; ray order inside a tile is preserved.  One
// workgroup owns one tile (a 16^3 block is sixteen of them; a frame at 5 cm has ~10^3 tiles with records, at 1 cm ~2 * 10^4,
// so the chip is full at every voxel size, and the near-camera blocks that every ray crosses are spread over 16 workgroups):
//   0. the tile's voxels are read into LDS with full-line loads;
//   1. every record of the tile is evaluated once (voxel centre, sdf, update weight) and classified: a SATURATING record
//      (saturating_update: provably leaves distance == truncation, adds an integer weight) only needs its weight summed --
//      per voxel, in LDS, with integer atomics (exact in any order); any other record marks its voxel "dirty";
//   2. per voxel: if no record was dirty and the voxel sits at +truncation with an integer weight (or is unobserved), its whole
//      run folds to  w <- min(max_weight, w + sum)  -- bit-identical to replaying it (exactness
//      arguments) -- which is most of a frame: the free space in front of the surfaces;
//   3. the records of the remaining ("hard") voxels -- the surface band -- are compacted in ray order, batches of 1024 are
//      sorted by voxel in LDS (stable counting sort: wave match-any ranks), and every hard voxel replays its records in
//      order with the reference's updateTsdfVoxel; the voxel state lives in LDS across batches;
//   4. the tile goes back to HBM with full-line stores.
// No float atomics; the result is the single-threaded reference order, bit for bit.  This replaces one or two of the global
// sort passes of the record pipeline and the 12-B scatter / gather of the per-record apply kernels.
constexpr u32 kSlabBits = 4;                               // z bits of the linear voxel index that belong to the tile id
constexpr u32 kTileShift = 12 - kSlabBits;                 // tile id = voxel id >> kTileShift = block ordinal << 4 | z
constexpr u32 kTileVox = 1u << kTileShift;                 // 256 voxels per tile
constexpr u32 kTilesPerBlock = 1u << kSlabBits;
constexpr u32 kHardBatch = 1024;
constexpr u32 kBT = 512, kBW = kBT / 64;  // two waves per SIMD: a tile is a chain of dependent global round trips

// Record range of every BUCKET = tile id & 4095 (tile id = key >> shift): the records are partitioned by the low 12 bits of
// the tile id in ONE stable pass whatever the number of touched blocks; up to 255 touched blocks a bucket is a tile, beyond
// that the tiles b, b + 4096, ... share bucket b and the apply takes them in turn.  (Invalid keys fall into bucket 4095 and
// are skipped there.)  mask == 0xFFFFFFFF: the key is the tile id itself (piece path), invalid keys have no range.
__global__ void __launch_bounds__(256) k_block_starts(RecordView V, u32* __restrict__ tile_beg, u32* __restrict__ tile_end, const Counters* cnt, u32 shift, u32 mask) {
  const u32 n = (cnt->err & kErrRecords) ? 0u : *V.d_n;
  const u32* __restrict__ key = V.key[V.info->parity & 1u];
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const u32 k = key[i];
    if (mask == 0xFFFFFFFFu && k == kInvalid) continue;
    const u32 t = (k >> shift) & mask;
    if (i == 0 || ((key[i - 1] >> shift) & mask) != t) tile_beg[t] = i;
    if (i + 1 == n || ((key[i + 1] >> shift) & mask) != t) tile_end[t] = i + 1;
  }
}

// Flush of a batch of `fill` hard records (block-wide; every thread calls it): sort the batch by voxel in LDS (stable
// counting sort: per-voxel counts, exclusive scan, wave match-any ranks, waves in turn), then every voxel replays its run
// in order with the reference's updateTsdfVoxel.  acc_cnt / acc_sum are scratch here (phases 1-2 are over).
template <u32 kT, u32 kTS>
__device__ __forceinline__ void tile_flush(const FrameParams& P, u32* blk, u32* acc_cnt, u32* acc_sum, const unsigned short* b_lin, const float* b_sdf,
                                           const float* b_uw, const u32* b_col, unsigned short* perm, u32* scan_lds, u32 fill, u32 tid, u32 lane, u32 wave) {
  constexpr u32 kTV = 1u << kTS;  // voxels per tile
  static_assert(kTV <= kT, "thread = voxel");
  if (tid < kTV) acc_cnt[tid] = 0;
  __syncthreads();
  for (u32 p = tid; p < fill; p += kT) atomicAdd(&acc_cnt[b_lin[p] & (kTV - 1u)], 1u);
  __syncthreads();
  {  // exclusive scan over the tile's voxels (thread = voxel)
    const u32 c = (tid < kTV) ? acc_cnt[tid] : 0u;
    u32 tot;
    const u32 ex = block_exclusive_scan<kT / 64>(c, &tot, scan_lds);
    if (tid < kTV) acc_sum[tid] = ex;
  }
  __syncthreads();
  const u32 chunk = ((fill + kT - 1u) / kT) * 64u;  // positions per wave, a multiple of 64
  for (u32 w = 0; w < kT / 64; ++w) {
    if (wave == w) {
      const u32 wbeg = min(fill, w * chunk), wend = min(fill, wbeg + chunk);
      for (u32 p0 = wbeg; p0 < wend; p0 += 64) {
        const u32 p = p0 + lane;
        const bool valid = p < wend;
        const u32 lin = valid ? (b_lin[p] & (kTV - 1u)) : 0u;
        u64 peers = __ballot(valid);
#pragma unroll
        for (u32 b = 0; b < kTS; ++b) {
          const bool bit = (lin >> b) & 1u;
          const u64 m = __ballot(bit);
          peers &= bit ? m : ~m;
        }
        const u64 lower = peers & ((1ull << lane) - 1ull);
        if (valid) perm[acc_sum[lin] + static_cast<u32>(__popcll(lower))] = static_cast<unsigned short>(p);
        wave_lds_handover();
        if (valid && lower == 0ull) acc_sum[lin] += static_cast<u32>(__popcll(peers));
        wave_lds_handover();
      }
    }
    __syncthreads();
  }
  {
    const u32 v = tid & (kTV - 1u);
    const u32 c = (tid < kTV) ? acc_cnt[v] : 0u;
    if (c) {
      const u32 e = acc_sum[v];
      Voxel vx{__uint_as_float(blk[3 * v]), __uint_as_float(blk[3 * v + 1]), blk[3 * v + 2]};
      for (u32 j = e - c; j < e; ++j) {
        const u32 idx = perm[j];
        const float uw = b_uw[idx];
        // a saturating record on a voxel that sits at +truncation only adds its weight: exactly what updateTsdfVoxel
        // computes there (distance provably stays == truncation -- saturating_update -- and the weight is the same
        // float addition), without its divisions: most records of a dirty voxel's run
        if ((b_lin[idx] & 0x8000u) && vx.d == P.trunc)
          vx.w = std_min(P.max_weight, vx.w + uw);
        else
          update_voxel(P, vx, b_sdf[idx], uw, b_col[idx]);
      }
      blk[3 * v] = __float_as_uint(vx.d);
      blk[3 * v + 1] = __float_as_uint(vx.w);
      blk[3 * v + 2] = vx.c;
    }
  }
  __syncthreads();
}

// compute_sdf with the ray's part (dv = point_G - origin, dist = |dv|) taken from RayArrays::q: the same operations on the same values
__device__ __forceinline__ float step_sdf(const FrameParams& P, F3 dv, float dist, int gx, int gy, int gz) {
  const F3 origin{P.tx, P.ty, P.tz};
  const F3 c{center_coord(gx, P.voxel_size), center_coord(gy, P.voxel_size), center_coord(gz, P.voxel_size)};
  const F3 v = c - origin;
  const float proj = dot3(v, dv) / dist;
  return dist - proj;
}

// what a record needs from its ray.  kQ (merged): one 32-B line written by the merge -- point_G - origin, its length, the
// weight -- instead of four 4-B gathers from four arrays; the ray gathers were a quarter of this kernel at 1 cm with one
// frame in flight and 40 % of it beside the other stages (ablation: DESIGN.md section 6)
struct RayOfRecord {
  F3 a;      // kQ: point_G - origin; else point_G
  float dist, w;
};
template <bool kQ>
__device__ __forceinline__ RayOfRecord ray_of_record(const RayArrays& R, u32 r) {
  if constexpr (kQ) {
    typedef float F4 __attribute__((ext_vector_type(4)));
    const F4* q = reinterpret_cast<const F4*>(R.q + static_cast<size_t>(r) * 8u);
    const F4 q0 = q[0];
    return RayOfRecord{F3{q0.x, q0.y, q0.z}, q0.w, q[1].x};
  } else {
    return RayOfRecord{F3{R.px[r], R.py[r], R.pz[r]}, 0.0f, R.w[r]};
  }
}
template <bool kQ>
__device__ __forceinline__ float sdf_of_record(const FrameParams& P, const RayOfRecord& y, int gx, int gy, int gz) {
  if constexpr (kQ)
    return step_sdf(P, y.a, y.dist, gx, gy, gz);
  else
    return compute_sdf(P, y.a, gx, gy, gz);
}

// ---- large tiles: phase 1 by the whole chip ------------------------------------------------------------------------------------
// The rays of a frame all leave from the sensor: the tiles around it are crossed by every ray (1 cm: 320 of a frame's 2 * 10^4 tiles
// have more than 8 192 records and hold up to 29 % of its 2.8 * 10^7 records), and k_apply_block gives a tile to one workgroup -- its
// duration was the largest tile's, 0.8 ms, whatever the others cost (which is why neither larger tiles nor a wave per small tile
// changed it).  Phase 1 is a sum: saturating weights and
// record counts per voxel, exact in any order.  So the tiles of more than two chunks (of kBigChunk records) are cut into chunks,
// every chunk is classified by a workgroup of its own into LDS and added to the tile's accumulators in global memory (integer
// atomics), and k_apply_block starts such a tile from those instead of running over its records.  Phases 2-4 are unchanged -- a voxel
// whose records did not all fold is replayed in order by the one workgroup, from the records, as before (the tiles around the sensor are
// free space: nothing to replay).
constexpr u32 kBigChunk = 4096, kBigChunkMin = 1024;  // (COX_BIG_CHUNK: the chunk, at least kBigChunkMin; tiles of more than two chunks are split)
constexpr u32 kBigCap = 8192;  // large tiles a frame may split; further ones are taken whole
struct BigTiles {
  u32* of_tile;   // [tiles] slot + 1 of a split tile, 0 otherwise; zero between frames
  u32* acc;       // [kBigCap][2][256] sum of saturating weights, count | dirty << 31; zero between frames
  uint2* chunks;  // (tile, first record)
  u32 chunk_cap;
  u32 chunk;      // records per chunk; tiles of more than 2 * chunk records are split
  u32* list;      // the tiles k_apply_block takes (more than wave_max records), in no particular order: cnt->n_block_tiles of them
  u32 wave_max;   // largest tile k_apply_wave takes
};
__global__ void __launch_bounds__(256) k_big_tiles(const u32* __restrict__ tile_beg, const u32* __restrict__ tile_end, const int4* __restrict__ ord_info, Counters* cnt,
                                                   BigTiles B) {
  const u32 n_tiles = ((cnt->err & kErrRecords) ? 0u : cnt->n_touched) * kTilesPerBlock;
  const u32 lane = threadIdx.x & 63u, stride = gridDim.x * blockDim.x;
  for (u32 t0 = blockIdx.x * blockDim.x + (threadIdx.x & ~63u); t0 < n_tiles; t0 += stride) {  // (wave-uniform bounds: the ballot below)
    const u32 tile = t0 + lane;
    u32 beg = 0, end = 0;
    if (tile < n_tiles) {
      beg = tile_beg[tile];
      end = tile_end[tile];
    }
    const u32 n = end > beg ? end - beg : 0u;
    // k_apply_block's list: the tiles too large for a wave, one atomic per wave that has any (one word takes ~88 atomics / us)
    const bool listed = n > B.wave_max;
    const u64 m = __ballot(listed);
    if (m) {
      u32 base = 0;
      if (lane == static_cast<u32>(__ffsll(static_cast<long long>(m))) - 1u) base = atomicAdd(&cnt->n_block_tiles, static_cast<u32>(__popcll(m)));
      base = static_cast<u32>(__shfl(static_cast<int>(base), __ffsll(static_cast<long long>(m)) - 1, 64));
      if (listed) B.list[base + static_cast<u32>(__popcll(m & ((1ull << lane) - 1ull)))] = tile;
    }
    if (n <= 2u * B.chunk) continue;
    if (static_cast<u32>(ord_info[tile >> kSlabBits].w) == kInvalid) continue;  // (k_apply_block skips such a tile: nobody would take the slot back)
    const u32 slot = atomicAdd(&cnt->n_big_tiles, 1u);
    if (slot >= kBigCap) continue;
    const u32 nch = (n + B.chunk - 1u) / B.chunk;
    const u32 base = atomicAdd(&cnt->n_big_chunks, nch);
    const bool fits = base + nch <= B.chunk_cap;  // (always: the list holds records / kBigChunkMin + kBigCap entries)
    if (fits) B.of_tile[tile] = slot + 1u;
    for (u32 c = 0; c < nch && base + c < B.chunk_cap; ++c) B.chunks[base + c] = make_uint2(fits ? tile : kInvalid, beg + c * B.chunk);
  }
}
template <bool kQ>
__global__ void __launch_bounds__(kBT) __attribute__((amdgpu_waves_per_eu(8, 8))) k_big_classify(const FrameParams* __restrict__ Pp, RayArrays R, const int4* __restrict__ ord_info, RecordView V,
                                                                                                   const u32* __restrict__ tile_end, const Counters* cnt, BigTiles B) {
  const FrameParams P = *Pp;
  __shared__ u32 acc_sum[kTileVox], acc_cnt[kTileVox];
  const u32 n_chunks = min(cnt->n_big_chunks, B.chunk_cap);
  const u32 par = V.info->parity & 1u;
  const u32* __restrict__ rec_key = V.key[par];
  const u32* __restrict__ rec_ray = V.ray[par];
  const u32 tid = threadIdx.x;
  for (u32 ch = blockIdx.x; ch < n_chunks; ch += gridDim.x) {
    const uint2 c = B.chunks[ch];
    const u32 tile = c.x;
    if (tile == kInvalid) continue;  // (uniform)
    const int4 info = ord_info[tile >> kSlabBits];
    const u32 beg = c.y, end = min(tile_end[tile], beg + B.chunk);
    const int gz = info.z + static_cast<int>(tile & (kTilesPerBlock - 1u));
    __syncthreads();  // the previous chunk's sums have been handed over
    if (tid < kTileVox) {
      acc_sum[tid] = 0;
      acc_cnt[tid] = 0;
    }
    __syncthreads();
    for (u32 i0 = beg + tid; i0 < end; i0 += 2 * kBT) {  // (k_apply_block's phase 1)
      const u32 i1 = i0 + kBT;
      const bool has1 = i1 < end;
      const u32 k0 = rec_key[i0], r0 = rec_ray[i0];
      const u32 k1 = has1 ? rec_key[i1] : 0u, r1 = has1 ? rec_ray[i1] : r0;
      const RayOfRecord y0 = ray_of_record<kQ>(R, r0), y1 = ray_of_record<kQ>(R, r1);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (u == 1 && !has1) break;
        const u32 lin = (u ? k1 : k0) & (kTileVox - 1u);
        const RayOfRecord& y = u ? y1 : y0;
        const float sdf = sdf_of_record<kQ>(P, y, info.x + static_cast<int>(lin & 15u), info.y + static_cast<int>(lin >> 4), gz);
        const float uw = update_weight(P, sdf, y.w);
        bool fold = foldable_update(P, sdf, uw);
        if (fold && atomicAdd(&acc_sum[lin], static_cast<u32>(uw)) >= (1u << 30)) fold = false;
        atomicAdd(&acc_cnt[lin], 1u);
        if (!fold) atomicOr(&acc_cnt[lin], 0x80000000u);
      }
    }
    __syncthreads();
    if (tid < kTileVox) {
      const u32 cv = acc_cnt[tid];
      if (cv) {
        u32* g = B.acc + static_cast<size_t>(B.of_tile[tile] - 1u) * (2u * kTileVox);
        const u32 sv = acc_sum[tid];
        const u32 old = sv ? atomicAdd(&g[tid], sv) : 0u;
        const bool over = old + sv < old || old + sv >= (1u << 30);  // the tile's sum must stay an exact u32 below 2^30: else the voxel is replayed
        atomicAdd(&g[kTileVox + tid], cv & 0x7FFFFFFFu);
        if ((cv >> 31) || over) atomicOr(&g[kTileVox + tid], 0x80000000u);
      }
    }
  }
}

template <bool kQ, u32 kTS, bool kBucket>
__global__ void __launch_bounds__(kBT) __attribute__((amdgpu_waves_per_eu(8, 8))) k_apply_block(const FrameParams* __restrict__ Pp, RayArrays R, LayerView L, const int4* __restrict__ ord_info, RecordView V,
                                                     u32* __restrict__ tile_beg, u32* __restrict__ tile_end, Counters* cnt, u32* layer_err, u32* __restrict__ h_nblocks,
                                                     u32 min_records, BigTiles big) {
  // big.of_tile != nullptr (tiles, not buckets): a tile with a slot there was classified by k_big_classify; phase 1 is its accumulators.
  // min_records > 0 (tiles, not buckets): only the tiles with at least that many records; the others are left, ranges and all, to
  // k_apply_wave behind this kernel.
  // kTS = log2(voxels per tile): 8 = one z slab of the block (16 tiles per block), 9 = two (fine voxels: half as many tiles,
  // each a chain of dependent round trips, and thread = voxel uses all 512 threads)
  constexpr u32 kTV = 1u << kTS, kTPB = 4096u >> kTS, kSlabs = kTV / 256u;
  const FrameParams P = *Pp;
  __shared__ u32 blk[kTV * kWordsPerVoxel];
  __shared__ u32 acc_sum[kTV];  // phases 1-2: sum of the saturating weights of a voxel; phase 3: end of the voxel's run in the sorted batch
  __shared__ u32 acc_cnt[kTV];  // phases 1-2: records of the voxel | dirty << 31; phase 3: records of the voxel in the batch
  __shared__ u32 hardbits[kTV / 32];
  __shared__ float b_sdf[kHardBatch], b_uw[kHardBatch];
  __shared__ u32 b_col[kHardBatch];
  __shared__ unsigned short b_lin[kHardBatch], perm[kHardBatch];
  __shared__ u32 wsum[kBW], scan_lds[kBW], any_hard_s, any_rec_s;
  // last kernel of the frame: make this frame's error bits sticky until the host next looks, and leave the layer's block
  // count where the host can read it without a sync (pinned word; it decides when to grow the pool)
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (cnt->err) atomicOr(layer_err, cnt->err);
    *h_nblocks = min(*L.d_nblocks, L.capacity);
  }
  const u32 n_tiles = ((cnt->err & kErrRecords) ? 0u : cnt->n_touched) * kTPB;
  const u32 par = V.info->parity & 1u;
  const u32* __restrict__ rec_key = V.key[par];
  const u32* __restrict__ rec_ray = V.ray[par];
  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const bool exact_cap = P.max_weight <= 16711680.0f;  // 2^24 - 2^16: w + u never leaves the exact integers before the cap applies
  u32 my_updates = 0, my_voxels = 0, my_maxrun = 0;
  // kBucket (records partitioned in one pass by tile id & 4095): a unit is a bucket; the tiles unit, unit + 4096, ... share its
  // record range and are taken in turn, each looking at its own records only.  Otherwise a unit is a tile.
  const bool listed = !kBucket && big.list != nullptr;
  const u32 n_units = kBucket ? min(n_tiles, 4096u) : listed ? min(cnt->n_block_tiles, n_tiles) : n_tiles;
  for (u32 ui = blockIdx.x; ui < n_units; ui += gridDim.x) {
    const u32 unit = listed ? big.list[ui] : ui;
    const u32 beg = tile_beg[unit], end = tile_end[unit];
    if (end <= beg) continue;  // (uniform) no record touches this slab of the block
    if (!kBucket && end - beg < min_records) continue;  // (uniform) a small tile: k_apply_wave's
   for (u32 tile = unit; tile < n_tiles; tile += (kBucket ? 4096u : n_tiles)) {
    __syncthreads();           // everybody has read its range and is done with the previous tile's LDS
    if (tid == 0) {
      tile_beg[unit] = 0;  // leave the tables empty for the next frame
      tile_end[unit] = 0;
      any_hard_s = 0;
      any_rec_s = 0;
    }
    const int4 info = ord_info[tile / kTPB];
    const u32 pool = static_cast<u32>(info.w);
    if (pool == kInvalid) continue;  // (uniform; such a block's records carry invalid keys anyway)
    const int gz0 = info.z + static_cast<int>((tile & (kTPB - 1u)) * kSlabs);  // first z slab of the tile
    u32* gblk = L.voxels + (static_cast<size_t>(pool) * kVoxelsPerBlock + (tile & (kTPB - 1u)) * kTV) * kWordsPerVoxel;
    for (u32 i = tid; i < kTV * kWordsPerVoxel; i += kBT) blk[i] = gblk[i];
    const u32 split = (!kBucket && kTS == kTileShift && big.of_tile) ? big.of_tile[tile] : 0u;  // (uniform)
    if (split) {
      u32* g = big.acc + static_cast<size_t>(split - 1u) * (2u * kTV);
      for (u32 v = tid; v < kTV; v += kBT) {
        acc_sum[v] = g[v];
        acc_cnt[v] = g[kTV + v];
        g[v] = 0;  // leave the accumulators empty for the next frame
        g[kTV + v] = 0;
      }
    } else {
      for (u32 v = tid; v < kTV; v += kBT) {
        acc_sum[v] = 0;
        acc_cnt[v] = 0;
      }
    }
    __syncthreads();
    if (split && tid == 0) big.of_tile[tile] = 0;  // (behind the barrier: every thread has read the slot) ... and the slot table
    // ---- 1. classify --------------------------------------------------------------------------------------------------
    for (u32 i0 = beg + tid; i0 < (split ? beg : end); i0 += 2 * kBT) {  // two records per thread in flight: the gathers are latency-bound
      const u32 i1 = i0 + kBT;
      const bool has1 = i1 < end;
      const u32 k0 = rec_key[i0], r0 = rec_ray[i0];
      const u32 k1 = has1 ? rec_key[i1] : 0u, r1 = has1 ? rec_ray[i1] : r0;
      const RayOfRecord y0 = ray_of_record<kQ>(R, r0), y1 = ray_of_record<kQ>(R, r1);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (u == 1 && !has1) break;
        if (kBucket && ((u ? k1 : k0) >> kTS) != tile) continue;  // another tile of the bucket (or an invalid key)
        const u32 lin = (u ? k1 : k0) & (kTV - 1u);
        const RayOfRecord& y = u ? y1 : y0;
        const float sdf = sdf_of_record<kQ>(P, y, info.x + static_cast<int>(lin & 15u), info.y + static_cast<int>((lin >> 4) & 15u), gz0 + static_cast<int>(lin >> 8));
        const float uw = update_weight(P, sdf, y.w);
        bool fold = foldable_update(P, sdf, uw);
        if (fold && atomicAdd(&acc_sum[lin], static_cast<u32>(uw)) >= (1u << 30)) fold = false;  // the sum must stay an exact u32
        atomicAdd(&acc_cnt[lin], 1u);
        if (!fold) atomicOr(&acc_cnt[lin], 0x80000000u);
      }
    }
    __syncthreads();
    // ---- 2. fold what folds (thread = voxel) ----------------------------------------------------------------------------
    {
      const u32 v = tid & (kTV - 1u);
      const u32 c = (tid < kTV) ? acc_cnt[v] : 0u, count = c & 0x7FFFFFFFu;
      bool hard = false;
      if (count) {
        my_updates += count;
        my_voxels += 1;
        my_maxrun = max(my_maxrun, count);
        const float d = __uint_as_float(blk[3 * v]), w = __uint_as_float(blk[3 * v + 1]);
        const u32 sum = acc_sum[v];
        hard = true;
        if (!(c >> 31) && sum < (1u << 30) && (w == 0.0f || d == P.trunc)) {
          if (w >= P.max_weight) {
            hard = false;  // min(max_weight, w + u) == max_weight for every u > 0: nothing changes
          } else if (w == truncf(w) && w >= 0.0f) {
            const unsigned long long total = static_cast<unsigned long long>(w) + sum;
            if (total < 16777216ull || exact_cap) {
              const float ft = static_cast<float>(total < 16777216ull ? static_cast<u32>(total) : 16777216u);
              blk[3 * v] = __float_as_uint(P.trunc);  // an unobserved voxel's first saturating update sets the distance to +truncation
              blk[3 * v + 1] = __float_as_uint(ft >= P.max_weight ? P.max_weight : ft);
              hard = false;
            }
          }
        }
      }
      const u64 hm = __ballot(hard);
      if (lane == 0 && tid < kTV) {
        hardbits[wave * 2] = static_cast<u32>(hm);
        hardbits[wave * 2 + 1] = static_cast<u32>(hm >> 32);
        if (hm) any_hard_s = 1;
      }
      if (kBucket && count) any_rec_s = 1;  // (same value from every writer)
    }
    __syncthreads();
    if (kBucket && !any_rec_s) continue;  // (uniform) a tile of the bucket without records of its own in this slab: nothing to write
    // ---- 3. the hard voxels: ordered replay ---------------------------------------------------------------------------
    if (any_hard_s) {
      u32 fill = 0;
      for (u32 base = beg;; base += kBT) {
        const bool done = base >= end;
        if (done || fill + kBT > kHardBatch) {
          // -- flush: sort the batch by voxel (stable), replay every voxel's run in order
          if (fill) {
            tile_flush<kBT, kTS>(P, blk, acc_cnt, acc_sum, b_lin, b_sdf, b_uw, b_col, perm, scan_lds, fill, tid, lane, wave);
            fill = 0;
          }
          if (done) break;
        }
        // -- one round: kBT consecutive records, the hard ones appended to the batch in order
        const u32 i = base + tid;
        bool keep = false;
        u32 lin = 0;
        if (i < end) {
          const u32 k = rec_key[i];
          lin = k & (kTV - 1u);
          keep = (!kBucket || (k >> kTS) == tile) && ((hardbits[lin >> 5] >> (lin & 31u)) & 1u);
        }
        const u64 m = __ballot(keep);
        if (lane == 0) wsum[wave] = static_cast<u32>(__popcll(m));
        __syncthreads();
        u32 pos = fill + static_cast<u32>(__popcll(m & ((1ull << lane) - 1ull)));
        for (u32 w = 0; w < wave; ++w) pos += wsum[w];
        if (keep) {
          const u32 r = rec_ray[i];
          const RayOfRecord y = ray_of_record<kQ>(R, r);
          const float sdf = sdf_of_record<kQ>(P, y, info.x + static_cast<int>(lin & 15u), info.y + static_cast<int>((lin >> 4) & 15u), gz0 + static_cast<int>(lin >> 8));
          const float uw = update_weight(P, sdf, y.w);
          b_lin[pos] = static_cast<unsigned short>(lin | (foldable_update(P, sdf, uw) ? 0x8000u : 0u));
          b_sdf[pos] = sdf;
          b_uw[pos] = uw;
          b_col[pos] = R.color[r];
        }
#pragma unroll
        for (u32 w = 0; w < kBW; ++w) fill += wsum[w];
        __syncthreads();
      }
    }
    // ---- 4. the tile goes back ----------------------------------------------------------------------------------------
    __syncthreads();
    for (u32 i = tid; i < kTV * kWordsPerVoxel; i += kBT) gblk[i] = blk[i];
   }
  }
  // statistics
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    my_updates += __shfl_xor(my_updates, off, 64);
    my_voxels += __shfl_xor(my_voxels, off, 64);
    my_maxrun = max(my_maxrun, static_cast<u32>(__shfl_xor(static_cast<int>(my_maxrun), off, 64)));
  }
  if (lane == 0 && my_voxels) {
    u32* sh = cnt->shard[(blockIdx.x * kBW + wave) & 63u];
    atomicAdd(&sh[kShUpdates], my_updates);
    atomicAdd(&sh[kShVoxels], my_voxels);
    atomicMax(&sh[kShMaxRun], my_maxrun);
  }
}

// ---- wave apply: k_apply_block's four phases, ONE WAVE per small tile (fine voxels) ---------------------------------------------
// At fine voxels a frame has ~2 * 10^4 tiles, most of them of a few hundred records (1 cm: 2.8 * 10^7 records in all), and a tile is a
// chain of dependent global round trips -- range -> block info -> voxels, records -> ray lines -> (hard voxels) records again -> ray
// lines, colours -- with a workgroup barrier between the phases.  With a 512-thread workgroup per tile a CU has FOUR tiles in flight and
// half the lanes of phase 1 have no record; the kernel is bound by that chain (0.8 ms at 1 cm for ~0.1 ms worth of instructions and
// 0.04 ms worth of bytes).  A first wave-per-tile kernel that kept k_apply_block's loops (64 records per round, phase 3 re-reading keys,
// rays, lines and colours round by round) was no faster: sixteen tiles in flight per CU, but three times the round trips per tile.
// This one is built around the round trips instead.  A wave owns a tile of at most kN records (larger tiles go to k_apply_block, which
// runs first: the tiles next to the camera hold 10^4 records and more):
//   T1  range and block info of the NEXT tile, fetched while this one is worked on;
//   T2  the tile's voxels (four per lane, in registers) and ALL its records (kN / 64 per lane), issued together;
//   T3  the rays' 32-B lines (point_G - origin, length, weight, colour: written by the merge), all of them together;
//       then every record is evaluated once and KEPT in LDS (voxel, sdf, update weight, colour) -- phases 2 and 3 never go back to
//       global memory: fold (thread = voxel), stable counting sort of the hard voxels' records by voxel, ordered replay;
//   T4  the tile goes back (nothing waits for it).
// No workgroup barrier anywhere (LDS traffic of one wave is in order).  Same operations on the same values in the same order per voxel
// as k_apply_block: bit-identical.  The blocks of the frame are dealt to the XCDs by ordinal (block ordinal mod 8 = XCD of the
// workgroup): the sixteen tiles of a block, which share their rays' lines, stay in one L2.
constexpr u32 kWaveTileWaves = 4;   // waves (= tiles in flight) per workgroup
template <u32 kN>
struct alignas(16) WaveTileLds {
  u32 acc_sum[kTileVox];
  u32 acc_cnt[kTileVox];
  u32 hardbits[kTileVox / 32];
  float r_sdf[kN], r_uw[kN];
  u32 r_col[kN];
  unsigned short r_lin[kN], perm[kN];
};

template <u32 kN>
__global__ void __launch_bounds__(kWaveTileWaves * 64) k_apply_wave(const FrameParams* __restrict__ Pp, RayArrays R, LayerView L, const int4* __restrict__ ord_info, RecordView V,
                                                                   u32* __restrict__ tile_beg, u32* __restrict__ tile_end, Counters* cnt, u32* layer_err,
                                                                   u32* __restrict__ h_nblocks) {
  typedef u32 U4 __attribute__((ext_vector_type(4)));
  typedef float F4 __attribute__((ext_vector_type(4)));
  constexpr u32 kPer = kN / 64;  // records per lane
  static_assert(kTileVox == 256 && kWordsPerVoxel == 3 && kN % 64 == 0, "four voxels per lane; whole rounds of records");
  const FrameParams P = *Pp;
  __shared__ WaveTileLds<kN> lds[kWaveTileWaves];
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // last kernel of the frame (see k_apply_block)
    if (cnt->err) atomicOr(layer_err, cnt->err);
    *h_nblocks = min(*L.d_nblocks, L.capacity);
  }
  const u32 n_blocks = (cnt->err & kErrRecords) ? 0u : cnt->n_touched;
  const u32 par = V.info->parity & 1u;
  const u32* __restrict__ rec_key = V.key[par];
  const u32* __restrict__ rec_ray = V.ray[par];
  const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  WaveTileLds<kN>& W = lds[wave];
  const bool exact_cap = P.max_weight <= 16711680.0f;
  u32 my_updates = 0, my_voxels = 0, my_maxrun = 0;
  // XCD x (= workgroup index mod 8) takes the blocks with ordinal mod 8 == x; its waves take those blocks' tiles in turn
  const u32 xcd = blockIdx.x & 7u, n_xcd_waves = (gridDim.x >> 3) * kWaveTileWaves;  // (the grid is a multiple of 8 workgroups)
  const u32 my_tiles = ((n_blocks + 7u - xcd) >> 3) * kTilesPerBlock;                 // tiles of this XCD's blocks
  auto tile_of = [&](u32 k) { return (((k >> kSlabBits) << 3) + xcd) * kTilesPerBlock + (k & (kTilesPerBlock - 1u)); };
  u32 k = (blockIdx.x >> 3) * kWaveTileWaves + wave;
  u32 nbeg = 0, nend = 0;
  int4 ninfo = make_int4(0, 0, 0, 0);
  if (k < my_tiles) {
    const u32 t = tile_of(k);
    nbeg = tile_beg[t];
    nend = tile_end[t];
    ninfo = ord_info[t >> kSlabBits];
  }
  for (; k < my_tiles; k += n_xcd_waves) {
    const u32 tile = tile_of(k);
    const u32 beg = nbeg, end = nend;
    const int4 info = ninfo;
    if (k + n_xcd_waves < my_tiles) {  // T1 of the next tile
      const u32 t = tile_of(k + n_xcd_waves);
      nbeg = tile_beg[t];
      nend = tile_end[t];
      ninfo = ord_info[t >> kSlabBits];
    }
    const u32 n = end - beg;
    if (end <= beg || n > kN) continue;  // (wave-uniform) no record touches this slab of the block, or k_apply_block has taken the tile
    if (lane == 0) {
      tile_beg[tile] = 0;  // leave the tables empty for the next frame
      tile_end[tile] = 0;
    }
    const u32 pool = static_cast<u32>(info.w);
    if (pool == kInvalid) continue;  // (uniform; such a block's records carry invalid keys anyway)
    const int gz = info.z + static_cast<int>(tile & (kTilesPerBlock - 1u));
    u32* gblk = L.voxels + (static_cast<size_t>(pool) * kVoxelsPerBlock + (tile & (kTilesPerBlock - 1u)) * kTileVox) * kWordsPerVoxel;
    // ---- T2: voxels (voxel lane + 64 q in registers) and records ----------------------------------------------------------
    u32 vd[4], vw[4], vc[4];
#pragma unroll
    for (u32 q = 0; q < 4; ++q) {
      const u32* g = gblk + 3u * (lane + 64u * q);
      vd[q] = g[0];
      vw[q] = g[1];
      vc[q] = g[2];
    }
    u32 rk[kPer], rr[kPer];
#pragma unroll
    for (u32 t = 0; t < kPer; ++t) {
      const u32 j = lane + 64u * t;
      const bool on = j < n;
      rk[t] = on ? rec_key[beg + j] : 0u;
      rr[t] = on ? rec_ray[beg + j] : 0u;
    }
    reinterpret_cast<U4*>(W.acc_sum)[lane] = U4{0u, 0u, 0u, 0u};
    reinterpret_cast<U4*>(W.acc_cnt)[lane] = U4{0u, 0u, 0u, 0u};
    wave_lds_handover();
    // ---- T3: the rays' lines; 1. classify, every record kept in LDS ---------------------------------------------------------
    F4 qa[kPer];
    float qw[kPer];
    u32 qc[kPer];
#pragma unroll
    for (u32 t = 0; t < kPer; ++t) {
      const F4* q = reinterpret_cast<const F4*>(R.q + static_cast<size_t>(rr[t]) * 8u);  // (ray 0 for the lanes beyond n: any valid line)
      qa[t] = q[0];
      const F4 q1 = q[1];
      qw[t] = q1.x;
      qc[t] = __float_as_uint(q1.y);
    }
#pragma unroll
    for (u32 t = 0; t < kPer; ++t) {
      const u32 j = lane + 64u * t;
      if (j < n) {
        const u32 lin = rk[t] & (kTileVox - 1u);
        const float sdf = step_sdf(P, F3{qa[t].x, qa[t].y, qa[t].z}, qa[t].w, info.x + static_cast<int>(lin & 15u), info.y + static_cast<int>(lin >> 4), gz);
        const float uw = update_weight(P, sdf, qw[t]);
        bool fold = foldable_update(P, sdf, uw);
        W.r_lin[j] = static_cast<unsigned short>(lin | (fold ? 0x8000u : 0u));
        W.r_sdf[j] = sdf;
        W.r_uw[j] = uw;
        W.r_col[j] = qc[t];
        if (fold && atomicAdd(&W.acc_sum[lin], static_cast<u32>(uw)) >= (1u << 30)) fold = false;  // the sum must stay an exact u32
        atomicAdd(&W.acc_cnt[lin], 1u);
        if (!fold) atomicOr(&W.acc_cnt[lin], 0x80000000u);
      }
    }
    wave_lds_handover();
    // ---- 2. fold what folds (four voxels per lane) ------------------------------------------------------------------------
    bool any_hard = false;
#pragma unroll
    for (u32 q = 0; q < 4; ++q) {
      const u32 v = lane + 64u * q;
      const u32 c = W.acc_cnt[v], count = c & 0x7FFFFFFFu;
      bool hard = false;
      if (count) {
        my_updates += count;
        my_voxels += 1;
        my_maxrun = max(my_maxrun, count);
        const float d = __uint_as_float(vd[q]), w = __uint_as_float(vw[q]);
        const u32 sum = W.acc_sum[v];
        hard = true;
        if (!(c >> 31) && sum < (1u << 30) && (w == 0.0f || d == P.trunc)) {
          if (w >= P.max_weight) {
            hard = false;  // min(max_weight, w + u) == max_weight for every u > 0: nothing changes
          } else if (w == truncf(w) && w >= 0.0f) {
            const unsigned long long total = static_cast<unsigned long long>(w) + sum;
            if (total < 16777216ull || exact_cap) {
              const float ft = static_cast<float>(total < 16777216ull ? static_cast<u32>(total) : 16777216u);
              vd[q] = __float_as_uint(P.trunc);  // an unobserved voxel's first saturating update sets the distance to +truncation
              vw[q] = __float_as_uint(ft >= P.max_weight ? P.max_weight : ft);
              hard = false;
            }
          }
        }
      }
      const u64 hm = __ballot(hard);
      if (lane == 0) {
        W.hardbits[2 * q] = static_cast<u32>(hm);
        W.hardbits[2 * q + 1] = static_cast<u32>(hm >> 32);
      }
      any_hard |= hm != 0ull;
    }
    // ---- 3. the hard voxels: stable counting sort of their records by voxel, ordered replay --------------------------------
    if (any_hard) {
      reinterpret_cast<U4*>(W.acc_cnt)[lane] = U4{0u, 0u, 0u, 0u};
      wave_lds_handover();
      const u32 rounds = (n + 63u) >> 6;
      for (u32 t = 0; t < rounds; ++t) {
        const u32 j = lane + 64u * t;
        if (j < n) {
          const u32 lin = W.r_lin[j] & (kTileVox - 1u);
          if ((W.hardbits[lin >> 5] >> (lin & 31u)) & 1u) atomicAdd(&W.acc_cnt[lin], 1u);
        }
      }
      wave_lds_handover();
      {  // exclusive scan over the 256 voxels: four consecutive voxels per lane
        const U4 c = reinterpret_cast<const U4*>(W.acc_cnt)[lane];
        const u32 sum = c.x + c.y + c.z + c.w;
        const u32 ex = wave_inclusive_scan(sum) - sum;
        reinterpret_cast<U4*>(W.acc_sum)[lane] = U4{ex, ex + c.x, ex + c.x + c.y, ex + c.x + c.y + c.z};
      }
      wave_lds_handover();
      for (u32 t = 0; t < rounds; ++t) {  // ranks, 64 records at a time in record (= ray) order
        const u32 j = lane + 64u * t;
        u32 lin = 0;
        bool valid = false;
        if (j < n) {
          lin = W.r_lin[j] & (kTileVox - 1u);
          valid = (W.hardbits[lin >> 5] >> (lin & 31u)) & 1u;
        }
        u64 peers = __ballot(valid);
        if (peers == 0ull) continue;  // (uniform)
#pragma unroll
        for (u32 b = 0; b < kTileShift; ++b) {
          const bool bit = (lin >> b) & 1u;
          const u64 m = __ballot(bit);
          peers &= bit ? m : ~m;
        }
        const u64 lower = peers & ((1ull << lane) - 1ull);
        if (valid) W.perm[W.acc_sum[lin] + static_cast<u32>(__popcll(lower))] = static_cast<unsigned short>(j);
        wave_lds_handover();
        if (valid && lower == 0ull) W.acc_sum[lin] += static_cast<u32>(__popcll(peers));
        wave_lds_handover();
      }
#pragma unroll
      for (u32 q = 0; q < 4; ++q) {
        const u32 v = lane + 64u * q;
        const u32 c = W.acc_cnt[v];
        if (c) {
          const u32 e = W.acc_sum[v];
          Voxel vx{__uint_as_float(vd[q]), __uint_as_float(vw[q]), vc[q]};
          for (u32 i = e - c; i < e; ++i) {
            const u32 idx = W.perm[i];
            const float uw = W.r_uw[idx];
            if ((W.r_lin[idx] & 0x8000u) && vx.d == P.trunc)  // (see tile_flush)
              vx.w = std_min(P.max_weight, vx.w + uw);
            else
              update_voxel(P, vx, W.r_sdf[idx], uw, W.r_col[idx]);
          }
          vd[q] = __float_as_uint(vx.d);
          vw[q] = __float_as_uint(vx.w);
          vc[q] = vx.c;
        }
      }
    }
    // ---- T4: the tile goes back --------------------------------------------------------------------------------------------
#pragma unroll
    for (u32 q = 0; q < 4; ++q) {
      u32* g = gblk + 3u * (lane + 64u * q);
      g[0] = vd[q];
      g[1] = vw[q];
      g[2] = vc[q];
    }
    wave_lds_handover();  // (the next tile reuses the wave's LDS)
  }
  // statistics
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    my_updates += __shfl_xor(my_updates, off, 64);
    my_voxels += __shfl_xor(my_voxels, off, 64);
    my_maxrun = max(my_maxrun, static_cast<u32>(__shfl_xor(static_cast<int>(my_maxrun), off, 64)));
  }
  if (lane == 0 && my_voxels) {
    u32* sh = cnt->shard[(blockIdx.x * kWaveTileWaves + wave) & 63u];
    atomicAdd(&sh[kShUpdates], my_updates);
    atomicAdd(&sh[kShVoxels], my_voxels);
    atomicMax(&sh[kShMaxRun], my_maxrun);
  }
}

// ---- piece apply: k_apply_block's four phases, the tile's steps gathered through its pieces ------------------------------------
// The pieces arrive sorted by tile, ray order inside a tile preserved (stable sort of slots that were laid out in ray order).
// A round takes up to 256 pieces, thread = piece: header (slot, ray, length), the piece's bytes with one or two 16-B loads,
// the ray once per piece -- a step costs one byte of HBM traffic instead of an 8-B record.  The round is then EXPANDED in LDS
// into its steps (exclusive scan of the lengths; step -> voxel byte, piece), in (piece, step) = ray order, and phases 1 and
// 3 run over the steps with one lane per step exactly like k_apply_block runs over records.  A tile of at most 256 pieces
// (nearly all of them at fine voxels) is expanded once; the next tile's range is fetched while this one is worked on.
constexpr u32 kPT = 256, kPW = kPT / 64;
constexpr u32 kPieceBatch = 768;
constexpr u32 kStepCap = kPT * 31;  // steps of one round
struct PieceView {
  const u32* key[2];
  const u32* start[2];
  const u32* raylen[2];
  const SortInfo* info;
};
typedef u32 U32x4 __attribute__((ext_vector_type(4)));

// f(j, lin) for every step j < len of this lane's piece, lanes in lockstep (j is wave-uniform; every lane of the wave must call)
template <typename F>
__device__ __forceinline__ void piece_for_each(u32 len, const u64 (&H)[4], F&& f) {
#pragma unroll
  for (u32 h = 0; h < 4; ++h) {
    if (__ballot(8u * h < len) == 0ull) break;
    u64 cur = H[h];
#pragma unroll 1
    for (u32 b = 0; b < 8; ++b) {
      const u32 j = 8u * h + b;
      const bool on = j < len;
      if (__ballot(on) == 0ull) break;
      const u32 lin = static_cast<u32>(cur) & 255u;
      cur >>= 8;
      if (on) f(j, lin);
    }
  }
}

__global__ void __launch_bounds__(kPT) k_apply_pieces(const FrameParams* __restrict__ Pp, RayArrays R, LayerView L, const int4* __restrict__ ord_info, PieceView V,
                                                      const uint8_t* __restrict__ lin8, u32* __restrict__ tile_beg, u32* __restrict__ tile_end, Counters* cnt,
                                                      u32* layer_err, u32* __restrict__ h_nblocks) {
  const FrameParams P = *Pp;
  __shared__ u32 blk[kTileVox * kWordsPerVoxel];
  __shared__ u32 acc_sum[kTileVox];
  __shared__ u32 acc_cnt[kTileVox];
  __shared__ u32 hardbits[kTileVox / 32];
  __shared__ uint8_t s_lin[kStepCap], s_pc[kStepCap];             // the round's steps: voxel byte, piece (= thread that loaded it)
  __shared__ float pr_x[kPT], pr_y[kPT], pr_z[kPT], pr_d[kPT], pr_w[kPT];  // the round's pieces: point_G - origin, its length, the ray's weight
  __shared__ u32 pr_ray[kPT];
  __shared__ float b_sdf[kPieceBatch], b_uw[kPieceBatch];
  __shared__ u32 b_col[kPieceBatch];
  __shared__ unsigned short b_lin[kPieceBatch], perm[kPieceBatch];
  __shared__ u32 wsum[kPW], scan_lds[kPW], any_hard_s;
  static_assert(kPT == kTileVox, "phase 2 and the flush use thread = voxel");
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // last kernel of the frame (see k_apply_block)
    if (cnt->err) atomicOr(layer_err, cnt->err);
    *h_nblocks = min(*L.d_nblocks, L.capacity);
  }
  const u32 n_tiles = ((cnt->err & kErrRecords) ? 0u : cnt->n_touched) * kTilesPerBlock;
  const u32 par = V.info->parity & 1u;
  const u32* __restrict__ p_start = V.start[par];
  const u32* __restrict__ p_raylen = V.raylen[par];
  const U32x4* __restrict__ lin128 = reinterpret_cast<const U32x4*>(lin8);
  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const bool exact_cap = P.max_weight <= 16711680.0f;
  u32 my_updates = 0, my_voxels = 0, my_maxrun = 0;
  u32 tile = blockIdx.x;
  u32 nbeg = 0, nend = 0;
  int4 ninfo = make_int4(0, 0, 0, 0);
  if (tile < n_tiles) {
    nbeg = tile_beg[tile];
    nend = tile_end[tile];
    ninfo = ord_info[tile >> kSlabBits];
  }
  for (; tile < n_tiles; tile += gridDim.x) {
    const u32 beg = nbeg, end = nend;
    const int4 info = ninfo;
    if (tile + gridDim.x < n_tiles) {  // the next tile's range: in flight while this tile is worked on
      nbeg = tile_beg[tile + gridDim.x];
      nend = tile_end[tile + gridDim.x];
      ninfo = ord_info[(tile + gridDim.x) >> kSlabBits];
    }
    if (end <= beg) continue;  // (uniform) no piece in this slab of the block
    __syncthreads();           // everybody has read the range and is done with the previous tile's LDS
    if (tid == 0) {
      tile_beg[tile] = 0;  // leave the tables empty for the next frame
      tile_end[tile] = 0;
      any_hard_s = 0;
    }
    const u32 pool = static_cast<u32>(info.w);
    if (pool == kInvalid) continue;  // (uniform; such a block's pieces carry invalid keys anyway)
    const int gz = info.z + static_cast<int>(tile & (kTilesPerBlock - 1u));
    u32* gblk = L.voxels + (static_cast<size_t>(pool) * kVoxelsPerBlock + (tile & (kTilesPerBlock - 1u)) * kTileVox) * kWordsPerVoxel;
    // expand the round of pieces [pb, pb + kPT) into s_lin / s_pc / pr_*; returns its number of steps (block-wide)
    auto expand_round = [&](u32 pb) -> u32 {
      const u32 pi = pb + tid;
      u32 len = 0;
      u64 H[4] = {0, 0, 0, 0};
      if (pi < end) {
        const u32 st = p_start[pi], rl = p_raylen[pi];
        const u32 r = rl >> kPieceLenBits;
        len = rl & ((1u << kPieceLenBits) - 1u);
        const U32x4 a = lin128[static_cast<size_t>(st) * 2u];
        U32x4 b = {0, 0, 0, 0};
        if (len > 16u) b = lin128[static_cast<size_t>(st) * 2u + 1u];
        typedef float F4 __attribute__((ext_vector_type(4)));
        const F4* q = reinterpret_cast<const F4*>(R.q + static_cast<size_t>(r) * 8u);
        const F4 q0 = q[0];
        pr_w[tid] = q[1].x;
        pr_x[tid] = q0.x;
        pr_y[tid] = q0.y;
        pr_z[tid] = q0.z;
        pr_d[tid] = q0.w;
        pr_ray[tid] = r;
        H[0] = a.x | (static_cast<u64>(a.y) << 32);
        H[1] = a.z | (static_cast<u64>(a.w) << 32);
        H[2] = b.x | (static_cast<u64>(b.y) << 32);
        H[3] = b.z | (static_cast<u64>(b.w) << 32);
      }
      u32 tot;
      const u32 off = block_exclusive_scan<kPW>(len, &tot, scan_lds);
      piece_for_each(len, H, [&](u32 j, u32 lin) {
        s_lin[off + j] = static_cast<uint8_t>(lin);
        s_pc[off + j] = static_cast<uint8_t>(tid);
      });
      __syncthreads();
      return tot;
    };
    for (u32 i = tid; i < kTileVox * kWordsPerVoxel; i += kPT) blk[i] = gblk[i];
    acc_sum[tid] = 0;
    acc_cnt[tid] = 0;
    const bool keeps = end - beg <= kPT;  // one round: its expansion serves phase 3 too
    u32 T = 0;
    // ---- 1. classify (lane = step) ------------------------------------------------------------------------------------
    for (u32 pb = beg; pb < end; pb += kPT) {
      if (pb != beg) __syncthreads();  // the previous round's steps have been consumed
      T = expand_round(pb);            // (its barrier also covers blk / acc_*)
      for (u32 s0 = tid; s0 < T; s0 += kPT) {
        const u32 lin = s_lin[s0], pc = s_pc[s0];
        const float sdf = step_sdf(P, F3{pr_x[pc], pr_y[pc], pr_z[pc]}, pr_d[pc], info.x + static_cast<int>(lin & 15u), info.y + static_cast<int>(lin >> 4), gz);
        const float uw = update_weight(P, sdf, pr_w[pc]);
        bool fold = foldable_update(P, sdf, uw);
        if (fold && atomicAdd(&acc_sum[lin], static_cast<u32>(uw)) >= (1u << 30)) fold = false;  // the sum must stay an exact u32
        atomicAdd(&acc_cnt[lin], 1u);
        if (!fold) atomicOr(&acc_cnt[lin], 0x80000000u);
      }
    }
    __syncthreads();
    // ---- 2. fold what folds (thread = voxel) ----------------------------------------------------------------------------
    {
      const u32 v = tid;
      const u32 c = acc_cnt[v], count = c & 0x7FFFFFFFu;
      bool hard = false;
      if (count) {
        my_updates += count;
        my_voxels += 1;
        my_maxrun = max(my_maxrun, count);
        const float d = __uint_as_float(blk[3 * v]), w = __uint_as_float(blk[3 * v + 1]);
        const u32 sum = acc_sum[v];
        hard = true;
        if (!(c >> 31) && sum < (1u << 30) && (w == 0.0f || d == P.trunc)) {
          if (w >= P.max_weight) {
            hard = false;  // min(max_weight, w + u) == max_weight for every u > 0: nothing changes
          } else if (w == truncf(w) && w >= 0.0f) {
            const unsigned long long total = static_cast<unsigned long long>(w) + sum;
            if (total < 16777216ull || exact_cap) {
              const float ft = static_cast<float>(total < 16777216ull ? static_cast<u32>(total) : 16777216u);
              blk[3 * v] = __float_as_uint(P.trunc);
              blk[3 * v + 1] = __float_as_uint(ft >= P.max_weight ? P.max_weight : ft);
              hard = false;
            }
          }
        }
      }
      const u64 hm = __ballot(hard);
      if (lane == 0) {
        hardbits[wave * 2] = static_cast<u32>(hm);
        hardbits[wave * 2 + 1] = static_cast<u32>(hm >> 32);
        if (hm) any_hard_s = 1;
      }
    }
    __syncthreads();
    // ---- 3. the hard voxels: ordered replay (lane = step, chunks of kPT steps in ray order) ---------------------------
    if (any_hard_s) {
      u32 fill = 0;
      for (u32 pb = beg; pb < end; pb += kPT) {
        if (!keeps) {
          __syncthreads();
          T = expand_round(pb);
        }
        for (u32 base = 0; base < T; base += kPT) {
          if (fill + kPT > kPieceBatch) {
            tile_flush<kPT, kTileShift>(P, blk, acc_cnt, acc_sum, b_lin, b_sdf, b_uw, b_col, perm, scan_lds, fill, tid, lane, wave);
            fill = 0;
          }
          const u32 s0 = base + tid;
          bool keep = false;
          u32 lin = 0;
          if (s0 < T) {
            lin = s_lin[s0];
            keep = (hardbits[lin >> 5] >> (lin & 31u)) & 1u;
          }
          const u64 m = __ballot(keep);
          if (lane == 0) wsum[wave] = static_cast<u32>(__popcll(m));
          __syncthreads();
          u32 pos = fill + static_cast<u32>(__popcll(m & ((1ull << lane) - 1ull)));
          for (u32 w = 0; w < wave; ++w) pos += wsum[w];
          if (keep) {
            const u32 pc = s_pc[s0];
            const float sdf = step_sdf(P, F3{pr_x[pc], pr_y[pc], pr_z[pc]}, pr_d[pc], info.x + static_cast<int>(lin & 15u), info.y + static_cast<int>(lin >> 4), gz);
            const float uw = update_weight(P, sdf, pr_w[pc]);
            b_lin[pos] = static_cast<unsigned short>(lin | (foldable_update(P, sdf, uw) ? 0x8000u : 0u));
            b_sdf[pos] = sdf;
            b_uw[pos] = uw;
            b_col[pos] = R.color[pr_ray[pc]];
          }
#pragma unroll
          for (u32 w = 0; w < kPW; ++w) fill += wsum[w];
          __syncthreads();
        }
      }
      if (fill) tile_flush<kPT, kTileShift>(P, blk, acc_cnt, acc_sum, b_lin, b_sdf, b_uw, b_col, perm, scan_lds, fill, tid, lane, wave);
    }
    // ---- 4. the tile goes back ----------------------------------------------------------------------------------------
    __syncthreads();
    for (u32 i = tid; i < kTileVox * kWordsPerVoxel; i += kPT) gblk[i] = blk[i];
  }
  // statistics
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    my_updates += __shfl_xor(my_updates, off, 64);
    my_voxels += __shfl_xor(my_voxels, off, 64);
    my_maxrun = max(my_maxrun, static_cast<u32>(__shfl_xor(static_cast<int>(my_maxrun), off, 64)));
  }
  if (lane == 0 && my_voxels) {
    u32* sh = cnt->shard[(blockIdx.x * kPW + wave) & 63u];
    atomicAdd(&sh[kShUpdates], my_updates);
    atomicAdd(&sh[kShVoxels], my_voxels);
    atomicMax(&sh[kShMaxRun], my_maxrun);
  }
}
