// Round 1's per-record apply kernels (COX_APPLY=records): k_apply_eval, k_apply_long -- part of cox_integrator.hip (included there, in this order: the kernels use what is defined above them in that file).
#pragma once

// ---- apply: per voxel, the running weighted-mean / clamp update in canonical ray order -----------------
// After the stable sort the records of one voxel are contiguous ("segment") and in ray order.
// k_apply_eval, one wave per 64 consecutive records:
//   1. every lane evaluates its own record (voxel centre, sdf, update weight, colour) -- the costly,
//      order-independent part -- fully in parallel;
//   2. the head lane of every segment that ends inside the wave replays its records in order out of the
//      neighbours' registers (__shfl), at most 63 dependent steps;
//   3. a segment that crosses a wave boundary ("long": the few near-camera voxels that every ray crosses)
//      is only summarised: per wave, for the piece at its front (continuing from the previous wave) and
//      the piece at its back (starting here): record count, whether every record is a provable no-op on a
//      voxel sitting at +trunc with an integer weight (saturating_update), and the weight sum.
// k_apply_long, one wave per long segment: walks the piece summaries (64 pieces = 4096 records per load),
//   folds runs of no-op pieces exactly and replays only the rest record by record.
// No float atomics anywhere; the result is bit-reproducible and equals the sequential reference order.
struct VoxelRef {
  u32* ptr;  // 3 words
  int gx, gy, gz;
  bool ok;
};
// one 16-B gather per record instead of three dependent ones (ordinal -> hash slot -> block key / pool index): the emit
// kernel leaves (16 * block index, pool index) of every block touched this frame in ord_info
__device__ __forceinline__ VoxelRef locate_voxel(const LayerView& L, const int4* __restrict__ ord_info, u32 vid) {
  VoxelRef v;
  const u32 ord = vid >> 12, lin = vid & 4095u;
  const int4 b = ord_info[ord];
  const u32 pool = static_cast<u32>(b.w);
  v.gx = b.x + static_cast<int>(lin & 15u);
  v.gy = b.y + static_cast<int>((lin >> 4) & 15u);
  v.gz = b.z + static_cast<int>(lin >> 8);
  v.ok = pool != kInvalid;
  v.ptr = L.voxels + (static_cast<size_t>(pool) * kVoxelsPerBlock + lin) * kWordsPerVoxel;
  return v;
}
// summary word of a piece: bits 0..6 record count (0..64), bit 31 "foldable"
constexpr u32 kPieceFoldable = 0x80000000u;
// a record can be folded when it provably keeps distance == trunc and adds an integer weight
// saturating_update's margin shrinks with the weight, so the threshold of weight 1 (P.sat1, rounded up by the host) covers every
// integer weight: a float compare per record instead of a double division.  The sliver trunc*(1 + margin(uw)) <= sdf < sat1 is
// merely replayed instead of folded -- any subset of the saturating records may be folded, the result is the same.
__device__ __forceinline__ bool foldable_update(const FrameParams& P, float sdf, float uw) {
  return uw >= 1.0f && uw == truncf(uw) && uw < 65536.0f && sdf >= P.sat1;
}

// fold a run of foldable pieces with total integer weight wsum; false when the voxel state does not allow it
__device__ __forceinline__ bool fold_pieces(const FrameParams& P, Voxel& v, u32 wsum) {
  if (v.d != P.trunc) return false;
  if (v.w >= P.max_weight) return true;  // min(max_weight, w + u) == max_weight for every u > 0
  if (v.w != truncf(v.w) || v.w + static_cast<float>(wsum) >= 16777216.0f) return false;
  v.w = std_min(P.max_weight, v.w + static_cast<float>(wsum));  // integer partial sums are exact
  return true;
}

__global__ void __launch_bounds__(256) k_apply_eval(const FrameParams* __restrict__ Pp, RayArrays R, LayerView L, const int4* __restrict__ touched_slots, RecordView V,
                                                    u32* __restrict__ piece_front, u32* __restrict__ piece_back, u32* __restrict__ piece_wsum,
                                                    Counters* cnt) {
  const FrameParams P = *Pp;
  __shared__ u32 blk_updates, blk_voxels, blk_long, blk_maxrun;
  if (threadIdx.x == 0) {
    blk_updates = 0;
    blk_voxels = 0;
    blk_long = 0;
    blk_maxrun = 0;
  }
  __syncthreads();
  const u32 n = uniform_u32((cnt->err & kErrRecords) ? 0u : *V.d_n);
  const u32 par = uniform_u32(V.info->parity & 1u);
  const u32* __restrict__ rec_key = V.key[par];
  const u32* __restrict__ rec_ray = V.ray[par];
  const u32 lane = lane_id();
  const u32 n_waves = (n + 63) >> 6;
  const u32 waves_total = (gridDim.x * blockDim.x) >> 6;
  u32 my_updates = 0, my_voxels = 0, my_long = 0, my_maxrun = 0;
  for (u32 wv = uniform_u32((blockIdx.x * blockDim.x + threadIdx.x) >> 6); wv < n_waves; wv += waves_total) {
    const u32 wave_base = wv << 6;
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
    bool fold = false;
    if (valid) {
      vr = locate_voxel(L, touched_slots, key);
      const u32 r = rec_ray[i];
      const F3 pg{R.px[r], R.py[r], R.pz[r]};
      sdf = compute_sdf(P, pg, vr.gx, vr.gy, vr.gz);
      uw = update_weight(P, sdf, R.w[r]);
      color = R.color[r];
      fold = foldable_update(P, sdf, uw);
    }
    // ---- segment geometry -----------------------------------------------------------------------------
    const u64 bmask = __ballot(boundary);
    const u64 later = (lane == 63) ? 0ull : (bmask >> (lane + 1));
    u32 len = 0;
    bool runs_on = false;  // the segment of this head continues in the next wave
    if (head) {
      if (later) {
        len = static_cast<u32>(__ffsll(static_cast<long long>(later)));
      } else {
        const u32 nxt = wave_base + 64;
        if (nxt >= n || rec_key[nxt] != key)
          len = min(nxt, n) - i;
        else
          runs_on = true;
      }
    }
    // A segment that merely straddles the wave boundary (ends inside the next wave) is finished here: the few
    // records of its tail are evaluated by this wave too.  Only segments that cover the whole next wave are "long".
    const u64 omask = __ballot(runs_on);
    bool strad = false;
    float sdf2 = 0.0f, uw2 = 0.0f;
    u32 color2 = 0, ll = 0;
    if (omask) {
      ll = static_cast<u32>(__ffsll(static_cast<long long>(omask))) - 1u;
      const u32 lkey = static_cast<u32>(__builtin_amdgcn_readlane(key, ll));
      const u32 j = wave_base + 64 + lane;
      const bool in2 = (j < n) && (rec_key[j] == lkey);
      const u32 e2 = static_cast<u32>(__popcll(__ballot(in2)));  // the tail is contiguous: lanes [0, e2)
      if (e2 < 64) {
        strad = true;
        const int gx = __builtin_amdgcn_readlane(vr.gx, ll), gy = __builtin_amdgcn_readlane(vr.gy, ll), gz = __builtin_amdgcn_readlane(vr.gz, ll);
        if (in2) {
          const u32 r = rec_ray[j];
          const F3 pg{R.px[r], R.py[r], R.pz[r]};
          sdf2 = compute_sdf(P, pg, gx, gy, gz);
          uw2 = update_weight(P, sdf2, R.w[r]);
          color2 = R.color[r];
        }
        if (lane == ll) len = (64u - ll) + e2;
      }
    }
    const bool is_long = runs_on && !strad;
    // ---- 2. short segments: head lanes replay their records in order ----------------------------------
    const bool run_short = head && !is_long && vr.ok;
    Voxel v{0.0f, 0.0f, 0u};
    if (run_short) {
      v.d = __uint_as_float(vr.ptr[0]);
      v.w = __uint_as_float(vr.ptr[1]);
      v.c = vr.ptr[2];
    }
    // Most records of a frame lie in free space (saturating_update: distance stays == trunc, an integer weight is
    // added).  A short segment that consists of such records only, on a voxel in the matching state, is folded in one
    // step exactly like the pieces of a long segment (fold_pieces) instead of being replayed record by record; what is
    // left to replay decides how long this wave's loop runs.
    bool folded = false;
    {
      const u64 foldmask = __ballot(fold);
      const u32 wi = fold ? static_cast<u32>(uw) : 0u;
      const u32 psum = wave_inclusive_scan(wi);
      const u32 seg_hi = static_cast<u32>(__shfl(static_cast<int>(psum), static_cast<int>((lane + len - 1u) & 63u), 64));
      const bool inside = run_short && len > 0 && lane + len <= 64u;  // the whole segment lies in this wave's lanes
      if (inside) {
        const u64 seg = ((len == 64u) ? ~0ull : ((1ull << len) - 1ull)) << lane;
        if ((foldmask & seg) == seg) folded = fold_pieces(P, v, seg_hi - (psum - wi));
      }
    }
    {  // longest run of updates on one voxel (statistics only; long runs report theirs from k_apply_long)
      u32 mr = (head && !is_long) ? len : 0u;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) mr = max(mr, static_cast<u32>(__shfl_xor(static_cast<int>(mr), off, 64)));
      my_maxrun = max(my_maxrun, mr);
    }
    u32 max_len = (run_short && !folded) ? len : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) max_len = max(max_len, static_cast<u32>(__shfl_xor(static_cast<int>(max_len), off, 64)));
    for (u32 k = 0; k < max_len; ++k) {
      const u32 idx = lane + k;
      const int src = static_cast<int>(idx & 63u);
      float s_k = __shfl(sdf, src, 64);
      float u_k = __shfl(uw, src, 64);
      u32 c_k = static_cast<u32>(__shfl(static_cast<int>(color), src, 64));
      if (strad && k + ll >= 64u) {  // wave-uniform: only then can a lane reach past the wave (lane ll is the only one that does)
        const float s_2 = __shfl(sdf2, src, 64);
        const float u_2 = __shfl(uw2, src, 64);
        const u32 c_2 = static_cast<u32>(__shfl(static_cast<int>(color2), src, 64));
        if (idx >= 64u) {
          s_k = s_2;
          u_k = u_2;
          c_k = c_2;
        }
      }
      if (run_short && !folded && k < len) update_voxel(P, v, s_k, u_k, c_k);
    }
    if (run_short) {
      vr.ptr[0] = __float_as_uint(v.d);
      vr.ptr[1] = __float_as_uint(v.w);
      vr.ptr[2] = v.c;
    }
    // ---- 3. summaries of the pieces of long segments --------------------------------------------------
    // front piece: lanes [0, e) that continue the previous wave's last segment (e = first boundary lane)
    const u32 e = bmask ? static_cast<u32>(__ffsll(static_cast<long long>(bmask))) - 1u : 64u;
    {
      const u64 fmask = (e == 64) ? ~0ull : ((1ull << e) - 1ull);
      const bool all_fold = (__ballot(fold) & fmask) == fmask;
      const float wf = (lane < e) ? uw : 0.0f;
      u32 wsum = all_fold ? static_cast<u32>(wf) : 0u;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) wsum += static_cast<u32>(__shfl_xor(static_cast<int>(wsum), off, 64));
      if (lane == 0) {
        piece_front[wv] = e | ((all_fold && e > 0) ? kPieceFoldable : 0u);
        piece_wsum[2 * wv] = wsum;
      }
    }
    // back piece: the long segment (at most one) that starts in this wave; 0 = none
    const u64 lmask = __ballot(is_long);
    if (lmask) {
      const u32 lh = static_cast<u32>(__ffsll(static_cast<long long>(lmask))) - 1u;
      const u64 bm = ~((1ull << lh) - 1ull);
      const bool all_fold = (__ballot(fold) & bm) == bm;
      const float wb = (lane >= lh) ? uw : 0.0f;
      u32 wsum = all_fold ? static_cast<u32>(wb) : 0u;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) wsum += static_cast<u32>(__shfl_xor(static_cast<int>(wsum), off, 64));
      if (lane == lh) {
        piece_back[wv] = (64u - lh) | (all_fold ? kPieceFoldable : 0u);
        piece_wsum[2 * wv + 1] = wsum;
      }
      my_long += 1;
    } else if (lane == 0) {
      piece_back[wv] = 0;
    }
    my_updates += static_cast<u32>(__popcll(__ballot(valid)));
    my_voxels += static_cast<u32>(__popcll(__ballot(head)));
  }
  if (lane == 0) {
    if (my_updates) atomicAdd(&blk_updates, my_updates);
    if (my_voxels) atomicAdd(&blk_voxels, my_voxels);
    if (my_long) atomicAdd(&blk_long, my_long);
    if (my_maxrun) atomicMax(&blk_maxrun, my_maxrun);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    u32* sh = cnt->shard[blockIdx.x & 63u];
    if (blk_updates) atomicAdd(&sh[kShUpdates], blk_updates);
    if (blk_voxels) atomicAdd(&sh[kShVoxels], blk_voxels);
    if (blk_long) atomicAdd(&sh[kShLong], blk_long);
    if (blk_maxrun) atomicMax(&sh[kShMaxRun], blk_maxrun);
  }
}

// replay records [a, a + count) of the long segment in order (count <= 64)
__device__ __forceinline__ void replay_piece(const FrameParams& P, const RayArrays& R, const u32* __restrict__ rec_ray, int gx, int gy, int gz, u32 a, u32 count,
                                             u32 lane, Voxel& v) {
  float s = 0.0f, u = 0.0f;
  u32 c = 0;
  if (lane < count) {
    const u32 r = rec_ray[a + lane];
    const F3 pg{R.px[r], R.py[r], R.pz[r]};
    s = compute_sdf(P, pg, gx, gy, gz);
    u = update_weight(P, s, R.w[r]);
    c = R.color[r];
  }
  for (u32 k = 0; k < count; ++k) update_voxel(P, v, readlane_f32(s, k), readlane_f32(u, k), static_cast<u32>(__builtin_amdgcn_readlane(c, k)));
}
__global__ void __launch_bounds__(256) k_apply_long(const FrameParams* __restrict__ Pp, RayArrays R, LayerView L, const int4* __restrict__ touched_slots, RecordView V,
                                                    const u32* __restrict__ piece_front, const u32* __restrict__ piece_back,
                                                    const u32* __restrict__ piece_wsum, Counters* cnt, u32* layer_err, u32* __restrict__ h_nblocks) {
  const FrameParams P = *Pp;
  // last kernel of the frame: make this frame's error bits sticky until the host next looks, and leave the layer's block
  // count where the host can read it without a sync (pinned word; it decides when to grow the pool)
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (cnt->err) atomicOr(layer_err, cnt->err);
    *h_nblocks = min(*L.d_nblocks, L.capacity);
  }
  const u32 n = uniform_u32((cnt->err & kErrRecords) ? 0u : *V.d_n);
  if (n == 0) return;
  const u32 par = uniform_u32(V.info->parity & 1u);
  const u32* __restrict__ rec_key = V.key[par];
  const u32* __restrict__ rec_ray = V.ray[par];
  const u32 lane = lane_id();
  const u32 n_waves = (n + 63) >> 6;
  const u32 waves_total = (gridDim.x * blockDim.x) >> 6;
  // every record wave with a back piece owns one long segment: one wave of this kernel per record wave, so that the long
  // segments of a frame (they cluster: the voxels of the cone in front of the sensor) are folded side by side
  for (u32 w0 = uniform_u32((blockIdx.x * blockDim.x + threadIdx.x) >> 6); w0 < n_waves; w0 += waves_total) {
    {
      const u32 pb = uniform_u32(piece_back[w0]);
      if (pb == 0u) continue;
      const u32 count0 = pb & 127u;
      const u32 start = (w0 << 6) + 64u - count0;
      const VoxelRef vr = locate_voxel(L, touched_slots, uniform_u32(rec_key[start]));
      if (!vr.ok) continue;
      Voxel v{__uint_as_float(vr.ptr[0]), __uint_as_float(vr.ptr[1]), vr.ptr[2]};
      // the piece at the back of the wave that holds the head
      if (!((pb & kPieceFoldable) && fold_pieces(P, v, uniform_u32(piece_wsum[2 * w0 + 1])))) replay_piece(P, R, rec_ray, vr.gx, vr.gy, vr.gz, start, count0, lane, v);
      // front pieces of the following waves; the segment ends with the first piece shorter than 64
      bool more = true;
      u32 seg_records = count0;  // statistics: length of this voxel's run
      for (u32 wbase = w0 + 1; more && wbase < n_waves; wbase += 64) {
        const u32 w = wbase + lane;
        const u32 pf = (w < n_waves) ? piece_front[w] : 0u;
        const u32 ws = (w < n_waves) ? piece_wsum[2 * w] : 0u;
        const u32 count = pf & 127u;
        const u64 end_mask = __ballot(count < 64u);
        const u32 n_use = end_mask ? static_cast<u32>(__ffsll(static_cast<long long>(end_mask))) : 64u;  // pieces [0, n_use) belong to the segment
        if (end_mask) more = false;
        {
          u32 c = (lane < n_use) ? count : 0u;
#pragma unroll
          for (int off = 32; off > 0; off >>= 1) c += static_cast<u32>(__shfl_xor(static_cast<int>(c), off, 64));
          seg_records += c;
        }
        const u64 use_mask = (n_use == 64) ? ~0ull : ((1ull << n_use) - 1ull);
        const u64 hard_mask = __ballot(!(pf & kPieceFoldable)) & use_mask;  // pieces that need a record-by-record replay
        u32 pos = 0;
        while (pos < n_use) {
          const u64 hard_from = hard_mask >> pos;
          const u32 run = hard_from ? static_cast<u32>(__ffsll(static_cast<long long>(hard_from))) - 1u : (n_use - pos);  // foldable pieces ahead
          if (run > 0) {
            u32 t = (lane >= pos && lane < pos + run) ? ws : 0u;  // total weight of pieces [pos, pos + run)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) t += static_cast<u32>(__shfl_xor(static_cast<int>(t), off, 64));
            if (fold_pieces(P, v, t)) {
              pos += run;
              continue;
            }
          }
          // replay piece `pos` (hard, or the voxel is not in a foldable state yet)
          const u32 cnt_p = static_cast<u32>(__builtin_amdgcn_readlane(count, pos));
          if (cnt_p) replay_piece(P, R, rec_ray, vr.gx, vr.gy, vr.gz, (wbase + pos) << 6, cnt_p, lane, v);
          pos += 1;
        }
      }
      if (lane == 0) {
        vr.ptr[0] = __float_as_uint(v.d);
        vr.ptr[1] = __float_as_uint(v.w);
        vr.ptr[2] = v.c;
        atomicMax(&cnt->shard[w0 & 63u][kShMaxRun], seg_records);
      }
    }
  }
}
