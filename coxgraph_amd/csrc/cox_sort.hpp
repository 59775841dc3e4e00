// Device-wide exclusive scan and stable LSD radix sort of (u32 key, u32 value) pairs for gfx950.
//
// Element counts AND the number of key bits may live in device memory, so a frame never needs a
// host round trip between the kernel that produces a count and the kernels that consume it.  The
// host only passes a *hint* that sizes the grids; every kernel grid-strides, so any hint is correct.
//
// The sort is STABLE: equal keys keep their input order.  The engine relies on that to get the
// canonical per-voxel update order without carrying the order in the key.
#pragma once
#include "cox_device.hpp"

namespace cox {

// ------------------------------------------------------------------------------------------------
// exclusive scan, u32, three launches: per-block scan, scan of block sums, add back
// ------------------------------------------------------------------------------------------------
constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanThreads * kScanItems;  // 2048

__device__ __forceinline__ u32 wave_inclusive_scan(u32 v) {
  const u32 lane = lane_id();
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const u32 o = __shfl_up(v, off, 64);
    if (lane >= static_cast<u32>(off)) v += o;
  }
  return v;
}
// block-wide exclusive scan of one value per thread; NW = waves per block; returns exclusive prefix
template <int NW>
__device__ __forceinline__ u32 block_exclusive_scan(u32 v, u32* total, u32* lds /*[NW]*/) {
  const u32 lane = lane_id();
  const u32 wave = threadIdx.x >> 6;
  const u32 inc = wave_inclusive_scan(v);
  __syncthreads();  // protect lds reuse across calls
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  u32 base = 0, tot = 0;
#pragma unroll
  for (u32 w = 0; w < NW; ++w) {
    const u32 s = lds[w];
    if (w < wave) base += s;
    tot += s;
  }
  *total = tot;
  return base + inc - v;
}

// n = *d_n if d_n != nullptr else n_max; grid-strides over tiles, so any grid size is correct
static __global__ void __launch_bounds__(kScanThreads) k_scan_blocks(const u32* __restrict__ in, u32* __restrict__ out, u32* __restrict__ block_sums,
                                                              const u32* __restrict__ d_n, u32 n_max) {
  __shared__ u32 lds[4];
  const u32 n = d_n ? min(*d_n, n_max) : n_max;
  const u32 n_tiles = (n + kScanTile - 1) / kScanTile;
  for (u32 tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const u32 base = tile * kScanTile + threadIdx.x * kScanItems;
    u32 v[kScanItems];
    u32 sum = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
      v[i] = (base + i < n) ? in[base + i] : 0u;
      sum += v[i];
    }
    u32 total;
    u32 ex = block_exclusive_scan<4>(sum, &total, lds);
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
      if (base + i < n) out[base + i] = ex;
      ex += v[i];
    }
    if (threadIdx.x == 0) block_sums[tile] = total;
  }
}
// single block: exclusive scan of the tile sums in place; writes the grand total to *d_total
static __global__ void __launch_bounds__(1024) k_scan_sums(u32* __restrict__ block_sums, const u32* __restrict__ d_n, u32 n_max, u32* __restrict__ d_total) {
  __shared__ u32 lds[16];
  const u32 n = d_n ? min(*d_n, n_max) : n_max;
  const u32 n_tiles = (n + kScanTile - 1) / kScanTile;
  u32 carry = 0;
  for (u32 base = 0; base < n_tiles; base += 1024) {
    const u32 i = base + threadIdx.x;
    const u32 v = (i < n_tiles) ? block_sums[i] : 0u;
    u32 total;
    const u32 ex = block_exclusive_scan<16>(v, &total, lds);
    if (i < n_tiles) block_sums[i] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0 && d_total) *d_total = carry;
}
static __global__ void __launch_bounds__(kScanThreads) k_scan_add(u32* __restrict__ out, const u32* __restrict__ block_sums, const u32* __restrict__ d_n,
                                                           u32 n_max) {
  const u32 n = d_n ? min(*d_n, n_max) : n_max;
  const u32 n_tiles = (n + kScanTile - 1) / kScanTile;
  for (u32 tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const u32 add = block_sums[tile];
    const u32 base = tile * kScanTile + threadIdx.x * kScanItems;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i)
      if (base + i < n) out[base + i] += add;
  }
}

struct ScanWorkspace {
  u32* block_sums = nullptr;  // capacity >= ceil(n_max / kScanTile)
};
static inline u32 scan_num_blocks(u32 n) { return (n + kScanTile - 1) / kScanTile; }
// out may alias in.  d_total (optional) receives the sum of the n inputs.  n = min(*d_n, n_max) when
// d_n is given; n_hint only sizes the grid.
static inline void exclusive_scan_u32(const u32* in, u32* out, const u32* d_n, u32 n_max, u32 n_hint, u32* d_total, const ScanWorkspace& ws,
                                      hipStream_t s) {
  const u32 nb = scan_num_blocks(n_hint ? n_hint : 1);
  const u32 grid = nb < 1 ? 1 : (nb > 4096 ? 4096 : nb);
  hipLaunchKernelGGL(k_scan_blocks, dim3(grid), dim3(kScanThreads), 0, s, in, out, ws.block_sums, d_n, n_max);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, s, ws.block_sums, d_n, n_max, d_total);
  hipLaunchKernelGGL(k_scan_add, dim3(grid), dim3(kScanThreads), 0, s, out, ws.block_sums, d_n, n_max);
}

// ------------------------------------------------------------------------------------------------
// stable LSD radix sort, RB-bit digits (RB = 8 or 11)
// ------------------------------------------------------------------------------------------------
constexpr int kRsThreads = 256;
constexpr int kRsWaves = 4;
constexpr int kRsRounds = 8;
constexpr int kRsWaveTile = 64 * kRsRounds;      // 512 elements per wave
constexpr int kRsTile = kRsWaveTile * kRsWaves;  // 2048 elements per tile (smaller: the 2048-bin histograms dominate; larger: too few workgroups)
constexpr int kRsMaxPasses = 4;
constexpr u32 kRsMaxTiles = 4096;  // beyond this many tiles the tile doubles: the 2^RB x tiles histogram matrix must stay small next to the data

// log2(tile / kRsTile): the smallest k with n <= kRsMaxTiles * (kRsTile << k).  Computed identically by all three kernels.
__device__ __forceinline__ u32 rs_tile_shift(u32 n) {
  u32 k = 0;
  while ((static_cast<u64>(kRsMaxTiles) * kRsTile << k) < n) ++k;
  return k;
}

// device-side description of one sort
struct SortInfo {
  u32 nbits;   // key bits to sort on (passes whose shift >= nbits do nothing)
  u32 parity;  // 0: result in buffer 0, 1: result in buffer 1 (written by the kernels)
  u32 base;    // the nbits bits start at this bit of the key (a stable partition by the key's upper part); 0 = whole key
  u32 pad;
};

// Digit layout of one sort: the nbits key bits are split evenly over ceil(nbits / RB) passes (23 bits with RB = 12 ->
// 12 + 12, 20 bits with RB = 11 -> 10 + 10), at least 6 bits per pass (the offsets kernel works in groups of 64 digits).
// A last digit that reaches above nbits is harmless: keys are < 2^nbits or all ones.  Returns false when `pass` has
// nothing to do.  Computed identically by the three kernels of a pass and by the host.
template <int RB>
__host__ __device__ __forceinline__ bool rs_pass_digits(u32 nbits, int pass, int* shift, u32* bits) {
  if (nbits == 0) return false;
  const u32 np = (nbits + RB - 1) / RB;
  u32 w = (nbits + np - 1) / np;
  if (w < 6) w = 6;
  *shift = pass * static_cast<int>(w);
  *bits = w;
  return static_cast<u32>(*shift) < nbits;
}

template <int RB>
__global__ void __launch_bounds__(kRsThreads) k_rs_hist(const u32* __restrict__ k0, const u32* __restrict__ k1, const u32* __restrict__ d_n, u32 n_max,
                                                        int pass, const SortInfo* __restrict__ info, int host_bits,
                                                        u32* __restrict__ counts /*[n_tiles][2^bits]*/, u32 tiles_cap, u32* __restrict__ totals /*[2^bits]*/) {
  __shared__ u32 h[1u << RB];    // RB = widest digit; this pass sorts on `bits` <= RB bits at `shift`
  __shared__ u32 acc[1u << RB];  // this workgroup's share of the digit totals: ONE atomic per digit and workgroup, not per tile
                                 // (12 000 tiles x 512 digits of atomics were most of this kernel at 2.5 * 10^7 records)
  int shift;
  u32 bits;
  if (!rs_pass_digits<RB>(host_bits >= 0 ? static_cast<u32>(host_bits) : info->nbits, pass, &shift, &bits)) return;
  if (host_bits < 0) shift += static_cast<int>(info->base);
  const u32 kDigits = 1u << bits;
  const u32* keys = (pass & 1) ? k1 : k0;
  const u32 n = d_n ? min(*d_n, n_max) : n_max;
  const u32 tile_elems = static_cast<u32>(kRsTile) << rs_tile_shift(n);
  const u32 n_tiles = (n + tile_elems - 1) / tile_elems;
  for (u32 d = threadIdx.x; d < kDigits; d += kRsThreads) acc[d] = 0;  // (digit d is always handled by the same thread)
  for (u32 tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    for (u32 d = threadIdx.x; d < kDigits; d += kRsThreads) h[d] = 0;
    __syncthreads();
    const u64 start = static_cast<u64>(tile) * tile_elems;
    for (u64 i = start + threadIdx.x; i < start + tile_elems && i < n; i += kRsThreads) atomicAdd(&h[(keys[i] >> shift) & (kDigits - 1)], 1u);
    __syncthreads();
    for (u32 d = threadIdx.x; d < kDigits; d += kRsThreads) {
      const u32 c = h[d];
      counts[static_cast<size_t>(tile) * kDigits + d] = c;  // tile-major: coalesced here and in the scatter kernel
      acc[d] += c;
    }
    __syncthreads();
  }
  for (u32 d = threadIdx.x; d < kDigits; d += kRsThreads)
    if (acc[d]) atomicAdd(&totals[d], acc[d]);
}

// counts[tile][digit] -> position of the first element of (tile, digit) in the sorted output =
// (sum of the totals of all lower digits) + (counts of the same digit in all lower tiles).
// One workgroup owns 64 digits (lane = digit, so every access is a coalesced 256-B row segment); its 16 waves
// split the tiles: partial sums, a prefix over the waves, then the running prefix is written back.
constexpr int kRsOffThreads = 1024;
template <int RB>
__global__ void __launch_bounds__(kRsOffThreads) k_rs_offsets(const u32* __restrict__ d_n, u32 n_max, int pass, SortInfo* __restrict__ info, int host_bits,
                                                              u32* __restrict__ counts, u32 tiles_cap, const u32* __restrict__ totals) {
  int shift;
  u32 bits;
  if (!rs_pass_digits<RB>(host_bits >= 0 ? static_cast<u32>(host_bits) : info->nbits, pass, &shift, &bits)) return;
  if (host_bits < 0) shift += static_cast<int>(info->base);
  const u32 kDigits = 1u << bits;
  if (blockIdx.x * 64u >= kDigits) return;  // the grid covers 2^RB digits
  constexpr u32 kWaves = kRsOffThreads / 64;
  __shared__ u32 lds[16];
  __shared__ u32 part[kWaves][64];
  __shared__ u32 group_base;
  const u32 n = d_n ? min(*d_n, n_max) : n_max;
  const u32 tile_elems = static_cast<u32>(kRsTile) << rs_tile_shift(n);
  const u32 n_tiles = (n + tile_elems - 1) / tile_elems;
  const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const u32 d0 = blockIdx.x * 64u;  // first digit of this group
  // sum of the totals of all digits below the group
  u32 below = 0;
  for (u32 j = threadIdx.x; j < d0; j += kRsOffThreads) below += totals[j];
  u32 tot;
  (void)block_exclusive_scan<kWaves>(below, &tot, lds);
  if (threadIdx.x == 0) group_base = tot;
  // exclusive prefix of the group's own totals (wave 0)
  const u32 my_total = totals[d0 + lane];
  const u32 digit_excl = wave_inclusive_scan(my_total) - my_total;
  // partial sums of this wave's tile range
  const u32 chunk = (n_tiles + kWaves - 1) / kWaves;
  const u32 t_begin = min(n_tiles, wave * chunk), t_end = min(n_tiles, t_begin + chunk);
  u32 sum = 0;
#pragma unroll 4
  for (u32 t = t_begin; t < t_end; ++t) sum += counts[static_cast<size_t>(t) * kDigits + d0 + lane];
  part[wave][lane] = sum;
  __syncthreads();
  u32 run = group_base + digit_excl;
  for (u32 w = 0; w < wave; ++w) run += part[w][lane];
  for (u32 t = t_begin; t < t_end; t += 4) {  // 4 independent loads in flight per step
    u32 v[4];
#pragma unroll
    for (u32 q = 0; q < 4; ++q) v[q] = (t + q < t_end) ? counts[static_cast<size_t>(t + q) * kDigits + d0 + lane] : 0u;
#pragma unroll
    for (u32 q = 0; q < 4; ++q) {
      if (t + q < t_end) counts[static_cast<size_t>(t + q) * kDigits + d0 + lane] = run;
      run += v[q];
    }
  }
  if (host_bits < 0 && blockIdx.x == 0 && threadIdx.x == 0) info->parity = (pass + 1) & 1;  // this pass runs: its output buffer is current
}

// wave-wide "which lanes hold my digit" (RB ballots), restricted to lanes with valid == true
template <int RB>
__device__ __forceinline__ u64 match_digit(u32 digit, bool valid, u32 bits) {
  u64 peers = __ballot(valid);
#pragma unroll
  for (int b = 0; b < RB; ++b) {
    if (static_cast<u32>(b) < bits) {  // wave-uniform
      const bool bit = (digit >> b) & 1u;
      const u64 m = __ballot(bit);
      peers &= bit ? m : ~m;
    }
  }
  return peers;
}

template <int RB>
__global__ void __launch_bounds__(kRsThreads) k_rs_scatter(u32* __restrict__ k0, u32* __restrict__ v0, u32* __restrict__ k1, u32* __restrict__ v1,
                                                           u32* __restrict__ x0, u32* __restrict__ x1 /* optional second value array */,
                                                           const u32* __restrict__ d_n, u32 n_max, int pass, const SortInfo* __restrict__ info, int host_bits,
                                                           const u32* __restrict__ offsets, u32 tiles_cap, u32* __restrict__ totals) {
  __shared__ u32 base[kRsWaves][1u << RB];  // first per-wave digit counts, then per-wave running output positions
  int shift;
  u32 bits;
  if (!rs_pass_digits<RB>(host_bits >= 0 ? static_cast<u32>(host_bits) : info->nbits, pass, &shift, &bits)) return;
  if (host_bits < 0) shift += static_cast<int>(info->base);
  const u32 kDigits = 1u << bits;
  // the digit totals of this pass were consumed by the offsets kernel: leave them zero for the next sort (no memset per sort)
  if (blockIdx.x == 0)
    for (u32 d = threadIdx.x; d < kDigits; d += kRsThreads) totals[d] = 0;
  const u32* keys_in = (pass & 1) ? k1 : k0;
  const u32* vals_in = (pass & 1) ? v1 : v0;
  u32* keys_out = (pass & 1) ? k0 : k1;
  u32* vals_out = (pass & 1) ? v0 : v1;
  const u32* xtra_in = (pass & 1) ? x1 : x0;
  u32* xtra_out = (pass & 1) ? x0 : x1;
  const bool has_x = x0 != nullptr;
  const u32 n = d_n ? min(*d_n, n_max) : n_max;
  const u32 tshift = rs_tile_shift(n);
  const u32 tile_elems = static_cast<u32>(kRsTile) << tshift;
  const u32 wave_elems = static_cast<u32>(kRsWaveTile) << tshift;  // contiguous elements per wave
  const u32 rounds = static_cast<u32>(kRsRounds) << tshift;
  const u32 n_tiles = (n + tile_elems - 1) / tile_elems;
  const u32 lane = lane_id();
  const u32 wave = threadIdx.x >> 6;
  for (u32 tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    for (u32 w = 0; w < kRsWaves; ++w)
      for (u32 d = threadIdx.x; d < kDigits; d += kRsThreads) base[w][d] = 0;
    __syncthreads();
    const u64 wstart = static_cast<u64>(tile) * tile_elems + static_cast<u64>(wave) * wave_elems;
    // pass 1: per-wave digit counts of this wave's contiguous sub-tile
    for (u32 r = 0; r < rounds; ++r) {
      const u64 i = wstart + static_cast<u64>(r) * 64 + lane;
      if (i < n) atomicAdd(&base[wave][(keys_in[i] >> shift) & (kDigits - 1)], 1u);
    }
    __syncthreads();
    // counts -> output positions: global position of (tile, digit) + counts of the lower waves
    for (u32 d = threadIdx.x; d < kDigits; d += kRsThreads) {
      u32 run = offsets[static_cast<size_t>(tile) * kDigits + d];
#pragma unroll
      for (u32 w = 0; w < kRsWaves; ++w) {
        const u32 c = base[w][d];
        base[w][d] = run;
        run += c;
      }
    }
    __syncthreads();
    // pass 2: stable placement.  Order = (tile, wave, round, lane) = input order.
    u32* my_base = base[wave];
    for (u32 r = 0; r < rounds; ++r) {
      const u64 i = wstart + static_cast<u64>(r) * 64 + lane;
      const bool valid = i < n;
      u32 key = 0, val = 0, xv = 0;
      if (valid) {
        key = keys_in[i];
        val = vals_in[i];
        if (has_x) xv = xtra_in[i];
      }
      const u32 digit = (key >> shift) & (kDigits - 1);
      const u64 peers = match_digit<RB>(digit, valid, bits);
      const u64 lower = peers & ((1ull << lane) - 1ull);
      if (valid) {
        const u32 pos = my_base[digit] + static_cast<u32>(__popcll(lower));
        keys_out[pos] = key;
        vals_out[pos] = val;
        if (has_x) xtra_out[pos] = xv;
      }
      wave_lds_handover();
      if (valid && lower == 0ull) my_base[digit] += static_cast<u32>(__popcll(peers));  // group leader
      wave_lds_handover();
    }
    __syncthreads();
  }
}

constexpr u32 kRsMaxDigitBits = 12;
struct SortWorkspace {
  u32* counts = nullptr;   // [tiles_cap][2^kRsMaxDigitBits]
  u32 tiles_cap = 0;       // >= ceil(capacity / kRsTile)
  u32* totals = nullptr;   // [kRsMaxPasses][2^kRsMaxDigitBits]: zero when allocated, left zero by every sort
};
static inline u32 sort_num_tiles(u64 n) { return static_cast<u32>((n + kRsTile - 1) / kRsTile); }
static inline size_t sort_counts_words(u32 tiles_cap) { return static_cast<size_t>(tiles_cap) << kRsMaxDigitBits; }
static inline size_t sort_totals_words() { return static_cast<size_t>(kRsMaxPasses) << kRsMaxDigitBits; }

// (x0, x1: an optional second value array that travels with the pairs.)
// Sorts (k0,v0) by key bits [0, nbits) in ceil(nbits / RB) passes of equal width (rs_pass_digits).  Buffers ping-pong;
// the buffer that holds the result is returned when nbits is known on the host (bits_on_device == false).  Otherwise
// nbits is read from info->nbits (written by an earlier kernel of the frame), max_passes passes are enqueued (surplus
// ones exit at once), the result buffer is reported in info->parity (device) and -1 is returned.  RB = 11 or 12.
template <int RB>
static inline int radix_sort_pairs(u32* k0, u32* v0, u32* k1, u32* v1, const u32* d_n, u32 n_max, u32 n_hint, int host_bits, bool bits_on_device,
                                   int max_passes, const SortWorkspace& ws, SortInfo* info, hipStream_t s, u32* x0 = nullptr, u32* x1 = nullptr) {
  static_assert(RB >= 6 && RB <= static_cast<int>(kRsMaxDigitBits), "digit width");
  const u32 nt = sort_num_tiles(n_hint ? n_hint : 1);
  const u32 grid = nt < 1 ? 1 : (nt > 8192 ? 8192 : nt);
  int passes = max_passes;
  if (!bits_on_device) passes = (host_bits + RB - 1) / RB;
  if (passes > kRsMaxPasses) passes = kRsMaxPasses;
  const int hb = bits_on_device ? -1 : host_bits;
  for (int p = 0; p < passes; ++p) {
    u32* totals = ws.totals + (static_cast<size_t>(p) << kRsMaxDigitBits);
    hipLaunchKernelGGL(k_rs_hist<RB>, dim3(grid < 1024 ? grid : 1024), dim3(kRsThreads), 0, s, k0, k1, d_n, n_max, p, info, hb, ws.counts, ws.tiles_cap, totals);
    hipLaunchKernelGGL(k_rs_offsets<RB>, dim3((1u << RB) / 64u), dim3(kRsOffThreads), 0, s, d_n, n_max, p, info, hb, ws.counts, ws.tiles_cap, totals);
    hipLaunchKernelGGL(k_rs_scatter<RB>, dim3(grid), dim3(kRsThreads), 0, s, k0, v0, k1, v1, x0, x1, d_n, n_max, p, info, hb, ws.counts, ws.tiles_cap, totals);
  }
  return bits_on_device ? -1 : (passes & 1);
}

}  // namespace cox
