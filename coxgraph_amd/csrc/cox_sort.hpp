// Device-wide exclusive scan and stable LSD radix sort of (u32 key, u32 value) pairs for gfx950.
//
// Element counts live in device memory (d_n) so that no host round trip is needed between the
// kernel that produces a count and the kernels that consume it; the host passes an upper bound
// (n_max) that only sizes the grid.
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
// block-wide exclusive scan of one value per thread (256 threads); returns exclusive prefix, total via *total
__device__ __forceinline__ u32 block_exclusive_scan_256(u32 v, u32* total, u32* lds /*[4]*/) {
  const u32 lane = lane_id();
  const u32 wave = threadIdx.x >> 6;
  const u32 inc = wave_inclusive_scan(v);
  __syncthreads();  // protect lds reuse across calls
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  u32 base = 0;
#pragma unroll
  for (u32 w = 0; w < 4; ++w) {
    const u32 s = lds[w];
    if (w < wave) base += s;
  }
  *total = lds[0] + lds[1] + lds[2] + lds[3];
  return base + inc - v;
}

// n = *d_n if d_n != nullptr else n_max
__global__ void __launch_bounds__(kScanThreads) k_scan_blocks(const u32* __restrict__ in, u32* __restrict__ out, u32* __restrict__ block_sums,
                                                              const u32* __restrict__ d_n, u32 n_max) {
  __shared__ u32 lds[4];
  const u32 n = d_n ? min(*d_n, n_max) : n_max;
  const u32 base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
  u32 v[kScanItems];
  u32 sum = 0;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    v[i] = (base + i < n) ? in[base + i] : 0u;
    sum += v[i];
  }
  u32 total;
  u32 ex = block_exclusive_scan_256(sum, &total, lds);
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    if (base + i < n) out[base + i] = ex;
    ex += v[i];
  }
  if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}
// single block: exclusive scan of block sums in place; writes the grand total to *d_total
__global__ void __launch_bounds__(kScanThreads) k_scan_sums(u32* __restrict__ block_sums, u32 n_blocks, u32* __restrict__ d_total) {
  __shared__ u32 lds[4];
  u32 carry = 0;
  for (u32 base = 0; base < n_blocks; base += kScanThreads) {
    const u32 i = base + threadIdx.x;
    const u32 v = (i < n_blocks) ? block_sums[i] : 0u;
    u32 total;
    const u32 ex = block_exclusive_scan_256(v, &total, lds);
    if (i < n_blocks) block_sums[i] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0 && d_total) *d_total = carry;
}
__global__ void __launch_bounds__(kScanThreads) k_scan_add(u32* __restrict__ out, const u32* __restrict__ block_sums, const u32* __restrict__ d_n,
                                                           u32 n_max) {
  const u32 n = d_n ? min(*d_n, n_max) : n_max;
  const u32 add = block_sums[blockIdx.x];
  const u32 base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i)
    if (base + i < n) out[base + i] += add;
}

struct ScanWorkspace {
  u32* block_sums = nullptr;  // capacity >= ceil(n_max / kScanTile)
};
static inline u32 scan_num_blocks(u32 n_max) { return (n_max + kScanTile - 1) / kScanTile; }
// out may alias in.  d_total (optional) receives the sum of all n inputs.
static inline void exclusive_scan_u32(const u32* in, u32* out, const u32* d_n, u32 n_max, u32* d_total, const ScanWorkspace& ws, hipStream_t s) {
  const u32 nb = scan_num_blocks(n_max);
  if (nb == 0) {
    if (d_total) (void)hipMemsetAsync(d_total, 0, sizeof(u32), s);
    return;
  }
  hipLaunchKernelGGL(k_scan_blocks, dim3(nb), dim3(kScanThreads), 0, s, in, out, ws.block_sums, d_n, n_max);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(kScanThreads), 0, s, ws.block_sums, nb, d_total);
  hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(kScanThreads), 0, s, out, ws.block_sums, d_n, n_max);
}

// ------------------------------------------------------------------------------------------------
// stable LSD radix sort, 8-bit digits
// ------------------------------------------------------------------------------------------------
constexpr int kRsThreads = 256;
constexpr int kRsWaves = 4;
constexpr int kRsRounds = 16;
constexpr int kRsWaveTile = 64 * kRsRounds;         // 1024 elements per wave
constexpr int kRsTile = kRsWaveTile * kRsWaves;     // 4096 elements per block
constexpr int kRsRadix = 256;

__global__ void __launch_bounds__(kRsThreads) k_rs_hist(const u32* __restrict__ keys, const u32* __restrict__ d_n, u32 n_max, int shift,
                                                        u32* __restrict__ counts /*[256][nb]*/, u32 nb) {
  __shared__ u32 h[kRsRadix];
  const u32 n = d_n ? min(*d_n, n_max) : n_max;
  h[threadIdx.x] = 0;
  __syncthreads();
  const u32 start = blockIdx.x * kRsTile;
  for (u32 i = start + threadIdx.x; i < start + kRsTile && i < n; i += kRsThreads) atomicAdd(&h[(keys[i] >> shift) & 255u], 1u);
  __syncthreads();
  counts[threadIdx.x * nb + blockIdx.x] = h[threadIdx.x];
}

// wave-wide "which lanes hold my digit" (8 ballots), restricted to lanes with valid == true
__device__ __forceinline__ u64 match_digit(u32 digit, bool valid) {
  u64 peers = __ballot(valid);
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const bool bit = (digit >> b) & 1u;
    const u64 m = __ballot(bit);
    peers &= bit ? m : ~m;
  }
  return peers;
}

__global__ void __launch_bounds__(kRsThreads) k_rs_scatter(const u32* __restrict__ keys_in, const u32* __restrict__ vals_in, u32* __restrict__ keys_out,
                                                           u32* __restrict__ vals_out, const u32* __restrict__ d_n, u32 n_max, int shift,
                                                           const u32* __restrict__ offsets /*scanned [256][nb]*/, u32 nb) {
  __shared__ u32 wave_cnt[kRsWaves][kRsRadix];
  __shared__ u32 base[kRsWaves][kRsRadix];
  const u32 n = d_n ? min(*d_n, n_max) : n_max;
  const u32 lane = lane_id();
  const u32 wave = threadIdx.x >> 6;
  for (u32 w = 0; w < kRsWaves; ++w) wave_cnt[w][threadIdx.x] = 0;
  __syncthreads();
  const u32 wstart = blockIdx.x * kRsTile + wave * kRsWaveTile;
  // pass 1: per-wave digit counts of this wave's contiguous sub-tile
  for (int r = 0; r < kRsRounds; ++r) {
    const u32 i = wstart + r * 64 + lane;
    if (i < n) atomicAdd(&wave_cnt[wave][(keys_in[i] >> shift) & 255u], 1u);
  }
  __syncthreads();
  {
    // thread t owns digit t: global base of (block, digit) + counts of the lower waves
    u32 run = offsets[threadIdx.x * nb + blockIdx.x];
    for (u32 w = 0; w < kRsWaves; ++w) {
      base[w][threadIdx.x] = run;
      run += wave_cnt[w][threadIdx.x];
    }
  }
  __syncthreads();
  // pass 2: stable placement.  Order = (block, wave, round, lane) = input order.
  volatile u32* my_base = base[wave];
  for (int r = 0; r < kRsRounds; ++r) {
    const u32 i = wstart + r * 64 + lane;
    const bool valid = i < n;
    u32 key = 0, val = 0;
    if (valid) {
      key = keys_in[i];
      val = vals_in[i];
    }
    const u32 digit = (key >> shift) & 255u;
    const u64 peers = match_digit(digit, valid);
    if (valid) {
      const u32 rank = __popcll(peers & ((1ull << lane) - 1ull));
      const u32 pos = my_base[digit] + rank;
      keys_out[pos] = key;
      vals_out[pos] = val;
    }
    __builtin_amdgcn_wave_barrier();
    if (valid && (peers & ((1ull << lane) - 1ull)) == 0ull) my_base[digit] += static_cast<u32>(__popcll(peers));  // group leader
    __builtin_amdgcn_wave_barrier();
  }
}

struct SortWorkspace {
  u32* counts = nullptr;  // capacity >= 256 * ceil(n_max / kRsTile)
  ScanWorkspace scan;     // capacity >= scan_num_blocks(256 * ceil(n_max / kRsTile))
};
static inline u32 sort_num_blocks(u32 n_max) { return (n_max + kRsTile - 1) / kRsTile; }

// Sorts by key bits [0, nbits).  Buffers ping-pong: input in (k0, v0); returns 0 if the result is
// in (k0, v0), 1 if it is in (k1, v1).
static inline int radix_sort_pairs(u32* k0, u32* v0, u32* k1, u32* v1, const u32* d_n, u32 n_max, int nbits, const SortWorkspace& ws,
                                   hipStream_t s) {
  const u32 nb = sort_num_blocks(n_max);
  if (nb == 0 || nbits <= 0) return 0;
  int cur = 0;
  for (int shift = 0; shift < nbits; shift += 8) {
    u32* ki = cur ? k1 : k0;
    u32* vi = cur ? v1 : v0;
    u32* ko = cur ? k0 : k1;
    u32* vo = cur ? v0 : v1;
    hipLaunchKernelGGL(k_rs_hist, dim3(nb), dim3(kRsThreads), 0, s, ki, d_n, n_max, shift, ws.counts, nb);
    exclusive_scan_u32(ws.counts, ws.counts, nullptr, 256u * nb, nullptr, ws.scan, s);
    hipLaunchKernelGGL(k_rs_scatter, dim3(nb), dim3(kRsThreads), 0, s, ki, vi, ko, vo, d_n, n_max, shift, ws.counts, nb);
    cur ^= 1;
  }
  return cur;
}

}  // namespace cox
