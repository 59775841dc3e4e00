// MI355X (gfx950) submap TSDF-to-TSDF registration cost behind include/coxgraph_hip.h.
//
// Replaces the work of voxgraph's RegistrationCostFunction::Evaluate for the constraint that
// coxgraph::server::PoseGraphInterface::addForceRegistrationConstraint requests
// (coxgraph/src/server/pose_graph_interface.cpp:88-105) and that PoseGraphInterface::optimize
// evaluates inside ceres::Solve (pose_graph_interface.cpp:32-49).
//
//   cox_reg_evaluate   Ceres-shaped: residuals[n], two row-major n x 4 Jacobians
//   cox_reg_normal_eq  fused: per-point residual + Jacobian rows stay in registers / LDS; each wave
//                      accumulates sum x x^T for x = [J_ref(4) J_read(4) r 1 w has_corr] with
//                      v_mfma_f64_16x16x4_f64 (the only MFMA use in the engine), partials are summed
//                      in a fixed order, so H, b, cost are reproducible run to run.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <new>

#include "../../include/coxgraph_hip.h"
#include "cox_device.hpp"
#include "cox_internal.hpp"

using namespace cox;

struct ReadingView {
  const u32* voxels;
  const u64* ht_keys;
  const u32* ht_vals;
  u32 ht_mask;
  float voxel_size, voxel_size_inv, block_size, block_size_inv;
};

// relative pose reading<-reference: float for the per-point transform, double for the Jacobians
struct RelPose {
  float R[9];
  float t[3];
  double cf, sf, cr, sr;
  double tf[3], tr[3];
};

static RelPose make_rel_pose(const double ref[4], const double read[4]) {
  RelPose P;
  P.cf = std::cos(ref[3]);
  P.sf = std::sin(ref[3]);
  P.cr = std::cos(read[3]);
  P.sr = std::sin(read[3]);
  for (int k = 0; k < 3; ++k) {
    P.tf[k] = ref[k];
    P.tr[k] = read[k];
  }
  const double c = P.cr * P.cf + P.sr * P.sf, s = P.cr * P.sf - P.sr * P.cf;
  const double Rd[9] = {c, -s, 0, s, c, 0, 0, 0, 1};
  for (int i = 0; i < 9; ++i) P.R[i] = static_cast<float>(Rd[i]);
  const double dx = ref[0] - read[0], dy = ref[1] - read[1], dz = ref[2] - read[2];
  P.t[0] = static_cast<float>(P.cr * dx + P.sr * dy);
  P.t[1] = static_cast<float>(-P.sr * dx + P.cr * dy);
  P.t[2] = static_cast<float>(dz);
  return P;
}

__constant__ float c_interp_table[8][8] = {{1, 0, 0, 0, 0, 0, 0, 0},   {-1, 0, 0, 0, 1, 0, 0, 0},   {-1, 0, 1, 0, 0, 0, 0, 0},
                                           {-1, 1, 0, 0, 0, 0, 0, 0},  {1, 0, -1, 0, -1, 0, 1, 0},  {1, -1, -1, 1, 0, 0, 0, 0},
                                           {1, -1, 0, 0, -1, 1, 0, 0}, {-1, 1, 1, -1, 1, -1, -1, 1}};

struct PointResult {
  double r;
  double J[8];  // J_ref(4), J_read(4), unscaled
  float w;
  bool has;
};

// voxblox Interpolator<TsdfVoxel>::getVoxelsAndQVector + trilinear value/gradient, then the voxgraph residual
__device__ __forceinline__ PointResult reg_point(const ReadingView& L, const RelPose& P, const float* __restrict__ pt, double no_corr_cost) {
  PointResult o;
  const float px = pt[0], py = pt[1], pz = pt[2], pd = pt[3], pw = pt[4];
  o.w = pw;
  o.has = false;
  o.r = static_cast<double>(pw) * no_corr_cost;
#pragma unroll
  for (int k = 0; k < 8; ++k) o.J[k] = 0.0;
  const float pos[3] = {(P.R[0] * px + P.R[1] * py) + P.R[2] * pz + P.t[0], (P.R[3] * px + P.R[4] * py) + P.R[5] * pz + P.t[1],
                        (P.R[6] * px + P.R[7] * py) + P.R[8] * pz + P.t[2]};
  float sc[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) sc[k] = pos[k] * L.block_size_inv;
  if (!(index_in_range(sc[0]) && index_in_range(sc[1]) && index_in_range(sc[2]))) return o;
  int b[3], vi[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) b[k] = grid_index(sc[k]);
  if (ht_find(L.ht_keys, L.ht_mask, pack_key(b[0], b[1], b[2])) == kInvalid) return o;  // block of the point must exist
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float origin = static_cast<float>(b[k]) * L.block_size;
    const float rel = pos[k] - origin;
    int v = grid_index(rel * L.voxel_size_inv);
    v = v > 15 ? 15 : (v < 0 ? 0 : v);
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
  const u32 base_slot = ht_find(L.ht_keys, L.ht_mask, pack_key(b[0], b[1], b[2]));
  if (base_slot == kInvalid) return o;
  const u32 base_pool = L.ht_vals[base_slot];
  if (base_pool == kInvalid) return o;
  float d[8], off[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float c0 = static_cast<float>(b[k]) * L.block_size + center_coord(vi[k], L.voxel_size);
    off[k] = (pos[k] - c0) * L.voxel_size_inv;
  }
  // the 8 neighbours: at most 8 blocks, one per corner of the 2x2x2 cell; resolve the (up to 7) other blocks first, then issue
  // all 16 voxel loads independently of each other (the early-outs of a straight transcription serialise them)
  u32 pools[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int mx = (c >> 2) & 1, my = (c >> 1) & 1, mz = c & 1;
    const bool need = (!mx || vi[0] == 15) && (!my || vi[1] == 15) && (!mz || vi[2] == 15);  // this block combination is touched
    u32 pool = base_pool;
    if (c != 0) {
      pool = kInvalid;
      if (need) {
        const u32 slot = ht_find(L.ht_keys, L.ht_mask, pack_key(b[0] + mx, b[1] + my, b[2] + mz));
        if (slot != kInvalid) pool = L.ht_vals[slot];
      }
    }
    pools[c] = pool;
  }
  float wv[8];
  bool all_blocks = true;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int v[3] = {vi[0] + ((i >> 2) & 1), vi[1] + ((i >> 1) & 1), vi[2] + (i & 1)};
    int sel = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (v[k] >= 16) {
        v[k] -= 16;
        sel |= 4 >> k;
      }
    const u32 pool = pools[sel];
    all_blocks = all_blocks && pool != kInvalid;
    const u32 safe = pool == kInvalid ? base_pool : pool;
    const u32* vox = L.voxels + (static_cast<size_t>(safe) * kVoxelsPerBlock + static_cast<u32>(v[0] + 16 * (v[1] + 16 * v[2]))) * kWordsPerVoxel;
    d[i] = __uint_as_float(vox[0]);
    wv[i] = __uint_as_float(vox[1]);
  }
  if (!all_blocks) return o;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (!(wv[i] > 0.0f)) return o;  // Interpolator<TsdfVoxel>::isVoxelValid
  float md[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c) s += c_interp_table[r][c] * d[c];
    md[r] = s;
  }
  const float dx = off[0], dy = off[1], dz = off[2];
  const float q[8] = {1.0f, dx, dy, dz, dx * dy, dy * dz, dz * dx, dx * dy * dz};
  const float qx[8] = {0, 1, 0, 0, dy, 0, dz, dy * dz};
  const float qy[8] = {0, 0, 1, 0, dx, dz, 0, dz * dx};
  const float qz[8] = {0, 0, 0, 1, 0, dy, dx, dx * dy};
  float val = 0.0f, gxf = 0.0f, gyf = 0.0f, gzf = 0.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i) val += q[i] * md[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    gxf += qx[i] * md[i];
    gyf += qy[i] * md[i];
    gzf += qz[i] * md[i];
  }
  gxf *= L.voxel_size_inv;
  gyf *= L.voxel_size_inv;
  gzf *= L.voxel_size_inv;
  o.has = true;
  o.r = static_cast<double>((pd - val) * pw);
  const double w = pw, gx = gxf, gy = gyf, gz = gzf;
  const double gRx = gx * P.cr - gy * P.sr, gRy = gx * P.sr + gy * P.cr, gRz = gz;
  const double dRfp_x = -P.sf * px - P.cf * py, dRfp_y = P.cf * px - P.sf * py;
  const double qxx = (P.cf * px - P.sf * py) + P.tf[0] - P.tr[0], qyy = (P.sf * px + P.cf * py) + P.tf[1] - P.tr[1];
  const double dpr_x = -P.sr * qxx + P.cr * qyy, dpr_y = -P.cr * qxx - P.sr * qyy;
  o.J[0] = -w * gRx;
  o.J[1] = -w * gRy;
  o.J[2] = -w * gRz;
  o.J[3] = -w * (gRx * dRfp_x + gRy * dRfp_y);
  o.J[4] = w * gRx;
  o.J[5] = w * gRy;
  o.J[6] = w * gRz;
  o.J[7] = -w * (gx * dpr_x + gy * dpr_y);
  return o;
}

// ---- Ceres-shaped evaluation ------------------------------------------------------------------------
constexpr int kRegThreads = 256;
constexpr int kPartial = 256;  // one 16x16 f64 tile per workgroup

__global__ void __launch_bounds__(kRegThreads) k_reg_evaluate(ReadingView L, RelPose P, const float* __restrict__ pts, const u32* __restrict__ sample_idx,
                                                              u32 n_res, double no_corr_cost, double* __restrict__ res, double* __restrict__ jf,
                                                              double* __restrict__ jr, double* __restrict__ block_w) {
  __shared__ double wsum[kRegThreads / 64];
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  double w = 0.0;
  if (i < n_res) {
    const u32 pi = sample_idx ? sample_idx[i] : i;
    const PointResult o = reg_point(L, P, pts + 5ull * pi, no_corr_cost);
    w = static_cast<double>(o.w);
    if (res) res[i] = o.r;
    if (jf) {
#pragma unroll
      for (int k = 0; k < 4; ++k) jf[4ull * i + k] = o.J[k];
    }
    if (jr) {
#pragma unroll
      for (int k = 0; k < 4; ++k) jr[4ull * i + k] = o.J[4 + k];
    }
  }
  // deterministic sum of the reference weights: lanes in order, waves in order, blocks on the host side
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) w += __shfl_down(w, off, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) block_w[blockIdx.x] = ((wsum[0] + wsum[1]) + wsum[2]) + wsum[3];
}
__global__ void k_reg_sum_blocks(const double* __restrict__ block_w, u32 nb, u32 n_res, double* __restrict__ scale_out) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    double s = 0.0;
    for (u32 b = 0; b < nb; ++b) s += block_w[b];
    scale_out[0] = s > 0.0 ? static_cast<double>(n_res) / s : 0.0;
    scale_out[1] = s;
  }
}
__global__ void __launch_bounds__(256) k_reg_scale(double* __restrict__ a, size_t n, const double* __restrict__ scale) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) a[i] *= scale[0];
}

// ---- fused normal equations ----------------------------------------------------------------------------
typedef double double4_t __attribute__((ext_vector_type(4)));

// (bx of nb: this workgroup's place among the workgroups of ITS constraint -- the whole grid for one constraint, a row of it in a batch)
__device__ __forceinline__ void reg_normal_eq_body(const ReadingView& L, const RelPose& P, const float* __restrict__ pts, const u32* __restrict__ sample_idx, u32 n_res,
                                                   double no_corr_cost, double* partials /*[nb][256]*/, double* __restrict__ out /*[256]*/, u32* ticket, u32 bx, u32 nb) {
  __shared__ double X[kRegThreads / 64][16][68];  // per wave: 16 components x 64 points (rows padded to 68: conflict-free writes, 2-way reads)
  __shared__ double tile[kRegThreads / 64][256];
  const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  double4_t acc = {0.0, 0.0, 0.0, 0.0};
  const u32 stride = nb * blockDim.x;
  const u32 n_iter = (n_res + stride - 1) / stride;
  for (u32 it = 0; it < n_iter; ++it) {
    const u32 i = it * stride + bx * blockDim.x + threadIdx.x;
    double x[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = 0.0;
    if (i < n_res) {
      const u32 pi = sample_idx ? sample_idx[i] : i;
      const PointResult o = reg_point(L, P, pts + 5ull * pi, no_corr_cost);
#pragma unroll
      for (int k = 0; k < 8; ++k) x[k] = o.J[k];
      x[8] = o.r;
      x[9] = 1.0;
      x[10] = static_cast<double>(o.w);
      x[11] = o.has ? 1.0 : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) X[wave][k][lane] = x[k];
    __builtin_amdgcn_wave_barrier();
    // sum_p x_p x_p^T over this wave's 64 points: 16 MFMA steps of K = 4 points each.
    // f64 16x16x4 operand map: lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15].
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const double a = X[wave][lane & 15u][4 * t + (lane >> 4)];
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
  }
  // f64 C/D map: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
  for (int r = 0; r < 4; ++r) tile[wave][((lane >> 4) + 4 * r) * 16 + (lane & 15u)] = acc[r];
  __syncthreads();
  partials[static_cast<size_t>(bx) * kPartial + threadIdx.x] =
      ((tile[0][threadIdx.x] + tile[1][threadIdx.x]) + tile[2][threadIdx.x]) + tile[3][threadIdx.x];
  // the workgroup that finishes last sums the partials of all workgroups in workgroup order: same bits every run, and no
  // second launch (the one-workgroup reduction kernel took longer than this kernel)
  __shared__ u32 last;
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    const u32 t = atomicAdd(ticket, 1u);
    last = (t == nb - 1u) ? 1u : 0u;
    if (last) *ticket = 0u;  // ready for the next launch on this stream
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;  // four independent chains; the order of the additions stays fixed
  u32 bq = 0;
  for (; bq + 4 <= nb; bq += 4) {
    s0 += __builtin_nontemporal_load(&partials[static_cast<size_t>(bq) * kPartial + threadIdx.x]);
    s1 += __builtin_nontemporal_load(&partials[static_cast<size_t>(bq + 1) * kPartial + threadIdx.x]);
    s2 += __builtin_nontemporal_load(&partials[static_cast<size_t>(bq + 2) * kPartial + threadIdx.x]);
    s3 += __builtin_nontemporal_load(&partials[static_cast<size_t>(bq + 3) * kPartial + threadIdx.x]);
  }
  for (; bq < nb; ++bq) s0 += __builtin_nontemporal_load(&partials[static_cast<size_t>(bq) * kPartial + threadIdx.x]);
  out[threadIdx.x] = (s0 + s1) + (s2 + s3);
}

__global__ void __launch_bounds__(kRegThreads) k_reg_normal_eq(ReadingView L, RelPose P, const float* __restrict__ pts, const u32* __restrict__ sample_idx,
                                                               u32 n_res, double no_corr_cost, double* partials /*[grid][256]*/, double* __restrict__ out /*[256]*/,
                                                               u32* ticket) {
  reg_normal_eq_body(L, P, pts, sample_idx, n_res, no_corr_cost, partials, out, ticket, blockIdx.x, gridDim.x);
}
// ONE launch for all the registration constraints of a pose-graph evaluation (configs[4] evaluates 28 of them per solver iteration,
// each a 15 us launch on a dozen workgroups of its own): blockIdx.y = constraint, its poses / pointers in a table in device memory,
// one ticketed reduction per constraint.
struct RegBatchEntry {
  ReadingView L;
  RelPose P;
  const float* pts;
  const u32* sample_idx;
  double no_corr_cost;
  u32 n_res, nb;  // residuals, workgroups of this constraint (<= gridDim.x)
};
__global__ void __launch_bounds__(kRegThreads) k_reg_normal_eq_batch(const RegBatchEntry* __restrict__ table, double* partials /*[n][gridDim.x][256]*/,
                                                                     double* __restrict__ out /*[n][256]*/, u32* tickets /*[n]*/) {
  const u32 c = blockIdx.y;
  const u32 nb = table[c].nb;
  if (blockIdx.x >= nb) return;
  const RegBatchEntry E = table[c];
  reg_normal_eq_body(E.L, E.P, E.pts, E.sample_idx, E.n_res, E.no_corr_cost, partials + static_cast<size_t>(c) * gridDim.x * kPartial, out + static_cast<size_t>(c) * kPartial,
                     tickets + c, blockIdx.x, nb);
}

// ---- WeightedSampler<RegistrationPoint>::getRandomItem, reproducible and on the device ---------------------------------
// voxgraph draws the registration points of an evaluation with replacement, proportionally to their weight (uniform in
// [0, sum w), upper_bound on the cumulative weights, unseeded std::mt19937).  Here: weights in fixed point (2^-20 units ->
// exact integer prefix sums whatever the order they are added in), draw i = splitmix64(seed, i) scaled to [0, total).
__device__ __forceinline__ u64 weight_fixed(float w) {
  if (!(w > 0.0f)) return 0;
  const double s = static_cast<double>(w) * 1048576.0;
  return s >= 1.8e19 ? ~0ull : static_cast<u64>(s);
}
// one workgroup: inclusive prefix sums of n fixed-point weights (a point set is built once per submap; n ~ 1e4..1e6)
__global__ void __launch_bounds__(1024) k_weight_cumsum(const float* __restrict__ pts, u64 n, u64* __restrict__ cum, u64* __restrict__ total) {
  __shared__ u64 wsum[16];
  __shared__ u64 carry_s;
  const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (u64 base = 0; base < n; base += 1024) {
    const u64 i = base + threadIdx.x;
    u64 v = (i < n) ? weight_fixed(pts[5 * i + 4]) : 0ull;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const u64 o = __shfl_up(v, off, 64);
      if (lane >= static_cast<u32>(off)) v += o;
    }
    if (lane == 63) wsum[wave] = v;
    __syncthreads();
    u64 below = carry_s;
    for (u32 w = 0; w < wave; ++w) below += wsum[w];
    if (i < n) cum[i] = below + v;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = below + v;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry_s;
}
__device__ __forceinline__ u64 splitmix64(u64 x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__global__ void __launch_bounds__(256) k_draw_samples(const u64* __restrict__ cum, u64 n, u64 total, u64 seed, u32* __restrict__ out, u64 n_res) {
  const u64 i = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n_res) return;
  if (total == 0) {
    out[i] = 0;
    return;
  }
  const u64 r = splitmix64(seed * 0x9E3779B97F4A7C15ull + i);
  const u64 u = __umul64hi(r, total);  // uniform in [0, total)
  u64 lo = 0, hi = n;                  // upper_bound: first index with cum > u
  while (lo < hi) {
    const u64 mid = (lo + hi) >> 1;
    if (cum[mid] > u)
      hi = mid;
    else
      lo = mid + 1;
  }
  out[i] = static_cast<u32>(lo);
}

// ---- host ----------------------------------------------------------------------------------------------------
struct cox_reg {
  const cox_regpoints* ref = nullptr;
  const cox_layer* reading = nullptr;
  double no_corr_cost = 0.0;
  hipStream_t stream = nullptr;
  u32* d_idx = nullptr;
  u64 idx_cap = 0;
  u32* d_stored = nullptr;  // cox_reg_set_samples: indices kept on the GPU, used when a call passes sample_idx == NULL
  u64 stored_cap = 0, stored_samples = 0;
  bool has_stored = false;
  bool pending = false;    // a cox_reg_normal_eq_begin without its finish
  u64 pending_n = 0;
  double *d_res = nullptr, *d_jf = nullptr, *d_jr = nullptr;
  u64 res_cap = 0, jf_cap = 0, jr_cap = 0;
  u32* d_ticket = nullptr;    // last-workgroup ticket of the fused reduction (zero between launches)
  double* d_small = nullptr;  // block sums / partials / results
  u64 small_cap = 0;
  double* h_small = nullptr;  // pinned, 256 + 2 doubles
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  double ms = 0.0;
  uint64_t launches = 0;
  bool stream_dirty = false;  // work enqueued on this handle's stream that a batch on another handle's stream has to wait for (sample draws / uploads)
  // scratch of cox_reg_normal_eq_batch (owned by the batch's first handle)
  RegBatchEntry* d_table = nullptr;
  RegBatchEntry* h_table = nullptr;  // pinned
  double* d_bpart = nullptr;
  double* d_bout = nullptr;
  double* h_bout = nullptr;  // pinned
  u32* d_btickets = nullptr;
  u64 batch_cap = 0, bpart_cap = 0;
};

template <typename T>
static int dev_grow(T** p, u64* cap, u64 need) {
  if (need <= *cap) return COX_OK;
  if (*p) (void)hipFree(*p);
  *p = nullptr;
  *cap = 0;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(p), need * sizeof(T));
  if (e != hipSuccess) return (e == hipErrorOutOfMemory) ? COX_ERR_OUT_OF_MEMORY : COX_ERR_NO_DEVICE;
  *cap = need;
  return COX_OK;
}
#define COX_TRY(expr)              \
  do {                             \
    int st_ = (expr);              \
    if (st_ != COX_OK) return st_; \
  } while (0)

static ReadingView reading_view(const cox_layer* L) {
  return ReadingView{L->voxels, L->ht_keys, L->ht_vals, L->ht_cap - 1, L->voxel_size, L->voxel_size_inv, L->block_size, L->block_size_inv};
}

static int stage_samples(cox_reg* G, const uint32_t* sample_idx, uint64_t n_res, const u32** d_idx_out) {
  *d_idx_out = nullptr;
  if (!sample_idx && G->has_stored) {
    if (n_res != G->stored_samples) return COX_ERR_INVALID_ARG;
    *d_idx_out = G->d_stored;
    return COX_OK;
  }
  if (!sample_idx) return (n_res == G->ref->n) ? COX_OK : COX_ERR_INVALID_ARG;
  for (uint64_t i = 0; i < n_res; ++i)
    if (sample_idx[i] >= G->ref->n) return COX_ERR_INVALID_ARG;
  COX_TRY(dev_grow(&G->d_idx, &G->idx_cap, n_res));
  COX_HIP(hipMemcpyAsync(G->d_idx, sample_idx, sizeof(u32) * n_res, hipMemcpyHostToDevice, G->stream));
  *d_idx_out = G->d_idx;
  return COX_OK;
}

extern "C" {

int cox_regpoints_create(int device, const float* xyz_dist_weight, uint64_t n, cox_regpoints_t** out) {
  COX_ENTRY();
  if (!out || (n && !xyz_dist_weight) || n > 0x7FFFFFFFull) return COX_ERR_INVALID_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return COX_ERR_NO_DEVICE;
  COX_HIP(hipSetDevice(device));
  cox_regpoints* R = new (std::nothrow) cox_regpoints();
  if (!R) return COX_ERR_OUT_OF_MEMORY;
  R->device = device;
  R->n = n;
  if (n) {
    if (hipMalloc(reinterpret_cast<void**>(&R->pts), sizeof(float) * 5 * n) != hipSuccess) {
      delete R;
      return COX_ERR_OUT_OF_MEMORY;
    }
    if (hipMemcpy(R->pts, xyz_dist_weight, sizeof(float) * 5 * n, hipMemcpyHostToDevice) != hipSuccess) {
      (void)hipFree(R->pts);
      delete R;
      return COX_ERR_NO_DEVICE;
    }
  }
  *out = R;
  return COX_OK;
}
// registration points that already sit in HBM (the submap exchange: another rank's set arrives by all-gather / peer copy)
int cox_regpoints_create_dev(int device, const float* xyz_dist_weight_dev, uint64_t n, cox_regpoints_t** out) {
  COX_ENTRY();
  if (!out || (n && !xyz_dist_weight_dev) || n > 0x7FFFFFFFull) return COX_ERR_INVALID_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return COX_ERR_NO_DEVICE;
  COX_HIP(hipSetDevice(device));
  COX_HIP(hipDeviceSynchronize());  // whatever produced the buffer is done after this
  cox_regpoints* R = new (std::nothrow) cox_regpoints();
  if (!R) return COX_ERR_OUT_OF_MEMORY;
  R->device = device;
  R->n = n;
  if (n) {
    if (hipMalloc(reinterpret_cast<void**>(&R->pts), sizeof(float) * 5 * n) != hipSuccess) {
      delete R;
      return COX_ERR_OUT_OF_MEMORY;
    }
    if (hipMemcpy(R->pts, xyz_dist_weight_dev, sizeof(float) * 5 * n, hipMemcpyDeviceToDevice) != hipSuccess) {
      (void)hipFree(R->pts);
      delete R;
      return COX_ERR_NO_DEVICE;
    }
  }
  *out = R;
  return COX_OK;
}
// device pointer to the n * 5 floats (valid until the set is destroyed): what a rank hands to the all-gather
int cox_regpoints_data_dev(const cox_regpoints_t* R, const float** xyz_dist_weight_dev, uint64_t* n) {
  if (!R || !xyz_dist_weight_dev) return COX_ERR_INVALID_ARG;
  *xyz_dist_weight_dev = R->pts;
  if (n) *n = R->n;
  return COX_OK;
}
// host copy of a set (tests, hand-over to a mesher): n * 5 floats
int cox_regpoints_download(const cox_regpoints_t* R, float* xyz_dist_weight, uint64_t cap, uint64_t* n) {
  COX_ENTRY();
  if (!R) return COX_ERR_INVALID_ARG;
  if (n) *n = R->n;
  if (!xyz_dist_weight) return COX_OK;
  if (cap < R->n) return COX_ERR_BUFFER_TOO_SMALL;
  COX_HIP(hipSetDevice(R->device));
  if (R->n) COX_HIP(hipMemcpy(xyz_dist_weight, R->pts, sizeof(float) * 5 * R->n, hipMemcpyDeviceToHost));
  return COX_OK;
}
// the point set on another GPU of the node (hipMemcpyPeer: xGMI; same device: a plain copy)
int cox_regpoints_clone_to_device(const cox_regpoints_t* src, int dst_device, cox_regpoints_t** out) {
  COX_ENTRY();
  if (!src || !out) return COX_ERR_INVALID_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || dst_device < 0 || dst_device >= ndev) return COX_ERR_NO_DEVICE;
  COX_HIP(hipSetDevice(dst_device));
  cox_regpoints* R = new (std::nothrow) cox_regpoints();
  if (!R) return COX_ERR_OUT_OF_MEMORY;
  R->device = dst_device;
  R->n = src->n;
  if (src->n) {
    if (hipMalloc(reinterpret_cast<void**>(&R->pts), sizeof(float) * 5 * src->n) != hipSuccess) {
      delete R;
      return COX_ERR_OUT_OF_MEMORY;
    }
    if (hipMemcpyPeer(R->pts, dst_device, src->pts, src->device, sizeof(float) * 5 * src->n) != hipSuccess) {
      (void)hipFree(R->pts);
      delete R;
      return COX_ERR_NO_DEVICE;
    }
  }
  *out = R;
  return COX_OK;
}
int cox_regpoints_size(const cox_regpoints_t* R, uint64_t* n) {
  if (!R || !n) return COX_ERR_INVALID_ARG;
  *n = R->n;
  return COX_OK;
}
void cox_regpoints_destroy(cox_regpoints_t* R) {
  if (!R) return;
  (void)hipSetDevice(R->device);
  if (R->pts) (void)hipFree(R->pts);
  if (R->cum) (void)hipFree(R->cum);
  delete R;
}

int cox_reg_create(const cox_regpoints_t* reference, const cox_layer_t* reading, const cox_reg_config* cfg, cox_reg_t** out) {
  COX_ENTRY();
  if (!reference || !reading || !out) return COX_ERR_INVALID_ARG;
  const cox_layer* L = reinterpret_cast<const cox_layer*>(reading);
  if (L->device != reference->device) return COX_ERR_INVALID_ARG;  // hand the submap over first: cox_layer_clone_to_device / cox_regpoints_clone_to_device
  COX_HIP(hipSetDevice(L->device));
  cox_reg* G = new (std::nothrow) cox_reg();
  if (!G) return COX_ERR_OUT_OF_MEMORY;
  G->ref = reference;
  G->reading = L;
  G->no_corr_cost = cfg ? cfg->no_correspondence_cost : 0.0;
  bool ok = hipStreamCreateWithFlags(&G->stream, hipStreamNonBlocking) == hipSuccess;
  ok = ok && hipHostMalloc(reinterpret_cast<void**>(&G->h_small), sizeof(double) * 258, hipHostMallocDefault) == hipSuccess;
  ok = ok && hipEventCreate(&G->ev0) == hipSuccess && hipEventCreate(&G->ev1) == hipSuccess;
  ok = ok && hipMalloc(reinterpret_cast<void**>(&G->d_ticket), sizeof(u32)) == hipSuccess && hipMemset(G->d_ticket, 0, sizeof(u32)) == hipSuccess &&
       hipDeviceSynchronize() == hipSuccess;
  if (!ok) {
    cox_reg_destroy(G);
    return COX_ERR_NO_DEVICE;
  }
  *out = G;
  return COX_OK;
}
void cox_reg_destroy(cox_reg_t* G) {
  if (!G) return;
  (void)hipSetDevice(G->reading->device);
  if (G->stream) (void)hipStreamSynchronize(G->stream);
  void* ptrs[] = {G->d_idx, G->d_stored, G->d_res, G->d_jf, G->d_jr, G->d_small, G->d_ticket, G->d_table, G->d_bpart, G->d_bout, G->d_btickets};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (G->h_small) (void)hipHostFree(G->h_small);
  if (G->h_table) (void)hipHostFree(G->h_table);
  if (G->h_bout) (void)hipHostFree(G->h_bout);
  if (G->ev0) (void)hipEventDestroy(G->ev0);
  if (G->ev1) (void)hipEventDestroy(G->ev1);
  if (G->stream) (void)hipStreamDestroy(G->stream);
  delete G;
}

int cox_reg_evaluate(cox_reg_t* G, const double pose_ref[4], const double pose_read[4], const uint32_t* sample_idx, uint64_t n_res, double* residuals,
                     double* jac_ref, double* jac_read) {
  COX_ENTRY();
  if (!G || !pose_ref || !pose_read || n_res > 0x7FFFFFFFull || G->pending) return COX_ERR_INVALID_ARG;
  COX_HIP(hipSetDevice(G->reading->device));
  if (n_res == 0) return sample_idx || G->has_stored || G->ref->n == 0 ? COX_OK : COX_ERR_INVALID_ARG;
  const u32* d_idx;
  COX_TRY(stage_samples(G, sample_idx, n_res, &d_idx));
  const u32 n = static_cast<u32>(n_res);
  const u32 nb = (n + kRegThreads - 1) / kRegThreads;
  COX_TRY(dev_grow(&G->d_res, &G->res_cap, n_res));
  COX_TRY(dev_grow(&G->d_jf, &G->jf_cap, 4 * n_res));
  COX_TRY(dev_grow(&G->d_jr, &G->jr_cap, 4 * n_res));
  COX_TRY(dev_grow(&G->d_small, &G->small_cap, static_cast<u64>(nb) + 2));
  const RelPose P = make_rel_pose(pose_ref, pose_read);
  hipStream_t s = G->stream;
  cox_layer_wait_writes(G->reading, s);  // frames still in flight on the reading layer
  COX_HIP(hipEventRecord(G->ev0, s));
  hipLaunchKernelGGL(k_reg_evaluate, dim3(nb), dim3(kRegThreads), 0, s, reading_view(G->reading), P, G->ref->pts, d_idx, n, G->no_corr_cost,
                     residuals ? G->d_res : nullptr, jac_ref ? G->d_jf : nullptr, jac_read ? G->d_jr : nullptr, G->d_small + 2);
  hipLaunchKernelGGL(k_reg_sum_blocks, dim3(1), dim3(64), 0, s, G->d_small + 2, nb, n, G->d_small);
  if (residuals) hipLaunchKernelGGL(k_reg_scale, dim3((n + 255) / 256), dim3(256), 0, s, G->d_res, static_cast<size_t>(n), G->d_small);
  if (jac_ref) hipLaunchKernelGGL(k_reg_scale, dim3((4 * n + 255) / 256), dim3(256), 0, s, G->d_jf, static_cast<size_t>(4) * n, G->d_small);
  if (jac_read) hipLaunchKernelGGL(k_reg_scale, dim3((4 * n + 255) / 256), dim3(256), 0, s, G->d_jr, static_cast<size_t>(4) * n, G->d_small);
  COX_HIP(hipEventRecord(G->ev1, s));
  if (residuals) COX_HIP(hipMemcpyAsync(residuals, G->d_res, sizeof(double) * n_res, hipMemcpyDeviceToHost, s));
  if (jac_ref) COX_HIP(hipMemcpyAsync(jac_ref, G->d_jf, sizeof(double) * 4 * n_res, hipMemcpyDeviceToHost, s));
  if (jac_read) COX_HIP(hipMemcpyAsync(jac_read, G->d_jr, sizeof(double) * 4 * n_res, hipMemcpyDeviceToHost, s));
  COX_HIP(hipStreamSynchronize(s));
  COX_HIP(hipGetLastError());
  float ms = 0.0f;
  if (hipEventElapsedTime(&ms, G->ev0, G->ev1) == hipSuccess) {
    G->ms += ms;
    G->launches += 1;
  }
  return COX_OK;
}

int cox_reg_set_samples(cox_reg_t* G, const uint32_t* sample_idx, uint64_t n_res) {
  COX_ENTRY();
  if (!G || n_res > 0x7FFFFFFFull || G->pending) return COX_ERR_INVALID_ARG;
  COX_HIP(hipSetDevice(G->reading->device));
  G->has_stored = false;
  G->stored_samples = 0;
  if (!sample_idx) return COX_OK;  // back to "all points in order"
  for (uint64_t i = 0; i < n_res; ++i)
    if (sample_idx[i] >= G->ref->n) return COX_ERR_INVALID_ARG;
  COX_TRY(dev_grow(&G->d_stored, &G->stored_cap, std::max<uint64_t>(n_res, 1)));
  COX_HIP(hipMemcpy(G->d_stored, sample_idx, sizeof(u32) * n_res, hipMemcpyHostToDevice));
  G->has_stored = true;
  G->stored_samples = n_res;
  return COX_OK;
}

int cox_reg_draw_samples(cox_reg_t* G, uint64_t n_res, uint64_t seed) {
  COX_ENTRY();
  if (!G || n_res > 0x7FFFFFFFull || G->pending) return COX_ERR_INVALID_ARG;
  cox_regpoints* R = const_cast<cox_regpoints*>(G->ref);
  COX_HIP(hipSetDevice(R->device));
  G->has_stored = false;
  G->stored_samples = 0;
  if (R->n == 0) return n_res == 0 ? COX_OK : COX_ERR_INVALID_ARG;
  hipStream_t s = G->stream;
  if (!R->cum) {
    u64* cum = nullptr;
    COX_HIP(hipMalloc(reinterpret_cast<void**>(&cum), sizeof(u64) * (R->n + 1)));
    hipLaunchKernelGGL(k_weight_cumsum, dim3(1), dim3(1024), 0, s, R->pts, R->n, cum, cum + R->n);
    COX_HIP(hipMemcpyAsync(&R->cum_total, cum + R->n, sizeof(u64), hipMemcpyDeviceToHost, s));
    COX_HIP(hipStreamSynchronize(s));
    R->cum = cum;
  }
  COX_TRY(dev_grow(&G->d_stored, &G->stored_cap, std::max<uint64_t>(n_res, 1)));
  if (n_res) hipLaunchKernelGGL(k_draw_samples, dim3(static_cast<u32>((n_res + 255) / 256)), dim3(256), 0, s, R->cum, R->n, R->cum_total, seed, G->d_stored, n_res);
  COX_HIP(hipGetLastError());
  G->stream_dirty = true;  // (a batch that runs on another handle's stream waits for the draw)
  G->has_stored = true;
  G->stored_samples = n_res;
  return COX_OK;
}
int cox_reg_get_samples(cox_reg_t* G, uint32_t* sample_idx, uint64_t cap, uint64_t* n_res) {
  COX_ENTRY();
  if (!G || !n_res) return COX_ERR_INVALID_ARG;
  *n_res = G->has_stored ? G->stored_samples : 0;
  if (!sample_idx || !G->has_stored) return COX_OK;
  if (cap < G->stored_samples) return COX_ERR_BUFFER_TOO_SMALL;
  COX_HIP(hipSetDevice(G->reading->device));
  COX_HIP(hipStreamSynchronize(G->stream));
  if (G->stored_samples) COX_HIP(hipMemcpy(sample_idx, G->d_stored, sizeof(u32) * G->stored_samples, hipMemcpyDeviceToHost));
  return COX_OK;
}

int cox_reg_normal_eq_begin(cox_reg_t* G, const double pose_ref[4], const double pose_read[4], const uint32_t* sample_idx, uint64_t n_res) {
  COX_ENTRY();
  if (!G || !pose_ref || !pose_read || n_res > 0x7FFFFFFFull || G->pending) return COX_ERR_INVALID_ARG;
  COX_HIP(hipSetDevice(G->reading->device));
  G->pending_n = n_res;
  if (n_res == 0) {
    if (!(sample_idx || G->has_stored || G->ref->n == 0)) return COX_ERR_INVALID_ARG;
    G->pending = true;
    return COX_OK;
  }
  const u32* d_idx;
  COX_TRY(stage_samples(G, sample_idx, n_res, &d_idx));
  const u32 n = static_cast<u32>(n_res);
  // one workgroup per 256 residuals up to 1024 workgroups (then they stride); the last one to finish reduces
  const u32 nb = std::min<u32>(1024, (n + kRegThreads - 1) / kRegThreads);
  COX_TRY(dev_grow(&G->d_small, &G->small_cap, static_cast<u64>(nb) * kPartial + kPartial));
  const RelPose P = make_rel_pose(pose_ref, pose_read);
  hipStream_t s = G->stream;
  cox_layer_wait_writes(G->reading, s);  // frames still in flight on the reading layer
  COX_HIP(hipEventRecord(G->ev0, s));
  hipLaunchKernelGGL(k_reg_normal_eq, dim3(nb), dim3(kRegThreads), 0, s, reading_view(G->reading), P, G->ref->pts, d_idx, n, G->no_corr_cost,
                     G->d_small + kPartial, G->d_small, G->d_ticket);
  COX_HIP(hipEventRecord(G->ev1, s));
  COX_HIP(hipMemcpyAsync(G->h_small, G->d_small, sizeof(double) * kPartial, hipMemcpyDeviceToHost, s));
  G->pending = true;
  return COX_OK;
}

int cox_reg_normal_eq_finish(cox_reg_t* G, double H[64], double b[8], double* cost, uint64_t* n_corr) {
  if (!G || !H || !b || !cost || !G->pending) return COX_ERR_INVALID_ARG;
  G->pending = false;
  for (int i = 0; i < 64; ++i) H[i] = 0.0;
  for (int i = 0; i < 8; ++i) b[i] = 0.0;
  *cost = 0.0;
  if (n_corr) *n_corr = 0;
  if (G->pending_n == 0) return COX_OK;
  COX_HIP(hipSetDevice(G->reading->device));
  COX_HIP(hipStreamSynchronize(G->stream));
  COX_HIP(hipGetLastError());
  float ms = 0.0f;
  if (hipEventElapsedTime(&ms, G->ev0, G->ev1) == hipSuccess) {
    G->ms += ms;
    G->launches += 1;
  }
  const double* D = G->h_small;  // D[row * 16 + col] = sum_p x[row] x[col]
  const double sum_w = D[9 * 16 + 10];
  const double scale = sum_w > 0.0 ? static_cast<double>(G->pending_n) / sum_w : 0.0;
  const double s2 = scale * scale;
  for (int r = 0; r < 8; ++r) {
    for (int c = 0; c < 8; ++c) H[8 * r + c] = D[r * 16 + c] * s2;
    b[r] = D[r * 16 + 8] * s2;
  }
  *cost = 0.5 * D[8 * 16 + 8] * s2;
  if (n_corr) *n_corr = static_cast<uint64_t>(D[9 * 16 + 11] + 0.5);
  return COX_OK;
}

int cox_reg_normal_eq(cox_reg_t* G, const double pose_ref[4], const double pose_read[4], const uint32_t* sample_idx, uint64_t n_res, double H[64],
                      double b[8], double* cost, uint64_t* n_corr) {
  if (!G || !pose_ref || !pose_read || !H || !b || !cost) return COX_ERR_INVALID_ARG;
  COX_TRY(cox_reg_normal_eq_begin(G, pose_ref, pose_read, sample_idx, n_res));
  return cox_reg_normal_eq_finish(G, H, b, cost, n_corr);
}

int cox_reg_normal_eq_batch(cox_reg_t* const* regs, uint64_t n, const double* poses_ref, const double* poses_read, double* H, double* b, double* cost,
                            uint64_t* n_corr) {
  COX_ENTRY();
  if (!regs || n == 0 || n > 65535 || !poses_ref || !poses_read || !H || !b || !cost) return COX_ERR_INVALID_ARG;
  cox_reg* G0 = regs[0];
  if (!G0) return COX_ERR_INVALID_ARG;
  const int device = G0->reading->device;
  for (uint64_t c = 0; c < n; ++c) {
    const cox_reg* G = regs[c];
    if (!G || G->pending || G->reading->device != device || G->ref->device != device) return COX_ERR_INVALID_ARG;  // one GPU per batch
    const uint64_t nr = G->has_stored ? G->stored_samples : G->ref->n;
    if (nr > 0x7FFFFFFFull) return COX_ERR_INVALID_ARG;
  }
  COX_HIP(hipSetDevice(device));
  hipStream_t s = G0->stream;
  if (G0->batch_cap < n) {
    COX_HIP(hipStreamSynchronize(s));
    if (G0->h_table) (void)hipHostFree(G0->h_table);
    if (G0->h_bout) (void)hipHostFree(G0->h_bout);
    for (void* p : {static_cast<void*>(G0->d_table), static_cast<void*>(G0->d_bout), static_cast<void*>(G0->d_btickets)})
      if (p) (void)hipFree(p);
    G0->h_table = nullptr, G0->h_bout = nullptr, G0->d_table = nullptr, G0->d_bout = nullptr, G0->d_btickets = nullptr, G0->batch_cap = 0;
    const u64 cap = std::max<u64>(n, 32);
    if (hipHostMalloc(reinterpret_cast<void**>(&G0->h_table), sizeof(RegBatchEntry) * cap, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&G0->h_bout), sizeof(double) * kPartial * cap, hipHostMallocDefault) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&G0->d_table), sizeof(RegBatchEntry) * cap) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&G0->d_bout), sizeof(double) * kPartial * cap) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&G0->d_btickets), sizeof(u32) * cap) != hipSuccess) {
      (void)hipGetLastError();
      return COX_ERR_OUT_OF_MEMORY;
    }
    COX_HIP(hipMemset(G0->d_btickets, 0, sizeof(u32) * cap));  // every launch leaves them zero again
    COX_HIP(hipDeviceSynchronize());
    G0->batch_cap = cap;
  }
  u32 nb_max = 1;
  for (uint64_t c = 0; c < n; ++c) {
    cox_reg* G = regs[c];
    const u32 nr = static_cast<u32>(G->has_stored ? G->stored_samples : G->ref->n);
    const u32 nb = std::max<u32>(1, std::min<u32>(1024, (nr + kRegThreads - 1) / kRegThreads));  // (the same partition as a call of its own: same bits)
    nb_max = std::max(nb_max, nb);
    RegBatchEntry& E = G0->h_table[c];
    E.L = reading_view(G->reading);
    E.P = make_rel_pose(poses_ref + 4 * c, poses_read + 4 * c);
    E.pts = G->ref->pts;
    E.sample_idx = G->has_stored ? G->d_stored : nullptr;
    E.no_corr_cost = G->no_corr_cost;
    E.n_res = nr;
    E.nb = nb;
    if (G != G0 && G->stream_dirty) {  // its sample draw runs on its own stream
      COX_HIP(hipStreamSynchronize(G->stream));
      G->stream_dirty = false;
    }
    cox_layer_wait_writes(G->reading, s);  // frames still in flight on the reading layer
  }
  G0->stream_dirty = false;
  COX_TRY(dev_grow(&G0->d_bpart, &G0->bpart_cap, static_cast<u64>(n) * nb_max * kPartial));
  COX_HIP(hipMemcpyAsync(G0->d_table, G0->h_table, sizeof(RegBatchEntry) * n, hipMemcpyHostToDevice, s));
  COX_HIP(hipEventRecord(G0->ev0, s));
  hipLaunchKernelGGL(k_reg_normal_eq_batch, dim3(nb_max, static_cast<u32>(n)), dim3(kRegThreads), 0, s, G0->d_table, G0->d_bpart, G0->d_bout, G0->d_btickets);
  COX_HIP(hipEventRecord(G0->ev1, s));
  COX_HIP(hipMemcpyAsync(G0->h_bout, G0->d_bout, sizeof(double) * kPartial * n, hipMemcpyDeviceToHost, s));
  COX_HIP(hipStreamSynchronize(s));
  COX_HIP(hipGetLastError());
  float ms = 0.0f;
  if (hipEventElapsedTime(&ms, G0->ev0, G0->ev1) == hipSuccess) {
    G0->ms += ms;
    G0->launches += 1;
  }
  for (uint64_t c = 0; c < n; ++c) {
    const double* D = G0->h_bout + kPartial * c;  // D[row * 16 + col] = sum_p x[row] x[col]
    const double nr = static_cast<double>(G0->h_table[c].n_res);
    const double sum_w = D[9 * 16 + 10];
    const double scale = (sum_w > 0.0 && nr > 0.0) ? nr / sum_w : 0.0;
    const double s2 = scale * scale;
    for (int r = 0; r < 8; ++r) {
      for (int q = 0; q < 8; ++q) H[64 * c + 8 * r + q] = nr > 0.0 ? D[r * 16 + q] * s2 : 0.0;
      b[8 * c + r] = nr > 0.0 ? D[r * 16 + 8] * s2 : 0.0;
    }
    cost[c] = nr > 0.0 ? 0.5 * D[8 * 16 + 8] * s2 : 0.0;
    if (n_corr) n_corr[c] = nr > 0.0 ? static_cast<uint64_t>(D[9 * 16 + 11] + 0.5) : 0;
  }
  return COX_OK;
}

int cox_reg_kernel_time(cox_reg_t* G, double* ms, uint64_t* launches, int reset) {
  COX_ENTRY();
  if (!G) return COX_ERR_INVALID_ARG;
  if (ms) *ms = G->ms;
  if (launches) *launches = G->launches;
  if (reset) {
    G->ms = 0.0;
    G->launches = 0;
  }
  return COX_OK;
}

}  // extern "C"
