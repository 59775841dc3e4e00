// Device-side building blocks of the TSDF engine (gfx950 only).
//
// Index-deciding float math mirrors voxblox's operation order exactly (see DESIGN.md "numerics
// contract"); this file must be compiled with -ffp-contract=off and IEEE-correct fp32 divide/sqrt
// (hipcc default, passed explicitly by the build) so that floor()/DDA decisions agree bit for bit
// with a host evaluation of the same formulas.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cox {

typedef uint32_t u32;
typedef uint64_t u64;

constexpr float kEps = 1e-6f;          // voxblox kEpsilon / kCoordinateEpsilon / kFloatEpsilon
constexpr int kVps = 16;               // voxels per side (the only value the reference uses)
constexpr int kVoxelsPerBlock = 4096;  // 16^3
constexpr int kWordsPerVoxel = 3;      // wire layout: distance bits, weight bits, a|b<<8|g<<16|r<<24
constexpr int kIdxBias = 1 << 20;      // packed keys hold indices in [-2^20, 2^20)
constexpr u64 kEmptyKey = ~0ull;
constexpr u32 kInvalid = 0xFFFFFFFFu;

// ---- frame parameters (one struct by value per kernel launch) ------------------------------
struct FrameParams {
  float qw, qx, qy, qz, tx, ty, tz;  // T_G_C
  float voxel_size, voxel_size_inv;
  float trunc, max_weight, min_ray, max_ray;
  float sparsity_factor, start_subsampling_inv;  // start_subsampling_inv = factor * voxel_size_inv
  u32 n_points;
  u32 frame_id;
  int use_const_weight, allow_clear, carving, use_dropoff, use_sparsity, anti_grazing, freespace;
  int cast_from_origin;
  // inputs of the frame (device pointers) -- part of the parameter block so that kernel arguments stay the same
  // from frame to frame and a stage can be replayed as a HIP graph
  const float* xyz;
  const uint8_t* rgba;
  u32 np2;  // power of two > n_points (bundling sort key layout)
  float sat1;  // saturating_update's threshold for an update weight of 1, rounded up: sdf >= sat1 saturates for EVERY weight >= 1
};

struct F3 {
  float x, y, z;
};
__device__ __forceinline__ F3 f3(float x, float y, float z) { return F3{x, y, z}; }
__device__ __forceinline__ F3 operator+(F3 a, F3 b) { return F3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return F3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ F3 operator*(F3 a, float s) { return F3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot3(F3 a, F3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ F3 cross3(F3 a, F3 b) { return F3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ float std_min(float a, float b) { return (b < a) ? b : a; }  // std::min
__device__ __forceinline__ float std_max(float a, float b) { return (a < b) ? b : a; }  // std::max

// Eigen Quaternion::_transformVector + translation (minkindr transform)
__device__ __forceinline__ F3 transform_point(const FrameParams& P, F3 v) {
  const F3 qv{P.qx, P.qy, P.qz};
  F3 uv = cross3(qv, v);
  uv = uv + uv;
  const F3 c = cross3(qv, uv);
  return F3{((v.x + P.qw * uv.x) + c.x) + P.tx, ((v.y + P.qw * uv.y) + c.y) + P.ty, ((v.z + P.qw * uv.z) + c.z) + P.tz};
}

// voxblox MixedThreadSafeIndex::getNextIndexImpl
__device__ __forceinline__ u32 mixed_index(u32 seq, u32 n) {
  const u32 groups = n >> 10;
  if ((groups << 10) <= seq) return seq;
  return (seq % groups) * 1024u + seq / groups;
}

// inverse of mixed_index: the sequence number at which point idx is visited
__device__ __forceinline__ u32 mixed_sequence(u32 idx, u32 n) {
  const u32 groups = n >> 10;
  if ((groups << 10) <= idx) return idx;
  return (idx & 1023u) * groups + (idx >> 10);
}

__device__ __forceinline__ int grid_index(float scaled) { return static_cast<int>(floorf(scaled + kEps)); }
__device__ __forceinline__ bool index_in_range(float scaled) {
  // conservative: floor(scaled + eps) must land in [-2^20, 2^20)
  return scaled > -1048575.0f && scaled < 1048575.0f;
}
// voxblox getCenterPointFromGridIndex: (float(idx) + 0.5) * size evaluated in double, narrowed to float.  With |idx| < 2^21
// (index_in_range) idx + 0.5 is a float, the double product of two floats is exact (48 significant bits), and narrowing it is
// ONE rounding to nearest: exactly what the float multiplication does.  Three float operations instead of a trip through
// double per coordinate -- this runs once per (ray, voxel) update.
__device__ __forceinline__ float center_coord(int idx, float size) { return (static_cast<float>(idx) + 0.5f) * size; }

__device__ __forceinline__ u64 pack_key(int x, int y, int z) {
  return static_cast<u64>(static_cast<u32>(x + kIdxBias)) | (static_cast<u64>(static_cast<u32>(y + kIdxBias)) << 21) |
         (static_cast<u64>(static_cast<u32>(z + kIdxBias)) << 42);
}
__host__ __device__ __forceinline__ void unpack_key(u64 k, int* x, int* y, int* z) {
  *x = static_cast<int>(k & 0x1FFFFF) - (1 << 20);
  *y = static_cast<int>((k >> 21) & 0x1FFFFF) - (1 << 20);
  *z = static_cast<int>((k >> 42) & 0x1FFFFF) - (1 << 20);
}
__device__ __forceinline__ u32 hash_key(u64 k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ull;
  k ^= k >> 33;
  return static_cast<u32>(k);
}

// LDS written and then read by the SAME wave needs no hardware barrier (the LDS pipeline is in order per wave), only a
// compiler fence.  (Do not use `volatile` pointers for this: they decay to generic pointers and compile to flat_load /
// flat_store sc0 sc1 instead of ds_read / ds_write.)
__device__ __forceinline__ void wave_lds_handover() {
  __atomic_signal_fence(__ATOMIC_SEQ_CST);
  __builtin_amdgcn_wave_barrier();
  __atomic_signal_fence(__ATOMIC_SEQ_CST);
}

__device__ __forceinline__ u32 lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// ---- open-addressing hash set/map on packed keys -------------------------------------------
// insert-or-find; returns the slot, *fresh = true when this thread claimed it.
// The slot is first read with an agent-scope load (served past the non-coherent L1): thousands of
// rays ask for the same few dozen blocks per frame and same-address CAS round trips serialise.
__device__ __forceinline__ u32 ht_insert(u64* keys, u32 mask, u64 key, bool* fresh) {
  u32 slot = hash_key(key) & mask;
  for (u32 probe = 0; probe <= mask; ++probe) {
    const u64 seen = __hip_atomic_load(&keys[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (seen == key) {
      *fresh = false;
      return slot;
    }
    if (seen != kEmptyKey) {
      slot = (slot + 1) & mask;
      continue;
    }
    const u64 prev = atomicCAS(reinterpret_cast<unsigned long long*>(&keys[slot]), static_cast<unsigned long long>(kEmptyKey),
                               static_cast<unsigned long long>(key));
    if (prev == kEmptyKey) {
      *fresh = true;
      return slot;
    }
    if (prev == key) {
      *fresh = false;
      return slot;
    }
    slot = (slot + 1) & mask;
  }
  *fresh = false;
  return kInvalid;
}
// read-only lookup (keys written by an earlier kernel)
__device__ __forceinline__ u32 ht_find(const u64* keys, u32 mask, u64 key) {
  u32 slot = hash_key(key) & mask;
  for (u32 probe = 0; probe <= mask; ++probe) {
    const u64 k = keys[slot];
    if (k == key) return slot;
    if (k == kEmptyKey) return kInvalid;
    slot = (slot + 1) & mask;
  }
  return kInvalid;
}

// ---- RayCaster (voxblox integrator_utils; quirks kept, see oracle/cox_oracle.hpp) -----------
struct Dda {
  int c[3];
  int sgn[3];
  float t_next[3];
  float t_step[3];
  u32 n_axis[3];  // |end index - start index| per axis
  u32 nsteps;     // indices this ray emits (ray_length_in_steps + 1), 0 for a NaN / out-of-range ray
  bool range_error;
};

__device__ __forceinline__ int signum(float x) { return (x == 0.0f) ? 0 : (x < 0.0f ? -1 : 1); }

__device__ __forceinline__ void dda_setup(Dda& d, const FrameParams& P, F3 point_G, bool is_clearing) {
  const F3 origin{P.tx, P.ty, P.tz};
  const F3 dv = point_G - origin;
  const float z = dot3(dv, dv);
  F3 unit = dv;
  if (z > 0.0f) {
    const float n = sqrtf(z);
    unit = F3{dv.x / n, dv.y / n, dv.z / n};
  }
  F3 ray_start, ray_end;
  if (is_clearing) {
    float len = sqrtf(z);
    len = std_min(std_max(len - P.trunc, 0.0f), P.max_ray);
    ray_end = origin + unit * len;
    ray_start = P.carving ? origin : ray_end;
  } else {
    ray_end = point_G + unit * P.trunc;
    ray_start = P.carving ? origin : (point_G - unit * P.trunc);
  }
  F3 s = ray_start * P.voxel_size_inv;
  F3 e = ray_end * P.voxel_size_inv;
  if (!P.cast_from_origin) {
    const F3 tmp = s;
    s = e;
    e = tmp;
  }
  d.nsteps = 0;
  d.range_error = false;
  d.n_axis[0] = d.n_axis[1] = d.n_axis[2] = 0;
  if (isnan(s.x) || isnan(s.y) || isnan(s.z) || isnan(e.x) || isnan(e.y) || isnan(e.z)) return;
  if (!(index_in_range(s.x) && index_in_range(s.y) && index_in_range(s.z) && index_in_range(e.x) && index_in_range(e.y) && index_in_range(e.z))) {
    d.range_error = true;
    return;
  }
  const float st[3] = {s.x, s.y, s.z};
  const float en[3] = {e.x, e.y, e.z};
  u32 len = 0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    d.c[k] = grid_index(st[k]);
    const int ei = grid_index(en[k]);
    const int diff = ei - d.c[k];
    d.n_axis[k] = static_cast<u32>(diff < 0 ? -diff : diff);
    len += d.n_axis[k];
    const float ray = en[k] - st[k];
    const int sg = signum(ray);
    d.sgn[k] = sg;
    const int corrected = sg > 0 ? sg : 0;
    const float shifted = st[k] - static_cast<float>(d.c[k]);
    const float dist = static_cast<float>(corrected) - shifted;
    d.t_next[k] = dist / ray;  // the upstream |ray| < 0 guard never fires: -inf / NaN are intended
    d.t_step[k] = static_cast<float>(sg) / ray;
  }
  d.nsteps = len + 1;
}
// advance to the next voxel (call after consuming d.c)
__device__ __forceinline__ void dda_step(Dda& d) {
  int k = 0;
  float best = d.t_next[0];
  if (d.t_next[1] < best) {
    best = d.t_next[1];
    k = 1;
  }
  if (d.t_next[2] < best) {
    k = 2;
  }
  // branch-free select keeps the arrays in registers
  d.c[0] += (k == 0) ? d.sgn[0] : 0;
  d.c[1] += (k == 1) ? d.sgn[1] : 0;
  d.c[2] += (k == 2) ? d.sgn[2] : 0;
  d.t_next[0] = (k == 0) ? d.t_next[0] + d.t_step[0] : d.t_next[0];
  d.t_next[1] = (k == 1) ? d.t_next[1] + d.t_step[1] : d.t_next[1];
  d.t_next[2] = (k == 2) ? d.t_next[2] + d.t_step[2] : d.t_next[2];
}

// ---- voxel update (voxblox TsdfIntegratorBase::updateTsdfVoxel) ------------------------------
struct Voxel {
  float d, w;
  u32 c;  // wire packing a | b<<8 | g<<16 | r<<24
};
// ray colour is kept in the same wire packing
__device__ __forceinline__ u32 blend_colors(u32 c1, float w1, u32 c2, float w2) {
  const float tot = w1 + w2;
  w1 /= tot;
  w2 /= tot;
  u32 out = 0;
#pragma unroll
  for (int sh = 0; sh < 32; sh += 8) {
    const float a = static_cast<float>(static_cast<int>((c1 >> sh) & 255u));
    const float b = static_cast<float>(static_cast<int>((c2 >> sh) & 255u));
    const float v = roundf(a * w1 + b * w2);
    out |= (static_cast<u32>(static_cast<int>(v)) & 255u) << sh;
  }
  return out;
}
// signed distance of a voxel centre along the ray (computeDistance)
__device__ __forceinline__ float compute_sdf(const FrameParams& P, F3 point_G, int gx, int gy, int gz) {
  const F3 origin{P.tx, P.ty, P.tz};
  const F3 c{center_coord(gx, P.voxel_size), center_coord(gy, P.voxel_size), center_coord(gz, P.voxel_size)};
  const F3 v = c - origin;
  const F3 dv = point_G - origin;
  const float dist = sqrtf(dot3(dv, dv));
  const float proj = dot3(v, dv) / dist;
  return dist - proj;
}
__device__ __forceinline__ float update_weight(const FrameParams& P, float sdf, float weight) {
  float uw = weight;
  const float eps = P.voxel_size;
  if (P.use_dropoff && sdf < -eps) {
    uw = weight * (P.trunc + sdf) / (P.trunc - eps);
    uw = std_max(uw, 0.0f);
  }
  if (P.use_sparsity) {
    if (fabsf(sdf) < P.trunc) uw *= P.sparsity_factor;
  }
  return uw;
}
__device__ __forceinline__ void update_voxel(const FrameParams& P, Voxel& v, float sdf, float uw, u32 ray_color) {
  const float nw = v.w + uw;
  if (nw < kEps) return;
  const float nsdf = (sdf * uw + v.d * v.w) / nw;
  if (fabsf(sdf) < P.trunc) v.c = blend_colors(v.c, v.w, ray_color, uw);
  v.d = (nsdf > 0.0f) ? std_min(P.trunc, nsdf) : std_max(-P.trunc, nsdf);
  v.w = std_min(P.max_weight, nw);
}
// true when an update with (sdf, uw) provably leaves distance == trunc unchanged for every voxel
// weight in [0, max_weight] (float error analysis in DESIGN.md): lets long all-free-space runs be
// folded without changing a single bit of the result.
__device__ __forceinline__ bool saturating_update(const FrameParams& P, float sdf, float uw) {
  if (!(uw > 0.0f)) return false;
  const double margin = 4.0 * 5.9604644775390625e-08 * (1.0 + static_cast<double>(P.max_weight) / static_cast<double>(uw));
  return static_cast<double>(sdf) >= static_cast<double>(P.trunc) * (1.0 + margin);
}

}  // namespace cox
