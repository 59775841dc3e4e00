// Recover mode on the GPU: voxblox_msgs/Mesh with per-triangle observation history -> one point cloud per frame id ->
// per-pose clouds in the camera frame -> the integrator, without leaving HBM.
//
// Replaces voxblox::MeshConverter (coxgraph/include/coxgraph/map_comm/mesh_converter.h:22-289) and the loop of
// TsdfRecover::processMesh (coxgraph/include/coxgraph/map_comm/tsdf_recover.h:59-99).  The reference walks the
// triangles one by one and appends to a std::map<uint8_t, cloud>; here
//   k_tri_count    thread per triangle: decoded vertices -> interpolated point count, history pair count
//   scan x2        offsets of each triangle's points / (triangle, frame id) pairs
//   k_tri_write    thread per triangle: vertices + interpolateTriangle points into a triangle-major buffer, pairs expanded
//   radix sort     pairs by frame id (8 bits, stable => triangle order inside a cloud is the message order)
//   k_pair_sizes + scan + k_key_begin   where each pair's points go, where each frame id's cloud starts
//   k_gather       wave per pair: copies the triangle's points into the cloud (coalesced)
//   k_transform    every pose's cloud moved by T_G_C^-1 (what getNextPointcloud does per call)
// Arithmetic follows the reference expression by expression (see oracle/cox_oracle_mesh.hpp for the list of kept
// quirks); results are bit-identical to the oracle.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <vector>

#include "cox_internal.hpp"
#include "cox_sort.hpp"

using namespace cox;

namespace {

constexpr u32 kMaxEdgePts = 1u << 16;  // a dist += voxel_size loop longer than this is refused (it would stall in the reference)
enum : u32 { kMeshErrEdge = 1u, kMeshErrHistory = 2u };

struct MeshView {
  float block_edge_length, interp_step;
  u32 n_blocks, n_tri;
  const int64_t* block_index;
  const u64* vertex_begin;
  const uint16_t *x, *y, *z;
  const uint8_t *r, *g, *b;
  const uint8_t* block_has_history;
  const u64* block_rec_begin;  // first recovered-point index of the block
  const u64* history_begin;
  const u32* history;
};

struct Tri {
  F3 p[3];
  u32 c[3];  // r | g<<8 | b<<16 | a<<24 (byte order of the integrator's rgba input)
  u32 block;
  bool active;
};

__device__ __forceinline__ u32 find_block(const MeshView& M, u64 vertex) {
  u32 lo = 0, hi = M.n_blocks;  // last b with vertex_begin[b] <= vertex
  while (hi - lo > 1) {
    const u32 mid = (lo + hi) >> 1;
    if (M.vertex_begin[mid] <= vertex) lo = mid;
    else hi = mid;
  }
  return lo;
}

// mesh_converter.h:96-112
__device__ __forceinline__ Tri load_triangle(const MeshView& M, u32 t) {
  Tri T;
  const u64 v0 = static_cast<u64>(t) * 3;
  T.block = find_block(M, v0);
  T.active = M.block_has_history[T.block] != 0;
  constexpr float conv = 2.0f / 65535;
  const float ix = static_cast<float>(M.block_index[3 * T.block]), iy = static_cast<float>(M.block_index[3 * T.block + 1]),
              iz = static_cast<float>(M.block_index[3 * T.block + 2]);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const u64 v = v0 + k;
    T.p[k].x = (static_cast<float>(M.x[v]) * conv + ix) * M.block_edge_length;
    T.p[k].y = (static_cast<float>(M.y[v]) * conv + iy) * M.block_edge_length;
    T.p[k].z = (static_cast<float>(M.z[v]) * conv + iz) * M.block_edge_length;
    T.c[k] = static_cast<u32>(M.r[v]) | (static_cast<u32>(M.g[v]) << 8) | (static_cast<u32>(M.b[v]) << 16) | (255u << 24);
  }
  return T;
}

__device__ __forceinline__ float norm3(F3 a) { return sqrtf(dot3(a, a)); }

// number of points `for (float dist = step; dist < len; dist += step)` produces  (mesh_converter.h:225,233,241)
__device__ __forceinline__ u32 edge_count(float step, float len, u32* err) {
  u32 c = 0;
  for (float dist = step; dist < len; dist += step) {
    if (++c > kMaxEdgePts) {
      *err |= kMeshErrEdge;
      return 0;
    }
  }
  return c;
}

// one edge of interpolateTriangle: points from + t / |t| * dist, colours blended by dist / |t|
__device__ __forceinline__ u32 edge_write(float step, F3 from, F3 t, u32 ca, u32 cb, float* __restrict__ xyz, u32* __restrict__ rgba, u32 at) {
  const float len = norm3(t);
  for (float dist = step; dist < len; dist += step) {
    const F3 dir{t.x / len, t.y / len, t.z / len};
    xyz[3 * static_cast<size_t>(at)] = from.x + dir.x * dist;
    xyz[3 * static_cast<size_t>(at) + 1] = from.y + dir.y * dist;
    xyz[3 * static_cast<size_t>(at) + 2] = from.z + dir.z * dist;
    rgba[at] = blend_colors(ca, 1 - dist / len, cb, dist / len);
    ++at;
  }
  return at;
}

__global__ void __launch_bounds__(256) k_tri_count(MeshView M, u32* __restrict__ tri_npts, u32* __restrict__ tri_npairs, u64* __restrict__ totals,
                                                   u32* __restrict__ d_err) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  u32 npts = 0, npairs = 0, err = 0;
  if (t < M.n_tri) {
    const Tri T = load_triangle(M, t);
    if (T.active) {
      const u32 n01 = edge_count(M.interp_step, norm3(T.p[1] - T.p[0]), &err);
      const u32 n02 = edge_count(M.interp_step, norm3(T.p[2] - T.p[0]), &err);
      const u32 n12 = edge_count(M.interp_step, norm3(T.p[2] - T.p[1]), &err);
      npts = 3 + n01 + 1 + n02 + n12;
      const u64 h0 = M.history_begin[t], h1 = M.history_begin[t + 1];
      if (h1 < h0 || ((h1 - h0) & 1ull)) err |= kMeshErrHistory;
      else {
        u64 pairs = 0;
        for (u64 k = h0; k < h1; k += 2) {
          const u32 a = M.history[k], b = M.history[k + 1];
          if (a <= b) pairs += static_cast<u64>(b - a) + 1;
        }
        if (pairs > 0x7fffffffull) err |= kMeshErrHistory;
        else npairs = static_cast<u32>(pairs);
      }
    }
    tri_npts[t] = npts;
    tri_npairs[t] = npairs;
  }
  if (err) atomicOr(d_err, err);
  // totals in 64 bit (the 32-bit scans below are only trusted once these fit)
  u64 a = npts, b = npairs, c = static_cast<u64>(npts) * npairs;
  for (int off = 32; off > 0; off >>= 1) {
    a += __shfl_down(a, off);
    b += __shfl_down(b, off);
    c += __shfl_down(c, off);
  }
  if (lane_id() == 0 && (a | b | c)) {
    atomicAdd(reinterpret_cast<unsigned long long*>(&totals[0]), static_cast<unsigned long long>(a));
    atomicAdd(reinterpret_cast<unsigned long long*>(&totals[1]), static_cast<unsigned long long>(b));
    atomicAdd(reinterpret_cast<unsigned long long*>(&totals[2]), static_cast<unsigned long long>(c));  // points over all clouds
  }
}

// mesh_converter.h:113-142 + interpolateTriangle (:212-277)
__global__ void __launch_bounds__(256) k_tri_write(MeshView M, const u32* __restrict__ tri_off, const u32* __restrict__ pair_off,
                                                   float* __restrict__ tp_xyz, u32* __restrict__ tp_rgba, u32* __restrict__ pair_key,
                                                   u32* __restrict__ pair_tri, float* __restrict__ rec_xyz, uint8_t* __restrict__ rec_rgb) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= M.n_tri) return;
  const Tri T = load_triangle(M, t);
  if (!T.active) return;
  // recovered_pointcloud: one XYZRGB point per vertex
  const u64 rec0 = M.block_rec_begin[T.block] + (static_cast<u64>(t) * 3 - M.vertex_begin[T.block]);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    rec_xyz[3 * (rec0 + k)] = T.p[k].x;
    rec_xyz[3 * (rec0 + k) + 1] = T.p[k].y;
    rec_xyz[3 * (rec0 + k) + 2] = T.p[k].z;
    rec_rgb[3 * (rec0 + k)] = static_cast<uint8_t>(T.c[k]);
    rec_rgb[3 * (rec0 + k) + 1] = static_cast<uint8_t>(T.c[k] >> 8);
    rec_rgb[3 * (rec0 + k) + 2] = static_cast<uint8_t>(T.c[k] >> 16);
  }
  // triangle, then e01 points, centroid, e02 points, e12 points
  u32 at = tri_off[t];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    tp_xyz[3 * static_cast<size_t>(at)] = T.p[k].x;
    tp_xyz[3 * static_cast<size_t>(at) + 1] = T.p[k].y;
    tp_xyz[3 * static_cast<size_t>(at) + 2] = T.p[k].z;
    tp_rgba[at] = T.c[k];
    ++at;
  }
  at = edge_write(M.interp_step, T.p[0], T.p[1] - T.p[0], T.c[0], T.c[1], tp_xyz, tp_rgba, at);
  const F3 sum = (T.p[0] + T.p[1]) + T.p[2];
  tp_xyz[3 * static_cast<size_t>(at)] = sum.x / 3.0f;
  tp_xyz[3 * static_cast<size_t>(at) + 1] = sum.y / 3.0f;
  tp_xyz[3 * static_cast<size_t>(at) + 2] = sum.z / 3.0f;
  tp_rgba[at] = blend_colors(T.c[2], static_cast<float>(1 / 3.0), blend_colors(T.c[0], 0.5f, T.c[1], 0.5f), static_cast<float>(2 / 3.0));
  ++at;
  at = edge_write(M.interp_step, T.p[0], T.p[2] - T.p[0], T.c[0], T.c[1], tp_xyz, tp_rgba, at);  // colours 0 and 1: as the reference
  at = edge_write(M.interp_step, T.p[1], T.p[2] - T.p[1], T.c[1], T.c[2], tp_xyz, tp_rgba, at);
  // inclusive runs -> (frame id & 255, triangle) pairs
  u32 p = pair_off[t];
  for (u64 k = M.history_begin[t]; k < M.history_begin[t + 1]; k += 2) {
    const u32 a = M.history[k], b = M.history[k + 1];
    if (a > b) continue;
    for (u32 j = a;; ++j) {
      pair_key[p] = j & 255u;
      pair_tri[p] = t;
      ++p;
      if (j == b) break;
    }
  }
}

__global__ void __launch_bounds__(256) k_pair_sizes(const u32* __restrict__ pair_tri, const u32* __restrict__ tri_npts, u32* __restrict__ size, u32 n) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) size[i] = tri_npts[pair_tri[i]];
}

// key_pt_begin[k] = first point of frame id k's cloud, k = 0..256
__global__ void __launch_bounds__(320) k_key_begin(const u32* __restrict__ keys_sorted, const u32* __restrict__ out_off, u32 n_pairs, u32 n_points,
                                                   u32* __restrict__ key_pt_begin) {
  const u32 k = threadIdx.x;
  if (k > 256) return;
  u32 lo = 0, hi = n_pairs;  // first pair with key >= k
  while (lo < hi) {
    const u32 mid = (lo + hi) >> 1;
    if (keys_sorted[mid] < k) lo = mid + 1;
    else hi = mid;
  }
  key_pt_begin[k] = (lo < n_pairs) ? out_off[lo] : n_points;
}

__global__ void __launch_bounds__(256) k_gather(const u32* __restrict__ pair_tri, const u32* __restrict__ out_off, const u32* __restrict__ tri_off,
                                                const u32* __restrict__ tri_npts, const float* __restrict__ tp_xyz, const u32* __restrict__ tp_rgba,
                                                float* __restrict__ cloud_xyz, u32* __restrict__ cloud_rgba, u32 n_pairs) {
  const u32 wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (wave >= n_pairs) return;
  const u32 t = pair_tri[wave];
  const size_t src = tri_off[t], dst = out_off[wave];
  const u32 n = tri_npts[t];
  for (u32 j = lane_id(); j < 3 * n; j += 64) cloud_xyz[3 * dst + j] = tp_xyz[3 * src + j];
  for (u32 j = lane_id(); j < n; j += 64) cloud_rgba[dst + j] = tp_rgba[src + j];
}

struct PoseSlice {
  float qw, qx, qy, qz, tx, ty, tz;  // T_C_G = T_G_C^-1
  u32 src, len, dst;
};

// transformPointcloud(T_Submap_C.inverse(), cloud, out)  (mesh_converter.h:200-203)
__global__ void __launch_bounds__(256) k_transform(const PoseSlice* __restrict__ poses, const float* __restrict__ cloud_xyz, float* __restrict__ out_xyz) {
  const PoseSlice P = poses[blockIdx.y];
  const F3 qv{P.qx, P.qy, P.qz};
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < P.len; i += gridDim.x * blockDim.x) {
    const size_t s = 3 * (static_cast<size_t>(P.src) + i), d = 3 * (static_cast<size_t>(P.dst) + i);
    const F3 v{cloud_xyz[s], cloud_xyz[s + 1], cloud_xyz[s + 2]};
    F3 uv = cross3(qv, v);
    uv = uv + uv;
    const F3 c = cross3(qv, uv);
    out_xyz[d] = ((v.x + P.qw * uv.x) + c.x) + P.tx;
    out_xyz[d + 1] = ((v.y + P.qw * uv.y) + c.y) + P.ty;
    out_xyz[d + 2] = ((v.z + P.qw * uv.z) + c.z) + P.tz;
  }
}

template <typename T>
int dev_alloc(T** p, size_t count) {
  *p = nullptr;
  if (count == 0) count = 1;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T));
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return (e == hipErrorOutOfMemory) ? COX_ERR_OUT_OF_MEMORY : COX_ERR_NO_DEVICE;
  }
  return COX_OK;
}
template <typename T>
int dev_upload(T** p, const T* host, size_t count, hipStream_t s) {
  const int rc = dev_alloc(p, count);
  if (rc != COX_OK) return rc;
  if (count) COX_HIP(hipMemcpyAsync(*p, host, count * sizeof(T), hipMemcpyHostToDevice, s));
  return COX_OK;
}
template <typename T>
void dev_free(T** p) {
  if (*p) (void)hipFree(*p);
  *p = nullptr;
}
#define COX_TRY(expr)              \
  do {                             \
    int st_ = (expr);              \
    if (st_ != COX_OK) return st_; \
  } while (0)

// host helpers: float rotate in the device's operation order (Eigen _transformVector)
void host_rotate(const float q[4], const float v[3], float out[3]) {
  const float qx = q[1], qy = q[2], qz = q[3];
  float uv[3] = {qy * v[2] - qz * v[1], qz * v[0] - qx * v[2], qx * v[1] - qy * v[0]};
  for (float& u : uv) u = u + u;
  const float c[3] = {qy * uv[2] - qz * uv[1], qz * uv[0] - qx * uv[2], qx * uv[1] - qy * uv[0]};
  for (int k = 0; k < 3; ++k) out[k] = (v[k] + q[0] * uv[k]) + c[k];
}

// ros::Time difference -> Duration::toSec()
double stamp_diff_sec(u32 sec_a, u32 nsec_a, u32 sec_b, u32 nsec_b) {
  const int64_t d = (static_cast<int64_t>(sec_a) * 1000000000ll + nsec_a) - (static_cast<int64_t>(sec_b) * 1000000000ll + nsec_b);
  int64_t s = d / 1000000000ll, ns = d % 1000000000ll;
  if (ns < 0) {
    ns += 1000000000ll;
    s -= 1;
  }
  return static_cast<double>(static_cast<int32_t>(s)) + 1e-9 * static_cast<double>(static_cast<int32_t>(ns));
}
// a double used as key of the reference's std::map<uint8_t, ...> (x86-64: truncate to int32, keep the low byte)
uint8_t frame_key(double id) {
  if (!(id > -2147483649.0 && id < 2147483648.0)) return 0;
  return static_cast<uint8_t>(static_cast<int32_t>(id));
}

}  // namespace

struct cox_meshconv {
  int device = 0;
  float interp_step = 0.2f;
  hipStream_t stream = nullptr;
  // message (device copies) -- setMesh
  bool has_mesh = false;
  float block_edge_length = 0;
  u32 n_blocks = 0, n_tri = 0;
  u64 n_recovered = 0;
  int64_t* d_block_index = nullptr;
  u64 *d_vertex_begin = nullptr, *d_block_rec_begin = nullptr, *d_history_begin = nullptr;
  uint16_t *d_x = nullptr, *d_y = nullptr, *d_z = nullptr;
  uint8_t *d_r = nullptr, *d_g = nullptr, *d_b = nullptr, *d_has = nullptr;
  u32* d_history = nullptr;
  // trajectory -- setTrajectory appends (only clear() empties it)
  std::vector<u32> sec, nsec;
  std::vector<float> T_G_C;
  // products of convertToPointCloud
  bool converted = false;
  u64 rec_count = 0;  // points in d_rec_* (survives clear())
  float *d_rec_xyz = nullptr, *d_cloud_xyz = nullptr, *d_pose_xyz = nullptr;
  uint8_t* d_rec_rgb = nullptr;
  u32* d_cloud_rgba = nullptr;
  u32 key_pt_begin[257] = {};
  std::vector<PoseSlice> slices;  // per pose
};

static void free_mesh(cox_meshconv* C) {
  dev_free(&C->d_block_index);
  dev_free(&C->d_vertex_begin);
  dev_free(&C->d_block_rec_begin);
  dev_free(&C->d_history_begin);
  dev_free(&C->d_x);
  dev_free(&C->d_y);
  dev_free(&C->d_z);
  dev_free(&C->d_r);
  dev_free(&C->d_g);
  dev_free(&C->d_b);
  dev_free(&C->d_has);
  dev_free(&C->d_history);
  C->has_mesh = false;
  C->n_blocks = C->n_tri = 0;
  C->n_recovered = 0;
}
// keep_recovered: clear() leaves the caller's recovered_pointcloud alone (mesh_converter.h:170-181)
static void free_clouds(cox_meshconv* C, bool keep_recovered = false) {
  if (!keep_recovered) {
    dev_free(&C->d_rec_xyz);
    dev_free(&C->d_rec_rgb);
    C->rec_count = 0;
  }
  dev_free(&C->d_cloud_xyz);
  dev_free(&C->d_cloud_rgba);
  dev_free(&C->d_pose_xyz);
  C->slices.clear();
  C->converted = false;
  std::fill(std::begin(C->key_pt_begin), std::end(C->key_pt_begin), 0u);
}

extern "C" {

int cox_meshconv_create(int device, float interpolate_voxel_size, cox_meshconv_t** out) {
  COX_ENTRY();
  if (!out || !(interpolate_voxel_size > 0.0f)) return COX_ERR_INVALID_ARG;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) {
    (void)hipGetLastError();
    return COX_ERR_NO_DEVICE;
  }
  COX_HIP(hipSetDevice(device));
  auto* C = new cox_meshconv();
  C->device = device;
  C->interp_step = interpolate_voxel_size;
  if (hipStreamCreateWithFlags(&C->stream, hipStreamNonBlocking) != hipSuccess) {
    delete C;
    return COX_ERR_NO_DEVICE;
  }
  *out = C;
  return COX_OK;
}

void cox_meshconv_destroy(cox_meshconv_t* C) {
  if (!C) return;
  (void)hipSetDevice(C->device);
  (void)hipStreamSynchronize(C->stream);
  free_mesh(C);
  free_clouds(C);
  (void)hipStreamDestroy(C->stream);
  delete C;
}

int cox_meshconv_set_mesh(cox_meshconv_t* C, const cox_mesh_msg* m) {
  COX_ENTRY();
  if (!C || !m) return COX_ERR_INVALID_ARG;
  if (m->n_poses == 0) return COX_OK;  // "received a mesh with empty trajectory": ignored (mesh_converter.h:56-60)
  if (m->n_blocks && (!m->block_index || !m->vertex_begin || !m->block_has_history)) return COX_ERR_INVALID_ARG;
  if (!m->stamp_sec || !m->stamp_nsec || !m->T_G_C) return COX_ERR_INVALID_ARG;
  if (m->n_blocks > 0x7fffffffull) return COX_ERR_UNSUPPORTED;
  const u64 nv = m->n_blocks ? m->vertex_begin[m->n_blocks] : 0;
  if (m->n_blocks && m->vertex_begin[0] != 0) return COX_ERR_INVALID_ARG;
  std::vector<u64> rec_begin(m->n_blocks + 1, 0);
  for (u64 b = 0; b < m->n_blocks; ++b) {
    const u64 v0 = m->vertex_begin[b], v1 = m->vertex_begin[b + 1];
    if (v1 < v0 || v0 % 3 || v1 % 3) return COX_ERR_INVALID_ARG;
    rec_begin[b + 1] = rec_begin[b] + (m->block_has_history[b] ? v1 - v0 : 0);
  }
  if (nv / 3 > 0x7fffffffull) return COX_ERR_UNSUPPORTED;
  if (nv && (!m->x || !m->y || !m->z || !m->r || !m->g || !m->b || !m->history_begin)) return COX_ERR_INVALID_ARG;
  const u64 nh = nv ? m->history_begin[nv / 3] : 0;
  if (nh && !m->history) return COX_ERR_INVALID_ARG;
  COX_HIP(hipSetDevice(C->device));
  COX_HIP(hipStreamSynchronize(C->stream));
  free_mesh(C);
  free_clouds(C);
  hipStream_t s = C->stream;
  COX_TRY(dev_upload(&C->d_block_index, m->block_index, 3 * m->n_blocks, s));
  COX_TRY(dev_upload(&C->d_vertex_begin, reinterpret_cast<const u64*>(m->vertex_begin), m->n_blocks + 1, s));
  COX_TRY(dev_upload(&C->d_block_rec_begin, rec_begin.data(), m->n_blocks + 1, s));
  COX_TRY(dev_upload(&C->d_has, m->block_has_history, m->n_blocks, s));
  COX_TRY(dev_upload(&C->d_x, m->x, nv, s));
  COX_TRY(dev_upload(&C->d_y, m->y, nv, s));
  COX_TRY(dev_upload(&C->d_z, m->z, nv, s));
  COX_TRY(dev_upload(&C->d_r, m->r, nv, s));
  COX_TRY(dev_upload(&C->d_g, m->g, nv, s));
  COX_TRY(dev_upload(&C->d_b, m->b, nv, s));
  COX_TRY(dev_upload(&C->d_history_begin, reinterpret_cast<const u64*>(m->history_begin), nv ? nv / 3 + 1 : 0, s));
  COX_TRY(dev_upload(&C->d_history, m->history, nh, s));
  COX_HIP(hipStreamSynchronize(s));  // the caller's buffers are free again, rec_begin goes out of scope
  C->block_edge_length = m->block_edge_length;
  C->n_blocks = static_cast<u32>(m->n_blocks);
  C->n_tri = static_cast<u32>(nv / 3);
  C->n_recovered = rec_begin[m->n_blocks];
  C->has_mesh = true;
  for (u64 i = 0; i < m->n_poses; ++i) {
    C->sec.push_back(m->stamp_sec[i]);
    C->nsec.push_back(m->stamp_nsec[i]);
    C->T_G_C.insert(C->T_G_C.end(), m->T_G_C + 7 * i, m->T_G_C + 7 * i + 7);
  }
  return COX_OK;
}

int cox_meshconv_convert(cox_meshconv_t* C, uint64_t* n_recovered_points, int* converted) {
  COX_ENTRY();
  if (!C) return COX_ERR_INVALID_ARG;
  if (n_recovered_points) *n_recovered_points = 0;
  if (converted) *converted = 0;
  COX_HIP(hipSetDevice(C->device));
  free_clouds(C);
  if (!C->has_mesh || C->n_blocks == 0) return COX_OK;  // "if (mesh_.mesh_blocks.empty()) return false"
  hipStream_t s = C->stream;
  const u32 nt = C->n_tri;
  MeshView M{C->block_edge_length, C->interp_step, C->n_blocks, nt,          C->d_block_index, C->d_vertex_begin, C->d_x,
             C->d_y,               C->d_z,         C->d_r,      C->d_g,      C->d_b,           C->d_has,          C->d_block_rec_begin,
             C->d_history_begin,   C->d_history};
  int rc = COX_OK;
  u32 *tri_npts = nullptr, *tri_npairs = nullptr, *tri_off = nullptr, *pair_off = nullptr, *d_err = nullptr;
  u32 *pk[2] = {nullptr, nullptr}, *pt[2] = {nullptr, nullptr}, *psize = nullptr, *out_off = nullptr, *d_key_begin = nullptr;
  u64* d_totals = nullptr;
  float* tp_xyz = nullptr;
  u32* tp_rgba = nullptr;
  PoseSlice* d_slices = nullptr;
  ScanWorkspace scanws;
  SortWorkspace sortws;
  auto cleanup = [&]() {
    (void)hipStreamSynchronize(s);
    dev_free(&tri_npts), dev_free(&tri_npairs), dev_free(&tri_off), dev_free(&pair_off), dev_free(&d_err);
    dev_free(&pk[0]), dev_free(&pk[1]), dev_free(&pt[0]), dev_free(&pt[1]), dev_free(&psize), dev_free(&out_off), dev_free(&d_key_begin);
    dev_free(&d_totals), dev_free(&tp_xyz), dev_free(&tp_rgba), dev_free(&d_slices);
    dev_free(&scanws.block_sums), dev_free(&sortws.counts), dev_free(&sortws.totals);
  };
  auto body = [&]() -> int {
    COX_TRY(dev_alloc(&C->d_rec_xyz, 3 * C->n_recovered));
    COX_TRY(dev_alloc(&C->d_rec_rgb, 3 * C->n_recovered));
    u64 totals[3] = {0, 0, 0};
    u32 n_pairs = 0, n_points = 0;
    if (nt) {
      COX_TRY(dev_alloc(&tri_npts, nt));
      COX_TRY(dev_alloc(&tri_npairs, nt));
      COX_TRY(dev_alloc(&tri_off, nt));
      COX_TRY(dev_alloc(&pair_off, nt));
      COX_TRY(dev_alloc(&d_err, 1));
      COX_TRY(dev_alloc(&d_totals, 3));
      COX_HIP(hipMemsetAsync(d_err, 0, sizeof(u32), s));
      COX_HIP(hipMemsetAsync(d_totals, 0, 3 * sizeof(u64), s));
      const u32 grid_t = (nt + 255) / 256;
      hipLaunchKernelGGL(k_tri_count, dim3(grid_t), dim3(256), 0, s, M, tri_npts, tri_npairs, d_totals, d_err);
      u32 err = 0;
      COX_HIP(hipMemcpyAsync(totals, d_totals, sizeof(totals), hipMemcpyDeviceToHost, s));
      COX_HIP(hipMemcpyAsync(&err, d_err, sizeof(u32), hipMemcpyDeviceToHost, s));
      COX_HIP(hipStreamSynchronize(s));
      if (err & kMeshErrHistory) return COX_ERR_INVALID_ARG;  // odd run list / absurd run length
      if (err & kMeshErrEdge) return COX_ERR_UNSUPPORTED;     // interpolate_voxel_size far too small for this mesh
      if (totals[0] >= 0xffffff00ull || totals[1] >= 0x7fffffffull || totals[2] >= 0xffffff00ull) return COX_ERR_UNSUPPORTED;
      n_pairs = static_cast<u32>(totals[1]);
      n_points = static_cast<u32>(totals[2]);
      COX_TRY(dev_alloc(&scanws.block_sums, scan_num_blocks(std::max(nt, n_pairs)) + 2));
      exclusive_scan_u32(tri_npts, tri_off, nullptr, nt, nt, nullptr, scanws, s);
      exclusive_scan_u32(tri_npairs, pair_off, nullptr, nt, nt, nullptr, scanws, s);
      COX_TRY(dev_alloc(&tp_xyz, 3 * totals[0]));
      COX_TRY(dev_alloc(&tp_rgba, totals[0]));
      for (int k = 0; k < 2; ++k) {
        COX_TRY(dev_alloc(&pk[k], n_pairs));
        COX_TRY(dev_alloc(&pt[k], n_pairs));
      }
      hipLaunchKernelGGL(k_tri_write, dim3(grid_t), dim3(256), 0, s, M, tri_off, pair_off, tp_xyz, tp_rgba, pk[0], pt[0], C->d_rec_xyz, C->d_rec_rgb);
    }
    COX_TRY(dev_alloc(&C->d_cloud_xyz, 3 * static_cast<size_t>(n_points)));
    COX_TRY(dev_alloc(&C->d_cloud_rgba, n_points));
    if (n_pairs) {
      sortws.tiles_cap = std::max<u32>(1, sort_num_tiles(n_pairs));
      COX_TRY(dev_alloc(&sortws.counts, sort_counts_words(sortws.tiles_cap)));
      COX_TRY(dev_alloc(&sortws.totals, sort_totals_words()));
      COX_HIP(hipMemsetAsync(sortws.totals, 0, sizeof(u32) * sort_totals_words(), s));
      const int res = radix_sort_pairs<11>(pk[0], pt[0], pk[1], pt[1], nullptr, n_pairs, n_pairs, 8, false, 1, sortws, nullptr, s);
      COX_TRY(dev_alloc(&psize, n_pairs));
      COX_TRY(dev_alloc(&out_off, n_pairs));
      COX_TRY(dev_alloc(&d_key_begin, 257));
      const u32 grid_p = (n_pairs + 255) / 256;
      hipLaunchKernelGGL(k_pair_sizes, dim3(grid_p), dim3(256), 0, s, pt[res], tri_npts, psize, n_pairs);
      exclusive_scan_u32(psize, out_off, nullptr, n_pairs, n_pairs, nullptr, scanws, s);
      hipLaunchKernelGGL(k_key_begin, dim3(1), dim3(320), 0, s, pk[res], out_off, n_pairs, n_points, d_key_begin);
      hipLaunchKernelGGL(k_gather, dim3((n_pairs + 3) / 4), dim3(256), 0, s, pt[res], out_off, tri_off, tri_npts, tp_xyz, tp_rgba, C->d_cloud_xyz,
                         C->d_cloud_rgba, n_pairs);
      COX_HIP(hipMemcpyAsync(C->key_pt_begin, d_key_begin, sizeof(C->key_pt_begin), hipMemcpyDeviceToHost, s));
      COX_HIP(hipStreamSynchronize(s));
    }
    // per pose: frame id -> cloud slice -> camera frame  (getNextPointcloud, mesh_converter.h:183-210)
    const size_t np = C->sec.size();
    C->slices.resize(np);
    u64 total = 0;
    for (size_t i = 0; i < np; ++i) {
      const double id = (C->sec[i] == C->sec[0] && C->nsec[i] == C->nsec[0])
                            ? 0.0
                            : std::round(stamp_diff_sec(C->sec[i], C->nsec[i], C->sec[0], C->nsec[0]) / 0.05);
      const u32 key = frame_key(id);
      const float* T = &C->T_G_C[7 * i];
      const float qi[4] = {T[0], -T[1], -T[2], -T[3]};  // minkindr inverse: conjugate rotation, -(R^T t)
      float rt[3];
      host_rotate(qi, T + 4, rt);
      PoseSlice& P = C->slices[i];
      P.qw = qi[0], P.qx = qi[1], P.qy = qi[2], P.qz = qi[3];
      P.tx = -rt[0], P.ty = -rt[1], P.tz = -rt[2];
      P.src = C->key_pt_begin[key];
      P.len = C->key_pt_begin[key + 1] - C->key_pt_begin[key];
      if (total + P.len >= 0xffffff00ull) return COX_ERR_UNSUPPORTED;
      P.dst = static_cast<u32>(total);
      total += P.len;
    }
    COX_TRY(dev_alloc(&C->d_pose_xyz, 3 * total));
    if (np && total) {
      COX_TRY(dev_upload(&d_slices, C->slices.data(), np, s));
      u32 max_len = 0;
      for (const PoseSlice& P : C->slices) max_len = std::max(max_len, P.len);
      const u32 gx = std::min<u32>(1024u, (max_len + 255) / 256);
      for (size_t base = 0; base < np; base += 32768) {
        const u32 cnt = static_cast<u32>(std::min<size_t>(32768, np - base));
        hipLaunchKernelGGL(k_transform, dim3(gx, cnt), dim3(256), 0, s, d_slices + base, C->d_cloud_xyz, C->d_pose_xyz);
      }
    }
    COX_HIP(hipStreamSynchronize(s));
    COX_HIP(hipGetLastError());
    return COX_OK;
  };
  rc = body();
  cleanup();
  if (rc != COX_OK) {
    free_clouds(C);
    return rc;
  }
  C->converted = true;
  C->rec_count = C->n_recovered;
  if (n_recovered_points) *n_recovered_points = C->n_recovered;
  if (converted) *converted = 1;
  return COX_OK;
}

int cox_meshconv_recovered(cox_meshconv_t* C, float* xyz, uint8_t* rgb, uint64_t cap, uint64_t* n) {
  COX_ENTRY();
  if (!C || !n) return COX_ERR_INVALID_ARG;
  *n = C->rec_count;
  if (!xyz && !rgb) return COX_OK;
  if (cap < *n) return COX_ERR_BUFFER_TOO_SMALL;
  if (*n == 0) return COX_OK;
  COX_HIP(hipSetDevice(C->device));
  if (xyz) COX_HIP(hipMemcpy(xyz, C->d_rec_xyz, sizeof(float) * 3 * *n, hipMemcpyDeviceToHost));
  if (rgb) COX_HIP(hipMemcpy(rgb, C->d_rec_rgb, 3 * *n, hipMemcpyDeviceToHost));
  return COX_OK;
}

int cox_meshconv_next(cox_meshconv_t* C, int32_t* i, float T_G_C[7], const float** xyz_dev, const uint8_t** rgba_dev, uint64_t* n, int* has_next) {
  if (!C || !i || !n || !has_next) return COX_ERR_INVALID_ARG;
  *n = 0;
  if (xyz_dev) *xyz_dev = nullptr;
  if (rgba_dev) *rgba_dev = nullptr;
  if (*i < 0 || static_cast<size_t>(*i) >= C->sec.size()) {
    *has_next = 0;
    return COX_OK;
  }
  *has_next = 1;
  if (T_G_C) std::copy(&C->T_G_C[7 * *i], &C->T_G_C[7 * *i] + 7, T_G_C);
  if (C->converted && static_cast<size_t>(*i) < C->slices.size()) {  // without convertToPointCloud every cloud is empty
    const PoseSlice& P = C->slices[*i];
    *n = P.len;
    if (xyz_dev) *xyz_dev = C->d_pose_xyz + 3 * static_cast<size_t>(P.dst);
    if (rgba_dev) *rgba_dev = reinterpret_cast<const uint8_t*>(C->d_cloud_rgba + P.src);
  }
  ++*i;
  return COX_OK;
}

int cox_meshconv_download(cox_meshconv_t* C, int32_t i, float* xyz, uint8_t* rgba, uint64_t cap, uint64_t* n) {
  COX_ENTRY();
  if (!C || !n || i < 0 || static_cast<size_t>(i) >= C->sec.size()) return COX_ERR_INVALID_ARG;
  *n = 0;
  if (!C->converted || static_cast<size_t>(i) >= C->slices.size()) return COX_OK;
  const PoseSlice& P = C->slices[i];
  *n = P.len;
  if (!xyz && !rgba) return COX_OK;
  if (cap < P.len) return COX_ERR_BUFFER_TOO_SMALL;
  if (P.len == 0) return COX_OK;
  COX_HIP(hipSetDevice(C->device));
  if (xyz) COX_HIP(hipMemcpy(xyz, C->d_pose_xyz + 3 * static_cast<size_t>(P.dst), sizeof(float) * 3 * P.len, hipMemcpyDeviceToHost));
  if (rgba) COX_HIP(hipMemcpy(rgba, C->d_cloud_rgba + P.src, sizeof(u32) * P.len, hipMemcpyDeviceToHost));
  return COX_OK;
}

int cox_meshconv_clear(cox_meshconv_t* C) {
  COX_ENTRY();
  if (!C) return COX_ERR_INVALID_ARG;
  COX_HIP(hipSetDevice(C->device));
  COX_HIP(hipStreamSynchronize(C->stream));
  free_mesh(C);
  free_clouds(C, true);
  C->sec.clear();
  C->nsec.clear();
  C->T_G_C.clear();
  return COX_OK;
}

int cox_recover_process_mesh(cox_meshconv_t* C, cox_integrator_t* integ, const cox_mesh_msg* mesh, uint64_t* n_recovered_points, uint64_t* n_integrated) {
  COX_ENTRY();
  if (!C || !integ || !mesh) return COX_ERR_INVALID_ARG;
  if (n_recovered_points) *n_recovered_points = 0;
  if (n_integrated) *n_integrated = 0;
  cox_layer* layer = cox_internal_integrator_layer(integ);
  if (layer->device != C->device) return COX_ERR_INVALID_ARG;
  COX_TRY(cox_integrator_sync(integ));
  COX_TRY(cox_layer_clear(layer));  // tsdf_map_->getTsdfLayerPtr()->removeAllBlocks()   tsdf_recover.h:62
  COX_TRY(cox_meshconv_set_mesh(C, mesh));
  int converted = 0;
  uint64_t n_rec = 0;
  COX_TRY(cox_meshconv_convert(C, &n_rec, &converted));
  int32_t i = 0;
  uint64_t n_int = 0;
  for (;;) {  // tsdf_recover.h:68-77
    float T[7];
    const float* xyz = nullptr;
    const uint8_t* rgba = nullptr;
    uint64_t n = 0;
    int has_next = 0;
    COX_TRY(cox_meshconv_next(C, &i, T, &xyz, &rgba, &n, &has_next));
    if (!has_next) break;
    if (n == 0) continue;
    COX_TRY(cox_integrate_points_dev(integ, T, xyz, rgba, n, 0));
    ++n_int;
  }
  COX_TRY(cox_integrator_sync(integ));  // the clouds are freed by clear() below
  COX_TRY(cox_meshconv_clear(C));
  if (n_recovered_points) *n_recovered_points = n_rec;
  if (n_integrated) *n_integrated = n_int;
  return COX_OK;
}

}  // extern "C"
