// RCCL behind the C ABI: the two collectives of the server's multi-GPU path, so that a C++ host
// (coxgraph_amd/host/coxgraph_hip_posegraph.hpp) does not need Python / torch.distributed for them.
//
//   cox_comm_allreduce_f64   the packed normal equations of one pose-graph evaluation: (4N)^2 + 4N + 1 doubles, summed over
//                            the ranks with ONE ncclAllReduce (KBs: latency-bound -- never one call per constraint)
//   cox_comm_allgather_dev   the submap exchange: every rank's wire arrays (block indices, voxel words, registration points)
//                            to every rank, device to device over xGMI
//
// What the reference does instead: the server PULLS whole submaps from its clients over TCPROS, one by one
// (coxgraph/src/server/client_handler.cpp:82-104, coxgraph_server.cpp:119-128,253-258) and Ceres sums the residual blocks in
// one process (coxgraph/include/coxgraph/server/backend/pose_graph.h:52-73).
//
// librccl.so is loaded on first use (dlopen): the engine itself has no link-time dependency on it, and a box without RCCL
// gets COX_ERR_UNSUPPORTED from these entry points instead of a library that does not load.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <new>

#include "../../include/coxgraph_hip.h"
#include "cox_internal.hpp"

namespace {
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};
RcclApi& rccl() {
  static RcclApi api = [] {
    RcclApi a;
    for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      a.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (a.lib) break;
    }
    if (!a.lib) return a;
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(a.lib, "ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(a.lib, "ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(a.lib, "ncclCommDestroy"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(a.lib, "ncclAllReduce"));
    a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(a.lib, "ncclAllGather"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(a.lib, "ncclGetErrorString"));
    a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce && a.AllGather;
    return a;
  }();
  return api;
}
}  // namespace

struct cox_comm {
  int device = 0, rank = 0, world = 1;
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  double* d_buf = nullptr;
  uint64_t cap = 0;
  hipEvent_t ev = nullptr;  // producer stream -> collective stream
};

#define COX_NCCL(call)                                                                                                                 \
  do {                                                                                                                                 \
    ncclResult_t r_ = (call);                                                                                                          \
    if (r_ != ncclSuccess) {                                                                                                           \
      fprintf(stderr, "[coxgraph_hip] %s:%d %s -> %s\n", __FILE__, __LINE__, #call, rccl().GetErrorString ? rccl().GetErrorString(r_) : "?"); \
      return COX_ERR_COMM;                                                                                                             \
    }                                                                                                                                  \
  } while (0)

extern "C" {

int cox_comm_unique_id(uint8_t id[COX_COMM_ID_BYTES]) {
  if (!id) return COX_ERR_INVALID_ARG;
  if (!rccl().ok) return COX_ERR_UNSUPPORTED;
  static_assert(COX_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
  ncclUniqueId u;
  COX_NCCL(rccl().GetUniqueId(&u));
  memcpy(id, u.internal, COX_COMM_ID_BYTES);
  return COX_OK;
}

int cox_comm_init_rank(int device, int rank, int world, const uint8_t id[COX_COMM_ID_BYTES], cox_comm_t** out) {
  COX_ENTRY();
  if (!out || !id || world < 1 || rank < 0 || rank >= world) return COX_ERR_INVALID_ARG;
  if (!rccl().ok) return COX_ERR_UNSUPPORTED;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return COX_ERR_NO_DEVICE;
  COX_HIP(hipSetDevice(device));
  cox_comm* C = new (std::nothrow) cox_comm();
  if (!C) return COX_ERR_OUT_OF_MEMORY;
  C->device = device;
  C->rank = rank;
  C->world = world;
  ncclUniqueId u;
  memcpy(u.internal, id, COX_COMM_ID_BYTES);
  if (hipStreamCreateWithFlags(&C->stream, hipStreamNonBlocking) != hipSuccess) {
    delete C;
    return COX_ERR_NO_DEVICE;
  }
  const ncclResult_t r = rccl().CommInitRank(&C->comm, world, u, rank);
  if (r != ncclSuccess) {
    fprintf(stderr, "[coxgraph_hip] ncclCommInitRank -> %s\n", rccl().GetErrorString ? rccl().GetErrorString(r) : "?");
    (void)hipStreamDestroy(C->stream);
    delete C;
    return COX_ERR_COMM;
  }
  if (hipEventCreateWithFlags(&C->ev, hipEventDisableTiming) != hipSuccess) C->ev = nullptr;
  *out = C;
  return COX_OK;
}

void cox_comm_destroy(cox_comm_t* C) {
  if (!C) return;
  (void)hipSetDevice(C->device);
  if (C->stream) (void)hipStreamSynchronize(C->stream);
  if (C->comm && rccl().ok) (void)rccl().CommDestroy(C->comm);
  if (C->d_buf) (void)hipFree(C->d_buf);
  if (C->ev) (void)hipEventDestroy(C->ev);
  if (C->stream) (void)hipStreamDestroy(C->stream);
  delete C;
}

int cox_comm_rank(const cox_comm_t* C, int* rank, int* world) {
  if (!C) return COX_ERR_INVALID_ARG;
  if (rank) *rank = C->rank;
  if (world) *world = C->world;
  return COX_OK;
}

int cox_comm_allreduce_f64(cox_comm_t* C, double* buf, uint64_t n) {
  COX_ENTRY();
  if (!C || (n && !buf)) return COX_ERR_INVALID_ARG;
  if (n == 0) return COX_OK;
  COX_HIP(hipSetDevice(C->device));
  if (n > C->cap) {
    if (C->d_buf) (void)hipFree(C->d_buf);
    C->d_buf = nullptr;
    C->cap = 0;
    COX_HIP(hipMalloc(reinterpret_cast<void**>(&C->d_buf), sizeof(double) * n));
    C->cap = n;
  }
  COX_HIP(hipMemcpyAsync(C->d_buf, buf, sizeof(double) * n, hipMemcpyHostToDevice, C->stream));
  COX_NCCL(rccl().AllReduce(C->d_buf, C->d_buf, n, ncclFloat64, ncclSum, C->comm, C->stream));
  COX_HIP(hipMemcpyAsync(buf, C->d_buf, sizeof(double) * n, hipMemcpyDeviceToHost, C->stream));
  COX_HIP(hipStreamSynchronize(C->stream));
  return COX_OK;
}

// producer_stream: the hipStream_t (as void*) whatever wrote send_dev was enqueued on (NULL = the legacy default stream); the
// collective waits for that stream by an event, not for the whole device
int cox_comm_allgather_dev_on(cox_comm_t* C, const void* send_dev, void* recv_dev, uint64_t bytes_per_rank, void* producer_stream) {
  COX_ENTRY();
  if (!C || (bytes_per_rank && (!send_dev || !recv_dev))) return COX_ERR_INVALID_ARG;
  if (bytes_per_rank == 0) return COX_OK;
  COX_HIP(hipSetDevice(C->device));
  if (C->ev) {
    COX_HIP(hipEventRecord(C->ev, static_cast<hipStream_t>(producer_stream)));
    COX_HIP(hipStreamWaitEvent(C->stream, C->ev, 0));
  } else {
    COX_HIP(hipStreamSynchronize(static_cast<hipStream_t>(producer_stream)));
  }
  COX_NCCL(rccl().AllGather(send_dev, recv_dev, bytes_per_rank, ncclUint8, C->comm, C->stream));
  COX_HIP(hipStreamSynchronize(C->stream));
  return COX_OK;
}
// the engine's own exports (cox_layer_export_dev, cox_regpoints_data_dev) are complete when they return and torch tensors are
// written on torch's current stream: without a stream to name, the legacy default stream orders behind every blocking stream
int cox_comm_allgather_dev(cox_comm_t* C, const void* send_dev, void* recv_dev, uint64_t bytes_per_rank) {
  return cox_comm_allgather_dev_on(C, send_dev, recv_dev, bytes_per_rank, nullptr);
}

}  // extern "C"
