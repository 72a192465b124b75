// Brute-force Hamming kNN-2 over several GPUs through the C ABI: every rank scans its shard of
// the database (ascending row ranges, rank order) for all queries, one ncclAllGather exchanges the
// per-query (best, second, index) triples (12 bytes x queries per rank) over xGMI, and the merge
// kernel folds the shards in rank order, so that the earlier shard keeps ties: the first-minimum
// rule of a single scan, bit for bit.  The reference has no multi-device code; this is the one
// path here with an exchange step.
//
// RCCL is bound at run time (dlopen of librccl.so.1), so the library has no link-time dependency
// on it and a process that already holds an RCCL (PyTorch-ROCm ships one under the same soname)
// shares that copy.
#include <dlfcn.h>
#include <string.h>

#include <mutex>

#include "orb_common.h"

namespace orbhip {
hipError_t launch_knn2_merge(int nshards, int nq, size_t shard_stride, const int* best, const int* second, const int* idx,
                             float ratio, int* ob, int* os, int* oi, uint8_t* oa, hipStream_t st);
}

using namespace orbhip;

namespace {

// the few RCCL entry points used, with the types of rccl.h spelled out (ncclComm_t is an opaque
// pointer, ncclUniqueId 128 opaque bytes passed by value, ncclInt32 = 2, ncclSuccess = 0)
struct UniqueId { char internal[128]; };
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
};
std::once_flag g_once;
Rccl g_rccl;

const Rccl* rccl() {
  std::call_once(g_once, [] {
    Rccl& r = g_rccl;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
    }
    if (!r.lib) return;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.lib, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.lib, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.lib, "ncclAllGather"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.GetErrorString;
  });
  return g_rccl.ok ? &g_rccl : nullptr;
}

int rccl_fail(const Rccl* r, int code, const char* what) {
  set_error(std::string(what) + ": " + (r && r->GetErrorString ? r->GetErrorString(code) : "RCCL error"));
  return ORBHIP_ERR_HIP;
}

struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int dev) { hipGetDevice(&prev); if (prev != dev) hipSetDevice(dev); else prev = -1; }
  ~DeviceGuard() { if (prev >= 0) hipSetDevice(prev); }
};

}  // namespace

struct OrbKnnComm {
  void* comm = nullptr;  // ncclComm_t
  bool owned = false;    // created here (destroyed with the handle) or handed in by the caller
  int rank = 0, world = 1, device = 0;
  int* d_local = nullptr;     // [3][cap_q]: this rank's best, second, global index
  int* d_gathered = nullptr;  // [world][3][cap_q]
  int cap_q = 0;
};

extern "C" {

int orb_knn2_comm_unique_id(uint8_t* id) {
  if (!id) { set_error("null argument"); return ORBHIP_ERR_ARG; }
  const Rccl* r = rccl();
  if (!r) { set_error("librccl.so.1 could not be loaded"); return ORBHIP_ERR_UNSUPPORTED; }
  UniqueId u;
  const int rc = r->GetUniqueId(&u);
  if (rc != 0) return rccl_fail(r, rc, "ncclGetUniqueId");
  memcpy(id, u.internal, sizeof(u.internal));
  return ORBHIP_OK;
}

int orb_knn2_comm_create(const uint8_t* id, int rank, int world, int device_id, OrbKnnComm** out) {
  if (!out) { set_error("null argument"); return ORBHIP_ERR_ARG; }
  *out = nullptr;
  if (!id || world < 1 || rank < 0 || rank >= world) { set_error("bad rank / world size"); return ORBHIP_ERR_ARG; }
  const Rccl* r = rccl();
  if (!r) { set_error("librccl.so.1 could not be loaded"); return ORBHIP_ERR_UNSUPPORTED; }
  DeviceGuard g(device_id);
  UniqueId u;
  memcpy(u.internal, id, sizeof(u.internal));
  void* comm = nullptr;
  const int rc = r->CommInitRank(&comm, world, u, rank);
  if (rc != 0) return rccl_fail(r, rc, "ncclCommInitRank");
  OrbKnnComm* c = new OrbKnnComm();
  c->comm = comm; c->owned = true; c->rank = rank; c->world = world; c->device = device_id;
  *out = c;
  return ORBHIP_OK;
}

int orb_knn2_comm_wrap(void* nccl_comm, int rank, int world, int device_id, OrbKnnComm** out) {
  if (!out) { set_error("null argument"); return ORBHIP_ERR_ARG; }
  *out = nullptr;
  if (!nccl_comm || world < 1 || rank < 0 || rank >= world) { set_error("bad communicator, rank or world size"); return ORBHIP_ERR_ARG; }
  if (!rccl()) { set_error("librccl.so.1 could not be loaded"); return ORBHIP_ERR_UNSUPPORTED; }
  OrbKnnComm* c = new OrbKnnComm();
  c->comm = nccl_comm; c->owned = false; c->rank = rank; c->world = world; c->device = device_id;
  *out = c;
  return ORBHIP_OK;
}

void orb_knn2_comm_destroy(OrbKnnComm* c) {
  if (!c) return;
  DeviceGuard g(c->device);
  hipDeviceSynchronize();
  hipFree(c->d_local);
  hipFree(c->d_gathered);
  if (c->owned && c->comm && rccl()) rccl()->CommDestroy(c->comm);
  delete c;
}

int orb_knn2_sharded(OrbKnnComm* c, const uint8_t* d_q, int n_q, const uint8_t* d_db_shard, int n_db_shard,
                     int db_index_base, float ratio, int* d_best, int* d_second, int* d_idx, uint8_t* d_accept,
                     void* stream) {
  if (!c) { set_error("null communicator"); return ORBHIP_ERR_ARG; }
  if (n_q <= 0) return ORBHIP_OK;
  if (!d_q || !d_best || !d_second || !d_idx || (n_db_shard > 0 && !d_db_shard) || n_db_shard < 0) {
    set_error("null argument"); return ORBHIP_ERR_ARG;
  }
  const Rccl* r = rccl();
  if (!r) { set_error("librccl.so.1 could not be loaded"); return ORBHIP_ERR_UNSUPPORTED; }
  DeviceGuard g(c->device);
  hipStream_t st = (hipStream_t)stream;
  if (n_q > c->cap_q) {  // the first call, or more queries than any call before: (re)allocate the exchange buffers
    ORB_HIP_CHECK(hipStreamSynchronize(st));
    hipFree(c->d_local); hipFree(c->d_gathered);
    c->d_local = nullptr; c->d_gathered = nullptr; c->cap_q = 0;
    ORB_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&c->d_local), sizeof(int) * 3 * (size_t)n_q));
    ORB_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&c->d_gathered), sizeof(int) * 3 * (size_t)n_q * c->world));
    c->cap_q = n_q;
  }
  int* lb = c->d_local;
  int rc = orb_knn2_device(c->device, d_q, n_q, d_db_shard, n_db_shard, db_index_base, ratio, lb, lb + n_q, lb + 2 * (size_t)n_q,
                           nullptr, stream);
  if (rc != ORBHIP_OK) return rc;
  // rank s's triple lands at d_gathered + s * 3 * n_q: [world][3][n_q]
  const int nrc = r->AllGather(lb, c->d_gathered, 3 * (size_t)n_q, /*ncclInt32*/ 2, c->comm, st);
  if (nrc != 0) return rccl_fail(r, nrc, "ncclAllGather");
  const int* gb = c->d_gathered;
  ORB_HIP_CHECK(launch_knn2_merge(c->world, n_q, 3 * (size_t)n_q, gb, gb + n_q, gb + 2 * (size_t)n_q, ratio, d_best, d_second,
                                  d_idx, d_accept, st));
  return ORBHIP_OK;
}

}  // extern "C"
