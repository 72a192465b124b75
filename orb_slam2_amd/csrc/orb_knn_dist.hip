// Brute-force Hamming kNN-2 over several GPUs through the C ABI: every rank scans its shard of
// the database (ascending row ranges, rank order) for all queries, one ncclAllGather exchanges the
// per-query results as 8-byte records (best | second << 16, global row: 8 bytes x queries per rank)
// over xGMI, and the merge kernel folds the shards in rank order, so that the earlier shard keeps
// ties: the first-minimum rule of a single scan, bit for bit.  The reference has no multi-device
// code; this is the one path here with an exchange step.
//
// RCCL is bound at run time (dlopen of librccl.so.1), so the library has no link-time dependency
// on it and a process that already holds an RCCL (PyTorch-ROCm ships one under the same soname)
// shares that copy.
#include <dlfcn.h>
#include <string.h>

#include <mutex>

#include "orb_common.h"

namespace orbhip {
hipError_t launch_knn2_pack(int nq, const int* best, const int* second, const int* idx, void* packed, hipStream_t st);
hipError_t launch_knn2_merge_packed(int nshards, int nq, const void* packed, float ratio, int* ob, int* os, int* oi,
                                    uint8_t* oa, hipStream_t st);
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
  int (*CommAbort)(void*) = nullptr;
  int (*CommCount)(void*, int*) = nullptr;
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
    r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(dlsym(r.lib, "ncclCommAbort"));
    r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(r.lib, "ncclCommCount"));
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
  int* d_local = nullptr;       // [3][cap_q]: this rank's best, second, global index
  void* d_packed = nullptr;     // [cap_q] 8-byte records of this rank
  void* d_gathered = nullptr;   // [world][n_q] records of all ranks
  int cap_q = 0;
  // the last call's timeline on its stream: start, shard scanned and packed, merged
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
  bool timed = false;
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
  hipFree(c->d_packed);
  hipFree(c->d_gathered);
  for (hipEvent_t e : c->ev) if (e) hipEventDestroy(e);
  if (c->owned && c->comm && rccl()) rccl()->CommDestroy(c->comm);
  delete c;
}

int orb_knn2_comm_info(OrbKnnComm* c, int* rank, int* world, int* ranks_in_collective) {
  if (!c) { set_error("null communicator"); return ORBHIP_ERR_ARG; }
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  if (ranks_in_collective) {
    const Rccl* r = rccl();
    int n = -1;
    if (r && r->CommCount) {
      const int rc = r->CommCount(c->comm, &n);
      if (rc != 0) return rccl_fail(r, rc, "ncclCommCount");
    }
    *ranks_in_collective = n;
  }
  return ORBHIP_OK;
}

int orb_knn2_sharded_times(OrbKnnComm* c, float* scan_ms, float* exchange_ms) {
  if (!c) { set_error("null communicator"); return ORBHIP_ERR_ARG; }
  if (!c->timed) { set_error("no orb_knn2_sharded call has run on this communicator"); return ORBHIP_ERR_ARG; }
  DeviceGuard g(c->device);
  ORB_HIP_CHECK(hipEventSynchronize(c->ev[2]));
  float a = 0.f, b = 0.f;
  ORB_HIP_CHECK(hipEventElapsedTime(&a, c->ev[0], c->ev[1]));
  ORB_HIP_CHECK(hipEventElapsedTime(&b, c->ev[1], c->ev[2]));
  if (scan_ms) *scan_ms = a;
  if (exchange_ms) *exchange_ms = b;
  return ORBHIP_OK;
}

int orb_knn2_sharded(OrbKnnComm* c, const uint8_t* d_q, int n_q, const uint8_t* d_db_shard, int n_db_shard,
                     int db_index_base, float ratio, int* d_best, int* d_second, int* d_idx, uint8_t* d_accept,
                     void* stream) {
  // Everything that can fail for reasons of this rank alone is checked before the first piece of
  // work, so that a rank either returns here - the caller's arguments are the same kind on every
  // rank, and n_q must be the same number - or goes on to the collective.
  if (!c) { set_error("null communicator"); return ORBHIP_ERR_ARG; }
  if (n_q <= 0) return ORBHIP_OK;
  if (!d_q || !d_best || !d_second || !d_idx || (n_db_shard > 0 && !d_db_shard) || n_db_shard < 0) {
    set_error("null argument"); return ORBHIP_ERR_ARG;
  }
  if (((uintptr_t)d_q | (uintptr_t)d_db_shard) & 15) { set_error("descriptor arrays must be 16-byte aligned"); return ORBHIP_ERR_ARG; }
  const Rccl* r = rccl();
  if (!r) { set_error("librccl.so.1 could not be loaded"); return ORBHIP_ERR_UNSUPPORTED; }
  DeviceGuard g(c->device);
  hipStream_t st = (hipStream_t)stream;
  // a failure from here on leaves the other ranks waiting in the collective: the communicator is
  // aborted (if it is the library's own), which makes their pending operation fail instead
  auto fail_collective = [&](int code) {
    if (c->owned && c->comm && r->CommAbort) { r->CommAbort(c->comm); c->comm = nullptr; }
    return code;
  };
  if (!c->comm) { set_error("the communicator was aborted by an earlier failure"); return ORBHIP_ERR_HIP; }
  for (hipEvent_t& e : c->ev)
    if (!e && hipEventCreate(&e) != hipSuccess) { e = nullptr; set_error("hipEventCreate failed"); return fail_collective(ORBHIP_ERR_HIP); }
  if (n_q > c->cap_q) {  // the first call, or more queries than any call before: (re)allocate the exchange buffers
    hipError_t e = hipStreamSynchronize(st);
    hipFree(c->d_local); hipFree(c->d_packed); hipFree(c->d_gathered);
    c->d_local = nullptr; c->d_packed = nullptr; c->d_gathered = nullptr; c->cap_q = 0;
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->d_local), sizeof(int) * 3 * (size_t)n_q);
    if (e == hipSuccess) e = hipMalloc(&c->d_packed, 8 * (size_t)n_q);
    if (e == hipSuccess) e = hipMalloc(&c->d_gathered, 8 * (size_t)n_q * c->world);
    if (e != hipSuccess) return fail_collective(hip_fail(e, "exchange buffers"));
    c->cap_q = n_q;
  }
  int* lb = c->d_local;
  hipError_t e = hipEventRecord(c->ev[0], st);
  int rc = ORBHIP_OK;
  if (e == hipSuccess)
    rc = orb_knn2_device(c->device, d_q, n_q, d_db_shard, n_db_shard, db_index_base, ratio, lb, lb + n_q, lb + 2 * (size_t)n_q,
                         nullptr, stream);
  if (rc != ORBHIP_OK) return fail_collective(rc);
  if (e == hipSuccess) e = launch_knn2_pack(n_q, lb, lb + n_q, lb + 2 * (size_t)n_q, c->d_packed, st);
  if (e == hipSuccess) e = hipEventRecord(c->ev[1], st);
  if (e != hipSuccess) return fail_collective(hip_fail(e, "shard scan"));
  // rank s's records land at d_gathered + s * n_q: [world][n_q], two 32-bit words each
  const int nrc = r->AllGather(c->d_packed, c->d_gathered, 2 * (size_t)n_q, /*ncclInt32*/ 2, c->comm, st);
  if (nrc != 0) return rccl_fail(r, nrc, "ncclAllGather");
  ORB_HIP_CHECK(launch_knn2_merge_packed(c->world, n_q, c->d_gathered, ratio, d_best, d_second, d_idx, d_accept, st));
  ORB_HIP_CHECK(hipEventRecord(c->ev[2], st));
  c->timed = true;
  return ORBHIP_OK;
}

}  // extern "C"
