// Host-side plumbing shared by the host-buffer entry points: a leased per-call context (stream +
// device arena) and bump allocation of a call's device temporaries from it.
#pragma once
#include <algorithm>
#include <mutex>
#include <vector>

#include "orb_common.h"

namespace orbhip {

// The host-buffer entry points may be called from several threads at once (the reference calls
// its matchers from the tracking, mapping and loop-closing threads).  Each call leases a context
// with a stream of its own and a device arena for its temporaries, so calls neither wait for one
// another nor for the extractor: nothing here uses the null stream or synchronises the device.
// Contexts go back to a pool and are reused by whichever thread calls next.
struct CallCtx {
  int dev = 0;
  bool pooled = false;  // the device has stream-ordered memory pools (hipMallocAsync / hipFreeAsync)
  hipStream_t stream = nullptr;
  unsigned char* arena = nullptr;
  size_t cap = 0;   // bytes of the arena
  size_t want = 0;  // bytes the largest call so far needed
};

inline hipError_t ctx_alloc(CallCtx* c, void** p, size_t bytes) {
  return c->pooled ? hipMallocAsync(p, bytes, c->stream) : hipMalloc(p, bytes);
}
inline void ctx_free(CallCtx* c, void* p) {
  if (c->pooled) hipFreeAsync(p, c->stream);
  else hipFree(p);
}

class CtxPool {
 public:
  CallCtx* acquire(int dev) {
    CallCtx* c = nullptr;
    {
      std::lock_guard<std::mutex> l(m_);
      auto& fl = free_[dev & 63];
      if (!fl.empty()) { c = fl.back(); fl.pop_back(); }
    }
    if (!c) {
      c = new CallCtx();
      c->dev = dev;
      if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return nullptr; }
      int pools = 0;
      const char* env = getenv("ORBHIP_ASYNC_ALLOC");
      if (hipDeviceGetAttribute(&pools, hipDeviceAttributeMemoryPoolsSupported, dev) != hipSuccess) pools = 0;
      c->pooled = pools != 0 && !(env && env[0] == '0');
    }
    if (c->want > c->cap) {
      // grow between calls: nothing of this context is in flight.  Allocation and release are ordered
      // on the context's own stream (hipMalloc / hipFree would wait for every stream of the device,
      // i.e. for the other threads' calls)
      if (c->arena) ctx_free(c, c->arena);
      c->arena = nullptr; c->cap = 0;
      const size_t n = c->want + c->want / 4 + 4096;
      if (ctx_alloc(c, reinterpret_cast<void**>(&c->arena), n) == hipSuccess) c->cap = n;
    }
    return c;
  }
  void release(CallCtx* c) {
    std::lock_guard<std::mutex> l(m_);
    free_[c->dev & 63].push_back(c);
  }

 private:
  std::mutex m_;
  std::vector<CallCtx*> free_[64];
};
CtxPool& ctx_pool();  // one pool per process, defined next to the matchers

// Device temporaries of one call: bump allocation from the leased arena (blocks that do not fit
// are allocated on their own and the arena grows before the next call), copies on the call's stream.
struct DevBuf {
  CallCtx* ctx = nullptr;
  size_t used = 0, total = 0;
  std::vector<void*> extra;
  DevBuf() {
    int dev = 0;
    hipGetDevice(&dev);
    ctx = ctx_pool().acquire(dev);
  }
  ~DevBuf() {
    if (!ctx) return;
    if (!extra.empty()) {
      if (!ctx->pooled) hipStreamSynchronize(ctx->stream);
      for (void* p : extra) ctx_free(ctx, p);
    }
    ctx->want = std::max(ctx->want, total);
    ctx_pool().release(ctx);
  }
  hipStream_t stream() const { return ctx ? ctx->stream : nullptr; }
  template <typename T, typename U>
  int up(const T* host, size_t n, U** out) {
    static_assert(sizeof(T) == sizeof(U), "element size mismatch");
    *out = nullptr;
    if (!ctx) { set_error("could not create a stream"); return ORBHIP_ERR_HIP; }
    const size_t bytes = (std::max<size_t>(n, 1) * sizeof(T) + 255) & ~(size_t)255;
    void* d = nullptr;
    total += bytes;
    if (used + bytes <= ctx->cap) {
      d = ctx->arena + used;
      used += bytes;
    } else {
      hipError_t e = ctx_alloc(ctx, &d, bytes);
      if (e != hipSuccess) return hip_fail(e, "hipMalloc");
      extra.push_back(d);
    }
    if (host && n) {
      hipError_t e = hipMemcpyAsync(d, host, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream);
      if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync H2D");
    }
    *out = reinterpret_cast<U*>(d);
    return ORBHIP_OK;
  }
  // results back to the host; valid after sync()
  template <typename T>
  int down(T* host, const T* dev, size_t n) {
    if (!n) return ORBHIP_OK;
    hipError_t e = hipMemcpyAsync(host, dev, n * sizeof(T), hipMemcpyDeviceToHost, ctx->stream);
    if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync D2H");
    return ORBHIP_OK;
  }
  int sync() {
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return hip_fail(e, "hipStreamSynchronize");
    return ORBHIP_OK;
  }
};

struct DevSel {
  int prev = -1;
  explicit DevSel(int dev) { hipGetDevice(&prev); if (prev != dev) hipSetDevice(dev); else prev = -1; }
  ~DevSel() { if (prev >= 0) hipSetDevice(prev); }
};

#define UP(buf, host, n, out)                         \
  do {                                                \
    int rc__ = (buf).up((host), (size_t)(n), &(out)); \
    if (rc__ != ORBHIP_OK) return rc__;               \
  } while (0)
#define DOWN(buf, host, dev, n)                               \
  do {                                                        \
    int rc__ = (buf).down((host), (dev), (size_t)(n));        \
    if (rc__ != ORBHIP_OK) return rc__;                       \
  } while (0)
#define SYNC(buf)                       \
  do {                                  \
    int rc__ = (buf).sync();            \
    if (rc__ != ORBHIP_OK) return rc__; \
  } while (0)


}  // namespace orbhip
