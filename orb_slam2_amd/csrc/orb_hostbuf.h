// Host-side plumbing shared by the host-buffer entry points: a leased per-call context (stream +
// device arena) and bump allocation of a call's device temporaries from it.
#pragma once
#include <string.h>

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
  // pinned host mirror of the arena's first hcap bytes: a call's small arrays are packed into it and
  // go up in ONE transfer (and come back in one), instead of one transfer per array - each transfer
  // from or to pageable memory costs the calling thread several microseconds
  unsigned char* hstage = nullptr;
  size_t hcap = 0;
};
constexpr size_t kStageMaxBytes = (size_t)4 << 20;   // the mirror is at most this large
constexpr size_t kPackMaxBytes = (size_t)1 << 20;    // arrays up to this size go through it

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
      const size_t hn = std::min(c->cap, kStageMaxBytes);
      if (hn > c->hcap) {
        if (c->hstage) hipHostFree(c->hstage);
        c->hstage = nullptr; c->hcap = 0;
        if (hipHostMalloc(reinterpret_cast<void**>(&c->hstage), hn, hipHostMallocDefault) == hipSuccess) c->hcap = hn;
        else (void)hipGetLastError();
      }
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
  size_t pend_lo = ~(size_t)0, pend_hi = 0;  // arena range packed into the pinned mirror and not yet sent
  struct Pull { void* host; size_t off, bytes; };
  std::vector<Pull> pulls;                   // results to fetch through the mirror at sync()
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
  // sends what up() has packed so far (one transfer)
  hipError_t flush() {
    if (pend_hi <= pend_lo) return hipSuccess;
    const hipError_t e = hipMemcpyAsync(ctx->arena + pend_lo, ctx->hstage + pend_lo, pend_hi - pend_lo, hipMemcpyHostToDevice,
                                        ctx->stream);
    pend_lo = ~(size_t)0; pend_hi = 0;
    return e;
  }
  // The call's stream, for its kernel launches: whatever up() has packed is on its way first.  (The
  // transfer covers the arena from the first to the last packed array, blocks without a host source
  // in between included: those receive stale bytes, before any kernel of the call can write them.)
  hipStream_t stream() {
    if (!ctx) return nullptr;
    if (flush() != hipSuccess) (void)hipGetLastError();  // the launch that follows reports the stream's state
    return ctx->stream;
  }
  template <typename T, typename U>
  int up(const T* host, size_t n, U** out) {
    static_assert(sizeof(T) == sizeof(U), "element size mismatch");
    *out = nullptr;
    if (!ctx) { set_error("could not create a stream"); return ORBHIP_ERR_HIP; }
    const size_t bytes = (std::max<size_t>(n, 1) * sizeof(T) + 255) & ~(size_t)255;
    void* d = nullptr;
    total += bytes;
    bool in_arena = false;
    size_t off = 0;
    if (used + bytes <= ctx->cap) {
      off = used;
      d = ctx->arena + used;
      used += bytes;
      in_arena = true;
    } else {
      hipError_t e = ctx_alloc(ctx, &d, bytes);
      if (e != hipSuccess) return hip_fail(e, "hipMalloc");
      extra.push_back(d);
    }
    if (host && n) {
      const size_t nb = n * sizeof(T);
      if (in_arena && nb <= kPackMaxBytes && off + nb <= ctx->hcap) {
        memcpy(ctx->hstage + off, host, nb);
        pend_lo = std::min(pend_lo, off);
        pend_hi = std::max(pend_hi, off + nb);
      } else {
        // a transfer of its own; what is packed so far goes first, so that its range cannot reach
        // across this array later
        hipError_t e = flush();
        if (e == hipSuccess) e = hipMemcpyAsync(d, host, nb, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync H2D");
      }
    }
    *out = reinterpret_cast<U*>(d);
    return ORBHIP_OK;
  }
  // results back to the host; valid after sync()
  template <typename T>
  int down(T* host, const T* dev, size_t n) {
    if (!n) return ORBHIP_OK;
    const size_t nb = n * sizeof(T);
    const unsigned char* p = reinterpret_cast<const unsigned char*>(dev);
    if (ctx->arena && p >= ctx->arena && nb <= kPackMaxBytes && (size_t)(p - ctx->arena) + nb <= ctx->hcap) {
      pulls.push_back(Pull{host, (size_t)(p - ctx->arena), nb});
      return ORBHIP_OK;
    }
    hipError_t e = hipMemcpyAsync(host, dev, nb, hipMemcpyDeviceToHost, ctx->stream);
    if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync D2H");
    return ORBHIP_OK;
  }
  int sync() {
    hipError_t e = flush();
    if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync H2D");
    size_t lo = ~(size_t)0, hi = 0, sum = 0;
    for (const Pull& q : pulls) { lo = std::min(lo, q.off); hi = std::max(hi, q.off + q.bytes); sum += q.bytes; }
    // one transfer for all of them unless they lie far apart in the arena
    const bool together = !pulls.empty() && hi - lo <= 2 * sum + 65536;
    if (together) e = hipMemcpyAsync(ctx->hstage + lo, ctx->arena + lo, hi - lo, hipMemcpyDeviceToHost, ctx->stream);
    else
      for (const Pull& q : pulls)
        if (e == hipSuccess) e = hipMemcpyAsync(q.host, ctx->arena + q.off, q.bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync D2H");
    e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return hip_fail(e, "hipStreamSynchronize");
    if (together)
      for (const Pull& q : pulls) memcpy(q.host, ctx->hstage + q.off, q.bytes);
    pulls.clear();
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
