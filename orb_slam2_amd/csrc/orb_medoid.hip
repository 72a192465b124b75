// MapPoint::ComputeDistinctiveDescriptors (reference src/MapPoint.cc:386-469) for many map
// points at once on gfx950: of the descriptors that observe a point, the one whose median
// Hamming distance to all of them (itself included) is smallest; the first such descriptor
// on ties ('median < BestMedian', :458).  The median is element (int)(0.5 * (N - 1)) of the
// sorted row (:455).
//
// Points with up to 64 observations (practically all of them) take one wavefront each:
// lane j keeps descriptor j in registers, descriptor i is broadcast with v_readlane, and the
// k-th smallest of the 64 distances is found by bisecting the value range with ballots.
// Larger points take a workgroup each and a 257-bin histogram in LDS.
#include "orb_common.h"
#include "orb_hostbuf.h"

#include <vector>

namespace orbhip {
namespace {

constexpr int MD_WAVES = 4;

__global__ __launch_bounds__(64 * MD_WAVES) void medoid_small_kernel(int n_sel, const int* __restrict__ sel,
                                                                    const int* __restrict__ start,
                                                                    const uint4* __restrict__ desc,
                                                                    int* __restrict__ best_idx,
                                                                    int* __restrict__ best_median) {
  const int slot = blockIdx.x * MD_WAVES + (threadIdx.x >> 6);
  if (slot >= n_sel) return;  // whole wave
  const int p = sel[slot];
  const int lane = threadIdx.x & 63;
  const int s = __builtin_amdgcn_readfirstlane(start[p]);
  const int n = __builtin_amdgcn_readfirstlane(start[p + 1]) - s;
  uint32_t d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (lane < n) {
    const uint4 a = desc[2 * (size_t)(s + lane)], b = desc[2 * (size_t)(s + lane) + 1];
    d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
  }
  const int k = (n - 1) >> 1;
  int best = 0x7FFFFFFF, bidx = 0;
  for (int i = 0; i < n; i++) {
    int dist = 0;
#pragma unroll
    for (int w = 0; w < 8; w++) dist += __popc(d[w] ^ (uint32_t)__builtin_amdgcn_readlane((int)d[w], i));
    if (lane >= n) dist = 0x7FFF;
    // smallest v with #(dist <= v) > k
    int lo = 0, hi = 256;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      const int cnt = __popcll(__builtin_amdgcn_ballot_w64(dist <= mid));
      if (cnt > k) hi = mid; else lo = mid + 1;
    }
    if (lo < best) { best = lo; bidx = i; }
  }
  if (lane == 0) {
    best_idx[p] = n > 0 ? bidx : -1;
    best_median[p] = n > 0 ? best : -1;
  }
}

__global__ __launch_bounds__(256) void medoid_large_kernel(int n_sel, const int* __restrict__ sel,
                                                           const int* __restrict__ start,
                                                           const uint4* __restrict__ desc, int* __restrict__ best_idx,
                                                           int* __restrict__ best_median) {
  __shared__ int hist[320];
  __shared__ int s_med;
  const int p = sel[blockIdx.x];
  const int s = start[p], n = start[p + 1] - s;
  const int k = (n - 1) >> 1;
  int best = 0x7FFFFFFF, bidx = 0;
  for (int i = 0; i < n; i++) {
    for (int b = threadIdx.x; b < 320; b += 256) hist[b] = 0;
    __syncthreads();
    const uint4 qa = desc[2 * (size_t)(s + i)], qb = desc[2 * (size_t)(s + i) + 1];
    for (int j = threadIdx.x; j < n; j += 256) {
      const uint4 a = desc[2 * (size_t)(s + j)], b = desc[2 * (size_t)(s + j) + 1];
      const int dist = __popc(a.x ^ qa.x) + __popc(a.y ^ qa.y) + __popc(a.z ^ qa.z) + __popc(a.w ^ qa.w) +
                       __popc(b.x ^ qb.x) + __popc(b.y ^ qb.y) + __popc(b.z ^ qb.z) + __popc(b.w ^ qb.w);
      atomicAdd(&hist[dist], 1);
    }
    __syncthreads();
    if (threadIdx.x < 64) {  // wave 0: lane l owns bins 5l .. 5l+4; the bin where the running count passes k
      const int l = threadIdx.x;
      int c[5], tot = 0;
#pragma unroll
      for (int u = 0; u < 5; u++) { c[u] = hist[5 * l + u]; tot += c[u]; }
      int incl = tot;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off, 64);
        if (l >= off) incl += v;
      }
      int run = incl - tot;
      if (run <= k && incl > k) {
#pragma unroll
        for (int u = 0; u < 5; u++) {
          if (run <= k && run + c[u] > k) s_med = 5 * l + u;
          run += c[u];
        }
      }
    }
    __syncthreads();
    const int med = s_med;
    if (med < best) { best = med; bidx = i; }
  }
  if (threadIdx.x == 0) { best_idx[p] = bidx; best_median[p] = best; }
}

}  // namespace
}  // namespace orbhip

using namespace orbhip;

extern "C" int orb_distinctive_descriptors(int device_id, int n_points, const int* start, const uint8_t* desc,
                                           int* best_idx, int* best_median) {
  if (n_points <= 0) return ORBHIP_OK;
  if (!start || !best_idx) { set_error("null argument"); return ORBHIP_ERR_ARG; }
  std::vector<int> small, large;
  if (start[0] != 0) { set_error("start[0] must be 0"); return ORBHIP_ERR_ARG; }
  for (int p = 0; p < n_points; p++) {
    const int n = start[p + 1] - start[p];
    if (n < 0) { set_error("start[] must not decrease"); return ORBHIP_ERR_ARG; }
    if (n == 0) { best_idx[p] = -1; if (best_median) best_median[p] = -1; }
    else (n <= 64 ? small : large).push_back(p);
  }
  const int total = start[n_points];
  if (total == 0) return ORBHIP_OK;
  if (!desc) { set_error("null descriptors"); return ORBHIP_ERR_ARG; }
  // a leased stream and arena, like the matchers: MapPoint::ComputeDistinctiveDescriptors runs on the
  // mapping thread beside the tracking thread's calls and must not wait for them
  DevSel g(device_id);
  DevBuf buf;
  int *d_start = nullptr, *d_sel = nullptr, *d_out = nullptr;
  uint8_t* d_desc = nullptr;
  const int nsel = (int)(small.size() + large.size());
  std::vector<int> sel(small);
  sel.insert(sel.end(), large.begin(), large.end());
  std::vector<int> out(2 * (size_t)n_points, -1);
  UP(buf, start, (size_t)n_points + 1, d_start);
  UP(buf, sel.data(), (size_t)nsel, d_sel);
  UP(buf, desc, (size_t)32 * total, d_desc);
  UP(buf, (const int*)nullptr, 2 * (size_t)n_points, d_out);
  ORB_HIP_CHECK(hipMemsetAsync(d_out, 0xFF, sizeof(int) * 2 * (size_t)n_points, buf.stream()));
  if (!small.empty()) {
    hipLaunchKernelGGL(medoid_small_kernel, dim3(((unsigned)small.size() + MD_WAVES - 1) / MD_WAVES), dim3(64 * MD_WAVES), 0,
                       buf.stream(), (int)small.size(), d_sel, d_start, reinterpret_cast<const uint4*>(d_desc), d_out,
                       d_out + n_points);
    ORB_HIP_CHECK(hipGetLastError());
  }
  if (!large.empty()) {
    hipLaunchKernelGGL(medoid_large_kernel, dim3((unsigned)large.size()), dim3(256), 0, buf.stream(), (int)large.size(),
                       d_sel + small.size(), d_start, reinterpret_cast<const uint4*>(d_desc), d_out, d_out + n_points);
    ORB_HIP_CHECK(hipGetLastError());
  }
  DOWN(buf, out.data(), d_out, 2 * (size_t)n_points);
  SYNC(buf);
  for (int p = 0; p < n_points; p++) {
    best_idx[p] = out[p];
    if (best_median) best_median[p] = out[(size_t)n_points + p];
  }
  return ORBHIP_OK;
}
