// Hamming matching on gfx950: DescriptorDistance batches, brute-force kNN-2, and the
// candidate scans of ORBmatcher::SearchByProjection / SearchForTriangulation.
//
// The reference matchers are greedy: a train keypoint taken by an earlier query is
// skipped by later ones.  Each matcher kernel therefore has a parallel phase (every
// query scans its candidates and keeps its few best, ordered exactly like the
// reference's sequential scan would rank them, and the queries that share a candidate
// are joined into components) and an ordered phase that replays the reference's query
// order on those lists: component by component, the components spread over the
// workgroup's wavefronts, 64 queries per wavefront and step.
#include <algorithm>
#include <atomic>
#include <mutex>
#include <vector>

#include "orb_common.h"
#include "orb_hostbuf.h"

namespace orbhip {
hipError_t launch_knn2_mfma(const uint8_t* d_q, int n_q, const uint8_t* d_db, int n_db, int index_base, float ratio,
                            int* d_best, int* d_second, int* d_idx, uint8_t* d_accept, hipStream_t st);

constexpr int TH_HIGH = 100, TH_LOW = 50, HISTO_LENGTH = 30;  // reference src/ORBmatcher.cc:41-43
constexpr int GRID_COLS = 64, GRID_ROWS = 48;                 // reference include/Frame.h:47,52

__device__ __forceinline__ int hamming256(const uint32_t* a, const uint4 b0, const uint4 b1) {
  int d = __popc(a[0] ^ b0.x);
  d += __popc(a[1] ^ b0.y);
  d += __popc(a[2] ^ b0.z);
  d += __popc(a[3] ^ b0.w);
  d += __popc(a[4] ^ b1.x);
  d += __popc(a[5] ^ b1.y);
  d += __popc(a[6] ^ b1.z);
  d += __popc(a[7] ^ b1.w);
  return d;
}

// ---------------------------------------------------------------------------
// DescriptorDistance for n pairs (reference src/ORBmatcher.cc:1991-2011)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hamming_pairs_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b,
                                                            int n, int* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint4 a0 = a[2 * i], a1 = a[2 * i + 1];
  const uint32_t q[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
  out[i] = hamming256(q, b[2 * i], b[2 * i + 1]);
}

// ---------------------------------------------------------------------------
// Brute-force kNN-2.  A thread owns KQ queries in registers; database rows are
// staged in LDS in tiles and read back as wave-uniform (broadcast) 16-byte words.
// Per pair: 8 xor + 8 accumulating popcounts + 3 for the running two smallest
// keys, where key = distance << 22 | row (so the first minimum wins ties, as in
// the reference's sequential scans with a strict '<').
// ---------------------------------------------------------------------------
constexpr int KNN_THREADS = 256;
constexpr int KNN_KQ = 4;
constexpr int KNN_TILE = 512;        // database rows per LDS tile
constexpr int KNN_ROWBITS = 22;      // rows per chunk: keys hold a chunk-local row index
constexpr int KNN_CHUNK = 1 << KNN_ROWBITS;

__global__ __launch_bounds__(KNN_THREADS) void knn2_kernel(const uint4* __restrict__ q, int nq,
                                                           const uint4* __restrict__ db, int ndb, int index_base,
                                                           float ratio, int* __restrict__ out_best,
                                                           int* __restrict__ out_second, int* __restrict__ out_idx,
                                                           uint8_t* __restrict__ out_accept) {
  __shared__ uint4 s_db[2 * KNN_TILE + 4];  // + two rows read ahead by the software pipeline
  const int tid = threadIdx.x;
  const int q0 = (blockIdx.x * KNN_THREADS + tid) * KNN_KQ;
  uint32_t qr[KNN_KQ][8];
#pragma unroll
  for (int k = 0; k < KNN_KQ; k++) {
    const int qi = min(q0 + k, nq - 1);
    const uint4 a0 = q[2 * qi], a1 = q[2 * qi + 1];
    qr[k][0] = a0.x; qr[k][1] = a0.y; qr[k][2] = a0.z; qr[k][3] = a0.w;
    qr[k][4] = a1.x; qr[k][5] = a1.y; qr[k][6] = a1.z; qr[k][7] = a1.w;
  }
  int bd[KNN_KQ], sd[KNN_KQ], bi[KNN_KQ];
#pragma unroll
  for (int k = 0; k < KNN_KQ; k++) { bd[k] = 256; sd[k] = 256; bi[k] = -1; }

  for (int cbase = 0; cbase < ndb; cbase += KNN_CHUNK) {
    const int cn = min(KNN_CHUNK, ndb - cbase);
    uint32_t k1[KNN_KQ], k2[KNN_KQ], s2d[KNN_KQ];  // s2d = distance part of k2
#pragma unroll
    for (int k = 0; k < KNN_KQ; k++) { k1[k] = 0xFFFFFFFFu; k2[k] = 0xFFFFFFFFu; s2d[k] = 0xFFFFFFFFu >> KNN_ROWBITS; }
    for (int t0 = 0; t0 < cn; t0 += KNN_TILE) {
      const int tn = min(KNN_TILE, cn - t0);
      __syncthreads();
      for (int i = tid; i < 2 * tn; i += KNN_THREADS) s_db[i] = db[2 * (size_t)(cbase + t0) + i];
      __syncthreads();
      // one row against the thread's queries
      auto score = [&](const uint4& b0, const uint4& b1, const uint32_t row) {
        const uint32_t bw[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        uint32_t dk[KNN_KQ];
        bool need = false;
#pragma unroll
        for (int k = 0; k < KNN_KQ; k++) {
          // 8 xor + 8 popcounts chained through v_bcnt's accumulator operand (the compiler
          // would otherwise split the chain and add the pieces with extra instructions)
          uint32_t d;
          asm("v_bcnt_u32_b32 %0, %1, 0" : "=v"(d) : "v"(qr[k][0] ^ bw[0]));
#pragma unroll
          for (int w = 1; w < 8; w++) {
            const uint32_t x = qr[k][w] ^ bw[w];
            asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(d) : "v"(x));
          }
          dk[k] = d;
          need = need || d < s2d[k];
        }
        // Rows come in ascending order, so a row changes (best, second) only if its distance
        // is below the second-best distance; after the first few hundred rows that is rare,
        // and the whole wavefront skips the update.
        if (__builtin_amdgcn_ballot_w64(need) != 0) {
#pragma unroll
          for (int k = 0; k < KNN_KQ; k++) {
            const uint32_t key = (dk[k] << KNN_ROWBITS) | row;
            // two smallest keys: k1 <= k2 always, so the new second is the median of the three
            asm("v_med3_u32 %0, %1, %0, %2" : "+v"(k2[k]) : "v"(k1[k]), "v"(key));
            k1[k] = min(k1[k], key);
            s2d[k] = k2[k] >> KNN_ROWBITS;
          }
        }
      };
      // Two rows per step, each read into its own registers one row ahead (a single set of
      // registers that is handed on costs nine moves per row).
      uint4 n0 = s_db[0], n1 = s_db[1], m0, m1;
      for (int j = 0; j < tn; j += 2) {
        m0 = s_db[2 * j + 2]; m1 = s_db[2 * j + 3];
        score(n0, n1, (uint32_t)(t0 + j));
        n0 = s_db[2 * j + 4]; n1 = s_db[2 * j + 5];
        if (j + 1 < tn) score(m0, m1, (uint32_t)(t0 + j + 1));
      }
    }
    // fold the chunk into the running (best, second, index): an earlier chunk keeps ties
#pragma unroll
    for (int k = 0; k < KNN_KQ; k++) {
      const int cb = k1[k] == 0xFFFFFFFFu ? 256 : (int)(k1[k] >> KNN_ROWBITS);
      const int cs = k2[k] == 0xFFFFFFFFu ? 256 : (int)(k2[k] >> KNN_ROWBITS);
      const int ci = (int)(k1[k] & (KNN_CHUNK - 1)) + cbase;
      if (cb < bd[k]) { sd[k] = min(bd[k], cs); bd[k] = cb; bi[k] = ci; }
      else sd[k] = min(sd[k], cb);
    }
  }
#pragma unroll
  for (int k = 0; k < KNN_KQ; k++) {
    const int qi = q0 + k;
    if (qi < nq) {
      out_best[qi] = bd[k];
      out_second[qi] = sd[k];
      out_idx[qi] = bi[k] < 0 ? -1 : bi[k] + index_base;
      if (out_accept) out_accept[qi] = (uint8_t)(bi[k] >= 0 && (float)bd[k] < __fmul_rn(ratio, (float)sd[k]));
    }
  }
}

// Merge of per-shard results, shards in ascending database order: equals one scan of
// the concatenated database (earlier shard keeps ties).
__global__ __launch_bounds__(256) void knn2_merge_kernel(int nshards, int nq, size_t stride, const int* __restrict__ best,
                                                         const int* __restrict__ second, const int* __restrict__ idx,
                                                         float ratio, int* __restrict__ ob, int* __restrict__ os,
                                                         int* __restrict__ oi, uint8_t* __restrict__ oa) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nq) return;
  int bd = 256, sd = 256, bi = -1;
  for (int s = 0; s < nshards; s++) {
    const int cb = best[s * stride + i], cs = second[s * stride + i], ci = idx[s * stride + i];  // stride: entries from one shard to the next
    if (ci >= 0 && cb < bd) { sd = min(bd, cs); bd = cb; bi = ci; }
    else if (ci >= 0) sd = min(sd, cb);
  }
  ob[i] = bd; os[i] = sd; oi[i] = bi;
  if (oa) oa[i] = (uint8_t)(bi >= 0 && (float)bd < __fmul_rn(ratio, (float)sd));
}

hipError_t launch_knn2_merge(int nshards, int nq, size_t shard_stride, const int* best, const int* second, const int* idx,
                             float ratio, int* ob, int* os, int* oi, uint8_t* oa, hipStream_t st) {
  hipLaunchKernelGGL(knn2_merge_kernel, dim3((nq + 255) / 256), dim3(256), 0, st, nshards, nq, shard_stride, best, second,
                     idx, ratio, ob, os, oi, oa);
  return hipGetLastError();
}

// The same merge on the 8-byte records the ranks of a multi-GPU search exchange: per query
// x = best | second << 16 (distances are at most 256), y = global row (0xFFFFFFFF = none).
__global__ __launch_bounds__(256) void knn2_pack_kernel(int nq, const int* __restrict__ best, const int* __restrict__ second,
                                                        const int* __restrict__ idx, uint2* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nq) return;
  out[i] = make_uint2((uint32_t)best[i] | ((uint32_t)second[i] << 16), (uint32_t)idx[i]);
}

__global__ __launch_bounds__(256) void knn2_merge_packed_kernel(int nshards, int nq, const uint2* __restrict__ rec,
                                                                float ratio, int* __restrict__ ob, int* __restrict__ os,
                                                                int* __restrict__ oi, uint8_t* __restrict__ oa) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nq) return;
  int bd = 256, sd = 256, bi = -1;
  for (int s = 0; s < nshards; s++) {
    const uint2 r = rec[(size_t)s * nq + i];
    const int cb = (int)(r.x & 0xFFFFu), cs = (int)(r.x >> 16), ci = (int)r.y;
    if (ci >= 0 && cb < bd) { sd = min(bd, cs); bd = cb; bi = ci; }
    else if (ci >= 0) sd = min(sd, cb);
  }
  ob[i] = bd; os[i] = sd; oi[i] = bi;
  if (oa) oa[i] = (uint8_t)(bi >= 0 && (float)bd < __fmul_rn(ratio, (float)sd));
}

hipError_t launch_knn2_pack(int nq, const int* best, const int* second, const int* idx, void* packed, hipStream_t st) {
  hipLaunchKernelGGL(knn2_pack_kernel, dim3((nq + 255) / 256), dim3(256), 0, st, nq, best, second, idx,
                     reinterpret_cast<uint2*>(packed));
  return hipGetLastError();
}

hipError_t launch_knn2_merge_packed(int nshards, int nq, const void* packed, float ratio, int* ob, int* os, int* oi,
                                    uint8_t* oa, hipStream_t st) {
  hipLaunchKernelGGL(knn2_merge_packed_kernel, dim3((nq + 255) / 256), dim3(256), 0, st, nshards, nq,
                     reinterpret_cast<const uint2*>(packed), ratio, ob, os, oi, oa);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Greedy matchers
// ---------------------------------------------------------------------------
constexpr int MK = 4;  // ranked candidates kept per query
constexpr int M_THREADS = 1024;
enum { MODE_FRAMES = 0, MODE_MAPPOINTS = 1, MODE_TRIANG = 2, MODE_WINDOW = 3, MODE_BOW = 4, MODE_BEST = 5, MODE_INIT = 6 };
// TRIANG and BOW take their candidates from vocabulary-node buckets instead of the grid and
// index the output by the feature of side 1
constexpr bool mode_bucketed(int mode) { return mode == MODE_TRIANG || mode == MODE_BOW; }
// the output is indexed by the query instead of the train keypoint
constexpr bool mode_out_by_query(int mode) { return mode_bucketed(mode) || mode == MODE_BEST || mode == MODE_INIT; }

struct MatchArgs {
  int max_q, max_t, nlevels, check_ori;
  int keyframe_origin;  // window queries subtract the int-truncated origin (KeyFrame::GetFeaturesInArea)
  int lists_in_lds;  // the ranked candidate lists of all queries fit in LDS next to the grid
  int desc_in_lds;   // the train descriptors are staged in LDS as well (small frames only: see launch_match)
  int queue_cap;     // entries per thread in the window scans' queue (set by launch_match)
  int ts, qs;  // element stride (4-byte units) of t_x/t_y/t_oct/t_angle and of q_oct/q_angle:
               // 1 for separate arrays, 6 for fields of OrbKeyPoint records
  const int* n_q; const int* n_t;
  // train side (current frame / frame F / keyframe 2)
  const float* t_x; const float* t_y; const int* t_oct; const float* t_angle;
  const uint8_t* t_desc; const int* t_obs;       // obs: -1 = no map point (TRIANG: 0 / 1 = has map point)
  // query side
  const int* q_obs; const float* q_xw; const uint8_t* q_desc; const int* q_oct; const float* q_angle;
  const float* q_x; const float* q_y; const float* q_viewcos; const uint8_t* q_track;
  const float* q_r; const int* q_minl; const int* q_maxl;  // WINDOW: radius and level range per query
  int max_dist;                                            // WINDOW / BOW: accept threshold
  double chi2;                                             // BEST: reprojection gate (0 = off), with sigma2[] = inverse level sigma2
  int* out_dist;                                           // BEST: distance of the best candidate per query
  const int* q_idx1; const int* q_b0; const int* q_b1; const int* items2;  // TRIANG buckets
  const float* tcw;
  float fx, fy, cx, cy;
  OrbFrameBounds b;
  float q_min_x, q_min_y;  // origin the window queries subtract: b.min_x / b.min_y for a Frame, truncated to int for a KeyFrame
  float sf[kMaxLevels];
  float sigma2[kMaxLevels];
  float f12[9];
  float ex, ey, th, nnratio;
  // scratch, per pair slices of max_q
  uint32_t* lists; int* q_ntot; float4* q_geo; int* q_slot; int* q_bin;
  // frames with more train keypoints than fit in LDS (match_kernel<MODE, true>): per pair a block of
  // big_stride bytes that holds the keypoint coordinates, grid items and claim / owner words
  unsigned char* big; size_t big_stride;
  // outputs
  int* out_match; int* n_matches;
};

// A ranked candidate is ONE word, distance << 16 | tie-break, smaller = better:
//   grid modes: the tie-break is the candidate's slot in the grid-ordered arrays - the reference
//     visits a window in that order and the first of equal distances wins - and the slot also
//     names the keypoint (gmeta[slot] & 0xFFFF);
//   SearchByBoW: the position in the vocabulary node's bucket (first minimum wins);
//   SearchForTriangulation: 0xFFFF - position (the reference keeps the LAST of equal distances).
// 0xFFFFFFFF = no candidate.  Four bytes per kept candidate instead of eight halve the lists, the
// largest thing the replay needs on chip.
struct Cand { uint32_t key[MK]; int ntot; };
constexpr uint32_t kNoCand = 0xFFFFFFFFu;

__device__ __forceinline__ void cand_init(Cand& c) {
#pragma unroll
  for (int k = 0; k < MK; k++) c.key[k] = kNoCand;
  c.ntot = 0;
}
__device__ __forceinline__ void cand_insert(Cand& c, uint32_t key) {
  c.ntot++;
  if (key >= c.key[MK - 1]) return;
  c.key[MK - 1] = key;
#pragma unroll
  for (int k = MK - 1; k > 0; k--) {
    if (c.key[k] < c.key[k - 1]) { const uint32_t tk = c.key[k]; c.key[k] = c.key[k - 1]; c.key[k - 1] = tk; }
  }
}

struct MatchShared {
  int* start;            // GRID_COLS*GRID_ROWS + 1
  unsigned short* queue; // queue_cap * M_THREADS: per thread, the slots of the items found in a window that still need their distance
  // Grid modes: the train keypoints in GRID ORDER (cell by cell, ascending index inside a cell), as
  // the scans walk them: coordinates in one 8-byte word, and index | octave << 16 | initial
  // occupancy << 24 in another, so that an item outside the query's window - four out of five
  // visited - costs one LDS read.  Bucketed modes: coordinates by keypoint index in kx / ky.
  float2* gxy;           // max_t
  uint32_t* gmeta;       // max_t
  float* kx; float* ky;  // max_t (bucketed modes; the same memory as gxy)
  unsigned char* koct;   // max_t
  unsigned char* occ;    // max_t: train keypoint can no longer be taken
  int* claim;            // max_t: lowest pending lane that wants to take the keypoint
  const uint4* tdesc;    // train descriptors, two uint4 each (LDS copy if it fits, else global)
  int* owner;            // INIT: query that currently holds the train keypoint, -1 = none
  uint32_t* lists;       // max_q * MK ranked candidates (LDS if it fits, else global scratch)
  int* ntot;             // max_q: candidates seen (bit 30: the query's map point has observations), -1 = inactive
};

// Words of the replay that several lanes of the replaying wavefront write and read in turn (claims,
// owners).  In LDS plain accesses do; in global memory (BIG) a plain load may be served by the
// vector cache while the atomic it follows was performed in L2, so loads and stores go to L2.
template <bool BIG>
__device__ __forceinline__ int word_load(const int* p) {
  if constexpr (BIG) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}
template <bool BIG>
__device__ __forceinline__ void word_store(int* p, int v) {
  if constexpr (BIG) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}

// Components of the conflict graph.  Two queries can influence each other only through a train
// keypoint both can reach (one takes it, the other finds it taken), so the sequential loop of the
// reference falls apart into independent loops, one per connected component of the graph "query -
// candidate keypoint".  The components are found with a union-find over the train keypoints, kept
// in S.claim: a word holds the keypoint's parent in its low 20 bits; the bits above are the lane
// field the replay uses (kNoLane when idle).  Phase A unites every candidate a query ranks with the
// query's first one; a pass after it points every word at its root.
constexpr int kUfMask = 0xFFFFF, kNoLane = 0x7FF << 20;
template <bool BIG>
__device__ __forceinline__ int uf_find(int* P, int x) {
  for (;;) {
    const int p = word_load<BIG>(&P[x]) & kUfMask;
    if (p == x) return x;
    x = p;
  }
}
template <bool BIG>
__device__ __forceinline__ void uf_union(int* P, int a, int b) {
  for (;;) {
    a = uf_find<BIG>(P, a);
    b = uf_find<BIG>(P, b);
    if (a == b) return;
    if (a < b) { const int t = a; a = b; b = t; }  // the larger root goes under the smaller one
    if (atomicCAS(&P[a], kNoLane | a, kNoLane | b) == (kNoLane | a)) return;
    // (somebody else moved a meanwhile: look again)
  }
}

// Frame::GetFeaturesInArea (reference src/Frame.cc:371-449) fused with the best /
// second scan of the caller.  Visits candidates in the reference's order
// (grid column, grid row, insertion order) and ranks them by (distance, visit order).
// Largest distance at which a candidate can still change a decision of the replay: one
// beyond the accept threshold is never matched, and as a second-best it weakens a match only
// if nnratio * distance does not exceed the best one's distance, which is at most the
// threshold.  Candidates beyond it are not ranked at all, so crowded windows of unrelated
// descriptors (distance around 128) neither fill the kept lists nor cause new scans.
template <int MODE>
__device__ __forceinline__ int scan_dist_cap(const MatchArgs& A) {
  auto with_ratio = [&](int th) {
    if (!(A.nnratio > 0.f)) return 256;
    const float c = (float)th / A.nnratio + 2.0f;
    return c < 256.f ? (int)c : 256;
  };
  if (MODE == MODE_FRAMES) return TH_HIGH;
  if (MODE == MODE_WINDOW) return A.max_dist;
  if (MODE == MODE_MAPPOINTS) return with_ratio(TH_HIGH);
  if (MODE == MODE_INIT) return with_ratio(TH_LOW);
  if (MODE == MODE_BOW) return with_ratio(A.max_dist);
  return 256;
}

constexpr int kQueueMin = 2, kQueueMax = 8;  // entries per thread in the window scans' queues
constexpr int kWorkBins = 256;
static_assert(kWorkBins % 64 == 0 && kWorkBins < M_THREADS, "one wavefront scans the bins; the first threads clear them");

// Cells of the grid that a query's window touches, Frame::GetFeaturesInArea (reference
// src/Frame.cc:376-391); false if it lies outside the grid.
__device__ __forceinline__ bool window_cells(const MatchArgs& A, const float4 geo, int& cx0, int& cx1, int& cy0, int& cy1) {
  const float x = geo.x, y = geo.y, r = geo.z;
  const float invW = (float)GRID_COLS / (A.b.max_x - A.b.min_x);
  const float invH = (float)GRID_ROWS / (A.b.max_y - A.b.min_y);
  cx0 = max(0, (int)floorf((x - A.q_min_x - r) * invW));
  if (cx0 >= GRID_COLS) return false;
  cx1 = min(GRID_COLS - 1, (int)ceilf((x - A.q_min_x + r) * invW));
  if (cx1 < 0) return false;
  cy0 = max(0, (int)floorf((y - A.q_min_y - r) * invH));
  if (cy0 >= GRID_ROWS) return false;
  cy1 = min(GRID_ROWS - 1, (int)ceilf((y - A.q_min_y + r) * invH));
  if (cy1 < 0) return false;
  return true;
}

// UNITE (phase A): the candidates that are ranked are also put into one component
template <int MODE, bool UNITE = false, bool BIG = false>
__device__ void scan_query(const MatchArgs& A, const MatchShared& S, int pair, int nt, const float4 geo,
                           const uint32_t* qd, int bucket0, int bucket1, float qx, float qy, Cand& c) {
  cand_init(c);
  int first = -1;  // UNITE: the first candidate ranked
  auto unite = [&](int j) {
    if constexpr (UNITE) {
      if (first < 0) first = j;
      else uf_union<BIG>(S.claim, first, j);
    }
  };
  const uint4* tdesc = S.tdesc;
  const int dcap = scan_dist_cap<MODE>(A);
  (void)pair;
  if constexpr (!mode_bucketed(MODE)) {
    const float x = geo.x, y = geo.y, r = geo.z;
    const int lv = __float_as_int(geo.w);
    const int minLevel = (lv & 0xFFFF) - 1024, maxLevel = (lv >> 16) - 1024;
    int nMinCellX, nMaxCellX, nMinCellY, nMaxCellY;
    if (!window_cells(A, geo, nMinCellX, nMaxCellX, nMinCellY, nMaxCellY)) return;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    // The cells iy = nMinCellY..nMaxCellY of one grid column are adjacent in the CSR array, so a
    // column of the window is one contiguous run of items.  The scan alternates between two loops:
    // the first walks the items and tests window, level and occupancy (two LDS reads per item, no
    // dependent chain; four items in five drop out here) and queues the survivors; the second
    // takes the queued items' distances and ranks them.  Side by side in one loop, every step of
    // the wavefront would pay the second part's chain of latencies (descriptor, union-find,
    // insertion) as soon as ONE of its 64 lanes has a survivor.
    const int qcap = A.queue_cap;
    unsigned short* const queue = S.queue + threadIdx.x;  // entry k at queue[k * M_THREADS]
    // the level range as two bounds every octave (0..255) can be compared with
    const int lo = bCheckLevels ? max(minLevel, 0) : 0;
    const int hi = bCheckLevels && maxLevel >= 0 ? min(maxLevel, 255) : 255;
    int ix = nMinCellX;
    int p = S.start[ix * GRID_ROWS + nMinCellY], pe = S.start[ix * GRID_ROWS + nMaxCellY + 1];
    bool more = true;
    while (more) {
      int n = 0;
      for (;;) {
        if (p >= pe) {
          if (++ix > nMaxCellX) { more = false; break; }
          p = S.start[ix * GRID_ROWS + nMinCellY]; pe = S.start[ix * GRID_ROWS + nMaxCellY + 1];
          continue;
        }
        const float2 xy = S.gxy[p];
        const uint32_t meta = S.gmeta[p];
        const int slot = p++;
        const int oct = (int)((meta >> 16) & 0xFFu);
        // window, level, and the occupancy before anything is decided (held in the word): tested
        // without branches, the wavefront's lanes differ at nearly every item
        bool ok = ((int)(fabsf(xy.x - x) < r) & (int)(fabsf(xy.y - y) < r) & (int)(oct >= lo) & (int)(oct <= hi)) != 0;
        if (MODE != MODE_INIT) ok &= meta < 0x1000000u;
        if constexpr (MODE == MODE_BEST) {
          if (ok && A.chi2 > 0) {  // reference src/ORBmatcher.cc:1158-1164
            const float ex = x - xy.x, ey = y - xy.y;
            const float e2 = ex * ex + ey * ey;
            if ((double)(e2 * A.sigma2[min(oct, kMaxLevels - 1)]) > A.chi2) ok = false;
          }
        }
        if (ok) queue[n * M_THREADS] = (unsigned short)slot;
        n += ok;
        if (n == qcap) break;
      }
      // the queued items' distances; the next one's descriptor is on its way while one is ranked
      uint32_t slot = 0, meta = 0;
      uint4 t0 = make_uint4(0, 0, 0, 0), t1 = t0;
      auto load = [&](int k) {
        slot = queue[k * M_THREADS];
        meta = S.gmeta[slot];
        const uint32_t j = meta & 0xFFFFu;
        t0 = tdesc[2 * j]; t1 = tdesc[2 * j + 1];
      };
      if (n > 0) load(0);
      for (int k = 0; k < n; k++) {
        const uint32_t c_slot = slot, c_meta = meta;
        const uint4 c0 = t0, c1 = t1;
        if (k + 1 < n) load(k + 1);
        const int j = (int)(c_meta & 0xFFFFu);
        const int d = hamming256(qd, c0, c1);
        if constexpr (MODE == MODE_INIT) {
          // occ = matched distance + 1 (0 = never matched): skip if vMatchedDistance[i2] <= dist
          // (reference src/ORBmatcher.cc:596)
          const int occ = (int)(c_meta >> 24);
          if (occ && occ - 1 <= d) continue;
        }
        if (d > dcap) continue;
        unite(j);
        cand_insert(c, ((uint32_t)d << 16) | c_slot);
      }
    }
    (void)nt;
  } else {
    // candidates: keyframe-2 features of the same vocabulary node, in order; the
    // reference keeps the LAST one among equal distances ('dist > bestDist' skips)
    // (the bucket's items and their descriptors come from global memory, one after the other: the
    // item two ahead and the descriptor one ahead are on their way while a distance is taken)
    int pos = 0;
    int j_next = bucket0 < bucket1 ? A.items2[bucket0] : 0;
    int j_after = bucket0 + 1 < bucket1 ? A.items2[bucket0 + 1] : 0;
    uint4 t0 = tdesc[2 * j_next], t1 = tdesc[2 * j_next + 1];
    for (int p = bucket0; p < bucket1; p++, pos++) {
      const int j = j_next;
      const uint4 c0 = t0, c1 = t1;
      j_next = j_after;
      if (p + 1 < bucket1) { t0 = tdesc[2 * j_next]; t1 = tdesc[2 * j_next + 1]; }
      j_after = p + 2 < bucket1 ? A.items2[p + 2] : 0;
      if (S.occ[j]) continue;
      const int d = hamming256(qd, c0, c1);
      if constexpr (MODE == MODE_BOW) {
        // SearchByBoW keeps best and second over the whole bucket; the first minimum wins
        if (d > dcap) continue;
        unite(j);
        cand_insert(c, ((uint32_t)d << 16) | (uint32_t)pos);
        continue;
      }
      if (d > TH_LOW) continue;
      const float x2 = S.kx[j], y2 = S.ky[j];
      const float distex = A.ex - x2, distey = A.ey - y2;
      if (distex * distex + distey * distey < 100 * A.sf[S.koct[j]]) continue;
      // CheckDistEpipolarLine, reference src/ORBmatcher.cc:180-207
      const float a = qx * A.f12[0] + qy * A.f12[3] + A.f12[6];
      const float b = qx * A.f12[1] + qy * A.f12[4] + A.f12[7];
      const float cc = qx * A.f12[2] + qy * A.f12[5] + A.f12[8];
      const float num = a * x2 + b * y2 + cc;
      const float den = a * a + b * b;
      if (den == 0) continue;
      const float dsqr = num * num / den;
      if (!((double)dsqr < 3.84 * (double)A.sigma2[S.koct[j]])) continue;
      unite(j);
      cand_insert(c, ((uint32_t)d << 16) | (uint32_t)(0xFFFF - pos));
    }
    (void)nt; (void)geo;
  }
}

// scan_query for the replaying wavefront: all 64 lanes scan the candidates of ONE query (its
// parameters are the same in every lane) against the live occupancy, 64 grid items at a time, and
// every lane returns the query's two best (key, keypoint), key[k] = 0xFFFFFFFF if there is none.
// A lane of its own takes several microseconds for this (forty items, each a chain of dependent
// LDS and memory latencies) while the other 63 wait; together it is one or two rounds of loads.
template <int MODE>
__device__ void scan_query_wave(const MatchArgs& A, const MatchShared& S, const float4 geo, const uint32_t* qd,
                                int bucket0, int bucket1, float qx, float qy, uint32_t (&key)[2], uint32_t (&idx)[2]) {
  const int lane = threadIdx.x & 63;
  uint32_t k0 = 0xFFFFFFFFu, k1 = 0xFFFFFFFFu, i0 = 0, i1 = 0;  // this lane's two smallest
  auto take = [&](uint32_t k, uint32_t j) {
    if (k < k0) { k1 = k0; i1 = i0; k0 = k; i0 = j; }
    else if (k < k1) { k1 = k; i1 = j; }
  };
  const uint4* tdesc = S.tdesc;
  const int dcap = scan_dist_cap<MODE>(A);
  if constexpr (!mode_bucketed(MODE)) {
    const float x = geo.x, y = geo.y, r = geo.z;
    const int lv = __float_as_int(geo.w);
    const int minLevel = (lv & 0xFFFF) - 1024, maxLevel = (lv >> 16) - 1024;
    int nMinCellX, nMaxCellX, nMinCellY, nMaxCellY;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    if (window_cells(A, geo, nMinCellX, nMaxCellX, nMinCellY, nMaxCellY)) {
      for (int ix = nMinCellX; ix <= nMaxCellX; ix++) {
        const int pe = S.start[ix * GRID_ROWS + nMaxCellY + 1];
        for (int pb = S.start[ix * GRID_ROWS + nMinCellY]; pb < pe; pb += 64) {
          const int p = pb + lane;
          const bool in = p < pe;
          const float2 xyj = in ? S.gxy[p] : make_float2(0.f, 0.f);
          const uint32_t meta = in ? S.gmeta[p] : 0u;
          const int j = (int)(meta & 0xFFFFu), oct = (int)((meta >> 16) & 0xFFu);
          const int occ = S.occ[j];  // the live occupancy
          const float kxj = xyj.x, kyj = xyj.y;
          bool pass = in && fabsf(kxj - x) < r && fabsf(kyj - y) < r;
          if (bCheckLevels && (oct < minLevel || (maxLevel >= 0 && oct > maxLevel))) pass = false;
          const int mypos = p;  // the slot in grid order breaks ties, as in scan_query
          if (pass && (MODE == MODE_INIT || !occ)) {
            const int d = hamming256(qd, tdesc[2 * j], tdesc[2 * j + 1]);
            const bool held = MODE == MODE_INIT && occ && occ - 1 <= d;  // reference src/ORBmatcher.cc:596
            if (!held && d <= dcap) take(((uint32_t)d << 16) | (uint32_t)mypos, (uint32_t)j);
          }
        }
      }
    }
  } else {
    for (int pb = bucket0; pb < bucket1; pb += 64) {
      const int p = pb + lane;
      if (p >= bucket1) continue;
      const int pos = p - bucket0;
      const int j = A.items2[p];
      if (S.occ[j]) continue;
      const int d = hamming256(qd, tdesc[2 * j], tdesc[2 * j + 1]);
      if constexpr (MODE == MODE_BOW) {
        if (d <= dcap) take(((uint32_t)d << 16) | (uint32_t)pos, (uint32_t)j);
        continue;
      }
      if (d > TH_LOW) continue;
      const float x2 = S.kx[j], y2 = S.ky[j];
      const float distex = A.ex - x2, distey = A.ey - y2;
      if (distex * distex + distey * distey < 100 * A.sf[S.koct[j]]) continue;
      const float a = qx * A.f12[0] + qy * A.f12[3] + A.f12[6];
      const float b = qx * A.f12[1] + qy * A.f12[4] + A.f12[7];
      const float cc = qx * A.f12[2] + qy * A.f12[5] + A.f12[8];
      const float num = a * x2 + b * y2 + cc;
      const float den = a * a + b * b;
      if (den == 0) continue;
      const float dsqr = num * num / den;
      if (!((double)dsqr < 3.84 * (double)A.sigma2[S.koct[j]])) continue;
      take(((uint32_t)d << 16) | (uint32_t)(0xFFFF - pos), (uint32_t)j);
    }
  }
  // the two smallest keys of the wavefront (keys are distinct: each holds its item's rank)
#pragma unroll
  for (int r = 0; r < 2; r++) {
    unsigned long long best = ((unsigned long long)k0 << 32) | i0;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const unsigned long long other = __shfl_xor(best, o);
      best = other < best ? other : best;
    }
    key[r] = (uint32_t)(best >> 32);
    idx[r] = (uint32_t)best;
    if (k0 == key[r] && k0 != 0xFFFFFFFFu) { k0 = k1; i0 = i1; k1 = 0xFFFFFFFFu; }
  }
}

__device__ __forceinline__ int rot_bin(float rot) {
  const float factor = HISTO_LENGTH / 360.0f;
  if (rot < 0.0f) rot += 360.0f;
  int bin = (int)roundf(rot * factor);
  if (bin == HISTO_LENGTH) bin = 0;
  return bin;
}

template <int MODE, bool BIG>
__global__ __launch_bounds__(M_THREADS) void match_kernel(MatchArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int s_hist[HISTO_LENGTH];
  __shared__ int s_nm, s_keep[3];
  __shared__ int s_bins[kWorkBins + 1];  // phase A: queries by the number of items they visit, most first
  __shared__ int s_next;                 // phase A: next 64 queries of that order to hand out
  __shared__ int s_chunk[M_THREADS / 64];
  const int pair = blockIdx.x, tid = threadIdx.x;
  const int nq = min(A.n_q[pair], A.max_q), nt = min(A.n_t[pair], A.max_t);
  MatchShared S;
  {
    unsigned char* p = smem;
    S.start = reinterpret_cast<int*>(p); p += sizeof(int) * (GRID_COLS * GRID_ROWS + 4);
    S.queue = reinterpret_cast<unsigned short*>(p); p += sizeof(unsigned short) * M_THREADS * (size_t)A.queue_cap;
    if constexpr (BIG) {
      // LDS keeps the grid's offsets and the two byte arrays every candidate test reads; the rest
      // lives in the pair's block of global scratch (laid out by match_big_stride)
      S.koct = p; p += (A.max_t + 3) & ~3;
      S.occ = p; p += (A.max_t + 3) & ~3;
      unsigned char* g = A.big + (size_t)pair * A.big_stride;
      S.gxy = reinterpret_cast<float2*>(g);
      S.kx = reinterpret_cast<float*>(g); g += sizeof(float) * A.max_t;
      S.ky = reinterpret_cast<float*>(g); g += sizeof(float) * A.max_t;
      S.gmeta = reinterpret_cast<uint32_t*>(g); g += sizeof(uint32_t) * A.max_t;
      S.claim = reinterpret_cast<int*>(g); g += sizeof(int) * A.max_t;
      S.owner = reinterpret_cast<int*>(g);
    } else {
      S.gxy = reinterpret_cast<float2*>(p);
      S.kx = reinterpret_cast<float*>(p); p += sizeof(float) * A.max_t;
      S.ky = reinterpret_cast<float*>(p); p += sizeof(float) * A.max_t;
      S.gmeta = reinterpret_cast<uint32_t*>(p); p += sizeof(uint32_t) * A.max_t;
      S.koct = p; p += (A.max_t + 3) & ~3;
      S.occ = p; p += (A.max_t + 3) & ~3;
      S.claim = reinterpret_cast<int*>(p); p += sizeof(int) * A.max_t;
      S.owner = reinterpret_cast<int*>(p);
      if (MODE == MODE_INIT) p += sizeof(int) * A.max_t;
    }
    p = reinterpret_cast<unsigned char*>((reinterpret_cast<uintptr_t>(p) + 15) & ~(uintptr_t)15);
    if (A.desc_in_lds) {
      S.tdesc = reinterpret_cast<const uint4*>(p); p += 32 * (size_t)A.max_t;
    } else {
      S.tdesc = reinterpret_cast<const uint4*>(A.t_desc) + 2 * (size_t)pair * A.max_t;
    }
    if (A.lists_in_lds) {
      S.lists = reinterpret_cast<uint32_t*>(p); p += sizeof(uint32_t) * MK * A.max_q;
      S.ntot = reinterpret_cast<int*>(p);
    } else {
      S.lists = A.lists + (size_t)pair * A.max_q * MK;
      S.ntot = A.q_ntot + (size_t)pair * A.max_q;
    }
  }
  const size_t tbase = (size_t)pair * A.max_t, qbase = (size_t)pair * A.max_q;
  int* out_match = A.out_match + (mode_out_by_query(MODE) ? qbase : tbase);
  // Development build (make EXTRA=-DORB_MATCH_PROFILE OUT=...): pair 0 prints the phases' durations in
  // ticks of the 100 MHz wall clock.  PT(k) stamps: 0 start, 1 grid built, 5 queries binned, 2 ranked,
  // 3 components shared out, 4 replayed.
#ifdef ORB_MATCH_PROFILE
  long long pt[6] = {0, 0, 0, 0, 0, 0};
#define PT(k) pt[k] = wall_clock64()
#else
#define PT(k)
#endif
  PT(0);

  // ---- phase 0: train keypoints to LDS; 64x48 grid as CSR in insertion order
  // (Frame::AssignFeaturesToGrid / PosInGrid, reference src/Frame.cc:190-213, 461-477)
  for (int j = tid; j < nt; j += M_THREADS) {
    if (mode_bucketed(MODE)) { S.kx[j] = A.t_x[(tbase + j) * A.ts]; S.ky[j] = A.t_y[(tbase + j) * A.ts]; }
    S.koct[j] = (unsigned char)min(max(A.t_oct[(tbase + j) * A.ts], 0), 255);
    S.occ[j] = (unsigned char)(A.t_obs[tbase + j] > 0);
    if (MODE == MODE_INIT) S.owner[j] = -1;
    S.claim[j] = kNoLane | j;  // every keypoint a component of its own, nobody replaying it
    if (!mode_out_by_query(MODE)) out_match[j] = -1;
  }
  if (A.desc_in_lds) {
    const uint4* g = reinterpret_cast<const uint4*>(A.t_desc) + 2 * (size_t)pair * A.max_t;
    uint4* d = const_cast<uint4*>(S.tdesc);
    for (int j = tid; j < 2 * nt; j += M_THREADS) d[j] = g[j];
  }
  if (mode_bucketed(MODE) || MODE == MODE_INIT)
    for (int i = tid; i < A.max_q; i += M_THREADS) out_match[i] = -1;
  if (tid < HISTO_LENGTH) s_hist[tid] = 0;
  if (tid <= kWorkBins) s_bins[tid] = 0;
  if (tid == 0) { s_nm = 0; s_next = 0; }
  if constexpr (!mode_bucketed(MODE)) {
    constexpr int NC = GRID_COLS * GRID_ROWS;
    // counts go to start[cell + 2]; after the inclusive scan start[cell + 1] is the
    // cell's first slot, and the fill's atomic increments turn it into the cell's end,
    // which leaves start[] as an ordinary CSR offset array
    for (int c = tid; c < NC + 2; c += M_THREADS) S.start[c] = 0;
    __syncthreads();
    const float invW = (float)GRID_COLS / (A.b.max_x - A.b.min_x);
    const float invH = (float)GRID_ROWS / (A.b.max_y - A.b.min_y);
    for (int j = tid; j < nt; j += M_THREADS) {
      const float kxj = A.t_x[(tbase + j) * A.ts], kyj = A.t_y[(tbase + j) * A.ts];
      const int px = (int)roundf((kxj - A.b.min_x) * invW), py = (int)roundf((kyj - A.b.min_y) * invH);
      if (px >= 0 && px < GRID_COLS && py >= 0 && py < GRID_ROWS) atomicAdd(&S.start[px * GRID_ROWS + py + 2], 1);
    }
    __syncthreads();
    if (tid < 64) {  // inclusive scan of the 3072 counts, 48 per lane
      int sum = 0;
      for (int k = 0; k < GRID_ROWS; k++) sum += S.start[tid * GRID_ROWS + k + 2];
      int incl = sum;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (tid >= o) incl += v; }
      int run = incl - sum;
      for (int k = 0; k < GRID_ROWS; k++) { run += S.start[tid * GRID_ROWS + k + 2]; S.start[tid * GRID_ROWS + k + 2] = run; }
    }
    __syncthreads();
    for (int j = tid; j < nt; j += M_THREADS) {
      const float kxj = A.t_x[(tbase + j) * A.ts], kyj = A.t_y[(tbase + j) * A.ts];  // (served by the caches: read a moment ago)
      const int px = (int)roundf((kxj - A.b.min_x) * invW), py = (int)roundf((kyj - A.b.min_y) * invH);
      if (px >= 0 && px < GRID_COLS && py >= 0 && py < GRID_ROWS) {
        const int slot = atomicAdd(&S.start[px * GRID_ROWS + py + 1], 1);
        S.gxy[slot] = make_float2(kxj, kyj);
        S.gmeta[slot] = (uint32_t)j | ((uint32_t)S.koct[j] << 16) | ((uint32_t)S.occ[j] << 24);
      }
    }
    __syncthreads();
    // insertion order inside a cell is ascending keypoint index (the low 16 bits of the second word)
    for (int c = tid; c < NC; c += M_THREADS) {
      const int b = S.start[c], e = S.start[c + 1];
      for (int i = b + 1; i < e; i++) {
        const uint32_t v = S.gmeta[i];
        const float2 w = S.gxy[i];
        int k = i - 1;
        while (k >= b && (S.gmeta[k] & 0xFFFFu) > (v & 0xFFFFu)) { S.gmeta[k + 1] = S.gmeta[k]; S.gxy[k + 1] = S.gxy[k]; k--; }
        S.gmeta[k + 1] = v; S.gxy[k + 1] = w;
      }
    }
  }
  __syncthreads();
  PT(1);

  // ---- phase A: every query ranks its candidates against the initial occupancy.
  // A wavefront scans 64 queries in lockstep and takes as long as the one with the most items to
  // visit; windows grow with the pyramid level (a level-7 window holds twelve times the items of a
  // level-0 one) and vocabulary buckets differ as much.  So the queries are first counted into
  // bins by the number of items they will visit and handed to the lanes in descending order of it:
  // the 64 queries of a wavefront then have about the same length.
  auto setup = [&](int i, float4& geo, int& b0, int& b1, float& qx, float& qy, const uint8_t*& qdesc_ptr) -> bool {
    bool valid = true;
    if constexpr (MODE == MODE_FRAMES) {
      // projection, reference src/ORBmatcher.cc:1669-1712
      valid = A.q_obs[qbase + i] >= 0;
      const float* T = A.tcw + 16 * (size_t)pair;
      const float* X = A.q_xw + 3 * (qbase + i);
      const float xc = (T[0] * X[0] + T[1] * X[1] + T[2] * X[2]) + T[3];
      const float yc = (T[4] * X[0] + T[5] * X[1] + T[6] * X[2]) + T[7];
      const float zc = (T[8] * X[0] + T[9] * X[1] + T[10] * X[2]) + T[11];
      const float invzc = (float)(1.0 / (double)zc);
      if (invzc < 0) valid = false;
      const float u = A.fx * xc * invzc + A.cx, v = A.fy * yc * invzc + A.cy;
      if (u < A.b.min_x || u > A.b.max_x || v < A.b.min_y || v > A.b.max_y || isnan(u) || isnan(v)) valid = false;
      int oct = A.q_oct[(qbase + i) * A.qs];
      oct = min(max(oct, 0), A.nlevels - 1);
      geo.x = u; geo.y = v; geo.z = A.th * A.sf[oct];
      geo.w = __int_as_float(((oct - 1 + 1024) & 0xFFFF) | ((oct + 1 + 1024) << 16));
      qdesc_ptr = A.q_desc + 32 * (qbase + i);
    } else if constexpr (MODE == MODE_MAPPOINTS) {
      // window, reference src/ORBmatcher.cc:75-96
      valid = A.q_track[qbase + i] != 0;
      int lvl = A.q_oct[qbase + i];
      lvl = min(max(lvl, 0), A.nlevels - 1);
      float r = (double)A.q_viewcos[qbase + i] > 0.998 ? 2.5f : 4.0f;
      if ((double)A.th != 1.0) r *= A.th;
      geo.x = A.q_x[qbase + i]; geo.y = A.q_y[qbase + i]; geo.z = r * A.sf[lvl];
      geo.w = __int_as_float(((lvl - 1 + 1024) & 0xFFFF) | ((lvl + 1024) << 16));
      qdesc_ptr = A.q_desc + 32 * (qbase + i);
    } else if constexpr (MODE == MODE_WINDOW || MODE == MODE_BEST || MODE == MODE_INIT) {
      valid = A.q_track[qbase + i] != 0;
      const int lo = min(max(A.q_minl[qbase + i], -1000), 1000), hi = min(max(A.q_maxl[qbase + i], -1000), 1000);
      geo.x = A.q_x[qbase + i]; geo.y = A.q_y[qbase + i]; geo.z = A.q_r[qbase + i];
      geo.w = __int_as_float(((lo + 1024) & 0xFFFF) | ((hi + 1024) << 16));
      qdesc_ptr = A.q_desc + 32 * (qbase + i);
    } else {
      const int idx1 = A.q_idx1[qbase + i];
      b0 = A.q_b0[qbase + i]; b1 = A.q_b1[qbase + i];
      // TRIANG: the feature has no map point yet; BOW: it has a good one
      valid = MODE == MODE_TRIANG ? A.q_obs[(size_t)pair * A.max_q + idx1] <= 0 : A.q_obs[(size_t)pair * A.max_q + idx1] > 0;
      qx = A.q_x[qbase + idx1]; qy = A.q_y[qbase + idx1];
      qdesc_ptr = A.q_desc + 32 * (qbase + idx1);
    }
    return valid;
  };
  int* const order = A.q_bin + qbase;  // (the rotation bins of phase C are not in use yet)
  for (int i = tid; i < nq; i += M_THREADS) {
    float4 geo = make_float4(0.f, 0.f, 0.f, 0.f);
    int b0 = 0, b1 = 0;
    float qx = 0.f, qy = 0.f;
    const uint8_t* qdesc_ptr;
    const bool valid = setup(i, geo, b0, b1, qx, qy, qdesc_ptr);
    int work = b1 - b0;
    if constexpr (!mode_bucketed(MODE)) {
      int cx0, cx1, cy0, cy1;
      work = 0;
      if (window_cells(A, geo, cx0, cx1, cy0, cy1))
        for (int ix = cx0; ix <= cx1; ix++) work += S.start[ix * GRID_ROWS + cy1 + 1] - S.start[ix * GRID_ROWS + cy0];
    }
    if (valid && work > 0) {
      if (!mode_bucketed(MODE)) A.q_geo[qbase + i] = geo;  // (the scan below reads it again, and so do the replay's rescans)
      S.ntot[i] = work;
      atomicAdd(&s_bins[kWorkBins - 1 - min(work, kWorkBins - 1)], 1);
      continue;
    }
    // nothing to scan: the query's results are final here
    if constexpr (MODE == MODE_BEST) {
      A.out_match[qbase + i] = -1;
      A.out_dist[qbase + i] = 0x7FFFFFFF;
      S.ntot[i] = -1;
    } else {
      A.q_geo[qbase + i] = geo;
      int obs_bit = 0;
      if (MODE == MODE_FRAMES || MODE == MODE_MAPPOINTS) obs_bit = A.q_obs[qbase + i] > 0 ? 0x40000000 : 0;
      S.ntot[i] = valid ? obs_bit : -1;
#pragma unroll
      for (int k = 0; k < MK; k++) S.lists[i * MK + k] = kNoCand;
      A.q_slot[qbase + i] = -1;
    }
  }
  __syncthreads();
  if (tid < 64) {  // exclusive scan of the bins, kWorkBins / 64 per lane
    constexpr int PER = kWorkBins / 64;
    int sum = 0;
    for (int k = 0; k < PER; k++) sum += s_bins[tid * PER + k];
    int incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (tid >= o) incl += v; }
    int run = incl - sum;
    for (int k = 0; k < PER; k++) { const int v = s_bins[tid * PER + k]; s_bins[tid * PER + k] = run; run += v; }
    if (tid == 63) s_bins[kWorkBins] = run;  // queries to scan
  }
  __syncthreads();
  const int n_scan = s_bins[kWorkBins];
  for (int i = tid; i < nq; i += M_THREADS) {
    // (this thread wrote the word itself: the count where the query is to be scanned, else a final
    // value, which is negative or has none of the count's bits - counts are below 2^30 and not 0)
    const int work = S.ntot[i];
    const bool scan = work > 0 && !(work & 0x40000000);
    if (scan) __hip_atomic_store(&order[atomicAdd(&s_bins[kWorkBins - 1 - min(work, kWorkBins - 1)], 1)], i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  PT(5);
  for (;;) {
    // A wavefront takes the next 64 queries of the order when it is done with its last ones: the
    // longest first, so the wavefronts end about together.  (Every lane reads the chunk's number
    // from the wavefront's word, and nothing below leaves the loop's body early: all 64 lanes
    // come back to this point together.)
    if ((tid & 63) == 0) s_chunk[tid >> 6] = atomicAdd(&s_next, 1);
    __builtin_amdgcn_wave_barrier();
    const int chunk = s_chunk[tid >> 6];
    __builtin_amdgcn_wave_barrier();
    if (chunk * 64 >= n_scan) break;
    const int k = chunk * 64 + (tid & 63);
    if (k < n_scan) {
      const int i = __hip_atomic_load(&order[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      float4 geo = make_float4(0.f, 0.f, 0.f, 0.f);
      int b0 = 0, b1 = 0;
      float qx = 0.f, qy = 0.f;
      const uint8_t* qdesc_ptr;
      if constexpr (mode_bucketed(MODE)) {
        setup(i, geo, b0, b1, qx, qy, qdesc_ptr);
      } else {
        geo = A.q_geo[qbase + i];  // (written before the barrier, and never read before in this launch)
        qdesc_ptr = A.q_desc + 32 * (qbase + i);
      }
      Cand c;
      cand_init(c);
      {
        const uint4 d0 = reinterpret_cast<const uint4*>(qdesc_ptr)[0], d1 = reinterpret_cast<const uint4*>(qdesc_ptr)[1];
        const uint32_t qd[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
        scan_query<MODE, MODE != MODE_BEST, BIG>(A, S, pair, nt, geo, qd, b0, b1, qx, qy, c);
      }
      if constexpr (MODE == MODE_BEST) {
        // independent queries: the first minimum is the result
        const bool any = c.key[0] != kNoCand;
        A.out_match[qbase + i] = any ? (int)(S.gmeta[c.key[0] & 0xFFFFu] & 0xFFFFu) : -1;
        A.out_dist[qbase + i] = any ? (int)(c.key[0] >> 16) : 0x7FFFFFFF;
      } else {
        if (mode_bucketed(MODE)) A.q_geo[qbase + i] = geo;
        int obs_bit = 0;
        if (MODE == MODE_FRAMES || MODE == MODE_MAPPOINTS) obs_bit = A.q_obs[qbase + i] > 0 ? 0x40000000 : 0;
        S.ntot[i] = c.ntot | obs_bit;
#pragma unroll
        for (int k2 = 0; k2 < MK; k2++) S.lists[i * MK + k2] = c.key[k2];
        A.q_slot[qbase + i] = -1;
      }
    }
  }
  __syncthreads();
  PT(2);
  if constexpr (MODE == MODE_BEST) return;
  // every keypoint's word straight at its component's root (one read per query in the replay)
  for (int j = tid; j < nt; j += M_THREADS) word_store<BIG>(&S.claim[j], kNoLane | uf_find<BIG>(S.claim, j));
  __syncthreads();
  // the keypoint a candidate's word names
  auto cand_keypoint = [&](uint32_t word, int b0) -> uint32_t {
    const uint32_t tie = word & 0xFFFFu;
    if constexpr (MODE == MODE_TRIANG) return (uint32_t)A.items2[b0 + (int)(0xFFFFu - tie)];
    else if constexpr (MODE == MODE_BOW) return (uint32_t)A.items2[b0 + (int)tie];
    else return S.gmeta[tie] & 0xFFFFu;
  };
  // Components are independent loops, so the wavefronts share them out: a query goes to the
  // wavefront its component's root hashes to (bits 24-27 of its count word; bits 0-15 are the
  // count, bit 30 the observation flag, a negative word an inactive query).
  constexpr int kWaves = M_THREADS / 64;
  static_assert(kWaves == 16, "four bits of the count word name the wavefront");
  // (count words in global memory are written by one thread and read by another several times over:
  // past the vector cache, which may still hold an older value)
  auto count_word = [&](int i) -> int {
    return A.lists_in_lds ? S.ntot[i] : __hip_atomic_load(&S.ntot[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  for (int i = tid; i < nq; i += M_THREADS) {
    const int w = count_word(i);
    const uint32_t first = S.lists[i * MK];
    if (w > 0 && (w & 0xFFFF) > 0 && first != kNoCand) {
      int b0 = 0;
      if constexpr (mode_bucketed(MODE)) b0 = A.q_b0[qbase + i];
      const int root = word_load<BIG>(&S.claim[cand_keypoint(first, b0)]) & kUfMask;
      S.ntot[i] = w | (((root ^ (root >> 4) ^ (root >> 8)) & (kWaves - 1)) << 24);
    } else if (w > 0) {
      S.ntot[i] = w & 0x40000000;  // nothing ranked: nothing to replay
    }
  }
  __syncthreads();
  PT(3);

  // ---- phase B: replay the reference's query order.  Every wavefront walks the queries in
  // ascending order, gathers its own 64 at a time and decides them; within such a step the queries of
  // one component are decided in ascending order, one per round (the lowest pending lane of the
  // component, found with an atomic minimum on the root's word), and different components, which
  // do not touch each other's keypoints, go side by side - in the step and across the wavefronts.
  {
    const int lane = tid & 63, wave = tid >> 6;
    // the (up to 64) queries gathered for a step: in the scans' queues, which are idle now
    static_assert(kQueueMin * M_THREADS * sizeof(unsigned short) >= 64 * kWaves * sizeof(int), "the queues' minimum holds every wavefront's gathered queries");
    int* const stage = reinterpret_cast<int*>(S.queue) + 64 * wave;
    int base = 0;
    unsigned long long taken = 0;  // lanes of the 64 queries at `base` that an earlier step has taken
    for (;;) {
      int have = 0;
      while (have < 64 && base < nq) {  // (wavefront-uniform)
        const int i = base + lane;
        const int w = i < nq ? count_word(i) : -1;
        const bool own = w > 0 && (w & 0xFFFF) > 0 && ((w >> 24) & (kWaves - 1)) == wave && !((taken >> lane) & 1);
        const unsigned long long m = __ballot(own);
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        const bool take = own && have + rank < 64;
        if (take) stage[have + rank] = i;
        if (have + __popcll(m) <= 64) { have += __popcll(m); base += 64; taken = 0; }
        else { taken |= __ballot(take); have = 64; }  // the rest of these 64 in the next step
      }
      __builtin_amdgcn_wave_barrier();
      const int n = have;
      if (n == 0) break;
      const bool active = lane < n;
      const int i = active ? stage[lane] : 0;
      __builtin_amdgcn_wave_barrier();
      const int nt_i = active ? count_word(i) : 0;
      const bool q_occupies = (nt_i & 0x40000000) != 0;
      const int c_ntot = nt_i & 0xFFFF;
      int b0 = 0, my_idx1 = 0;
      if constexpr (mode_bucketed(MODE)) {
        if (active) { b0 = A.q_b0[qbase + i]; my_idx1 = A.q_idx1[qbase + i]; }
      }
      uint32_t ck[MK], cj[MK];
#pragma unroll
      for (int k = 0; k < MK; k++) ck[k] = active ? S.lists[i * MK + k] : kNoCand;
#pragma unroll
      for (int k = 0; k < MK; k++) cj[k] = ck[k] != kNoCand ? cand_keypoint(ck[k], b0) : 0u;
      // the query's component (the words' low bits no longer change)
      const int root = active ? word_load<BIG>(&S.claim[cj[0]]) & kUfMask : 0;
      bool resolved = !active;
      while (true) {
        const unsigned long long pending = __ballot(!resolved);
        if (pending == 0) break;
        // whose turn: the lowest pending lane of every component
        if (!resolved) {
          const int before = atomicMin(&S.claim[root], (lane << 20) | root);
          // (global words: the loads below must not overtake the atomics, so their answers are waited for)
          if constexpr (BIG) asm volatile("" ::"v"(before));
        }
        __builtin_amdgcn_wave_barrier();
        const bool mine = !resolved && (word_load<BIG>(&S.claim[root]) >> 20) == lane;
        __builtin_amdgcn_wave_barrier();
        if (mine) word_store<BIG>(&S.claim[root], kNoLane | root);
        // first and second candidate that are still free
        int s1 = -1, s2 = -1, d1 = 256, d2 = 256;
        bool need_rescan = false;
        if (mine) {
          int found = 0;
#pragma unroll
          for (int k = 0; k < MK; k++) {
            const int o_k = ck[k] != kNoCand ? (int)S.occ[cj[k]] : 1;
            const bool avail_k = MODE == MODE_INIT ? (o_k == 0 || o_k - 1 > (int)(ck[k] >> 16)) : o_k == 0;
            if (ck[k] != kNoCand && avail_k) {
              if (found == 0) { s1 = (int)cj[k]; d1 = (int)(ck[k] >> 16); }
              else if (found == 1) { s2 = (int)cj[k]; d2 = (int)(ck[k] >> 16); }
              found++;
            }
          }
          const int want = (MODE == MODE_MAPPOINTS || MODE == MODE_BOW || MODE == MODE_INIT) ? 2 : 1;
          // c_ntot > MK: candidates beyond the kept ones exist, so the list is not exhaustive
          need_rescan = found < want && c_ntot > MK;
        }
        // A kept list that ran out: the query's candidates are scanned again, against the current
        // occupancy - by the whole wavefront, one such query after the other.
        for (unsigned long long rm = __ballot(need_rescan); rm; rm &= rm - 1) {  // uniform
          const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)rm) - 1);
          const int qi = __builtin_amdgcn_readlane(i, src);
          const float4 geo = A.q_geo[qbase + qi];
          const uint8_t* qp;
          int b0 = 0, b1 = 0; float qx = 0.f, qy = 0.f;
          if constexpr (mode_bucketed(MODE)) {
            const int idx1 = A.q_idx1[qbase + qi];
            b0 = A.q_b0[qbase + qi]; b1 = A.q_b1[qbase + qi];
            qx = A.q_x[qbase + idx1]; qy = A.q_y[qbase + idx1];
            qp = A.q_desc + 32 * (qbase + idx1);
          } else {
            qp = A.q_desc + 32 * (qbase + qi);
          }
          const uint4 e0 = reinterpret_cast<const uint4*>(qp)[0], e1 = reinterpret_cast<const uint4*>(qp)[1];
          const uint32_t qd[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
          uint32_t rk[2], ri[2];
          scan_query_wave<MODE>(A, S, geo, qd, b0, b1, qx, qy, rk, ri);
          if (lane == src) {
            s1 = s2 = -1; d1 = d2 = 256;
            if (rk[0] != kNoCand) { s1 = (int)ri[0]; d1 = (int)(rk[0] >> 16); }
            if (rk[1] != kNoCand) { s2 = (int)ri[1]; d2 = (int)(rk[1] >> 16); }
          }
        }
        // the query's decision (everything before it in its component has been decided)
        bool accept = false, occupy = false;
        if (mine && s1 >= 0) {
          if constexpr (MODE == MODE_FRAMES) {
            accept = d1 <= TH_HIGH;
            occupy = accept && q_occupies;
          } else if constexpr (MODE == MODE_MAPPOINTS) {
            accept = d1 <= TH_HIGH;
            if (accept) {
              const int l1 = S.koct[s1], l2 = s2 >= 0 ? (int)S.koct[s2] : -1;
              if (l1 == l2 && (float)d1 > A.nnratio * (float)d2) accept = false;
            }
            occupy = accept && q_occupies;
          } else if constexpr (MODE == MODE_WINDOW) {
            accept = d1 <= A.max_dist;
            occupy = accept;
          } else if constexpr (MODE == MODE_INIT) {
            // reference src/ORBmatcher.cc:613-616 (bestDist2 = INT_MAX without a second candidate)
            accept = d1 <= TH_LOW && (float)d1 < (float)(s2 >= 0 ? d2 : 0x7FFFFFFF) * A.nnratio;
            occupy = accept;
          } else if constexpr (MODE == MODE_BOW) {
            // reference src/ORBmatcher.cc:309-312 resp. :783-785 (d2 = 256 without a second candidate)
            accept = d1 <= A.max_dist && (float)d1 < A.nnratio * (float)d2;
            occupy = accept;
          } else {
            accept = true;  // thresholds were applied during the scan
            occupy = true;
          }
        }
        if (mine) {
          resolved = true;
          if (accept) {
            if constexpr (MODE == MODE_INIT) {
              // a keypoint matched before goes to the new query (reference :619-628)
              const int prev = word_load<BIG>(&S.owner[s1]);
              if (prev >= 0) { out_match[prev] = -1; atomicSub(&s_nm, 1); }
              word_store<BIG>(&S.owner[s1], i);
              S.occ[s1] = (unsigned char)(d1 + 1);
              out_match[i] = s1;
            } else if (occupy) {
              S.occ[s1] = 1;
            }
            atomicAdd(&s_nm, 1);
            A.q_slot[qbase + i] = s1;
            if constexpr (MODE == MODE_INIT) {
              // stored above
            } else if constexpr (mode_bucketed(MODE)) {
              out_match[my_idx1] = s1;
            } else {
              atomicMax(&out_match[s1], i);  // a later query overwrites an earlier one
            }
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  PT(4);
#ifdef ORB_MATCH_PROFILE
  if (pair == 0 && tid == 0)
    printf("match profile (100 MHz ticks): setup+grid %lld  count+sort %lld  scan %lld  roots+share %lld  replay (wavefront 0) %lld\n",
           pt[1] - pt[0], pt[5] - pt[1], pt[2] - pt[5], pt[3] - pt[2], pt[4] - pt[3]);
#endif
  __syncthreads();

  // ---- phase C: rotation consistency (reference src/ORBmatcher.cc:1765-1785, 1014-1034,
  // ComputeThreeMaxima :1943-1985)
  if (A.check_ori && MODE != MODE_MAPPOINTS) {
    for (int i = tid; i < nq; i += M_THREADS) {
      const int s1 = A.q_slot[qbase + i];
      int b = -1;
      if (s1 >= 0) {
        const float qa = mode_bucketed(MODE) ? A.q_angle[qbase + A.q_idx1[qbase + i]] : A.q_angle[(qbase + i) * A.qs];
        b = rot_bin(qa - A.t_angle[(tbase + s1) * A.ts]);
        atomicAdd(&s_hist[b], 1);
      }
      A.q_bin[qbase + i] = b;
    }
    __syncthreads();
    if (tid == 0) {
      int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
      for (int i = 0; i < HISTO_LENGTH; i++) {
        const int s = s_hist[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
      }
      if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
      else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
      s_keep[0] = ind1; s_keep[1] = ind2; s_keep[2] = ind3;
    }
    __syncthreads();
    for (int i = tid; i < nq; i += M_THREADS) {
      const int b = A.q_bin[qbase + i];
      if (b >= 0 && b != s_keep[0] && b != s_keep[1] && b != s_keep[2]) {
        if constexpr (MODE == MODE_INIT) {
          // the histogram holds every accepted query; one whose keypoint was taken over
          // later is no longer a match (reference src/ORBmatcher.cc:668-673)
          if (out_match[i] >= 0) { out_match[i] = -1; atomicSub(&s_nm, 1); }
        } else {
          if (mode_bucketed(MODE)) out_match[A.q_idx1[qbase + i]] = -1;
          else out_match[A.q_slot[qbase + i]] = -2;
          atomicSub(&s_nm, 1);
        }
      }
    }
    __syncthreads();
  }
  if (tid == 0) A.n_matches[pair] = s_nm;
}

size_t match_lds_bytes(int max_t, int max_q, bool lists_in_lds, bool desc_in_lds, bool owner = false) {
  // (with the queues' minimum; launch_match gives the queue what is left over)
  size_t n = sizeof(int) * (GRID_COLS * GRID_ROWS + 4) + sizeof(unsigned short) * kQueueMin * M_THREADS +
             2 * sizeof(float) * (size_t)max_t + sizeof(uint32_t) * (size_t)max_t + 2 * (size_t)((max_t + 3) & ~3) +
             sizeof(int) * (size_t)max_t + 32;
  if (lists_in_lds) n += (sizeof(uint32_t) * MK + sizeof(int)) * (size_t)max_q;
  if (desc_in_lds) n += 32 * (size_t)max_t;
  if (owner) n += sizeof(int) * (size_t)max_t;
  return n;
}

// LDS of match_kernel<MODE, true> and the size of its per-pair block of global scratch
size_t match_big_lds_bytes(int max_t) {
  return sizeof(int) * (GRID_COLS * GRID_ROWS + 4) + sizeof(unsigned short) * kQueueMin * M_THREADS + 2 * (size_t)((max_t + 3) & ~3) + 32;
}
size_t match_big_stride(int max_t) {
  return (5 * sizeof(int) * (size_t)max_t + 255) & ~(size_t)255;
}

// ---------------------------------------------------------------------------
// Stereo row-band search: per left keypoint the best right keypoint by Hamming distance
// among those whose row band floor(y - r)..ceil(y + r), r = 2 * scaleFactor[octave],
// contains the left keypoint's row, within one pyramid level and the disparity range.
// (This fork of the reference has no stereo matcher; the search is the coarse stage of
// upstream ORB-SLAM2's stereo association with this tree's DescriptorDistance and
// thresholds, src/ORBmatcher.cc:41-43.)  Left keypoints are independent of each other,
// and the first minimum in ascending right index wins, so a workgroup per pair ranks
// (distance, right index) in parallel.  Right keypoints are bucketed by their own row;
// the rows a band can reach are adjacent, so the candidates of a left keypoint are one
// contiguous run of the bucket array.
// ---------------------------------------------------------------------------
struct StereoArgs {
  int max_l, max_r, rows, nlevels, band, desc_in_lds, th_dist;
  const int* n_l; const int* n_r;
  const float* l_x; const float* l_y; const int* l_oct; const uint8_t* l_desc;
  const float* r_x; const float* r_y; const int* r_oct; const uint8_t* r_desc;
  float sf[kMaxLevels];
  float min_d, max_d;
  int* best_idx; int* best_dist; int* n_matches;
};

__global__ __launch_bounds__(M_THREADS) void stereo_kernel(StereoArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int s_nm;
  const int pair = blockIdx.x, tid = threadIdx.x;
  const int nl = min(A.n_l[pair], A.max_l), nr = min(A.n_r[pair], A.max_r);
  unsigned char* p = smem;
  int* start = reinterpret_cast<int*>(p); p += sizeof(int) * ((size_t)A.rows + 4);
  float* rx = reinterpret_cast<float*>(p); p += sizeof(float) * A.max_r;
  float* ry = reinterpret_cast<float*>(p); p += sizeof(float) * A.max_r;
  unsigned short* items = reinterpret_cast<unsigned short*>(p); p += sizeof(unsigned short) * ((A.max_r + 1) & ~1);
  unsigned char* roct = p; p += (A.max_r + 3) & ~3;
  p = reinterpret_cast<unsigned char*>((reinterpret_cast<uintptr_t>(p) + 15) & ~(uintptr_t)15);
  const uint4* rdesc = reinterpret_cast<const uint4*>(A.r_desc) + 2 * (size_t)pair * A.max_r;
  if (A.desc_in_lds) {
    uint4* d = reinterpret_cast<uint4*>(p);
    for (int j = tid; j < 2 * nr; j += M_THREADS) d[j] = rdesc[j];
    rdesc = d;
  }
  const size_t lbase = (size_t)pair * A.max_l, rbase = (size_t)pair * A.max_r;
  // right keypoints into row buckets (counts at start[row + 2], scan, fill: as the grid of match_kernel)
  for (int r = tid; r < A.rows + 2; r += M_THREADS) start[r] = 0;
  if (tid == 0) s_nm = 0;
  __syncthreads();
  for (int j = tid; j < nr; j += M_THREADS) {
    rx[j] = A.r_x[rbase + j]; ry[j] = A.r_y[rbase + j];
    roct[j] = (unsigned char)min(max(A.r_oct[rbase + j], 0), A.nlevels - 1);
    atomicAdd(&start[min(max((int)ry[j], 0), A.rows - 1) + 2], 1);
  }
  __syncthreads();
  if (tid < 64) {
    const int per = (A.rows + 63) / 64;
    int sum = 0;
    for (int k = 0; k < per; k++) { const int r = tid * per + k; if (r < A.rows) sum += start[r + 2]; }
    int incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (tid >= o) incl += v; }
    int run = incl - sum;
    for (int k = 0; k < per; k++) { const int r = tid * per + k; if (r < A.rows) { run += start[r + 2]; start[r + 2] = run; } }
  }
  __syncthreads();
  for (int j = tid; j < nr; j += M_THREADS)
    items[atomicAdd(&start[min(max((int)ry[j], 0), A.rows - 1) + 1], 1)] = (unsigned short)j;
  __syncthreads();

  for (int i = tid; i < nl; i += M_THREADS) {
    const int levelL = A.l_oct[lbase + i];
    const float vL = A.l_y[lbase + i], uL = A.l_x[lbase + i];
    const int row = (int)vL;
    uint32_t best = ((uint32_t)TH_HIGH << 16) | 0xFFFFu;
    const float minU = uL - A.max_d, maxU = uL - A.min_d;
    if (row >= 0 && row < A.rows && !(maxU < 0)) {
      const uint4* ld = reinterpret_cast<const uint4*>(A.l_desc) + 2 * (lbase + i);
      const uint4 d0 = ld[0], d1 = ld[1];
      const uint32_t qd[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
      const int lo = max(row - A.band, 0), hi = min(row + A.band, A.rows - 1);
      for (int q = start[lo]; q < start[hi + 1]; q++) {
        const int j = items[q];
        const int o = roct[j];
        if (o < levelL - 1 || o > levelL + 1) continue;
        const float r = 2.0f * A.sf[o];
        const int maxr = (int)ceilf(ry[j] + r), minr = (int)floorf(ry[j] - r);
        if (row < minr || row > maxr) continue;
        const float uR = rx[j];
        if (!(uR >= minU && uR <= maxU)) continue;
        const uint32_t d = (uint32_t)hamming256(qd, rdesc[2 * j], rdesc[2 * j + 1]);
        best = min(best, (d << 16) | (uint32_t)j);
      }
    }
    const int bd = (int)(best >> 16);
    const bool hit = bd < A.th_dist && bd < TH_HIGH;
    A.best_dist[lbase + i] = min(bd, TH_HIGH);
    A.best_idx[lbase + i] = hit ? (int)(best & 0xFFFFu) : -1;
    if (hit) atomicAdd(&s_nm, 1);
  }
  __syncthreads();
  if (tid == 0) A.n_matches[pair] = s_nm;
}

size_t stereo_lds_bytes(int rows, int max_r, bool desc_in_lds) {
  size_t n = sizeof(int) * ((size_t)rows + 4) + 2 * sizeof(float) * (size_t)max_r +
             sizeof(unsigned short) * (size_t)((max_r + 1) & ~1) + (size_t)((max_r + 3) & ~3) + 32;
  if (desc_in_lds) n += 32 * (size_t)max_r;
  return n;
}

CtxPool& ctx_pool() {
  static CtxPool pool;
  return pool;
}

constexpr size_t kScratchPerQuery = sizeof(uint2) * MK + sizeof(int) * 3 + sizeof(float4);

void carve_scratch(MatchArgs& A, void* scratch, size_t nq) {
  unsigned char* p = reinterpret_cast<unsigned char*>(scratch);
  A.q_geo = reinterpret_cast<float4*>(p); p += sizeof(float4) * nq;
  A.lists = reinterpret_cast<uint32_t*>(p); p += sizeof(uint2) * MK * nq;  // (four bytes a candidate are in use)
  A.q_ntot = reinterpret_cast<int*>(p); p += sizeof(int) * nq;
  A.q_slot = reinterpret_cast<int*>(p); p += sizeof(int) * nq;
  A.q_bin = reinterpret_cast<int*>(p);
}

constexpr size_t kMatchLdsLimit = 156 * 1024;
constexpr size_t kMatchLdsComfort = 96 * 1024;  // up to here the descriptors come along into LDS
constexpr int kFewPairs = 16;

// true if the train side does not fit match_kernel's LDS-resident form (about 7900 keypoints; 6500
// for SearchForInitialization, which also keeps an owner per keypoint) and the call needs the
// per-pair block of global scratch of the large form
template <int MODE>
bool match_needs_big(int max_t, int max_q) {
  return match_lds_bytes(max_t, max_q, false, false, MODE == MODE_INIT) > kMatchLdsLimit;
}

template <int MODE, bool BIG>
int launch_match_as(const MatchArgs& A, int n_pairs, size_t lds, hipStream_t st) {
  static std::atomic<bool> raised[64];  // per device: raise the function's dynamic LDS limit once
  int dev = 0;
  ORB_HIP_CHECK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !raised[dev].load(std::memory_order_acquire)) {
    ORB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&match_kernel<MODE, BIG>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMatchLdsLimit));
    if (dev >= 0 && dev < 64) raised[dev].store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL((match_kernel<MODE, BIG>), dim3(n_pairs), dim3(M_THREADS), lds, st, A);
  ORB_HIP_CHECK(hipGetLastError());
  return ORBHIP_OK;
}

// The per-thread queues of the window scans: kQueueMin entries each are part of the sizes above;
// up to kQueueMax where the LDS has room left.
void give_queue_the_rest(MatchArgs& A, size_t& lds) {
  const size_t one = sizeof(unsigned short) * M_THREADS;
  const int extra = lds < kMatchLdsLimit ? (int)std::min<size_t>((kMatchLdsLimit - lds) / one, kQueueMax - kQueueMin) : 0;
  A.queue_cap = kQueueMin + extra;
  lds += one * extra;
}

template <int MODE>
int launch_match(const MatchArgs& A0, int n_pairs, hipStream_t st) {
  MatchArgs A = A0;
  A.queue_cap = kQueueMin;
  if (!A.keyframe_origin) { A.q_min_x = A.b.min_x; A.q_min_y = A.b.min_y; }
  else { A.q_min_x = (float)(int)A.b.min_x; A.q_min_y = (float)(int)A.b.min_y; }  // KeyFrame::mnMinX / mnMinY are const int
  constexpr bool own = MODE == MODE_INIT;
  if (match_needs_big<MODE>(A.max_t, A.max_q)) {
    if (!A.big) { set_error("too many train keypoints for the on-chip grid and no scratch for the large form was given"); return ORBHIP_ERR_CAPACITY; }
    A.desc_in_lds = 0; A.lists_in_lds = 0;
    size_t lds = match_big_lds_bytes(A.max_t);
    if (lds > kMatchLdsLimit) { set_error("too many train keypoints (limit 65535)"); return ORBHIP_ERR_UNSUPPORTED; }
    give_queue_the_rest(A, lds);
    return launch_match_as<MODE, true>(A, n_pairs, lds, st);
  }
  // On chip beside the grid: the ranked lists, which the replay - every step a chain of dependent
  // reads - walks in order.  The train descriptors only for the vocabulary-bucket searches, which
  // take a distance to every item of a bucket: a window search takes two or three per query, less
  // than staging the descriptors costs (one pair of 2000 x 2000: 0.093 ms staged, 0.076 ms from
  // global memory, where they stay in L2).  And a workgroup that fills the LDS keeps the blocks of
  // every other kernel off its compute unit - an extraction beside a batch of 255 pairs took 4 %
  // longer with 148 KB here than with 108 - so batches stage descriptors only while all of it
  // stays small; a call for a few pairs holds few compute units and may fill the LDS.
  A.lists_in_lds = match_lds_bytes(A.max_t, A.max_q, true, false, own) <= kMatchLdsLimit;
  A.desc_in_lds = mode_bucketed(MODE) && match_lds_bytes(A.max_t, A.max_q, A.lists_in_lds != 0, true, own) <=
                                             (n_pairs <= kFewPairs ? kMatchLdsLimit : kMatchLdsComfort);
  size_t lds = match_lds_bytes(A.max_t, A.max_q, A.lists_in_lds != 0, A.desc_in_lds != 0, own);
  if (!mode_bucketed(MODE)) give_queue_the_rest(A, lds);
  return launch_match_as<MODE, false>(A, n_pairs, lds, st);
}

template <int MODE>
int run_match(MatchArgs& A, int n_pairs, DevBuf& buf) {
  const size_t nq = (size_t)n_pairs * A.max_q;
  unsigned char* scratch;
  UP(buf, (const unsigned char*)nullptr, nq * kScratchPerQuery + 64, scratch);
  carve_scratch(A, scratch, nq);
  if (match_needs_big<MODE>(A.max_t, A.max_q)) {
    A.big_stride = match_big_stride(A.max_t);
    UP(buf, (const unsigned char*)nullptr, A.big_stride * (size_t)n_pairs, A.big);
  }
  if (A.ts == 0) A.ts = 1;
  if (A.qs == 0) A.qs = 1;
  return launch_match<MODE>(A, n_pairs, buf.stream());  // the caller copies the results back and waits
}

// Merge-walk of two ordered DBoW2 feature vectors (reference src/ORBmatcher.cc:251-352,
// :730-817, :894-1011): the side-1 features of every common node, in the reference's
// visiting order, each with the CSR range of its side-2 bucket.
int bucket_queries(int n1, int n2, int n_nodes1, const int* node1, const int* start1, const int* items1, int n_nodes2,
                   const int* node2, const int* start2, const int* items2, std::vector<int>& q_idx1,
                   std::vector<int>& q_b0, std::vector<int>& q_b1) {
  for (int a = 0, b = 0; a < n_nodes1 && b < n_nodes2;) {
    if (node1[a] == node2[b]) {
      for (int p = start1[a]; p < start1[a + 1]; p++) {
        const int idx1 = items1[p];
        if (idx1 < 0 || idx1 >= n1) { set_error("feature index out of range"); return ORBHIP_ERR_ARG; }
        q_idx1.push_back(idx1); q_b0.push_back(start2[b]); q_b1.push_back(start2[b + 1]);
      }
      a++; b++;
    } else if (node1[a] < node2[b]) {
      a = (int)(std::lower_bound(node1 + a, node1 + n_nodes1, node2[b]) - node1);
    } else {
      b = (int)(std::lower_bound(node2 + b, node2 + n_nodes2, node1[a]) - node2);
    }
  }
  const int nitems2 = start2[n_nodes2];
  for (int p = 0; p < nitems2; p++)
    if (items2[p] < 0 || items2[p] >= n2) { set_error("feature index out of range"); return ORBHIP_ERR_ARG; }
  return ORBHIP_OK;
}

}  // namespace orbhip

// Scratch of orb_search_by_projection_frames_noalloc: the per-query work arrays of match_kernel
// (window geometry, ranked candidate lists, counts, chosen slot, rotation bin) for up to
// ORBHIP_NOALLOC_MAX_QUERIES queries.  Every GPU the library runs on holds its own copy of this
// array at its own address until the process ends.  In host code the name below is only a handle
// for hipGetSymbolAddress; kernels and copies get the address that call returned.
static __device__ __align__(256) unsigned char scratch[(size_t)ORBHIP_NOALLOC_MAX_QUERIES * orbhip::kScratchPerQuery + 64];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
void* g_scratch_ptr[kMaxDevices] = {};  // device address of `scratch` on each GPU, 0 = not asked yet

// Address of `scratch` on the current device `dev`, looked up once per device.
int noalloc_scratch_address(int dev, void** out) {
  if (dev < 0 || dev >= kMaxDevices) { orbhip::set_error("device index out of range"); return ORBHIP_ERR_ARG; }
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[dev]) {
    void* p = nullptr;
    hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return orbhip::hip_fail(e, "hipGetSymbolAddress(scratch)");
    if (!p) { orbhip::set_error("hipGetSymbolAddress returned a null address"); return ORBHIP_ERR_HIP; }
    g_scratch_ptr[dev] = p;
  }
  *out = g_scratch_ptr[dev];
  return ORBHIP_OK;
}
}  // namespace

using namespace orbhip;

extern "C" {

int orb_hamming_distance_pairs(int device_id, const uint8_t* a, const uint8_t* b, int n, int* out) {
  if (n <= 0) return ORBHIP_OK;
  if (!a || !b || !out) { set_error("null argument"); return ORBHIP_ERR_ARG; }
  DevSel g(device_id);
  DevBuf buf;
  uint8_t *da, *db; int* dout;
  UP(buf, a, (size_t)n * 32, da);
  UP(buf, b, (size_t)n * 32, db);
  UP(buf, (const int*)nullptr, n, dout);
  hipLaunchKernelGGL(hamming_pairs_kernel, dim3((n + 255) / 256), dim3(256), 0, buf.stream(), (const uint4*)da, (const uint4*)db, n, dout);
  ORB_HIP_CHECK(hipGetLastError());
  DOWN(buf, out, dout, n);
  SYNC(buf);
  return ORBHIP_OK;
}

int orb_knn2_device_algo(int device_id, const uint8_t* d_q, int n_q, const uint8_t* d_db, int n_db, int db_index_base,
                         float ratio, int* d_best, int* d_second, int* d_idx, uint8_t* d_accept, void* stream, int algo) {
  if (n_q <= 0) return ORBHIP_OK;
  if (!d_q || !d_best || !d_second || !d_idx || (n_db > 0 && !d_db) || n_db < 0) { set_error("null argument"); return ORBHIP_ERR_ARG; }
  if (((uintptr_t)d_q | (uintptr_t)d_db) & 15) { set_error("descriptor arrays must be 16-byte aligned"); return ORBHIP_ERR_ARG; }
  if (algo < 0 || algo > 2) { set_error("algo must be 0 (auto), 1 (popcount) or 2 (matrix cores)"); return ORBHIP_ERR_ARG; }
  DevSel g(device_id);
  // the matrix-core kernel pays off once a workgroup's 512 queries meet a few thousand rows
  const bool mfma = algo == 2 || (algo == 0 && n_db >= 2048 && n_q >= 256);
  if (mfma && n_db > 0) {
    ORB_HIP_CHECK(launch_knn2_mfma(d_q, n_q, d_db, n_db, db_index_base, ratio, d_best, d_second, d_idx, d_accept,
                                   (hipStream_t)stream));
    return ORBHIP_OK;
  }
  const int per_block = KNN_THREADS * KNN_KQ;
  hipLaunchKernelGGL(knn2_kernel, dim3((n_q + per_block - 1) / per_block), dim3(KNN_THREADS), 0, (hipStream_t)stream,
                     (const uint4*)d_q, n_q, (const uint4*)d_db, n_db, db_index_base, ratio, d_best, d_second, d_idx,
                     d_accept);
  ORB_HIP_CHECK(hipGetLastError());
  return ORBHIP_OK;
}

int orb_knn2_device(int device_id, const uint8_t* d_q, int n_q, const uint8_t* d_db, int n_db, int db_index_base,
                    float ratio, int* d_best, int* d_second, int* d_idx, uint8_t* d_accept, void* stream) {
  return orb_knn2_device_algo(device_id, d_q, n_q, d_db, n_db, db_index_base, ratio, d_best, d_second, d_idx, d_accept,
                              stream, 0);
}

int orb_knn2_merge_device(int device_id, int n_shards, int n_q, const int* d_best, const int* d_second,
                          const int* d_idx, float ratio, int* d_out_best, int* d_out_second, int* d_out_idx,
                          uint8_t* d_out_accept, void* stream) {
  if (n_q <= 0) return ORBHIP_OK;
  if (n_shards < 1 || !d_best || !d_second || !d_idx || !d_out_best || !d_out_second || !d_out_idx) { set_error("null argument"); return ORBHIP_ERR_ARG; }
  DevSel g(device_id);
  ORB_HIP_CHECK(launch_knn2_merge(n_shards, n_q, (size_t)n_q, d_best, d_second, d_idx, ratio, d_out_best, d_out_second,
                                  d_out_idx, d_out_accept, (hipStream_t)stream));
  return ORBHIP_OK;
}

int orb_knn2_pack_device(int device_id, int n_q, const int* d_best, const int* d_second, const int* d_idx,
                         void* d_packed, void* stream) {
  if (n_q <= 0) return ORBHIP_OK;
  if (!d_best || !d_second || !d_idx || !d_packed) { set_error("null argument"); return ORBHIP_ERR_ARG; }
  if ((uintptr_t)d_packed & 7) { set_error("packed records must be 8-byte aligned"); return ORBHIP_ERR_ARG; }
  DevSel g(device_id);
  ORB_HIP_CHECK(launch_knn2_pack(n_q, d_best, d_second, d_idx, d_packed, (hipStream_t)stream));
  return ORBHIP_OK;
}

int orb_knn2_merge_packed_device(int device_id, int n_shards, int n_q, const void* d_packed, float ratio,
                                 int* d_out_best, int* d_out_second, int* d_out_idx, uint8_t* d_out_accept,
                                 void* stream) {
  if (n_q <= 0) return ORBHIP_OK;
  if (n_shards < 1 || !d_packed || !d_out_best || !d_out_second || !d_out_idx) { set_error("null argument"); return ORBHIP_ERR_ARG; }
  if ((uintptr_t)d_packed & 7) { set_error("packed records must be 8-byte aligned"); return ORBHIP_ERR_ARG; }
  DevSel g(device_id);
  ORB_HIP_CHECK(launch_knn2_merge_packed(n_shards, n_q, d_packed, ratio, d_out_best, d_out_second, d_out_idx,
                                         d_out_accept, (hipStream_t)stream));
  return ORBHIP_OK;
}

int orb_knn2_bruteforce(int device_id, const uint8_t* q, int n_q, const uint8_t* db, int n_db, float ratio, int* best,
                        int* second, int* idx, uint8_t* accept) {
  if (n_q <= 0) return ORBHIP_OK;
  if (!q || !best || !second || !idx || (n_db > 0 && !db)) { set_error("null argument"); return ORBHIP_ERR_ARG; }
  DevSel g(device_id);
  DevBuf buf;
  uint8_t *dq, *ddb, *dacc; int *db_, *ds_, *di_;
  UP(buf, q, (size_t)n_q * 32, dq);
  UP(buf, db, (size_t)n_db * 32, ddb);
  UP(buf, (const int*)nullptr, n_q, db_);
  UP(buf, (const int*)nullptr, n_q, ds_);
  UP(buf, (const int*)nullptr, n_q, di_);
  UP(buf, (const uint8_t*)nullptr, n_q, dacc);
  int rc = orb_knn2_device(device_id, dq, n_q, ddb, n_db, 0, ratio, db_, ds_, di_, dacc, buf.stream());
  if (rc != ORBHIP_OK) return rc;
  DOWN(buf, best, db_, n_q);
  DOWN(buf, second, ds_, n_q);
  DOWN(buf, idx, di_, n_q);
  if (accept) DOWN(buf, accept, dacc, n_q);
  SYNC(buf);
  return ORBHIP_OK;
}

int orb_search_by_projection_frames(int device_id, int n_pairs, int max_last, int max_cur, const float* tcw, float fx,
                                    float fy, float cx, float cy, OrbFrameBounds bounds, const float* scale_factors,
                                    int nlevels, const int* n_last, const int* last_mp_obs, const float* last_xw,
                                    const uint8_t* last_mp_desc, const int* last_octave, const float* last_angle,
                                    const int* n_cur, const float* cur_x, const float* cur_y, const int* cur_octave,
                                    const float* cur_angle, const uint8_t* cur_desc, const int* cur_obs, float th,
                                    int check_orientation, int* cur_match, int* n_matches) {
  if (n_pairs <= 0) return ORBHIP_OK;
  if (max_last <= 0 || max_cur <= 0 || max_cur > 65535 || nlevels < 1 || nlevels > kMaxLevels) { set_error("size out of range"); return ORBHIP_ERR_ARG; }
  if (!tcw || !scale_factors || !n_last || !last_mp_obs || !last_xw || !last_mp_desc || !last_octave || !last_angle ||
      !n_cur || !cur_x || !cur_y || !cur_octave || !cur_angle || !cur_desc || !cur_obs || !cur_match || !n_matches) {
    set_error("null argument"); return ORBHIP_ERR_ARG;
  }
  DevSel g(device_id);
  DevBuf buf;
  MatchArgs A{};
  A.max_q = max_last; A.max_t = max_cur; A.nlevels = nlevels; A.check_ori = check_orientation;
  A.fx = fx; A.fy = fy; A.cx = cx; A.cy = cy; A.b = bounds; A.th = th;
  for (int i = 0; i < nlevels; i++) A.sf[i] = scale_factors[i];
  const size_t nq = (size_t)n_pairs * max_last, nt = (size_t)n_pairs * max_cur;
  UP(buf, n_last, n_pairs, A.n_q); UP(buf, n_cur, n_pairs, A.n_t);
  UP(buf, tcw, (size_t)n_pairs * 16, A.tcw);
  UP(buf, last_mp_obs, nq, A.q_obs); UP(buf, last_xw, nq * 3, A.q_xw); UP(buf, last_mp_desc, nq * 32, A.q_desc);
  UP(buf, last_octave, nq, A.q_oct); UP(buf, last_angle, nq, A.q_angle);
  UP(buf, cur_x, nt, A.t_x); UP(buf, cur_y, nt, A.t_y); UP(buf, cur_octave, nt, A.t_oct);
  UP(buf, cur_angle, nt, A.t_angle); UP(buf, cur_desc, nt * 32, A.t_desc); UP(buf, cur_obs, nt, A.t_obs);
  UP(buf, (const int*)nullptr, nt, A.out_match); UP(buf, (const int*)nullptr, n_pairs, A.n_matches);
  int rc = run_match<MODE_FRAMES>(A, n_pairs, buf);
  if (rc != ORBHIP_OK) return rc;
  DOWN(buf, cur_match, A.out_match, nt);
  DOWN(buf, n_matches, A.n_matches, n_pairs);
  SYNC(buf);
  return ORBHIP_OK;
}

size_t orb_match_scratch_bytes(int n_pairs, int max_queries) {
  if (n_pairs <= 0 || max_queries <= 0) return 64;
  return (size_t)n_pairs * max_queries * kScratchPerQuery + 64;
}

size_t orb_match_scratch_bytes_for(int n_pairs, int max_last, int max_cur) {
  size_t n = (orb_match_scratch_bytes(n_pairs, max_last) + 255) & ~(size_t)255;
  if (n_pairs > 0 && max_cur > 0 && match_needs_big<MODE_FRAMES>(max_cur, max_last)) n += match_big_stride(max_cur) * (size_t)n_pairs;
  return n;
}

// Shared body of the two device-record forms below: argument checks, the MatchArgs of
// MODE_FRAMES on OrbKeyPoint records, launch on the caller's stream.  `scratch` is a device
// address holding orb_match_scratch_bytes(n_pairs, max_last) bytes.
static int frames_device_launch(int n_pairs, int max_last, int max_cur, const float* d_tcw, float fx, float fy, float cx,
                                float cy, OrbFrameBounds bounds, const float* scale_factors, int nlevels,
                                const int* d_n_last, const OrbKeyPoint* d_last_kps, const int* d_last_mp_obs,
                                const float* d_last_xw, const uint8_t* d_last_mp_desc, const int* d_n_cur,
                                const OrbKeyPoint* d_cur_kps, const uint8_t* d_cur_desc, const int* d_cur_obs, float th,
                                int check_orientation, void* scratch, size_t scratch_bytes, int* d_cur_match,
                                int* d_n_matches, hipStream_t stream) {
  if (((uintptr_t)d_last_mp_desc | (uintptr_t)d_cur_desc | (uintptr_t)scratch) & 15) { set_error("descriptor and scratch arrays must be 16-byte aligned"); return ORBHIP_ERR_ARG; }
  MatchArgs A{};
  A.max_q = max_last; A.max_t = max_cur; A.nlevels = nlevels; A.check_ori = check_orientation;
  A.fx = fx; A.fy = fy; A.cx = cx; A.cy = cy; A.b = bounds; A.th = th;
  for (int i = 0; i < nlevels; i++) A.sf[i] = scale_factors[i];
  A.ts = A.qs = (int)(sizeof(OrbKeyPoint) / 4);
  A.n_q = d_n_last; A.n_t = d_n_cur; A.tcw = d_tcw;
  A.q_obs = d_last_mp_obs; A.q_xw = d_last_xw; A.q_desc = d_last_mp_desc;
  A.q_oct = &reinterpret_cast<const int*>(d_last_kps)[5];
  A.q_angle = &reinterpret_cast<const float*>(d_last_kps)[3];
  const float* ck = reinterpret_cast<const float*>(d_cur_kps);
  A.t_x = ck; A.t_y = ck + 1; A.t_angle = ck + 3; A.t_oct = reinterpret_cast<const int*>(ck) + 5;
  A.t_desc = d_cur_desc; A.t_obs = d_cur_obs;
  A.out_match = d_cur_match; A.n_matches = d_n_matches;
  carve_scratch(A, scratch, (size_t)n_pairs * max_last);
  if (match_needs_big<MODE_FRAMES>(max_cur, max_last)) {
    // more current-frame keypoints than the on-chip grid holds: the large form's block lies behind
    // the per-query arrays, if the caller sized the scratch with orb_match_scratch_bytes_for
    if (scratch_bytes < orb_match_scratch_bytes_for(n_pairs, max_last, max_cur)) {
      set_error("max_cur = " + std::to_string(max_cur) + " keypoints do not fit the on-chip grid (about 7900): size the "
                "scratch with orb_match_scratch_bytes_for(n_pairs, max_last, max_cur)");
      return ORBHIP_ERR_CAPACITY;
    }
    A.big = reinterpret_cast<unsigned char*>(scratch) + ((orb_match_scratch_bytes(n_pairs, max_last) + 255) & ~(size_t)255);
    A.big_stride = match_big_stride(max_cur);
  }
  return launch_match<MODE_FRAMES>(A, n_pairs, stream);
}

static bool frames_device_args_ok(int max_last, int max_cur, int nlevels, const void* const* ptrs, int nptrs) {
  if (max_last <= 0 || max_cur <= 0 || max_cur > 65535 || nlevels < 1 || nlevels > kMaxLevels) { set_error("size out of range"); return false; }
  for (int i = 0; i < nptrs; i++)
    if (!ptrs[i]) { set_error("null argument"); return false; }
  return true;
}

int orb_search_by_projection_frames_device(int device_id, int n_pairs, int max_last, int max_cur, const float* d_tcw,
                                           float fx, float fy, float cx, float cy, OrbFrameBounds bounds,
                                           const float* scale_factors, int nlevels, const int* d_n_last,
                                           const OrbKeyPoint* d_last_kps, const int* d_last_mp_obs,
                                           const float* d_last_xw, const uint8_t* d_last_mp_desc, const int* d_n_cur,
                                           const OrbKeyPoint* d_cur_kps, const uint8_t* d_cur_desc,
                                           const int* d_cur_obs, float th, int check_orientation, void* d_scratch,
                                           size_t scratch_bytes, int* d_cur_match, int* d_n_matches, void* stream) {
  if (n_pairs <= 0) return ORBHIP_OK;
  const void* ptrs[] = {d_tcw, scale_factors, d_n_last, d_last_kps, d_last_mp_obs, d_last_xw, d_last_mp_desc,
                        d_n_cur, d_cur_kps, d_cur_desc, d_cur_obs, d_scratch, d_cur_match, d_n_matches};
  if (!frames_device_args_ok(max_last, max_cur, nlevels, ptrs, (int)(sizeof(ptrs) / sizeof(ptrs[0])))) return ORBHIP_ERR_ARG;
  if (scratch_bytes < orb_match_scratch_bytes(n_pairs, max_last)) { set_error("scratch too small"); return ORBHIP_ERR_CAPACITY; }
  DevSel g(device_id);
  return frames_device_launch(n_pairs, max_last, max_cur, d_tcw, fx, fy, cx, cy, bounds, scale_factors, nlevels, d_n_last,
                              d_last_kps, d_last_mp_obs, d_last_xw, d_last_mp_desc, d_n_cur, d_cur_kps, d_cur_desc,
                              d_cur_obs, th, check_orientation, d_scratch, scratch_bytes, d_cur_match, d_n_matches,
                              (hipStream_t)stream);
}

int orb_search_by_projection_frames_noalloc(int device_id, int n_pairs, int max_last, int max_cur, const float* d_tcw,
                                            float fx, float fy, float cx, float cy, OrbFrameBounds bounds,
                                            const float* scale_factors, int nlevels, const int* d_n_last,
                                            const OrbKeyPoint* d_last_kps, const int* d_last_mp_obs,
                                            const float* d_last_xw, const uint8_t* d_last_mp_desc, const int* d_n_cur,
                                            const OrbKeyPoint* d_cur_kps, const uint8_t* d_cur_desc,
                                            const int* d_cur_obs, float th, int check_orientation, int* d_cur_match,
                                            int* d_n_matches, void* stream) {
  if (n_pairs <= 0) return ORBHIP_OK;
  const void* ptrs[] = {d_tcw, scale_factors, d_n_last, d_last_kps, d_last_mp_obs, d_last_xw, d_last_mp_desc,
                        d_n_cur, d_cur_kps, d_cur_desc, d_cur_obs, d_cur_match, d_n_matches};
  if (!frames_device_args_ok(max_last, max_cur, nlevels, ptrs, (int)(sizeof(ptrs) / sizeof(ptrs[0])))) return ORBHIP_ERR_ARG;
  if ((long long)n_pairs * max_last > ORBHIP_NOALLOC_MAX_QUERIES) {
    set_error("n_pairs * max_last = " + std::to_string((long long)n_pairs * max_last) +
              " exceeds the built-in scratch of ORBHIP_NOALLOC_MAX_QUERIES = " +
              std::to_string(ORBHIP_NOALLOC_MAX_QUERIES) + " queries; use orb_search_by_projection_frames_device");
    return ORBHIP_ERR_CAPACITY;
  }
  if (match_needs_big<MODE_FRAMES>(max_cur, max_last)) {
    set_error("max_cur = " + std::to_string(max_cur) + " keypoints do not fit the on-chip grid (about 7900) and the large "
              "form needs scratch per frame pair; use orb_search_by_projection_frames_device");
    return ORBHIP_ERR_CAPACITY;
  }
  DevSel g(device_id);
  void* sp = nullptr;
  int rc = noalloc_scratch_address(device_id, &sp);
  if (rc != ORBHIP_OK) return rc;
  // nothing in the scratch has to start at zero: the kernel writes every entry of a pair's
  // slice before it reads it
  return frames_device_launch(n_pairs, max_last, max_cur, d_tcw, fx, fy, cx, cy, bounds, scale_factors, nlevels, d_n_last,
                              d_last_kps, d_last_mp_obs, d_last_xw, d_last_mp_desc, d_n_cur, d_cur_kps, d_cur_desc,
                              d_cur_obs, th, check_orientation, sp, sizeof(scratch), d_cur_match, d_n_matches,
                              (hipStream_t)stream);
}

int orb_search_by_projection_mappoints(int device_id, OrbFrameBounds bounds, const float* scale_factors, int nlevels,
                                       int n_f, const float* f_x, const float* f_y, const int* f_octave,
                                       const uint8_t* f_desc, const int* f_obs, int n_mp, const uint8_t* mp_track,
                                       const int* mp_level, const float* mp_viewcos, const float* mp_projx,
                                       const float* mp_projy, const uint8_t* mp_desc, const int* mp_obs, float th,
                                       float nnratio, int* f_match, int* n_matches) {
  if (!n_matches) { set_error("null argument"); return ORBHIP_ERR_ARG; }
  *n_matches = 0;
  if (n_f <= 0) return ORBHIP_OK;
  if (n_f > 65535 || nlevels < 1 || nlevels > kMaxLevels) { set_error("size out of range"); return ORBHIP_ERR_ARG; }
  if (!f_match || !scale_factors || !f_x || !f_y || !f_octave || !f_desc || !f_obs) { set_error("null argument"); return ORBHIP_ERR_ARG; }
  if (n_mp <= 0) { for (int i = 0; i < n_f; i++) f_match[i] = -1; return ORBHIP_OK; }
  if (!mp_track || !mp_level || !mp_viewcos || !mp_projx || !mp_projy || !mp_desc || !mp_obs) { set_error("null argument"); return ORBHIP_ERR_ARG; }
  DevSel g(device_id);
  DevBuf buf;
  MatchArgs A{};
  A.max_q = n_mp; A.max_t = n_f; A.nlevels = nlevels; A.check_ori = 0;
  A.b = bounds; A.th = th; A.nnratio = nnratio;
  for (int i = 0; i < nlevels; i++) A.sf[i] = scale_factors[i];
  UP(buf, &n_mp, 1, A.n_q); UP(buf, &n_f, 1, A.n_t);
  UP(buf, mp_track, n_mp, A.q_track); UP(buf, mp_level, n_mp, A.q_oct); UP(buf, mp_viewcos, n_mp, A.q_viewcos);
  UP(buf, mp_projx, n_mp, A.q_x); UP(buf, mp_projy, n_mp, A.q_y); UP(buf, mp_desc, (size_t)n_mp * 32, A.q_desc);
  UP(buf, mp_obs, n_mp, A.q_obs);
  UP(buf, f_x, n_f, A.t_x); UP(buf, f_y, n_f, A.t_y); UP(buf, f_octave, n_f, A.t_oct);
  UP(buf, f_desc, (size_t)n_f * 32, A.t_desc); UP(buf, f_obs, n_f, A.t_obs);
  A.t_angle = A.t_x;
  UP(buf, (const int*)nullptr, n_f, A.out_match); UP(buf, (const int*)nullptr, 1, A.n_matches);
  int rc = run_match<MODE_MAPPOINTS>(A, 1, buf);
  if (rc != ORBHIP_OK) return rc;
  DOWN(buf, f_match, A.out_match, n_f);
  DOWN(buf, n_matches, A.n_matches, 1);
  SYNC(buf);
  return ORBHIP_OK;
}

static int match_windowed_impl(int keyframe, int device_id, OrbFrameBounds bounds, int n_t, const float* t_x, const float* t_y,
                       const int* t_octave, const float* t_angle, const uint8_t* t_desc, const uint8_t* t_taken,
                       int n_q, const uint8_t* q_active, const float* q_u, const float* q_v, const float* q_radius,
                       const int* q_min_level, const int* q_max_level, const float* q_angle, const uint8_t* q_desc,
                       int max_dist, int check_orientation, int* t_match, int* n_matches) {
  if (!n_matches) { set_error("null argument"); return ORBHIP_ERR_ARG; }
  *n_matches = 0;
  if (n_t <= 0) return ORBHIP_OK;
  if (n_t > 65535) { set_error("size out of range"); return ORBHIP_ERR_ARG; }
  if (!t_match || !t_x || !t_y || !t_octave || !t_desc || !t_taken) { set_error("null argument"); return ORBHIP_ERR_ARG; }
  if (n_q <= 0) { for (int i = 0; i < n_t; i++) t_match[i] = -1; return ORBHIP_OK; }
  if (!q_active || !q_u || !q_v || !q_radius || !q_min_level || !q_max_level || !q_desc ||
      (check_orientation && (!q_angle || !t_angle))) { set_error("null argument"); return ORBHIP_ERR_ARG; }
  DevSel g(device_id);
  DevBuf buf;
  MatchArgs A{};
  A.max_q = n_q; A.max_t = n_t; A.nlevels = 1; A.check_ori = check_orientation; A.b = bounds; A.max_dist = max_dist;
  A.keyframe_origin = keyframe;
  std::vector<int> taken(n_t);
  for (int i = 0; i < n_t; i++) taken[i] = t_taken[i] ? 1 : -1;
  UP(buf, &n_q, 1, A.n_q); UP(buf, &n_t, 1, A.n_t);
  UP(buf, q_active, n_q, A.q_track); UP(buf, q_u, n_q, A.q_x); UP(buf, q_v, n_q, A.q_y); UP(buf, q_radius, n_q, A.q_r);
  UP(buf, q_min_level, n_q, A.q_minl); UP(buf, q_max_level, n_q, A.q_maxl); UP(buf, q_desc, (size_t)n_q * 32, A.q_desc);
  UP(buf, t_x, n_t, A.t_x); UP(buf, t_y, n_t, A.t_y); UP(buf, t_octave, n_t, A.t_oct);
  UP(buf, t_desc, (size_t)n_t * 32, A.t_desc); UP(buf, taken.data(), n_t, A.t_obs);
  if (check_orientation) { UP(buf, q_angle, n_q, A.q_angle); UP(buf, t_angle, n_t, A.t_angle); }
  else { A.q_angle = A.q_x; A.t_angle = A.t_x; }
  UP(buf, (const int*)nullptr, n_t, A.out_match); UP(buf, (const int*)nullptr, 1, A.n_matches);
  int rc = run_match<MODE_WINDOW>(A, 1, buf);
  if (rc != ORBHIP_OK) return rc;
  DOWN(buf, t_match, A.out_match, n_t);
  DOWN(buf, n_matches, A.n_matches, 1);
  SYNC(buf);
  return ORBHIP_OK;
}

int orb_match_windowed(int device_id, OrbFrameBounds bounds, int n_t, const float* t_x, const float* t_y,
                       const int* t_octave, const float* t_angle, const uint8_t* t_desc, const uint8_t* t_taken,
                       int n_q, const uint8_t* q_active, const float* q_u, const float* q_v, const float* q_radius,
                       const int* q_min_level, const int* q_max_level, const float* q_angle, const uint8_t* q_desc,
                       int max_dist, int check_orientation, int* t_match, int* n_matches) {
  return match_windowed_impl(0, device_id, bounds, n_t, t_x, t_y, t_octave, t_angle, t_desc, t_taken, n_q, q_active, q_u, q_v,
                             q_radius, q_min_level, q_max_level, q_angle, q_desc, max_dist, check_orientation, t_match,
                             n_matches);
}

int orb_match_windowed_keyframe(int device_id, OrbFrameBounds bounds, int n_t, const float* t_x, const float* t_y,
                                const int* t_octave, const float* t_angle, const uint8_t* t_desc,
                                const uint8_t* t_taken, int n_q, const uint8_t* q_active, const float* q_u,
                                const float* q_v, const float* q_radius, const int* q_min_level,
                                const int* q_max_level, const float* q_angle, const uint8_t* q_desc, int max_dist,
                                int check_orientation, int* t_match, int* n_matches) {
  return match_windowed_impl(1, device_id, bounds, n_t, t_x, t_y, t_octave, t_angle, t_desc, t_taken, n_q, q_active, q_u, q_v,
                             q_radius, q_min_level, q_max_level, q_angle, q_desc, max_dist, check_orientation, t_match,
                             n_matches);
}

static int window_best_impl(int keyframe, int device_id, OrbFrameBounds bounds, int n_t, const float* t_x, const float* t_y,
                    const int* t_octave, const uint8_t* t_desc, const float* inv_level_sigma2, int nlevels, double chi2,
                    int n_q, const uint8_t* q_active, const float* q_u, const float* q_v, const float* q_radius,
                    const int* q_level, const uint8_t* q_desc, int* best_idx, int* best_dist) {
  if (n_q <= 0) return ORBHIP_OK;
  if (!best_idx || !best_dist) { set_error("null argument"); return ORBHIP_ERR_ARG; }
  for (int i = 0; i < n_q; i++) { best_idx[i] = -1; best_dist[i] = 0x7FFFFFFF; }
  if (n_t <= 0) return ORBHIP_OK;
  if (n_t > 65535) { set_error("size out of range"); return ORBHIP_ERR_ARG; }
  if (!t_x || !t_y || !t_octave || !t_desc || !q_active || !q_u || !q_v || !q_radius || !q_level || !q_desc) {
    set_error("null argument"); return ORBHIP_ERR_ARG;
  }
  if (chi2 > 0 && (!inv_level_sigma2 || nlevels < 1 || nlevels > kMaxLevels)) { set_error("level table out of range"); return ORBHIP_ERR_ARG; }
  DevSel g(device_id);
  DevBuf buf;
  MatchArgs A{};
  A.max_q = n_q; A.max_t = n_t; A.nlevels = 1; A.b = bounds; A.chi2 = chi2 > 0 ? chi2 : 0.0;
  A.keyframe_origin = keyframe;
  if (chi2 > 0) {
    for (int i = 0; i < kMaxLevels; i++) A.sigma2[i] = inv_level_sigma2[std::min(i, nlevels - 1)];
    for (int i = 0; i < n_t; i++)
      if (t_octave[i] < 0 || t_octave[i] >= nlevels) { set_error("octave out of range"); return ORBHIP_ERR_ARG; }
  }
  // the level test of the callers, kpLevel < n - 1 || kpLevel > n -> skip, as a level range
  std::vector<int> lo(n_q), hi(n_q), free_slots(n_t, -1);
  for (int i = 0; i < n_q; i++) { hi[i] = std::max(q_level[i], 0); lo[i] = q_level[i] - 1; }
  for (int i = 0; i < n_q; i++)
    if (q_level[i] < 0) { set_error("negative level"); return ORBHIP_ERR_ARG; }
  UP(buf, &n_q, 1, A.n_q); UP(buf, &n_t, 1, A.n_t);
  UP(buf, q_active, n_q, A.q_track); UP(buf, q_u, n_q, A.q_x); UP(buf, q_v, n_q, A.q_y); UP(buf, q_radius, n_q, A.q_r);
  UP(buf, lo.data(), n_q, A.q_minl); UP(buf, hi.data(), n_q, A.q_maxl); UP(buf, q_desc, (size_t)n_q * 32, A.q_desc);
  UP(buf, t_x, n_t, A.t_x); UP(buf, t_y, n_t, A.t_y); UP(buf, t_octave, n_t, A.t_oct);
  UP(buf, t_desc, (size_t)n_t * 32, A.t_desc); UP(buf, free_slots.data(), n_t, A.t_obs);
  A.q_angle = A.q_x; A.t_angle = A.t_x;
  UP(buf, (const int*)nullptr, n_q, A.out_match); UP(buf, (const int*)nullptr, n_q, A.out_dist);
  UP(buf, (const int*)nullptr, 1, A.n_matches);
  int rc = run_match<MODE_BEST>(A, 1, buf);
  if (rc != ORBHIP_OK) return rc;
  DOWN(buf, best_idx, A.out_match, n_q);
  DOWN(buf, best_dist, A.out_dist, n_q);
  SYNC(buf);
  return ORBHIP_OK;
}

int orb_window_best(int device_id, OrbFrameBounds bounds, int n_t, const float* t_x, const float* t_y,
                    const int* t_octave, const uint8_t* t_desc, const float* inv_level_sigma2, int nlevels, double chi2,
                    int n_q, const uint8_t* q_active, const float* q_u, const float* q_v, const float* q_radius,
                    const int* q_level, const uint8_t* q_desc, int* best_idx, int* best_dist) {
  return window_best_impl(0, device_id, bounds, n_t, t_x, t_y, t_octave, t_desc, inv_level_sigma2, nlevels, chi2, n_q,
                          q_active, q_u, q_v, q_radius, q_level, q_desc, best_idx, best_dist);
}

int orb_window_best_keyframe(int device_id, OrbFrameBounds bounds, int n_t, const float* t_x, const float* t_y,
                             const int* t_octave, const uint8_t* t_desc, const float* inv_level_sigma2, int nlevels,
                             double chi2, int n_q, const uint8_t* q_active, const float* q_u, const float* q_v,
                             const float* q_radius, const int* q_level, const uint8_t* q_desc, int* best_idx,
                             int* best_dist) {
  return window_best_impl(1, device_id, bounds, n_t, t_x, t_y, t_octave, t_desc, inv_level_sigma2, nlevels, chi2, n_q,
                          q_active, q_u, q_v, q_radius, q_level, q_desc, best_idx, best_dist);
}

int orb_search_for_initialization(int device_id, OrbFrameBounds bounds, int n1, const int* octave1, const float* angle1,
                                  const uint8_t* desc1, const float* prev_x, const float* prev_y, int n2,
                                  const float* x2, const float* y2, const int* octave2, const float* angle2,
                                  const uint8_t* desc2, float window_size, float nnratio, int check_orientation,
                                  int* matches12, int* n_matches) {
  if (!n_matches) { set_error("null argument"); return ORBHIP_ERR_ARG; }
  *n_matches = 0;
  if (n1 <= 0) return ORBHIP_OK;
  if (!matches12) { set_error("null argument"); return ORBHIP_ERR_ARG; }
  for (int i = 0; i < n1; i++) matches12[i] = -1;
  if (n2 <= 0) return ORBHIP_OK;
  if (n2 > 65535) { set_error("size out of range"); return ORBHIP_ERR_ARG; }
  if (!octave1 || !angle1 || !desc1 || !prev_x || !prev_y || !x2 || !y2 || !octave2 || !angle2 || !desc2) {
    set_error("null argument"); return ORBHIP_ERR_ARG;
  }
  DevSel g(device_id);
  DevBuf buf;
  MatchArgs A{};
  A.max_q = n1; A.max_t = n2; A.nlevels = 1; A.check_ori = check_orientation; A.b = bounds; A.nnratio = nnratio;
  // only features of the original image are matched (level1 > 0 -> continue, reference :569),
  // against level 0 of the other frame (GetFeaturesInArea(..., level1, level1), :575)
  std::vector<uint8_t> act(n1);
  std::vector<float> r(n1, window_size);
  std::vector<int> zero(std::max(n1, n2), 0), free_slots(n2, -1);
  for (int i = 0; i < n1; i++) act[i] = octave1[i] > 0 ? 0 : 1;
  UP(buf, &n1, 1, A.n_q); UP(buf, &n2, 1, A.n_t);
  UP(buf, act.data(), n1, A.q_track); UP(buf, prev_x, n1, A.q_x); UP(buf, prev_y, n1, A.q_y); UP(buf, r.data(), n1, A.q_r);
  UP(buf, zero.data(), n1, A.q_minl); A.q_maxl = A.q_minl;
  UP(buf, desc1, (size_t)n1 * 32, A.q_desc); UP(buf, angle1, n1, A.q_angle);
  UP(buf, x2, n2, A.t_x); UP(buf, y2, n2, A.t_y); UP(buf, octave2, n2, A.t_oct); UP(buf, angle2, n2, A.t_angle);
  UP(buf, desc2, (size_t)n2 * 32, A.t_desc); UP(buf, free_slots.data(), n2, A.t_obs);
  UP(buf, (const int*)nullptr, n1, A.out_match); UP(buf, (const int*)nullptr, 1, A.n_matches);
  int rc = run_match<MODE_INIT>(A, 1, buf);
  if (rc != ORBHIP_OK) return rc;
  DOWN(buf, matches12, A.out_match, n1);
  DOWN(buf, n_matches, A.n_matches, 1);
  SYNC(buf);
  return ORBHIP_OK;
}

int orb_search_for_triangulation(int device_id, int n1, const float* x1, const float* y1, const int* octave1,
                                 const float* angle1, const uint8_t* desc1, const uint8_t* has_mp1, int n2,
                                 const float* x2, const float* y2, const int* octave2, const float* angle2,
                                 const uint8_t* desc2, const uint8_t* has_mp2, int n_nodes1, const int* node1,
                                 const int* start1, const int* items1, int n_nodes2, const int* node2,
                                 const int* start2, const int* items2, const float* f12, float ex, float ey,
                                 const float* scale_factors2, const float* level_sigma2_2, int nlevels,
                                 int check_orientation, int* matches12, int* n_matches) {
  (void)octave1;
  if (!n_matches) { set_error("null argument"); return ORBHIP_ERR_ARG; }
  *n_matches = 0;
  if (n1 <= 0) return ORBHIP_OK;
  if (!matches12) { set_error("null argument"); return ORBHIP_ERR_ARG; }
  for (int i = 0; i < n1; i++) matches12[i] = -1;
  if (n2 <= 0 || n_nodes1 <= 0 || n_nodes2 <= 0) return ORBHIP_OK;
  if (n2 > 65535 || nlevels < 1 || nlevels > kMaxLevels) { set_error("size out of range"); return ORBHIP_ERR_ARG; }
  if (!x1 || !y1 || !angle1 || !desc1 || !has_mp1 || !x2 || !y2 || !octave2 || !angle2 || !desc2 || !has_mp2 ||
      !node1 || !start1 || !items1 || !node2 || !start2 || !items2 || !f12 || !scale_factors2 || !level_sigma2_2) {
    set_error("null argument"); return ORBHIP_ERR_ARG;
  }
  std::vector<int> q_idx1, q_b0, q_b1;
  {
    const int rcq = bucket_queries(n1, n2, n_nodes1, node1, start1, items1, n_nodes2, node2, start2, items2, q_idx1, q_b0, q_b1);
    if (rcq != ORBHIP_OK) return rcq;
  }
  const int nq = (int)q_idx1.size();
  if (nq == 0) return ORBHIP_OK;
  const int nitems2 = start2[n_nodes2];
  DevSel g(device_id);
  DevBuf buf;
  MatchArgs A{};
  const int max_q = std::max(nq, n1);  // per-pair slices hold both query-indexed and idx1-indexed arrays
  A.max_q = max_q; A.max_t = n2; A.nlevels = nlevels; A.check_ori = check_orientation;
  A.ex = ex; A.ey = ey;
  for (int i = 0; i < 9; i++) A.f12[i] = f12[i];
  for (int i = 0; i < nlevels; i++) { A.sf[i] = scale_factors2[i]; A.sigma2[i] = level_sigma2_2[i]; }
  std::vector<int> obs1(max_q, 0), obs2(n2), oct2(n2);
  for (int i = 0; i < n1; i++) obs1[i] = has_mp1[i] ? 1 : 0;
  for (int i = 0; i < n2; i++) { obs2[i] = has_mp2[i] ? 1 : 0; oct2[i] = std::min(std::max(octave2[i], 0), nlevels - 1); }
  std::vector<float> qx(max_q, 0.f), qy(max_q, 0.f), qa(max_q, 0.f);
  std::vector<uint8_t> qd((size_t)max_q * 32, 0);
  std::copy(x1, x1 + n1, qx.begin()); std::copy(y1, y1 + n1, qy.begin()); std::copy(angle1, angle1 + n1, qa.begin());
  std::copy(desc1, desc1 + (size_t)n1 * 32, qd.begin());
  q_idx1.resize(max_q, 0); q_b0.resize(max_q, 0); q_b1.resize(max_q, 0);
  UP(buf, &nq, 1, A.n_q); UP(buf, &n2, 1, A.n_t);
  UP(buf, obs1.data(), max_q, A.q_obs); UP(buf, qx.data(), max_q, A.q_x); UP(buf, qy.data(), max_q, A.q_y);
  UP(buf, qa.data(), max_q, A.q_angle); UP(buf, qd.data(), (size_t)max_q * 32, A.q_desc);
  UP(buf, q_idx1.data(), max_q, A.q_idx1); UP(buf, q_b0.data(), max_q, A.q_b0); UP(buf, q_b1.data(), max_q, A.q_b1);
  UP(buf, items2, nitems2, A.items2);
  UP(buf, x2, n2, A.t_x); UP(buf, y2, n2, A.t_y); UP(buf, oct2.data(), n2, A.t_oct); UP(buf, angle2, n2, A.t_angle);
  UP(buf, desc2, (size_t)n2 * 32, A.t_desc); UP(buf, obs2.data(), n2, A.t_obs);
  UP(buf, (const int*)nullptr, max_q, A.out_match); UP(buf, (const int*)nullptr, 1, A.n_matches);
  int rc = run_match<MODE_TRIANG>(A, 1, buf);
  if (rc != ORBHIP_OK) return rc;
  std::vector<int> m(max_q);
  DOWN(buf, m.data(), A.out_match, max_q);
  DOWN(buf, n_matches, A.n_matches, 1);
  SYNC(buf);
  std::copy(m.begin(), m.begin() + n1, matches12);
  return ORBHIP_OK;
}

int orb_search_by_bow(int device_id, int n1, const float* angle1, const uint8_t* desc1, const uint8_t* ok1, int n2,
                      const float* angle2, const uint8_t* desc2, const uint8_t* ok2, int n_nodes1, const int* node1,
                      const int* start1, const int* items1, int n_nodes2, const int* node2, const int* start2,
                      const int* items2, float nnratio, int max_dist, int check_orientation, int* matches12,
                      int* n_matches) {
  if (!n_matches) { set_error("null argument"); return ORBHIP_ERR_ARG; }
  *n_matches = 0;
  if (n1 <= 0) return ORBHIP_OK;
  if (!matches12) { set_error("null argument"); return ORBHIP_ERR_ARG; }
  for (int i = 0; i < n1; i++) matches12[i] = -1;
  if (n2 <= 0 || n_nodes1 <= 0 || n_nodes2 <= 0) return ORBHIP_OK;
  if (n2 > 65535) { set_error("size out of range"); return ORBHIP_ERR_ARG; }
  if (!angle1 || !desc1 || !ok1 || !angle2 || !desc2 || !ok2 || !node1 || !start1 || !items1 || !node2 || !start2 ||
      !items2) {
    set_error("null argument"); return ORBHIP_ERR_ARG;
  }
  std::vector<int> q_idx1, q_b0, q_b1;
  {
    const int rcq = bucket_queries(n1, n2, n_nodes1, node1, start1, items1, n_nodes2, node2, start2, items2, q_idx1, q_b0, q_b1);
    if (rcq != ORBHIP_OK) return rcq;
  }
  const int nq = (int)q_idx1.size();
  if (nq == 0) return ORBHIP_OK;
  const int nitems2 = start2[n_nodes2];
  DevSel g(device_id);
  DevBuf buf;
  MatchArgs A{};
  const int max_q = std::max(nq, n1);  // per-pair slices hold both query-indexed and idx1-indexed arrays
  A.max_q = max_q; A.max_t = n2; A.nlevels = 1; A.check_ori = check_orientation;
  A.nnratio = nnratio; A.max_dist = max_dist;
  // a side-2 feature that may not be matched at all starts out taken
  std::vector<int> obs1(max_q, 0), obs2(n2), oct2(n2, 0);
  for (int i = 0; i < n1; i++) obs1[i] = ok1[i] ? 1 : 0;
  for (int i = 0; i < n2; i++) obs2[i] = ok2[i] ? 0 : 1;
  std::vector<float> qa(max_q, 0.f);
  std::vector<uint8_t> qd((size_t)max_q * 32, 0);
  std::copy(angle1, angle1 + n1, qa.begin());
  std::copy(desc1, desc1 + (size_t)n1 * 32, qd.begin());
  q_idx1.resize(max_q, 0); q_b0.resize(max_q, 0); q_b1.resize(max_q, 0);
  UP(buf, &nq, 1, A.n_q); UP(buf, &n2, 1, A.n_t);
  UP(buf, obs1.data(), max_q, A.q_obs); UP(buf, qa.data(), max_q, A.q_angle); UP(buf, qd.data(), (size_t)max_q * 32, A.q_desc);
  A.q_x = A.q_angle; A.q_y = A.q_angle;  // positions play no part in this search
  UP(buf, q_idx1.data(), max_q, A.q_idx1); UP(buf, q_b0.data(), max_q, A.q_b0); UP(buf, q_b1.data(), max_q, A.q_b1);
  UP(buf, items2, nitems2, A.items2);
  UP(buf, angle2, n2, A.t_angle); A.t_x = A.t_angle; A.t_y = A.t_angle;
  UP(buf, oct2.data(), n2, A.t_oct);
  UP(buf, desc2, (size_t)n2 * 32, A.t_desc); UP(buf, obs2.data(), n2, A.t_obs);
  UP(buf, (const int*)nullptr, max_q, A.out_match); UP(buf, (const int*)nullptr, 1, A.n_matches);
  int rc = run_match<MODE_BOW>(A, 1, buf);
  if (rc != ORBHIP_OK) return rc;
  std::vector<int> m(max_q);
  DOWN(buf, m.data(), A.out_match, max_q);
  DOWN(buf, n_matches, A.n_matches, 1);
  SYNC(buf);
  std::copy(m.begin(), m.begin() + n1, matches12);
  return ORBHIP_OK;
}

int orb_stereo_match(int device_id, int rows, int n_left, const float* l_x, const float* l_y, const int* l_octave,
                     const uint8_t* l_desc, int n_right, const float* r_x, const float* r_y, const int* r_octave,
                     const uint8_t* r_desc, const float* scale_factors, int nlevels, float min_d, float max_d,
                     int th_dist, int* best_idx, int* best_dist, int* n_matches) {
  if (!n_matches) { set_error("null argument"); return ORBHIP_ERR_ARG; }
  *n_matches = 0;
  if (n_left <= 0) return ORBHIP_OK;
  if (!best_idx || !best_dist || !l_x || !l_y || !l_octave || !l_desc || !scale_factors) {
    set_error("null argument"); return ORBHIP_ERR_ARG;
  }
  for (int i = 0; i < n_left; i++) { best_idx[i] = -1; best_dist[i] = TH_HIGH; }
  if (n_right <= 0) return ORBHIP_OK;
  if (!r_x || !r_y || !r_octave || !r_desc) { set_error("null argument"); return ORBHIP_ERR_ARG; }
  if (n_right > 65535 || rows < 1 || rows > 16384 || nlevels < 1 || nlevels > kMaxLevels) {
    set_error("size out of range"); return ORBHIP_ERR_ARG;
  }
  StereoArgs A{};
  A.max_l = n_left; A.max_r = n_right; A.rows = rows; A.nlevels = nlevels; A.th_dist = th_dist;
  A.min_d = min_d; A.max_d = max_d;
  float sfmax = 0.f;
  for (int i = 0; i < nlevels; i++) { A.sf[i] = scale_factors[i]; sfmax = std::max(sfmax, scale_factors[i]); }
  A.band = (int)std::ceil(2.0f * sfmax) + 2;  // rows between a right keypoint's own row and the ends of its band
  constexpr size_t kLdsLimit = 156 * 1024;
  A.desc_in_lds = stereo_lds_bytes(rows, n_right, true) <= kLdsLimit;
  const size_t lds = stereo_lds_bytes(rows, n_right, A.desc_in_lds != 0);
  if (lds > kLdsLimit) { set_error("too many right keypoints or rows for the LDS-resident row buckets"); return ORBHIP_ERR_UNSUPPORTED; }
  DevSel g(device_id);
  DevBuf buf;
  UP(buf, &n_left, 1, A.n_l); UP(buf, &n_right, 1, A.n_r);
  UP(buf, l_x, n_left, A.l_x); UP(buf, l_y, n_left, A.l_y); UP(buf, l_octave, n_left, A.l_oct);
  UP(buf, l_desc, (size_t)n_left * 32, A.l_desc);
  UP(buf, r_x, n_right, A.r_x); UP(buf, r_y, n_right, A.r_y); UP(buf, r_octave, n_right, A.r_oct);
  UP(buf, r_desc, (size_t)n_right * 32, A.r_desc);
  UP(buf, (const int*)nullptr, n_left, A.best_idx); UP(buf, (const int*)nullptr, n_left, A.best_dist);
  UP(buf, (const int*)nullptr, 1, A.n_matches);
  ORB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stereo_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
  hipLaunchKernelGGL(stereo_kernel, dim3(1), dim3(M_THREADS), lds, buf.stream(), A);
  ORB_HIP_CHECK(hipGetLastError());
  DOWN(buf, best_idx, A.best_idx, n_left);
  DOWN(buf, best_dist, A.best_dist, n_left);
  DOWN(buf, n_matches, A.n_matches, 1);
  SYNC(buf);
  return ORBHIP_OK;
}

}  // extern "C"
