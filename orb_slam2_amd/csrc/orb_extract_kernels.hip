// HIP kernels of the ORB extractor for gfx950 (MI355X): pyramid, cell-wise FAST,
// quad-tree keypoint distribution, 7x7 blur, orientation + rBRIEF.
//
// Every kernel works on a batch.  The frame is blockIdx.x, the fastest-varying block
// coordinate: workgroups are dealt to the 8 XCDs round-robin, so with the frame in x all
// workgroups of one frame land on one XCD and share its L2 (halos, overlapping patches).
//  Integer stages
// are bit-exact restatements of the OpenCV arithmetic the reference calls; float
// stages use _rn intrinsics so that no multiply-add is contracted.
#include "orb_common.h"

#include <vector>

namespace orbhip {

__device__ __forceinline__ int refl101(int i, int n) {
  i = i < 0 ? -i : i;
  return i >= n ? 2 * n - 2 - i : i;
}

// ---------------------------------------------------------------------------
// Pyramid.
// Level 0: copy of the input with a 19-px BORDER_REFLECT_101 frame
// (reference src/ORBextractor.cc:1406-1408).  One thread writes 16 consecutive bytes of
// a padded slab row.  Input rows have any alignment (the stride need not be a multiple
// of 4), so interior pieces are read as five aligned dwords and shifted into place;
// pieces that touch the frame or the row ends go byte by byte.  The interior pieces are
// numbered first (row by row), the others after them, so that all but one wavefront hold a
// single kind and none runs both paths.
// ---------------------------------------------------------------------------
constexpr int L0_PIECES = 4;  // 16-byte pieces per thread of the level-0 copy (2: 1 % slower, 8: the same)
struct L0Map {
  int ca, ni;      // pieces [ca, ca + ni) of a row are interior (ni = 0: none)
  int nb;          // the other pieces of a row
  uint32_t ni_m, nb_m;  // 2^32 / n + 1: t / n == umulhi(t, m) for t * n < 2^32
  int n_int, total;     // interior pieces, all pieces (of one frame)
};
__global__ __launch_bounds__(256) void pyr_level0_kernel(const uint8_t* __restrict__ img, size_t stride,
                                                         size_t frame_stride, uint8_t* __restrict__ pyr,
                                                         unsigned long long slab, LevelGeom L, L0Map M,
                                                         int* __restrict__ cand_cnt, int nlevels) {
  // the frame's candidate counters start at zero (FAST, later in the stream, adds to them)
  if (blockIdx.y == 0 && threadIdx.x < nlevels) cand_cnt[blockIdx.x * nlevels + threadIdx.x] = 0;
  // a thread moves L0_PIECES pieces a fixed distance apart; all their loads are issued before
  // the first result is needed, so that more bytes are in flight per wavefront
  const int half = (M.total + L0_PIECES - 1) / L0_PIECES;
  const int t0 = blockIdx.y * 256 + threadIdx.x;
  if (t0 >= half) return;
  const int f = blockIdx.x;
  uint32_t x[L0_PIECES][5];
  const uint8_t* S[L0_PIECES];
  int px0[L0_PIECES], row[L0_PIECES], c16[L0_PIECES];
  bool live[L0_PIECES], fastp[L0_PIECES], own[L0_PIECES];
  uint32_t sh[L0_PIECES];
#pragma unroll
  for (int u = 0; u < L0_PIECES; u++) {
    const int t = t0 + u * half;
    own[u] = false;
    x[u][0] = 0;
    live[u] = t < M.total;
    const int tt = live[u] ? t : 0;
    fastp[u] = tt < M.n_int;
    int c;
    if (fastp[u]) {
      row[u] = (int)__umulhi((uint32_t)tt, M.ni_m);
      c = M.ca + tt - row[u] * M.ni;
    } else {
      const int v = tt - M.n_int;
      row[u] = (int)__umulhi((uint32_t)v, M.nb_m);
      const int j = v - row[u] * M.nb;
      c = j < M.ca ? j : j + M.ni;
    }
    c16[u] = c * 16;
    const int iy = refl101(row[u] - kEdge, L.h);
    S[u] = img + (size_t)f * frame_stride + (size_t)iy * stride;
    px0[u] = c16[u] - kXOff;
    if (fastp[u]) {
      // four dwords in one load (any dword alignment); the fifth is the next lane's first when
      // that lane holds the next piece of the same row, and a load of its own otherwise
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4), aligned(4)));
      const uintptr_t a = reinterpret_cast<uintptr_t>(S[u] + px0[u]);
      const uint32_t* q = reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3);
      sh[u] = (uint32_t)(a & 3);
      const u32x4 X = *reinterpret_cast<const u32x4*>(q);
      x[u][0] = X.x; x[u][1] = X.y; x[u][2] = X.z; x[u][3] = X.w;
      own[u] = (threadIdx.x & 63) == 63 || c + 1 >= M.ca + M.ni || t0 + 1 >= half;  // (the next lane may have left)
      x[u][4] = 0;
      if (own[u]) x[u][4] = q[4];
    }
  }
#pragma unroll
  for (int u = 0; u < L0_PIECES; u++) {  // (all lanes of the wavefront take part in the exchange)
    const uint32_t nxt = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * ((threadIdx.x & 63) + 1), (int)x[u][0]);
    if (fastp[u] && !own[u]) x[u][4] = nxt;
  }
#pragma unroll
  for (int u = 0; u < L0_PIECES; u++) {
    if (!live[u]) continue;
    uint4 out;
    if (fastp[u]) {
      out.x = __builtin_amdgcn_alignbyte(x[u][1], x[u][0], sh[u]);
      out.y = __builtin_amdgcn_alignbyte(x[u][2], x[u][1], sh[u]);
      out.z = __builtin_amdgcn_alignbyte(x[u][3], x[u][2], sh[u]);
      out.w = __builtin_amdgcn_alignbyte(x[u][4], x[u][3], sh[u]);
    } else {
      uint32_t o[4] = {0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < 16; k++) {
        const int px = px0[u] + k;
        if (px >= -kEdge && px < L.w + kEdge) o[k >> 2] |= (uint32_t)S[u][refl101(px, L.w)] << (8 * (k & 3));
      }
      out = make_uint4(o[0], o[1], o[2], o[3]);
    }
    *reinterpret_cast<uint4*>(pyr + (size_t)f * slab + L.off + (size_t)row[u] * L.pitch + c16[u]) = out;
  }
}

// Level 0 from an interleaved colour image: cv::cvtColor(..., CV_RGB2GRAY / CV_BGR2GRAY /
// CV_RGBA2GRAY / CV_BGRA2GRAY) of the reference's Tracking::GrabImage* (src/Tracking.cc:203-216)
// folded into the copy above, so the grey image never exists in memory.  8-bit cvtColor is
// integer: (c0*k0 + c1*k1 + c2*k2 + 2^(s-1)) >> s with BT.601 weights that sum to 2^s, so the
// result fits a byte: s = 15, R 9798, G 19235, B 3735 in OpenCV 4.x; s = 14, 4899 / 9617 / 1868
// before.  Alpha is ignored.
// A thread writes 16 slab bytes from 16*CH source bytes of any alignment.
template <int CH>
__global__ __launch_bounds__(256) void pyr_level0_color_kernel(const uint8_t* __restrict__ img, size_t stride,
                                                               size_t frame_stride, uint8_t* __restrict__ pyr,
                                                               unsigned long long slab, LevelGeom L, int k0, int k1,
                                                               int k2, int rnd, int shift, int* __restrict__ cand_cnt,
                                                               int nlevels) {
  if (blockIdx.y == 0 && threadIdx.x < nlevels) cand_cnt[blockIdx.x * nlevels + threadIdx.x] = 0;
  const int nchunk = L.pitch >> 4;
  const int t = blockIdx.y * 256 + threadIdx.x;
  const int row = t / nchunk;  // padded row
  if (row >= L.h + 2 * kEdge) return;
  const int c16 = (t - row * nchunk) * 16;
  const int f = blockIdx.x;
  const int iy = refl101(row - kEdge, L.h);
  const uint8_t* S = img + (size_t)f * frame_stride + (size_t)iy * stride;
  const int px0 = c16 - kXOff;
  uint32_t o[4] = {0, 0, 0, 0};
  if (px0 >= 4 && px0 + 20 <= L.w) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(S + (size_t)px0 * CH);
    const uint32_t* q = reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3);
    const uint32_t sh = (uint32_t)(a & 3);
    uint32_t x[4 * CH + 1], w[4 * CH];
#pragma unroll
    for (int j = 0; j <= 4 * CH; j++) x[j] = q[j];
#pragma unroll
    for (int j = 0; j < 4 * CH; j++) w[j] = __builtin_amdgcn_alignbyte(x[j + 1], x[j], sh);
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int b = k * CH;
      const int c0 = (w[b >> 2] >> (8 * (b & 3))) & 255;
      const int c1 = (w[(b + 1) >> 2] >> (8 * ((b + 1) & 3))) & 255;
      const int c2 = (w[(b + 2) >> 2] >> (8 * ((b + 2) & 3))) & 255;
      o[k >> 2] |= (uint32_t)((c0 * k0 + c1 * k1 + c2 * k2 + rnd) >> shift) << (8 * (k & 3));
    }
  } else {
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int px = px0 + k;
      if (px >= -kEdge && px < L.w + kEdge) {
        const uint8_t* c = S + (size_t)refl101(px, L.w) * CH;
        o[k >> 2] |= (uint32_t)((c[0] * k0 + c[1] * k1 + c[2] * k2 + rnd) >> shift) << (8 * (k & 3));
      }
    }
  }
  *reinterpret_cast<uint4*>(pyr + (size_t)f * slab + L.off + (size_t)row * L.pitch + c16) =
      make_uint4(o[0], o[1], o[2], o[3]);
}

// Level l >= 1: cv::resize(INTER_LINEAR, 8U fixed point) of level l-1's un-padded
// pixels, then the reflected frame of level l itself
// (reference src/ORBextractor.cc:1367-1389).  Frame pixels are recomputed from the
// mirrored coordinate instead of being copied, so one pass writes the whole slab row.
__global__ __launch_bounds__(256) void pyr_resize_kernel(uint8_t* __restrict__ pyr, unsigned long long slab,
                                                         LevelGeom Ls, LevelGeom Ld,
                                                         const XTab* __restrict__ xt,
                                                         const YTab* __restrict__ yt) {
  const int c4 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (c4 >= Ld.pitch) return;
  const int row = blockIdx.y;
  const int f = blockIdx.z;
  const int iy = refl101(row - kEdge, Ld.h);
  const YTab ty = yt[iy];
  const uint8_t* src = pyr + (size_t)f * slab + Ls.off + kXOff;
  const uint8_t* S0 = src + (size_t)(ty.sy0 + kEdge) * Ls.pitch;
  const uint8_t* S1 = src + (size_t)(ty.sy1 + kEdge) * Ls.pitch;
  const int b0 = ty.b0, b1 = ty.b1;
  uint32_t out = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int px = c4 + k - kXOff;
    if (px >= -kEdge && px < Ld.w + kEdge) {
      const XTab tx = xt[refl101(px, Ld.w)];
      const int sx = tx.sx;
      const int sx1 = min(sx + 1, Ls.w - 1);
      const int r0 = S0[sx] * tx.a0 + S0[sx1] * tx.a1;
      const int r1 = S1[sx] * tx.a0 + S1[sx1] * tx.a1;
      int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
      v = min(max(v, 0), 255);
      out |= (uint32_t)v << (8 * k);
    }
  }
  *reinterpret_cast<uint32_t*>(pyr + (size_t)f * slab + Ld.off + (size_t)row * Ld.pitch + c4) = out;
}

// Tiled form of the resize for scale factors up to about 1.45: a workgroup produces
// PR_TR rows of the level x PR_TC slab columns (the frame's columns with them, at their
// mirrored coordinates; the frame's rows as copies).  The source rows it needs are staged in LDS
// with 16-byte loads; a thread owns 4 adjacent output columns (one dword store per
// row), reads its taps from LDS and reuses the horizontal result of a source row
// when the next output row starts on it.  xtp[] holds, per SLAB column, the source
// column and the second tap's weight (first weight = 2048 - it; 0xFFFF marks columns
// outside the 19-px frame, which are written as 0).
constexpr int PR_TC = 1024, PR_TR = 8, PR_TR_SMALL = 4, PR_THREADS = 256;  // tile height: batches / a few frames (6 to 10 rows measure the same, 16 and 4 are 3 % slower)
constexpr int PR_SROWS = 32, PR_MAXPITCH = 1536;  // limits of the staged source region
static_assert(sizeof(YTab) == 8, "a row's vertical taps are fetched as one 64-bit word");

// STAGED: the caller has put the source rows into s_src already (pyr_level01_kernel)
template <bool STAGED>
__device__ __forceinline__ void pr_tile(uint8_t* __restrict__ pyr, unsigned long long slab, const LevelGeom& Ls,
                                        const LevelGeom& Ld, const uint32_t* __restrict__ xtp,
                                        const YTab* __restrict__ yt, const int2* __restrict__ xrange,
                                        const int2* __restrict__ yrange, int PR_SPITCH, int tile_rows, int gx, int gy,
                                        int f, uint8_t* s_src) {
  const int tid = threadIdx.x;
  const int2 X = xrange[gx], Y = yrange[gy];
  const int a0 = (X.x + kXOff) & ~15;                  // first staged slab column
  if constexpr (!STAGED) {
    const int n16 = (X.y + kXOff - a0 + 16) >> 4;        // 16-byte pieces per row
    const int nrows = Y.y - Y.x + 1;
    const uint8_t* src = pyr + (size_t)f * slab + Ls.off + (size_t)(Y.x + kEdge) * Ls.pitch + a0;
    for (int r = tid >> 6; r < nrows; r += PR_THREADS / 64)
      for (int c = tid & 63; c < n16; c += 64)
        *reinterpret_cast<uint4*>(&s_src[r * PR_SPITCH + 16 * c]) =
            *reinterpret_cast<const uint4*>(src + (size_t)r * Ls.pitch + 16 * c);
  }
  // the rows' vertical taps: lane i of every wavefront fetches those of the tile's row i now (the
  // loop below reads them with v_readlane instead of waiting for a scalar load per row)
  // (tiles cover the level's own rows; the rows of the reflected frame above and below are copies
  // of rows 1 .. 19 and h - 20 .. h - 2 and are stored along with those)
  const int rbeg = gy * tile_rows;
  unsigned long long ytl = 0;
  if ((tid & 63) < tile_rows)
    ytl = *reinterpret_cast<const unsigned long long*>(&yt[min(rbeg + (tid & 63), Ld.h - 1)]);
  __syncthreads();
  const int c = gx * PR_TC + 4 * tid;
  if (c >= Ld.pitch) return;
  const uint4 e = *reinterpret_cast<const uint4*>(&xtp[c]);
  const uint32_t ev[4] = {e.x, e.y, e.z, e.w};
  int col[4], w0[4], w1[4];
  bool valid[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int a1 = (int)(ev[k] >> 16);
    valid[k] = a1 != 0xFFFF;
    col[k] = valid[k] ? (int)(ev[k] & 0xFFFF) + kXOff - a0 : 0;
    w1[k] = valid[k] ? a1 : 0;
    w0[k] = valid[k] ? 2048 - a1 : 0;
  }
  // cv::resize's vertical step is ((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16), rounded by
  // (+2) >> 2 and saturated.  With the weight moved up 12 bits and the horizontal sum kept as
  // (h >> 4) << 4 (both fit 24 bits) each product-and-shift is one v_mul_hi_u32_u24; the result
  // never leaves 0..255 (the weights of a pair sum to 2048, or 2049 when both round up, and
  // h <= 255 * 2049, so the two terms add up to at most 1020), so no clamp is needed.
  auto mulhi24 = [](uint32_t a, uint32_t b) { return (uint32_t)(((unsigned long long)(a & 0xFFFFFFu) * (b & 0xFFFFFFu)) >> 32); };
  uint32_t hprev[4] = {0, 0, 0, 0};
  int prev_row = -1;  // source row whose horizontal result is in hprev
  uint8_t* dst = pyr + (size_t)f * slab + Ld.off + c;
  const int rend = min(gy * tile_rows + tile_rows, Ld.h);
  for (int r = rbeg; r < rend; r++) {
    const uint32_t ylo = __builtin_amdgcn_readlane((uint32_t)ytl, r - rbeg);
    const uint32_t yhi = __builtin_amdgcn_readlane((uint32_t)(ytl >> 32), r - rbeg);
    YTab ty;
    ty.sy0 = (uint16_t)ylo; ty.sy1 = (uint16_t)(ylo >> 16); ty.b0 = (int16_t)yhi; ty.b1 = (int16_t)(yhi >> 16);
    const int s0 = ty.sy0, s1 = ty.sy1;
    uint32_t h0[4], h1[4];
    if (s0 == prev_row) {
#pragma unroll
      for (int k = 0; k < 4; k++) h0[k] = hprev[k];
    } else {
      const uint8_t* R = &s_src[(s0 - Y.x) * PR_SPITCH];
#pragma unroll
      for (int k = 0; k < 4; k++) h0[k] = (uint32_t)(R[col[k]] * w0[k] + R[col[k] + 1] * w1[k]) & 0xFFFFF0u;
    }
    if (s1 == s0) {
#pragma unroll
      for (int k = 0; k < 4; k++) h1[k] = h0[k];
    } else {
      const uint8_t* R = &s_src[(s1 - Y.x) * PR_SPITCH];
#pragma unroll
      for (int k = 0; k < 4; k++) h1[k] = (uint32_t)(R[col[k]] * w0[k] + R[col[k] + 1] * w1[k]) & 0xFFFFF0u;
    }
    const uint32_t b0 = (uint32_t)ty.b0 << 12, b1 = (uint32_t)ty.b1 << 12;
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t v = (mulhi24(b0, h0[k]) + mulhi24(b1, h1[k]) + 2u) >> 2;
      out |= v << (8 * k);
      hprev[k] = h1[k];
    }
    prev_row = s1;
    *reinterpret_cast<uint32_t*>(dst + (size_t)(r + kEdge) * Ld.pitch) = out;
    const int up = kEdge - r, down = kEdge + 2 * (Ld.h - 1) - r;  // the row's mirror images in the frame (uniform)
    if (r >= 1 && up >= 0) *reinterpret_cast<uint32_t*>(dst + (size_t)up * Ld.pitch) = out;
    if (r <= Ld.h - 2 && down < Ld.h + 2 * kEdge) *reinterpret_cast<uint32_t*>(dst + (size_t)down * Ld.pitch) = out;
  }
}


__global__ __launch_bounds__(PR_THREADS) void pyr_resize_tile_kernel(uint8_t* __restrict__ pyr,
                                                                     unsigned long long slab, LevelGeom Ls,
                                                                     LevelGeom Ld, const uint32_t* __restrict__ xtp,
                                                                     const YTab* __restrict__ yt,
                                                                     const int2* __restrict__ xrange,
                                                                     const int2* __restrict__ yrange, int PR_SPITCH,
                                                                     int tile_rows) {
  // staged source rows: the launch sizes this to the level's widest / tallest tile
  extern __shared__ __attribute__((aligned(16))) uint8_t s_src[];
  pr_tile<false>(pyr, slab, Ls, Ld, xtp, yt, xrange, yrange, PR_SPITCH, tile_rows, blockIdx.z, blockIdx.y, blockIdx.x, s_src);
}

// Levels 0 and 1 in one pass over the input image (one-channel input, batches): a workgroup is a
// tile of level 1 as above, but takes the level-0 rows it needs from the IMAGE, and on the way
// writes them into level 0's slab - each level-0 row by the first tile that needs it, each 16-byte
// piece of it by the tile column whose staged range begins there - together with the reflected
// frame: the columns left and right of a row with the row, the 19 rows above and below as copies
// of rows 1 .. 19 and h - 20 .. h - 2.  Level 0 is then never read back for the resize, and the
// pyramid is one launch shorter.  The host checks that consecutive tiles' source rows leave no gap
// (every image row is then written by exactly one tile row).
// own_rows[gy].x = first level-0 row tile row gy writes (own_rows[ngy].x = h).
__global__ __launch_bounds__(PR_THREADS) void pyr_level01_kernel(
    const uint8_t* __restrict__ img, size_t stride, size_t frame_stride, uint8_t* __restrict__ pyr,
    unsigned long long slab, LevelGeom L0, LevelGeom L1, const uint32_t* __restrict__ xtp, const YTab* __restrict__ yt,
    const int2* __restrict__ xrange, const int2* __restrict__ yrange, const int2* __restrict__ own_rows, int PR_SPITCH,
    int tile_rows, int* __restrict__ cand_cnt, int nlevels) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_src[];
  const int f = blockIdx.x, gy = blockIdx.y, gx = blockIdx.z, ngx = gridDim.z, tid = threadIdx.x;
  // the frame's candidate counters start at zero (FAST, later in the stream, adds to them)
  if (gy == 0 && gx == 0 && tid < nlevels) cand_cnt[f * nlevels + tid] = 0;
  const int2 X = xrange[gx], Y = yrange[gy];
  const int npieces = L0.pitch >> 4;
  const int st0 = ((X.x + kXOff) & ~15) >> 4;             // staged pieces [st0, st1) of a slab row
  const int st1 = st0 + ((X.y + kXOff - 16 * st0 + 16) >> 4);
  const int ow0 = gx == 0 ? 0 : st0;                      // pieces this tile column writes to level 0
  const int ow1 = gx + 1 < ngx ? ((xrange[gx + 1].x + kXOff) & ~15) >> 4 : npieces;
  const int c_lo = min(ow0, st0), c_n = max(ow1, st1) - c_lo;
  const int r_own1 = own_rows[gy + 1].x;                  // rows [Y.x, r_own1) are written, [Y.x, Y.y] staged
  const int r_n = max(r_own1, Y.y + 1) - Y.x;
  const uint8_t* const fimg = img + (size_t)f * frame_stride;
  uint8_t* const dst0 = pyr + (size_t)f * slab + L0.off;
  // where a finished 16-byte piece goes: the staged rows in LDS, level 0's slab, or both
  auto put = [&](int r, int c, const uint4 out) {
    if (c >= st0 && c < st1 && r <= Y.y) *reinterpret_cast<uint4*>(&s_src[(r - Y.x) * PR_SPITCH + 16 * (c - st0)]) = out;
    if (c >= ow0 && c < ow1 && r < r_own1) {
      uint8_t* d = dst0 + 16 * c;
      *reinterpret_cast<uint4*>(d + (size_t)(r + kEdge) * L0.pitch) = out;
      const int up = kEdge - r, down = kEdge + 2 * (L0.h - 1) - r;  // the row's mirror images in the frame
      if (r >= 1 && up >= 0) *reinterpret_cast<uint4*>(d + (size_t)up * L0.pitch) = out;
      if (r <= L0.h - 2 && down < L0.h + 2 * kEdge) *reinterpret_cast<uint4*>(d + (size_t)down * L0.pitch) = out;
    }
  };
  // Interior pieces (16 c - kXOff >= 4 and 16 c - kXOff + 20 <= w: the five aligned dwords around
  // the piece lie inside the image row) are numbered apart from the others, so that a wavefront
  // runs one of the two paths, not both.
  const int cf0 = max(c_lo, (kXOff + 4 + 15) / 16), cf1 = min(c_lo + c_n, (L0.w + kXOff - 20) / 16 + 1);
  // (a single such piece goes the other way: the division below needs nf >= 2)
  const int nf = cf1 - cf0 >= 2 ? cf1 - cf0 : 0;
  if (nf > 0) {
    const uint32_t nf_m = (uint32_t)((1ull << 32) / (unsigned)nf + 1);  // t / nf by multiplication (t * nf < 2^32)
    const int total = r_n * nf;
    for (int t0 = tid; t0 < total; t0 += 2 * PR_THREADS) {
      // two pieces per step, their loads issued together: four dwords in one load (any dword
      // alignment) and the one after them, shifted into place
      uint32_t x[2][5];
      int rr[2], cc[2];
      bool live[2];
      uint32_t sh[2];
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int t = t0 + u * PR_THREADS;
        live[u] = t < total;
        const int tt = live[u] ? t : 0;
        const int ri = (int)__umulhi((uint32_t)tt, nf_m);
        rr[u] = Y.x + ri;
        cc[u] = cf0 + tt - ri * nf;
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4), aligned(4)));
        const uintptr_t a = reinterpret_cast<uintptr_t>(fimg + (size_t)rr[u] * stride + (16 * cc[u] - kXOff));
        const uint32_t* q = reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3);
        sh[u] = (uint32_t)(a & 3);
        const u32x4 V = *reinterpret_cast<const u32x4*>(q);
        x[u][0] = V.x; x[u][1] = V.y; x[u][2] = V.z; x[u][3] = V.w;
        x[u][4] = q[4];
      }
#pragma unroll
      for (int u = 0; u < 2; u++) {
        if (!live[u]) continue;
        uint4 out;
        out.x = __builtin_amdgcn_alignbyte(x[u][1], x[u][0], sh[u]);
        out.y = __builtin_amdgcn_alignbyte(x[u][2], x[u][1], sh[u]);
        out.z = __builtin_amdgcn_alignbyte(x[u][3], x[u][2], sh[u]);
        out.w = __builtin_amdgcn_alignbyte(x[u][4], x[u][3], sh[u]);
        put(rr[u], cc[u], out);
      }
    }
  }
  {
    // the pieces on the frame or across a row's end: byte by byte, mirrored at the image's edges
    const int nleft = nf > 0 ? cf0 - c_lo : c_n, nslow = c_n - nf;
    const int total = r_n * nslow;
    for (int t = tid; t < total; t += PR_THREADS) {
      const int ri = t / nslow, j = t - ri * nslow;
      const int r = Y.x + ri, c = j < nleft ? c_lo + j : cf1 + (j - nleft);
      const uint8_t* S = fimg + (size_t)r * stride;
      const int px0 = 16 * c - kXOff;
      uint32_t o[4] = {0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < 16; k++) {
        const int px = px0 + k;
        if (px >= -kEdge && px < L0.w + kEdge) o[k >> 2] |= (uint32_t)S[refl101(px, L0.w)] << (8 * (k & 3));
      }
      put(r, c, make_uint4(o[0], o[1], o[2], o[3]));
    }
  }
  // (pr_tile's barrier comes before its first read of s_src)
  pr_tile<true>(pyr, slab, L0, L1, xtp, yt, xrange, yrange, PR_SPITCH, tile_rows, gx, gy, f, s_src);
}

// Two levels in one launch, for launches of a few frames, where every level is a launch of a
// few microseconds that has to wait for the one before: workgroups with blockIdx.y < ngy1 are
// tiles of level l (as above); the others are tiles of level l + 1, computed from level l - 1 as
// well: the rows of level l that a tile needs are made in LDS first (all interior columns; the
// same arithmetic, so the same bytes as the level-l tiles write), then the tile from them.
// Level l + 1 must fit one column tile.
__global__ __launch_bounds__(PR_THREADS) void pyr_resize_dual_kernel(
    uint8_t* __restrict__ pyr, unsigned long long slab, LevelGeom L0, LevelGeom L1, LevelGeom L2,
    const uint32_t* __restrict__ xtp1, const uint32_t* __restrict__ xtp2, const YTab* __restrict__ yt,
    const int2* __restrict__ xrange1, const int2* __restrict__ yrange1, const int2* __restrict__ yrange2, int spitch1,
    int tile_rows, int ngy1, int spitch0, int pitchB, int offB) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_dual[];
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  if ((int)blockIdx.y < ngy1) {  // uniform
    pr_tile<false>(pyr, slab, L0, L1, xtp1, yt + L1.ytab_off, xrange1, yrange1, spitch1, tile_rows, blockIdx.z, blockIdx.y, f, s_dual);
    return;
  }
  if (blockIdx.z != 0) return;
  const int gy = blockIdx.y - ngy1;
  const int2 Y2 = yrange2[gy];  // rows of level l this tile of level l + 1 reads
  const YTab* yt1 = yt + L1.ytab_off;
  const int nrows1 = Y2.y - Y2.x + 1;
  // their vertical taps, one row per lane
  unsigned long long y1l = 0;
  if (lane < nrows1) y1l = *reinterpret_cast<const unsigned long long*>(&yt1[Y2.x + lane]);
  const int tmin = (int)((uint32_t)__builtin_amdgcn_readlane((uint32_t)y1l, 0) & 0xFFFFu);
  const int tmax = (int)((uint32_t)__builtin_amdgcn_readlane((uint32_t)y1l, nrows1 - 1) >> 16);
  // stage rows tmin .. tmax of level l - 1, interior columns (and the one after them)
  uint8_t* const s_src = s_dual;
  uint8_t* const s_b = s_dual + offB;
  {
    const int n16 = (L0.w + 1 + 15) >> 4, nrows0 = tmax - tmin + 1;
    const uint8_t* src = pyr + (size_t)f * slab + L0.off + (size_t)(tmin + kEdge) * L0.pitch + kXOff;
    for (int r = tid >> 6; r < nrows0; r += PR_THREADS / 64)
      for (int c = lane; c < n16; c += 64)
        *reinterpret_cast<uint4*>(&s_src[r * spitch0 + 16 * c]) =
            *reinterpret_cast<const uint4*>(src + (size_t)r * L0.pitch + 16 * c);
  }
  __syncthreads();
  auto mulhi24 = [](uint32_t a, uint32_t b) { return (uint32_t)(((unsigned long long)(a & 0xFFFFFFu) * (b & 0xFFFFFFu)) >> 32); };
  // one pass of the resize: rows [ra, rb) (the taps of row ra + i in lane i of ytl), source rows in S
  // (row sbase at offset 0), four columns whose table entries are e; returns through `emit(row, dword)`
  auto pass = [&](const uint8_t* S, int sp, int sbase, const uint4 e, unsigned long long ytl, int nrows, auto emit) {
    const uint32_t ev[4] = {e.x, e.y, e.z, e.w};
    int col[4], w0[4], w1[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int a1 = (int)(ev[k] >> 16);
      const bool valid = a1 != 0xFFFF;
      col[k] = valid ? (int)(ev[k] & 0xFFFF) : 0;
      w1[k] = valid ? a1 : 0;
      w0[k] = valid ? 2048 - a1 : 0;
    }
    uint32_t hprev[4] = {0, 0, 0, 0};
    int prev_row = -1;
    for (int i = 0; i < nrows; i++) {
      const uint32_t ylo = __builtin_amdgcn_readlane((uint32_t)ytl, i);
      const uint32_t yhi = __builtin_amdgcn_readlane((uint32_t)(ytl >> 32), i);
      const int s0 = (int)(ylo & 0xFFFFu), s1 = (int)(ylo >> 16);
      uint32_t h0[4], h1[4];
      if (s0 == prev_row) {
#pragma unroll
        for (int k = 0; k < 4; k++) h0[k] = hprev[k];
      } else {
        const uint8_t* R = S + (s0 - sbase) * sp;
#pragma unroll
        for (int k = 0; k < 4; k++) h0[k] = (uint32_t)(R[col[k]] * w0[k] + R[col[k] + 1] * w1[k]) & 0xFFFFF0u;
      }
      if (s1 == s0) {
#pragma unroll
        for (int k = 0; k < 4; k++) h1[k] = h0[k];
      } else {
        const uint8_t* R = S + (s1 - sbase) * sp;
#pragma unroll
        for (int k = 0; k < 4; k++) h1[k] = (uint32_t)(R[col[k]] * w0[k] + R[col[k] + 1] * w1[k]) & 0xFFFFF0u;
      }
      const uint32_t b0 = (uint32_t)(yhi & 0xFFFFu) << 12, b1 = (uint32_t)(yhi >> 16) << 12;
      uint32_t out = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        out |= ((mulhi24(b0, h0[k]) + mulhi24(b1, h1[k]) + 2u) >> 2) << (8 * k);
        hprev[k] = h1[k];
      }
      prev_row = s1;
      emit(i, out);
    }
  };
  // level l: rows Y2.x .. Y2.y, interior columns, into s_b
  for (int cg = tid; 4 * cg < L1.w; cg += PR_THREADS)
    pass(s_src, spitch0, tmin, *reinterpret_cast<const uint4*>(&xtp1[kXOff + 4 * cg]), y1l, nrows1,
         [&](int i, uint32_t v) { *reinterpret_cast<uint32_t*>(&s_b[i * pitchB + 4 * cg]) = v; });
  __syncthreads();
  // level l + 1: the tile's rows from them
  const int c = 4 * tid;
  if (c >= L2.pitch) return;
  const int rbeg = gy * tile_rows, rend = min(rbeg + tile_rows, L2.h);
  unsigned long long y2l = 0;
  if (lane < rend - rbeg) y2l = *reinterpret_cast<const unsigned long long*>(&yt[L2.ytab_off + rbeg + lane]);
  uint8_t* dst = pyr + (size_t)f * slab + L2.off + c;
  pass(s_b, pitchB, Y2.x, *reinterpret_cast<const uint4*>(&xtp2[c]), y2l, rend - rbeg, [&](int i, uint32_t v) {
    const int r = rbeg + i;
    *reinterpret_cast<uint32_t*>(dst + (size_t)(r + kEdge) * L2.pitch) = v;
    const int up = kEdge - r, down = kEdge + 2 * (L2.h - 1) - r;
    if (r >= 1 && up >= 0) *reinterpret_cast<uint32_t*>(dst + (size_t)up * L2.pitch) = v;
    if (r <= L2.h - 2 && down < L2.h + 2 * kEdge) *reinterpret_cast<uint32_t*>(dst + (size_t)down * L2.pitch) = v;
  });
}

// ---------------------------------------------------------------------------
// FAST-9/16 per cell with the iniThFAST -> minThFAST fallback and cell-local
// 3x3 non-maximum suppression (reference src/ORBextractor.cc:1077-1146 calling
// cv::FAST).  A workgroup stages a run of cells of one cell row in LDS.
//
// For a pixel with centre c and ring p[0..15], let d[k] = c - p[k],
//   A = max_k min(d[k..k+8]),  B = max_k min(-d[k..k+8]),  V = max(A, B).
// The pixel is a corner at threshold t iff V > t, and OpenCV's corner score is
// then V - 1; so one V map serves both thresholds.
// ---------------------------------------------------------------------------
constexpr int F_THREADS = 256;
constexpr int F_USEW = 240;   // widest run of cell interiors per workgroup
constexpr int F_PITCH = 256;  // LDS row pitch: a wavefront's 64 lanes x 4 columns are a staged row (the dwords left and right of
                              // a row that the quick test reads belong to the rows above and below: never a pixel of the run)
constexpr int F_MAXH = 66;    // tallest cell view (hCell <= 59, + 6)
constexpr int F_MAXCELLS = 16;
constexpr int F_QCAP = 64 + 256;  // a wave's queue: carried-over remainder + one row of survivors (re-queued entries while there is room)
constexpr int F_CCAP = 512;       // corners kept per wavefront before falling back to a dense sweep
constexpr int F_OVERFLOW = 1 << 20;  // a corner count that says: the list overflowed
static_assert(3 + F_USEW + 6 <= 256, "a staged row must fit 64 lanes x 4 columns");

typedef short fast_s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ fast_s2 as_s2(uint32_t v) { return __builtin_bit_cast(fast_s2, v); }
__device__ __forceinline__ uint32_t as_u32(fast_s2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ fast_s2 pk_min(fast_s2 a, fast_s2 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ fast_s2 pk_max(fast_s2 a, fast_s2 b) { return __builtin_elementwise_max(a, b); }

// One polarity of V for the pixel at p (see above): A = c - min_k max(p[k..k+8]) for a ring
// darker than the centre, B = max_k min(p[k..k+8]) - c = (255 - c) - min_k max(255 - p[k..k+8])
// for a brighter one, so B is A on the complemented bytes.  A pixel is a corner by at most
// one of the two (two arcs of 9 do not fit in 16), and the four compass points say which can be.
// Opposite ring pixels share a register, (p[k], p[k+8]) on packed 16-bit lanes, so eight
// registers hold the ring and each step of the sliding maximum is eight operations; where a
// window wraps past p[15] the neighbour register is used with its halves exchanged.
__device__ __forceinline__ fast_s2 swap2(fast_s2 a) { return __builtin_shufflevector(a, a, 1, 0); }
template <int FP>
__device__ __forceinline__ int fast_v1(const uint8_t* p /* centre in LDS */, int t, bool dark_only, bool& again) {
  // all seventeen reads relative to the ring's first byte in memory (p[9]), so that every offset
  // is a non-negative constant that fits the LDS instruction's offset field: one address, no adds
  constexpr int B = 3 * FP + 1;
  typedef __attribute__((address_space(3))) const uint8_t* lds_bytes;
  uint32_t ba = (uint32_t)(uintptr_t)(lds_bytes)p - B;
  asm("" : "+v"(ba));  // opaque: otherwise the subtraction is folded back into seven negative offsets, an add each
  const lds_bytes b = (lds_bytes)(uintptr_t)ba;
  const int c = b[B];
  const int offs[16] = {B + 3 * FP,     B + 3 * FP + 1, B + 2 * FP + 2, B + FP + 3,
                        B + 3,               B - FP + 3,     B - 2 * FP + 2, B - 3 * FP + 1,
                        B - 3 * FP,     B - 3 * FP - 1, B - 2 * FP - 2, B - FP - 3,
                        B - 3,               B + FP - 3,     B + 2 * FP - 2, B + 3 * FP - 1};
  uint32_t r[16];
#pragma unroll
  for (int k = 0; k < 16; k++) r[k] = b[offs[k]];
  // which polarity can hold: an arc of 9 contains two adjacent compass points
  const bool br = (int)min(max(r[0], r[8]), max(r[4], r[12])) > c + t;
  const bool dk = (int)max(min(r[0], r[8]), min(r[4], r[12])) < c - t;
  const bool bright = br && !dark_only;
  again = bright && dk;  // both can: the caller runs this pixel once more for the darker ring
  const uint32_t flip = bright ? 0x00FF00FFu : 0u;
  fast_s2 Q[8], m2[8], m4[8], m8[8];
#pragma unroll
  for (int k = 0; k < 8; k++) Q[k] = as_s2((r[k] | (r[k + 8] << 16)) ^ flip);
#pragma unroll
  for (int k = 0; k < 8; k++) m2[k] = pk_max(Q[k], k < 7 ? Q[k + 1] : swap2(Q[0]));
#pragma unroll
  for (int k = 0; k < 8; k++) m4[k] = pk_max(m2[k], k < 6 ? m2[k + 2] : swap2(m2[k - 6]));
#pragma unroll
  for (int k = 0; k < 8; k++) m8[k] = pk_max(m4[k], k < 4 ? m4[k + 4] : swap2(m4[k - 4]));
  fast_s2 best = {32767, 32767};
#pragma unroll
  for (int k = 0; k < 8; k++) best = pk_min(best, pk_max(m8[k], swap2(Q[k])));  // arcs k and k + 8
  return (bright ? 255 - c : c) - min((int)best.x, (int)best.y);
}

// NMS decision for the pixel whose V byte is at q: strictly greater than the 8
// neighbours' scores, where neighbours in another cell (hasL / hasR false) or on the
// 3-px rim (V byte 0) count as 0.  Scores are V - 1, so "score > 0" is "V >= 2".
template <int FP>
__device__ __forceinline__ bool fast_nms_keep(const uint8_t* q, int v, bool hasL, bool hasR) {
  // all eight loads first (the rim keeps them inside the map): one LDS round trip, not eight
  const int u = q[-FP], d = q[FP];
  const int l0 = q[-1], l1 = q[-FP - 1], l2 = q[FP - 1];
  const int r0 = q[1], r1 = q[-FP + 1], r2 = q[FP + 1];
  const int mc = max(u, d);
  const int ml = hasL ? max(l0, max(l1, l2)) : 0;
  const int mr = hasR ? max(r0, max(r1, r2)) : 0;
  return v > max(mc, max(ml, mr));
}

// floor(off / FP) for off < 32768, and floor(n / d) for n, d < 256 with m = 65536 / d + 1, by
// multiplication (the identities are checked over their whole ranges in tests/test_arith_cpu.py).
template <int FP>
__device__ __forceinline__ int div_pitch(int off) {
  static_assert(FP == 256 || FP == 224, "div_pitch knows these two pitches");
  return FP == 256 ? off >> 8 : (int)(((uint32_t)off * 18725u) >> 22);
}
__device__ __forceinline__ int div_small(int n, int m) { return (n * m) >> 16; }

// the workgroup's LDS: arrays declared by the kernel (one copy, whichever pitch the body is built for)
struct FastShared {
  uint8_t* dyn;                  // staged view and V map
  uint16_t (*list)[F_QCAP];      // per wavefront: quick-test survivors ...
  uint16_t (*corner)[F_CCAP];    // ... and corners (V > t), both as the pixel's byte offset in the staged
                                 // view (bit 15 of a survivor: test the darker ring only)
  int* ncorner;
  int* cnt;
  int* vars;  // total, base, emit of the dense sweep
};

struct FastPlan { unsigned long long slab; int cand_total, nlevels; };  // what the kernel needs of the plan
template <int FP>  // LDS row pitch of the staged view and of the V map
__device__ __forceinline__ void fast_body(const uint8_t* __restrict__ pyr, const FastPlan& P, const FastWork& item,
                                          uint32_t* __restrict__ cand_all, int* __restrict__ cand_cnt, int iniTh,
                                          int minTh, int lds_rows, const FastShared S) {
  // staged view and its V map: lds_rows rows each (the tallest cell view of the class)
  uint8_t* const s_dyn = S.dyn;
  uint8_t* const s_img = s_dyn;
  // the V map is indexed like the view, but only its rows 2 .. th - 3 exist (corners lie in rows
  // 3 .. th - 4, the suppression looks one row further): its first two rows would be the view's last two
  uint8_t* const s_v = s_dyn + (lds_rows - 2) * FP;
  uint16_t (*const s_list)[F_QCAP] = S.list;
  uint16_t (*const s_corner)[F_CCAP] = S.corner;
  int* const s_ncorner = S.ncorner;
  int* const s_cnt = S.cnt;
  int &s_total = S.vars[0], &s_base = S.vars[1], &s_emit = S.vars[2];

  const int f = blockIdx.x;
  // the wave index as a scalar: loops over "rows of this wave" are then uniform for the compiler,
  // and the queue length and list addresses stay in scalar registers
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // (the view's geometry comes worked out with the item: see FastWork)
  const int wCell = item.wCell;
  const int wCellM = item.wCellM;  // div_small(n, wCellM) = n / wCell
  const int iniY = item.iniY, xb = item.xb;
  const int tw = item.tw, th = item.th;
  const int wi = tw - 6, hi = th - 6;  // interior (pixels that can be corners); wi <= F_USEW
  if (wi <= 0 || hi <= 0) return;      // uniform: a view shorter than 7 rows yields nothing

  // stage the view, dword-aligned on the slab: a wave per row.  LDS column 0 is the slab column
  // of the view's first dword; interior pixel x sits in LDS column x + xoff.
  const int xoff = item.xoff;
  const int ndw = item.ndw;
  struct { int pitch, cand_off, cand_cap; } L = {item.pitch, item.cand_off, item.cand_cap};
  const uint8_t* base = pyr + (size_t)f * P.slab + (uint32_t)item.base_off;
  {
    // all of this wave's rows are requested before the first one is waited for (one memory
    // latency per workgroup, not one per four rows), and the V map is cleared in between
    constexpr int W = F_THREADS / 64, NR = (F_MAXH + W - 1) / W;
    const int cc = min(lane, ndw - 1);
    const uint8_t* src = base + (size_t)wave * L.pitch + 4 * cc;
    uint32_t v[NR];
#pragma unroll
    for (int k = 0; k < NR; k++) {
      v[k] = 0;
      if (wave + k * W < th) v[k] = *reinterpret_cast<const uint32_t*>(src + (size_t)(k * W) * L.pitch);  // uniform
    }
#pragma unroll 1
    for (int r = wave + 2; r < th - 2; r += W) {
      if (4 * lane < FP) *reinterpret_cast<uint32_t*>(&s_v[r * FP + 4 * lane]) = 0;
    }
#pragma unroll
    for (int k = 0; k < NR; k++) {
      const int r = wave + k * W;
      if (r < th) {
        if (4 * lane < FP) *reinterpret_cast<uint32_t*>(&s_img[r * FP + 4 * lane]) = v[k];
      }
    }
  }
  if (tid < F_MAXCELLS) s_cnt[tid] = 0;
  __syncthreads();

  // number of set bits of a wave mask below this lane (v_mbcnt_lo / v_mbcnt_hi)
  auto below = [](unsigned long long m) {
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
  };
  uint32_t* out = cand_all + (size_t)f * P.cand_total + L.cand_off;
  int* const out_cnt = &cand_cnt[f * P.nlevels + item.level];
  // level coordinates (relative to the border origin) and record of the pixel at byte offset off
  auto record = [&](int off, int ey, int v) {
    const int ex = off - ey * FP - xoff;
    const int xr = xb + 3 + ex - kBorder, yr = iniY + ey - kBorder;  // ey counts staged rows: interior row + 3
    return (uint32_t)xr | ((uint32_t)yr << kCoordBits) | ((uint32_t)(v - 1) << 24);
  };
  unsigned cellmask = (1u << item.ncells) - 1;  // cells to (re)compute in this pass
  // LDS byte address of this wave's queue (a scalar)
  const uint32_t qbase = __builtin_amdgcn_readfirstlane(
      (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)&s_list[wave][0]);
  // the same for its corner list and for the V map
  const uint32_t cbase = __builtin_amdgcn_readfirstlane(
      (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)&s_corner[wave][0]);
  const uint32_t vbase = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)s_v);
  for (int pass = 0; pass < 2; pass++) {
    const int t = pass == 0 ? iniTh : minTh;
    int qn = 0;             // wave-uniform: entries waiting in this wave's queue
    int ncorner = 0;        // wave-uniform: entries in this wave's corner list, or F_OVERFLOW: list too small, sweep densely

    // full test of queue entries [head, head + n) (n <= 64), corners appended
    auto drain = [&](int head, int n) {
      const int k = head + lane;
      bool again = false;
      int e = 0, off = 0, v = 0;  // v = 0: no corner (idle lanes)
      if (lane < n) {
        e = s_list[wave][k];
        off = e & 0x7FFF;
        v = fast_v1<FP>(&s_img[off], t, (e & 0x8000) != 0, again);
      }
      // a pixel that either polarity could make a corner (rare) goes to the queue's tail once
      // more, marked (bit 15) for the darker ring
      const unsigned long long ma = __builtin_amdgcn_ballot_w64(again);
      if (ma) {
        if (qn + __popcll(ma) <= F_QCAP) {
          if (again) s_list[wave][qn + below(ma)] = (uint16_t)(e | 0x8000);
          qn += __popcll(ma);
        } else if (again) {  // no room left in the queue: the second polarity right here
          bool unused;
          v = max(v, fast_v1<FP>(&s_img[off], t, true, unused));
        }
      }
      ncorner = __builtin_amdgcn_readfirstlane(ncorner);  // (the compiler does not see that it is the same in all lanes)
      if (ncorner <= F_CCAP - 64) {
        // corners (V > t): V into the map (which starts at 0: staging, or the clearing before the
        // second pass) and the pixel's offset to the wave's corner list, both under exec = the
        // corner mask and without a branch, as in the quick test's compaction
        uint32_t caddr = cbase + 2u * (uint32_t)ncorner, tmp, tmp2;
        unsigned long long save;
        int cnt;
        asm volatile(
            "s_mov_b64 %[save], exec\n\t"
            "v_cmp_lt_i32 vcc, %[t], %[v]\n\t"
            "s_mov_b64 exec, vcc\n\t"
            "v_add_u32 %[tmp2], %[vbase], %[off]\n\t"
            "v_mbcnt_lo_u32_b32 %[tmp], vcc_lo, 0\n\t"
            "v_mbcnt_hi_u32_b32 %[tmp], vcc_hi, %[tmp]\n\t"
            "ds_write_b8 %[tmp2], %[v]\n\t"
            "v_lshl_add_u32 %[tmp], %[tmp], 1, %[ca]\n\t"
            "ds_write_b16 %[tmp], %[off]\n\t"
            "s_mov_b64 exec, %[save]\n\t"
            "s_bcnt1_i32_b64 %[cnt], vcc\n\t"
            "s_lshl1_add_u32 %[ca], %[cnt], %[ca]\n\t"
            : [ca] "+s"(caddr), [tmp] "=&v"(tmp), [tmp2] "=&v"(tmp2), [save] "=&s"(save), [cnt] "=&s"(cnt)
            : [t] "v"(t), [v] "v"(v), [off] "v"(off), [vbase] "s"(vbase)
            : "vcc", "scc", "memory");
        ncorner = (int)((caddr - cbase) >> 1);
      } else {  // the list is (nearly) full: the same with the room checked
        const bool corner = v > t;
        if (corner) s_v[off] = (uint8_t)v;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(corner);
        const int nc = __popcll(m);
        if (ncorner + nc > F_CCAP) ncorner = F_OVERFLOW;
        else if (corner) s_corner[wave][ncorner + below(m)] = (uint16_t)off;
        if (ncorner != F_OVERFLOW) ncorner += nc;
      }
    };

    // ---- V map for the masked cells, one run of adjacent masked cells at a time (the first
    // pass has a single run: the whole strip).  A lane tests four adjacent pixels; a run that
    // needs no more than 16 or 32 lanes per row puts 4 or 2 rows side by side in the wavefront,
    // so the second pass costs what its cells cost, not what the strip costs.  Survivors of the
    // quick reject go to the wave's queue, which is drained 64 entries at a time (full test on
    // all lanes); the remainder is carried on.  No workgroup barrier inside: the queues are
    // private to the wave and LDS operations of one wave complete in order.
    for (unsigned runs = cellmask; runs;) {  // uniform
      const int c0 = __builtin_ctz(runs);
      const int len = __builtin_ctz(~(runs >> c0));
      runs &= ~(((1u << len) - 1u) << c0);
      const int x0 = c0 * wCell, x1 = min((c0 + len) * wCell, wi);  // interior columns of the run
      const int lc0 = (x0 + xoff) & ~3;                             // its first LDS dword
      const int nl = ((x1 + xoff) - lc0 + 3) >> 2;                  // lanes a row needs (<= 64)
      const int lsh = nl <= 16 ? 4 : nl <= 32 ? 5 : 6;
      const int rps = 64 >> lsh;  // rows per wavefront step
      const int li = lane & ((1 << lsh) - 1), sub = lane >> lsh;
      const int col4 = lc0 + 4 * li;
      // per lane and pixel the threshold the quick test compares with: t, or 32767 for a pixel
      // outside the run, which then never passes (two pixels per register, like the test values)
      uint32_t tv[2];
#pragma unroll
      for (int h2 = 0; h2 < 2; h2++) {
        uint32_t w = 0;
#pragma unroll
        for (int jj = 0; jj < 2; jj++) {
          const int x = col4 + 2 * h2 + jj - xoff;
          const bool mine = li < nl && x >= x0 && x < x1;
          w |= (uint32_t)(mine ? t : 32767) << (16 * jj);
        }
        tv[h2] = w;
      }
      // byte offset of the lane's first pixel; a lane whose row lies below the view (the last step
      // of a run with several rows per step) reads the last row and never passes
      int rowoff_next = (wave * rps + sub + 3) * FP + col4;
      const int rowoff_last = (hi - 1 + 3) * FP + col4;
      for (int yb = wave * rps; yb < hi; yb += (F_THREADS / 64) * rps) {
        if (yb + rps > hi) {  // uniform, and only in the last step of a run with several rows per step
          asm volatile("");   // (keeps this a branch: as selects it would cost every step four instructions)
          if (rowoff_next > rowoff_last) { rowoff_next = rowoff_last; tv[0] = tv[1] = 0x7FFF7FFFu; }
        }
        const uint32_t tv01 = tv[0], tv23 = tv[1];
        const int rowoff = rowoff_next;
        rowoff_next += (F_THREADS / 64) * rps * FP;
        // the five dwords around the lane's pixels, addressed from the topmost one (offsets that
        // fit the LDS instructions' offset fields; opaque, or the compiler re-bases them with adds)
        typedef __attribute__((address_space(3))) const uint32_t* lds_words;
        uint32_t ra = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint8_t*)s_img + rowoff - 3 * FP;
        asm("" : "+v"(ra));
        const lds_words rw = (lds_words)(uintptr_t)ra;
        const uint32_t U = rw[0];
        const uint32_t Lw = rw[3 * FP / 4 - 1];
        const uint32_t C = rw[3 * FP / 4];
        const uint32_t R = rw[3 * FP / 4 + 1];
        const uint32_t D = rw[6 * FP / 4];
        // per pixel: some ring pixel pair at the compass points is brighter / darker than the centre
        // by more than t.  Nine contiguous ring pixels always contain two adjacent compass points:
        // (b0&b4)|(b4&b8)|(b8&b12)|(b12&b0) == (b0|b8)&(b4|b12), for "brighter" as for "darker", i.e.
        //   max( min(max(p0,p8), max(p4,p12)) - c,  c - max(min(p0,p8), min(p4,p12)) ) > t.
        // Two pixels at a time on packed 16-bit lanes: v_perm spreads the bytes of the five dwords
        // (the pixels 3 to the left / right come straight out of the neighbouring dwords).
        uint32_t dq[2];
#pragma unroll
        for (int h2 = 0; h2 < 2; h2++) {
          const uint32_t s01 = h2 ? 0x0c030c02u : 0x0c010c00u;
          const fast_s2 c2 = as_s2(__builtin_amdgcn_perm(0u, C, s01));
          const fast_s2 p0 = as_s2(__builtin_amdgcn_perm(0u, D, s01)), p8 = as_s2(__builtin_amdgcn_perm(0u, U, s01));
          const fast_s2 p12 = as_s2(__builtin_amdgcn_perm(C, Lw, h2 ? 0x0c040c03u : 0x0c020c01u));  // pixels x-3
          const fast_s2 p4 = as_s2(__builtin_amdgcn_perm(R, C, h2 ? 0x0c060c05u : 0x0c040c03u));    // pixels x+3
          const fast_s2 hiq = pk_min(pk_max(p0, p8), pk_max(p4, p12));
          const fast_s2 loq = pk_max(pk_min(p0, p8), pk_min(p4, p12));
          dq[h2] = as_u32(pk_max(hiq - c2, c2 - loq));
        }
        // Compaction of the (up to) four passing pixels per lane into the wave's queue.  Written out
        // by hand: per pixel one compare into vcc, the lane's rank among the passing lanes
        // (v_mbcnt), one LDS write under exec = vcc and the queue's end moved on by the mask's
        // population count - no branch (an LDS write with exec = 0 does nothing) and four scalar
        // instructions, where the compiler's version of "if (pass) queue[n + rank] = x" takes a
        // branch and eleven.
        {
          uint32_t qaddr = qbase + 2u * (uint32_t)qn, tmp, tmp2;
          unsigned long long save;
          int cnt;
#define ORB_FAST_PUSH(TV, DQ, SEL, VAL_INSTR, VAL)                                  \
  "v_cmp_lt_i16_sdwa vcc, %[" TV "], %[" DQ "] src0_sel:" SEL " src1_sel:" SEL "\n\t" \
  "s_mov_b64 exec, vcc\n\t"                                                       \
  VAL_INSTR                                                                        \
  "v_mbcnt_lo_u32_b32 %[tmp], vcc_lo, 0\n\t"                                      \
  "v_mbcnt_hi_u32_b32 %[tmp], vcc_hi, %[tmp]\n\t"                                 \
  "v_lshl_add_u32 %[tmp], %[tmp], 1, %[qa]\n\t"                                   \
  "ds_write_b16 %[tmp], %[" VAL "]\n\t"                                           \
  "s_mov_b64 exec, %[save]\n\t"                                                   \
  "s_bcnt1_i32_b64 %[cnt], vcc\n\t"                                               \
  "s_lshl1_add_u32 %[qa], %[cnt], %[qa]\n\t"
          asm volatile("s_mov_b64 %[save], exec\n\t"  //
                       ORB_FAST_PUSH("tv01", "d01", "WORD_0", "", "val")  //
                       ORB_FAST_PUSH("tv01", "d01", "WORD_1", "v_add_u32 %[tmp2], 1, %[val]\n\t", "tmp2")  //
                       ORB_FAST_PUSH("tv23", "d23", "WORD_0", "v_add_u32 %[tmp2], 2, %[val]\n\t", "tmp2")  //
                       ORB_FAST_PUSH("tv23", "d23", "WORD_1", "v_add_u32 %[tmp2], 3, %[val]\n\t", "tmp2")
                       : [qa] "+s"(qaddr), [tmp] "=&v"(tmp), [tmp2] "=&v"(tmp2), [save] "=&s"(save), [cnt] "=&s"(cnt)
                       : [tv01] "v"(tv01), [tv23] "v"(tv23), [d01] "v"(dq[0]), [d23] "v"(dq[1]), [val] "v"(rowoff)
                       : "vcc", "scc", "memory");
#undef ORB_FAST_PUSH
          qn = (int)((qaddr - qbase) >> 1);
        }
        if (qn >= 64) {
          int head = 0;
          for (; head + 64 <= qn; head += 64) drain(head, 64);
          const int rem = qn - head;
          const int e = lane < rem ? (int)s_list[wave][head + lane] : 0;
          if (lane < rem) s_list[wave][lane] = (uint16_t)e;
          qn = rem;
        }
      }
    }
    for (int head = 0; head < qn;) {  // qn may grow by re-queued entries
      const int n = min(64, qn - head);
      drain(head, n);
      head += n;
    }
    if (lane == 0) s_ncorner[wave] = ncorner == F_OVERFLOW ? -1 : ncorner;
    __syncthreads();
    // ---- cell-local NMS over the corner lists; a corner that survives is counted for its cell
    // and written out at once (a cell that is computed again in the second pass has, by
    // definition, no survivor in the first).  The survivors are first moved to the front of the
    // wave's list (in place: an entry is read before any lane can overwrite it), then the wave
    // reserves their room with ONE atomic and writes them; a wave waits for the atomic's return
    // once, not once per 64 corners.  The order within a level's list is arbitrary, the
    // quad-tree kernel does not depend on it.
    bool any_overflow = false;
#pragma unroll
    for (int w = 0; w < F_THREADS / 64; w++) any_overflow |= s_ncorner[w] < 0;
    if (!any_overflow) {
      int nkeep = 0;                              // uniform
      for (int k0 = 0; k0 < ncorner; k0 += 64) {  // uniform
        const int k = k0 + lane;
        const int off = k < ncorner ? (int)s_corner[wave][k] : 3 * FP + 4;
        const int ey = div_pitch<FP>(off);
        const int ex = off - ey * FP - xoff;
        const int ecj = div_small(ex, wCellM), exc = ex - ecj * wCell;
        const uint8_t* q = &s_v[off];
        const int v = q[0];
        const bool keep = k < ncorner && v >= 2 && fast_nms_keep<FP>(q, v, exc > 0, exc < wCell - 1);
        if (keep) atomicAdd(&s_cnt[ecj], 1);
        const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
        if (keep) s_corner[wave][nkeep + below(m)] = (uint16_t)off;
        nkeep += __popcll(m);
      }
      if (nkeep) {
        int wbase = 0;
        if (lane == 0) wbase = atomicAdd(out_cnt, nkeep);
        wbase = __builtin_amdgcn_readfirstlane(wbase);
        for (int k0 = 0; k0 < nkeep; k0 += 64) {
          const int k = k0 + lane;
          if (k < nkeep && wbase + k < L.cand_cap) {
            const int off = s_corner[wave][k];
            out[wbase + k] = record(off, div_pitch<FP>(off), s_v[off]);
          }
        }
      }
      __syncthreads();
    } else {
      // a corner list overflowed: dense sweep of the V map, a thread per interior column;
      // count, one reservation for the workgroup, then the same sweep writes
      const int x = tid;
      const bool mycol = x < wi && ((cellmask >> div_small(x, wCellM)) & 1u);
      const int cj = div_small(min(x, wi - 1), wCellM), xc = x - cj * wCell;
      if (tid == 0) { s_total = 0; s_emit = 0; }
      __syncthreads();
      if (mycol) {
        int n = 0;
#pragma unroll 1
        for (int y = 0; y < hi; y++) {
          const uint8_t* q = &s_v[(y + 3) * FP + x + xoff];
          const int v = q[0];
          if (v >= 2 && fast_nms_keep<FP>(q, v, xc > 0, xc < wCell - 1)) n++;
        }
        if (n) { atomicAdd(&s_cnt[cj], n); atomicAdd(&s_total, n); }
      }
      __syncthreads();
      if (tid == 0) s_base = s_total ? atomicAdd(out_cnt, s_total) : 0;
      __syncthreads();
      if (mycol && s_total) {
        const int gbase = s_base;
#pragma unroll 1
        for (int y = 0; y < hi; y++) {
          const int off = (y + 3) * FP + x + xoff;
          const uint8_t* q = &s_v[off];
          const int v = q[0];
          if (v >= 2 && fast_nms_keep<FP>(q, v, xc > 0, xc < wCell - 1)) {
            const int slot = gbase + atomicAdd(&s_emit, 1);
            if (slot < L.cand_cap) out[slot] = record(off, y + 3, v);
          }
        }
      }
      __syncthreads();
    }
    if (pass == 1) break;
    unsigned need = 0;
#pragma unroll 1
    for (int c = 0; c < item.ncells; c++)
      if (s_cnt[c] == 0) need |= 1u << c;
    if (need == 0) break;  // uniform
    cellmask = need;
    // the cells that get the lower threshold start over: clear their V columns (their corners,
    // all suppressed, are dropped with the lists)
    {
      const int x = tid;
      if (x < wi && ((need >> div_small(x, wCellM)) & 1u)) {
#pragma unroll 1
        for (int y = 0; y < hi; y++) s_v[(y + 3) * FP + x + xoff] = 0;
      }
    }
    __syncthreads();
  }
}


// Views of up to wide_rows rows are staged at a pitch of 256 bytes (runs of up to F_USEW columns);
// taller ones at 224 (their runs are cut shorter by the host), which makes them fit the same
// LDS budget: one launch, and as many workgroups per CU for the tall views as for the others.
constexpr int F_PITCH_NARROW = 224;
constexpr int F_USEW_NARROW = 208;
static_assert(3 + F_USEW_NARROW + 6 + 3 <= F_PITCH_NARROW, "a narrow staged row must fit its pitch");
__global__ __launch_bounds__(F_THREADS) void fast_kernel(const uint8_t* __restrict__ pyr, FastPlan P,
                                                         const FastWork* __restrict__ items,
                                                         uint32_t* __restrict__ cand_all,
                                                         int* __restrict__ cand_cnt, int iniTh, int minTh,
                                                         int wide_rows, int narrow_rows) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
  __shared__ uint16_t s_list[F_THREADS / 64][F_QCAP];
  __shared__ uint16_t s_corner[F_THREADS / 64][F_CCAP];
  __shared__ int s_ncorner[F_THREADS / 64];
  __shared__ int s_cnt[F_MAXCELLS];
  __shared__ int s_vars[3];
  const FastShared S{s_dyn, s_list, s_corner, s_ncorner, s_cnt, s_vars};
  const FastWork item = items[blockIdx.y];
  if (item.hCell + 6 <= wide_rows) fast_body<F_PITCH>(pyr, P, item, cand_all, cand_cnt, iniTh, minTh, wide_rows, S);
  else fast_body<F_PITCH_NARROW>(pyr, P, item, cand_all, cand_cnt, iniTh, minTh, narrow_rows, S);
}

// ---------------------------------------------------------------------------
// Quad-tree keypoint distribution (reference src/ORBextractor.cc:690-1036 and
// ExtractorNode::DivideNode :604-676), one workgroup per (level, frame).
//
// The reference keeps a std::list of nodes, pushes children to the front and
// finally walks the list.  Restated without the list:
//  * every node carries a key that grows with creation time (roots: reversed index),
//    so the list order is "descending key";
//  * a full pass splits every node that holds more than one point; an "ordered"
//    round splits them in (count, key)-descending order and stops at the first
//    split that brings the node total to N -- which nodes get split follows from a
//    prefix sum of (children - 1) over that order;
//  * points never move: each carries the slot of its current node.
// Ties of the reference's sort of pair<count, node*> are resolved as if node
// addresses grew with allocation order (later-created node first).
// ---------------------------------------------------------------------------
constexpr int OT_MAXT = 1024;  // workgroup size: 512 for ordinary frames, 1024 when levels hold tens of thousands of candidates

template <int OT_MAXS>  // node slots
struct OtShared {
  short x0[OT_MAXS], x1[OT_MAXS], y0[OT_MAXS], y1[OT_MAXS];
  int count[OT_MAXS];
  int key[OT_MAXS];
  int ccount[OT_MAXS * 4];  // points per child; the output stage keeps the survivors' keys here
  union {
    short cslot[OT_MAXS * 4];          // slot of each child of a split node (steps 6-7)
    unsigned long long skey[OT_MAXS];  // sort keys of an ordered round (step 3); best point per node (output stage)
  };
  short elist[OT_MAXS];
  short order[OT_MAXS];
  short cumg[OT_MAXS];   // prefix sums over the processing order: at most 3 resp. 4 per node
  short cumc[OT_MAXS];
  short bykey[OT_MAXS];  // slots of the nodes created by the last pass, by key - first key of that pass
  unsigned char split[OT_MAXS];
  int wsum[OT_MAXT / 64];
  int vars[16];
};
enum { V_SIZE = 0, V_NSLOTS, V_NEXTSEQ, V_MODE, V_FINISH, V_NE, V_NEWEXP, V_PN, V_GROW, V_NCHILD, V_K };

constexpr int OT_RPT = 8;  // points per thread that the register-resident form can hold

// Visits the points of this thread: f(candidate record, node label&).  With REG the
// records and labels live in registers (a thread always handles the same points); otherwise
// they are read from, and labels written back to, global memory.
template <bool REG, int OT_THREADS, typename F>
__device__ __forceinline__ void ot_for_points(int n, int tid, const uint32_t* __restrict__ pts, uint32_t* __restrict__ pn,
                                              uint32_t (&rp)[OT_RPT], uint32_t (&rn)[OT_RPT], F f) {
  if constexpr (REG) {
#pragma unroll
    for (int k = 0; k < OT_RPT; k++)
      if (tid + k * OT_THREADS < n) f(rp[k], rn[k]);
  } else {
    // OT_BATCH points per step, loads first, so that their latencies overlap
    constexpr int OT_BATCH = 16;
    for (int p0 = tid; p0 < n; p0 += OT_BATCH * OT_THREADS) {
      uint32_t w0[OT_BATCH], v[OT_BATCH];
#pragma unroll
      for (int u = 0; u < OT_BATCH; u++) {
        const int p = min(p0 + u * OT_THREADS, n - 1);
        w0[u] = pn[p];
        v[u] = pts[p];
      }
#pragma unroll
      for (int u = 0; u < OT_BATCH; u++) {
        const int p = p0 + u * OT_THREADS;
        if (p < n) {
          uint32_t w = w0[u];
          f(v[u], w);
          if (w != w0[u]) pn[p] = w;
        }
      }
    }
  }
}

template <int OT_MAXS, int OT_THREADS, bool REG>
__device__ __forceinline__ void octree_body(OtShared<OT_MAXS>& S, const uint32_t* __restrict__ pts,
                                            uint32_t* __restrict__ pn, uint32_t* __restrict__ sel,
                                            int* __restrict__ sel_cnt, const Plan& P, const int n, const int l,
                                            const int f, const int tid) {
  const LevelGeom& L = P.lv[l];
  uint32_t rp[OT_RPT], rn[OT_RPT];
  if constexpr (REG) {
#pragma unroll
    for (int k = 0; k < OT_RPT; k++) {
      const int p = tid + k * OT_THREADS;
      rp[k] = p < n ? pts[p] : 0u;
      rn[k] = 0u;
    }
  }
  const int N = L.quota;
  const int nIni = L.nIni;
  const float hX = L.hX;

  for (int s = tid; s < OT_MAXS; s += OT_THREADS) S.count[s] = 0;
  if (tid < nIni) {
    S.x0[tid] = (short)(int)__fmul_rn(hX, (float)tid);
    S.x1[tid] = (short)(int)__fmul_rn(hX, (float)(tid + 1));
    S.y0[tid] = 0;
    S.y1[tid] = (short)(L.maxBY - kBorder);
    S.key[tid] = nIni - 1 - tid;
    S.bykey[nIni - 1 - tid] = (short)tid;
  }
  __syncthreads();
  if constexpr (!REG)
    for (int p = tid; p < n; p += OT_THREADS) pn[p] = 0xFFFFFFFFu;  // so that the first label is always stored
  ot_for_points<REG, OT_THREADS>(n, tid, pts, pn, rp, rn, [&](uint32_t v, uint32_t& node) {
    const int x = v & kCoordMask;
    int r = (int)__fdiv_rn((float)x, hX);
    r = min(max(r, 0), nIni - 1);
    node = (uint32_t)r;
    atomicAdd(&S.count[r], 1);
  });
  __syncthreads();
  if (tid == 0) {
    int sz = 0;
    for (int r = 0; r < nIni; r++) sz += S.count[r] > 0;
    S.vars[V_SIZE] = sz;
    S.vars[V_NSLOTS] = nIni;
    S.vars[V_NEXTSEQ] = nIni;
    S.vars[V_MODE] = 0;
    S.vars[V_FINISH] = 0;
    S.vars[V_K] = nIni;
  }
  __syncthreads();

  while (true) {
    const int finish = S.vars[V_FINISH];
    const int nslots = S.vars[V_NSLOTS], size = S.vars[V_SIZE], mode = S.vars[V_MODE];
    const int nextseq = S.vars[V_NEXTSEQ];
    __syncthreads();
    if (finish) break;
    if (tid == 0) { S.vars[V_NE] = 0; S.vars[V_NEWEXP] = 0; }
    __syncthreads();
    // 1. nodes that can be split.  In a full pass every such node is split, in list order
    // (descending key).  All of them were created by the previous pass, whose children sit
    // in bykey[] by ascending key, so the processing order is a compaction of that array
    // read backwards.  An ordered round collects them unordered and ranks them in step 3.
    int E = 0;
    for (int s = tid; s < nslots; s += OT_THREADS) S.split[s] = 0;
    if (mode == 0) {
      const int K = S.vars[V_K];
      const int lane = tid & 63, wave = tid >> 6;
      for (int hi = K; hi > 0; hi -= OT_THREADS) {
        const int idx = hi - 1 - tid;
        int s = 0;
        bool fl = false;
        if (idx >= 0) { s = S.bykey[idx]; fl = S.count[s] > 1; }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(fl);
        if (lane == 0) S.wsum[wave] = __popcll(m);
        __syncthreads();
        int woff = 0, all = 0;
#pragma unroll
        for (int w = 0; w < OT_THREADS / 64; w++) { const int c = S.wsum[w]; all += c; if (w < wave) woff += c; }
        if (fl) {
          S.order[E + woff + __popcll(m & ((1ull << lane) - 1))] = (short)s;
          S.ccount[4 * s] = 0; S.ccount[4 * s + 1] = 0; S.ccount[4 * s + 2] = 0; S.ccount[4 * s + 3] = 0;
        }
        E += all;
        __syncthreads();
      }
    } else {
      for (int s = tid; s < nslots; s += OT_THREADS) {
        if (S.count[s] > 1) {
          S.elist[atomicAdd(&S.vars[V_NE], 1)] = (short)s;
          S.ccount[4 * s] = 0; S.ccount[4 * s + 1] = 0; S.ccount[4 * s + 2] = 0; S.ccount[4 * s + 3] = 0;
        }
      }
      __syncthreads();
      E = S.vars[V_NE];
    }
    // 2. quadrant of every point of such a node (DivideNode's point loop)
    ot_for_points<REG, OT_THREADS>(n, tid, pts, pn, rp, rn, [&](uint32_t v, uint32_t& node) {
      const int s = node & 0xFFFF;
      if (S.count[s] > 1) {
        const int x = v & kCoordMask, y = (v >> kCoordBits) & kCoordMask;
        const int hx = (S.x1[s] - S.x0[s] + 1) >> 1, hy = (S.y1[s] - S.y0[s] + 1) >> 1;
        const int q = (x < S.x0[s] + hx ? 0 : 1) + (y < S.y0[s] + hy ? 0 : 2);
        node = (uint32_t)s | ((uint32_t)q << 16);
        atomicAdd(&S.ccount[4 * s + q], 1);
      }
    });
    if (mode == 1)
      for (int i = tid; i < E; i += OT_THREADS) {
        const int s = S.elist[i];
        S.skey[i] = ((unsigned long long)(unsigned)S.count[s] << 32) | (unsigned)S.key[s];
      }
    __syncthreads();
    // 3. children per node along the processing order; an ordered round first ranks the
    // nodes by (count, key) descending
    for (int i = tid; i < E; i += OT_THREADS) {
      int r = i, s;
      if (mode == 0) {
        s = S.order[i];
      } else {
        const unsigned long long k = S.skey[i];
        r = 0;
#pragma unroll 8
        for (int j = 0; j < E; j++) r += S.skey[j] > k;  // (unrolled: the loads of 8 keys are in flight together)
        s = S.elist[i];
        S.order[r] = (short)s;
      }
      const int c = (S.ccount[4 * s] > 0) + (S.ccount[4 * s + 1] > 0) + (S.ccount[4 * s + 2] > 0) + (S.ccount[4 * s + 3] > 0);
      S.cumg[r] = (short)(c - 1);
      S.cumc[r] = (short)c;
    }
    __syncthreads();
    // 4./5. prefix sums over the processing order; in an ordered round stop at the
    // first split that reaches N nodes
    if (tid < 64) {
      int cg = 0, cc = 0, pn_ = -1;
      for (int base = 0; base < E; base += 64) {
        const int i = base + tid;
        int g = i < E ? S.cumg[i] : 0, c = i < E ? S.cumc[i] : 0;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int g2 = __shfl_up(g, o), c2 = __shfl_up(c, o);
          if (tid >= o) { g += g2; c += c2; }
        }
        g += cg; c += cc;
        if (i < E) { S.cumg[i] = (short)g; S.cumc[i] = (short)c; }
        if (mode == 1 && pn_ < 0) {
          const unsigned long long hit = __builtin_amdgcn_ballot_w64(i < E && size + g >= N);
          if (hit) pn_ = base + __ffsll((long long)hit);  // index of first hit + 1
        }
        cg = __shfl(g, 63); cc = __shfl(c, 63);
      }
      if (tid == 0) {
        const int Pn = pn_ < 0 ? E : pn_;
        S.vars[V_PN] = Pn;
        S.vars[V_GROW] = Pn ? S.cumg[Pn - 1] : 0;
        S.vars[V_NCHILD] = Pn ? S.cumc[Pn - 1] : 0;
      }
    }
    __syncthreads();
    const int Pn = S.vars[V_PN];
    // 6. split the first Pn nodes of the order
    for (int r = tid; r < Pn; r += OT_THREADS) {
      const int s = S.order[r];
      const int X0 = S.x0[s], X1 = S.x1[s], Y0 = S.y0[s], Y1 = S.y1[s];
      const int hx = (X1 - X0 + 1) >> 1, hy = (Y1 - Y0 + 1) >> 1;
      const int gbase = r ? S.cumg[r - 1] : 0, cbase = r ? S.cumc[r - 1] : 0;
      int cc[4];
#pragma unroll
      for (int q = 0; q < 4; q++) cc[q] = S.ccount[4 * s + q];
      S.split[s] = 1;
      int k = 0, newexp = 0;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        if (cc[q] > 0) {
          const int slot = k == 0 ? s : nslots + gbase + (k - 1);
          S.cslot[4 * s + q] = (short)slot;
          S.x0[slot] = (short)((q & 1) ? X0 + hx : X0);
          S.x1[slot] = (short)((q & 1) ? X1 : X0 + hx);
          S.y0[slot] = (short)((q & 2) ? Y0 + hy : Y0);
          S.y1[slot] = (short)((q & 2) ? Y1 : Y0 + hy);
          S.count[slot] = cc[q];
          S.key[slot] = nextseq + cbase + k;
          S.bykey[cbase + k] = (short)slot;
          newexp += cc[q] > 1;
          k++;
        }
      }
      if (newexp) atomicAdd(&S.vars[V_NEWEXP], newexp);
    }
    __syncthreads();
    // 7. move the points of split nodes to their child
    ot_for_points<REG, OT_THREADS>(n, tid, pts, pn, rp, rn, [&](uint32_t, uint32_t& node) {
      const int s = node & 0xFFFF;
      if (S.split[s]) node = (uint32_t)(unsigned short)S.cslot[4 * s + (node >> 16)];
    });
    __syncthreads();
    // 8. loop control (src/ORBextractor.cc:891-897, 911, 992-993)
    if (tid == 0) {
      const int newsize = size + S.vars[V_GROW];
      S.vars[V_SIZE] = newsize;
      S.vars[V_NSLOTS] = nslots + S.vars[V_GROW];
      S.vars[V_NEXTSEQ] = nextseq + S.vars[V_NCHILD];
      S.vars[V_K] = S.vars[V_NCHILD];
      if (newsize >= N || newsize == size) S.vars[V_FINISH] = 1;
      else if (mode == 0 && newsize + 3 * S.vars[V_NEWEXP] > N) S.vars[V_MODE] = 1;
    }
    __syncthreads();
  }

  // best point of every node: highest response, first in the reference's candidate
  // order (cell row, cell column, y, x) on ties (src/ORBextractor.cc:1007-1032)
  const int nslots = S.vars[V_NSLOTS];
  for (int s = tid; s < nslots; s += OT_THREADS) S.skey[s] = 0;
  __syncthreads();
  ot_for_points<REG, OT_THREADS>(n, tid, pts, pn, rp, rn, [&](uint32_t v, uint32_t& node) {
    const int s = node & 0xFFFF;
    const int x = v & kCoordMask, y = (v >> kCoordBits) & kCoordMask;
    const unsigned long long canon =
        ((unsigned long long)(((y - 3) / L.hCell) * 256 + (x - 3) / L.wCell) << 24) | ((unsigned)y << 12) | (unsigned)x;
    const unsigned long long k = ((unsigned long long)(v >> 24) << 40) | (((1ull << 40) - 1) - canon);
    atomicMax(&S.skey[s], k);
  });
  __syncthreads();
  // surviving nodes, densely: slot list and keys; the output position of a node is its
  // rank by key, descending (the order of the reference's list)
  if (tid == 0) S.vars[V_NE] = 0;
  __syncthreads();
  for (int s = tid; s < nslots; s += OT_THREADS)
    if (S.count[s] > 0) S.elist[atomicAdd(&S.vars[V_NE], 1)] = (short)s;
  __syncthreads();
  const int M = S.vars[V_NE];
  // Keys are distinct creation numbers below V_NEXTSEQ, so the rank of a node is the number of
  // surviving keys above its own: a bitmap of the surviving keys, the population counts of its
  // words summed from the top, and one word's bits above the key.  (The node count bounds the
  // keys only loosely: a node whose points all fall into one quadrant gets a new key every
  // pass; where the bitmap would not fit, every node counts the greater keys instead.)
  const int nkeys = S.vars[V_NEXTSEQ];
  const int nwords = (nkeys + 63) >> 6;
  unsigned long long* bm = reinterpret_cast<unsigned long long*>(S.ccount);  // nwords words, then nwords ints
  int* above = S.ccount + 2 * nwords;
  const bool use_bitmap = 3 * nwords <= OT_MAXS * 4;
  if (use_bitmap) {
    for (int w = tid; w < nwords; w += OT_THREADS) bm[w] = 0;
    __syncthreads();
    for (int i = tid; i < M; i += OT_THREADS) {
      const int k = S.key[S.elist[i]];
      atomicOr(&bm[k >> 6], 1ull << (k & 63));
    }
    __syncthreads();
    for (int w = tid; w < nwords; w += OT_THREADS) {
      int c = 0;
      for (int v = w + 1; v < nwords; v++) c += __popcll(bm[v]);
      above[w] = c;
    }
    __syncthreads();
  } else {
    int* kd = S.ccount;
    for (int i = tid; i < ((M + 3) & ~3); i += OT_THREADS) kd[i] = i < M ? S.key[S.elist[i]] : -1;
    __syncthreads();
  }
  for (int i = tid; i < M; i += OT_THREADS) {
    const int s = S.elist[i];
    int r = 0;
    if (use_bitmap) {
      const int k = S.key[s];
      r = above[k >> 6] + __popcll((bm[k >> 6] >> (k & 63)) >> 1);
    } else {
      const int* kd = S.ccount;
      const int k = kd[i];
#pragma unroll 4
      for (int j = 0; j < M; j += 4) {
        const int4 q = *reinterpret_cast<const int4*>(&kd[j]);
        r += (q.x > k) + (q.y > k) + (q.z > k) + (q.w > k);
      }
    }
    const unsigned long long b = S.skey[s];
    const unsigned long long canon = ((1ull << 40) - 1) - (b & ((1ull << 40) - 1));
    const uint32_t x = (uint32_t)(canon & 0xFFF), y = (uint32_t)((canon >> 12) & 0xFFF);
    if (r < L.sel_cap) sel[r] = x | (y << kCoordBits) | ((uint32_t)(b >> 40) << 24);
  }
  if (tid == 0) sel_cnt[f * P.nlevels + l] = min(S.vars[V_SIZE], L.sel_cap);
}

template <int OT_MAXS, int OT_THREADS>
__global__ __launch_bounds__(OT_THREADS) void octree_kernel(const uint32_t* __restrict__ cand_all,
                                                            const int* __restrict__ cand_cnt,
                                                            uint32_t* __restrict__ pnode_all,
                                                            uint32_t* __restrict__ sel_all,
                                                            int* __restrict__ sel_cnt, Plan P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ot_smem[];
  OtShared<OT_MAXS>& S = *reinterpret_cast<OtShared<OT_MAXS>*>(ot_smem);
  const int l = blockIdx.y, f = blockIdx.x, tid = threadIdx.x;
  const LevelGeom& L = P.lv[l];
  const uint32_t* pts = cand_all + (size_t)f * P.cand_total + L.cand_off;
  uint32_t* pn = pnode_all + (size_t)f * P.cand_total + L.cand_off;
  uint32_t* sel = sel_all + (size_t)f * P.sel_total + L.sel_off;
  const int n = min(cand_cnt[f * P.nlevels + l], L.cand_cap);
  if (n <= OT_RPT * OT_THREADS) octree_body<OT_MAXS, OT_THREADS, true>(S, pts, pn, sel, sel_cnt, P, n, l, f, tid);
  else octree_body<OT_MAXS, OT_THREADS, false>(S, pts, pn, sel, sel_cnt, P, n, l, f, tid);
}

// ---------------------------------------------------------------------------
// cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101), 8-bit fixed-point path
// (called at reference src/ORBextractor.cc:1300-1305).  Taps are 8.8 fixed point
// and sum to 256; the row pass keeps 8.8, the column pass 16.16, result
// (acc + 2^15) >> 16.  The padded pyramid already holds the reflected border.
// ---------------------------------------------------------------------------
constexpr int B_TW = 128, B_TH = 48, B_THREADS = 256;
constexpr int B_ROWS = B_TH + 6;  // staged rows: 3 above, 3 below
constexpr int B_INP = 160;        // staged bytes per row: level x0-16 .. x0+143

typedef unsigned short blur_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t blur_dot2(uint32_t a, uint32_t b, uint32_t c) {
  return __builtin_amdgcn_udot2(__builtin_bit_cast(blur_us2, a), __builtin_bit_cast(blur_us2, b), c, false);
}

// Row pass with v_dot4_u32_u8 (taps are bytes), two image rows per thread so that the
// 8.8 results of vertically adjacent rows share a dword; column pass with
// v_dot2_u32_u16 on those row pairs.
// SAT: the taps sum to more than 256 (OpenCV 4.0 - 4.4 rounds every tap: 18 34 49 55 49 34 18),
// so a result can pass 255 and is saturated like cv::saturate_cast does.
template <bool SAT>
__global__ __launch_bounds__(B_THREADS) void blur_kernel(const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur,
                                                         Plan P, const BlurItem* __restrict__ items, int nitems,
                                                         int fchunk, int nframes, int t0, int t1, int t2, int t3) {
  __shared__ __attribute__((aligned(16))) uint8_t s_in[B_ROWS * B_INP];
  __shared__ __attribute__((aligned(16))) uint32_t s_hp[(B_ROWS / 2) * B_TW];  // h[2p][x] | h[2p+1][x] << 16
  // grid.x = the frames of one chunk, grid.y = tile, grid.z = chunk: all tiles of a chunk of
  // frames are dispatched before the next chunk starts (see launch_blur)
  const BlurItem it = items[blockIdx.y];
  const int f = blockIdx.z * fchunk + blockIdx.x, tid = threadIdx.x;
  if (f >= nframes) return;
  const LevelGeom& L = P.lv[it.level];
  const int x0 = it.tx * B_TW, y0 = it.ty * B_TH;
  const int tw = min(B_TW, L.w - x0), th = min(B_TH, L.h - y0);
  const int tw4 = (tw + 3) >> 2;
  const int n16 = (4 * tw4 + 24 + 15) >> 4;  // 16-byte pieces per staged row
  const uint8_t* base = pyr + (size_t)f * P.slab + L.off + (size_t)(y0 - 3 + kEdge) * L.pitch + (x0 - 16 + kXOff);
  for (int i = tid; i < (th + 6) * 10; i += B_THREADS) {
    const int r = i / 10, c = i - r * 10;
    if (c < n16)
      *reinterpret_cast<uint4*>(&s_in[r * B_INP + 16 * c]) =
          *reinterpret_cast<const uint4*>(base + (size_t)r * L.pitch + 16 * c);
  }
  __syncthreads();
  const uint32_t T1 = (uint32_t)t0 | ((uint32_t)t1 << 8) | ((uint32_t)t2 << 16) | ((uint32_t)t3 << 24);
  const uint32_t T2 = (uint32_t)t2 | ((uint32_t)t1 << 8) | ((uint32_t)t0 << 16);
  const int c = tid & 31, rsub = tid >> 5;
  if (c < tw4) {
    for (int pr = rsub; pr < (th + 6 + 1) / 2; pr += B_THREADS / 32) {
      uint32_t h[2][4];
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const uint32_t* w = reinterpret_cast<const uint32_t*>(&s_in[(2 * pr + k) * B_INP + 4 * c + 12]);
        const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
        // pixel j uses staged bytes j+1 .. j+7 of (w0, w1, w2)
        h[k][0] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 1), T1, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 1), T2, 0u, false), false);
        h[k][1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 2), T1, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 2), T2, 0u, false), false);
        h[k][2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 3), T1, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 3), T2, 0u, false), false);
        h[k][3] = __builtin_amdgcn_udot4(w1, T1, __builtin_amdgcn_udot4(w2, T2, 0u, false), false);
      }
      uint4 o;
      o.x = h[0][0] | (h[1][0] << 16); o.y = h[0][1] | (h[1][1] << 16);
      o.z = h[0][2] | (h[1][2] << 16); o.w = h[0][3] | (h[1][3] << 16);
      *reinterpret_cast<uint4*>(&s_hp[pr * B_TW + 4 * c]) = o;
    }
  }
  __syncthreads();
  if (c >= tw4) return;
  // tap pairs for an output row whose first tap row is even / odd in the staged numbering
  const uint32_t TPe0 = (uint32_t)t0 | ((uint32_t)t1 << 16), TPe1 = (uint32_t)t2 | ((uint32_t)t3 << 16);
  const uint32_t TPe2 = (uint32_t)t2 | ((uint32_t)t1 << 16), TPe3 = (uint32_t)t0;
  const uint32_t TPo0 = (uint32_t)t0 << 16, TPo1 = (uint32_t)t1 | ((uint32_t)t2 << 16);
  const uint32_t TPo2 = (uint32_t)t3 | ((uint32_t)t2 << 16), TPo3 = (uint32_t)t1 | ((uint32_t)t0 << 16);
  uint8_t* dst = blur + (size_t)f * P.slab + L.off + (size_t)(y0 + kEdge) * L.pitch + x0 + kXOff + 4 * c;
  for (int y = rsub; y < th; y += B_THREADS / 32) {
    const bool odd = y & 1;
    const uint32_t tp0 = odd ? TPo0 : TPe0, tp1 = odd ? TPo1 : TPe1, tp2 = odd ? TPo2 : TPe2, tp3 = odd ? TPo3 : TPe3;
    const uint4* hp = reinterpret_cast<const uint4*>(&s_hp[(y >> 1) * B_TW + 4 * c]);
    const uint4 v0 = hp[0], v1 = hp[B_TW / 4], v2 = hp[2 * (B_TW / 4)], v3 = hp[3 * (B_TW / 4)];
    uint32_t a0 = blur_dot2(v3.x, tp3, blur_dot2(v2.x, tp2, blur_dot2(v1.x, tp1, blur_dot2(v0.x, tp0, 32768u))));
    uint32_t a1 = blur_dot2(v3.y, tp3, blur_dot2(v2.y, tp2, blur_dot2(v1.y, tp1, blur_dot2(v0.y, tp0, 32768u))));
    uint32_t a2 = blur_dot2(v3.z, tp3, blur_dot2(v2.z, tp2, blur_dot2(v1.z, tp1, blur_dot2(v0.z, tp0, 32768u))));
    uint32_t a3 = blur_dot2(v3.w, tp3, blur_dot2(v2.w, tp2, blur_dot2(v1.w, tp1, blur_dot2(v0.w, tp0, 32768u))));
    if (SAT) { a0 = min(a0, 0xFFFFFFu); a1 = min(a1, 0xFFFFFFu); a2 = min(a2, 0xFFFFFFu); a3 = min(a3, 0xFFFFFFu); }
    // results are < 2^24: byte 2 of each accumulator is the output pixel
    const uint32_t o = ((a0 >> 16) & 0xFF) | ((a1 >> 8) & 0xFF00) | (a2 & 0xFF0000) | ((a3 << 8) & 0xFF000000u);
    *reinterpret_cast<uint32_t*>(dst + (size_t)y * L.pitch) = o;
  }
}

// ---------------------------------------------------------------------------
// Orientation (IC_Angle, reference src/ORBextractor.cc:96-143, with
// cv::fastAtan2) and steered BRIEF (computeOrbDescriptor, :155-206), one wavefront
// per keypoint; also the epilogue of operator() (:1164-1179, 1323-1337).
// ---------------------------------------------------------------------------
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
  const float s = (float)(180.0 / 3.14159265358979323846);
  const float p1 = 0.9997878412794807f * s, p3 = -0.3258083974640975f * s;
  const float p5 = 0.1555786518463281f * s, p7 = -0.04432655554792128f * s;
  const float ax = fabsf(x), ay = fabsf(y);
  const float eps = 2.2204460492503131e-16f;
  float a, c, c2;
  if (ax >= ay) {
    c = __fdiv_rn(ay, __fadd_rn(ax, eps));
    c2 = __fmul_rn(c, c);
    a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
  } else {
    c = __fdiv_rn(ax, __fadd_rn(ay, eps));
    c2 = __fmul_rn(c, c);
    a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
  }
  if (x < 0) a = __fsub_rn(180.f, a);
  if (y < 0) a = __fsub_rn(360.f, a);
  return a;
}

constexpr int D_THREADS = 64;         // workgroup = one wavefront (wavefronts do not cooperate; a launch of many frames
                                      // runs 2 % faster when a finished wavefront makes room at once, not its workgroup)
constexpr int D_THREADS_SMALL = 256;  // ... four for launches of up to 8 frames
constexpr int D_KPW = 8;        // keypoints per wavefront (batches)
constexpr int D_KPW_SMALL = 4;  // ... when a launch holds only a few frames: more, shorter wavefronts
constexpr int D_RAW_ROWS = 32, D_BLR_ROWS = 48, D_PITCH = 64;  // staged rows (16 per load instruction), bytes per row
constexpr int D_PAD = 16;                 // bytes in front of the level patch: its row 0 is read from offset -1
// Fused form (the blurred patch is made in LDS from the level, no blurred level in memory):
// The three pitches are chosen for the LDS banks (dword banks: 32 for 4-byte accesses and for stores,
// 64 for 16-byte reads; lanes are served in fixed groups).  Row pass: lane = item (pair of rows pr,
// column group g), 10 groups per pair; with 21 dwords per row a pair is 42 = 10 mod 32, so a lane's
// bank is its item number and the 32 lanes of a group never share one (at 20 dwords every read took
// two turns).  The price is that staged rows are only dword-aligned: they are stored as dwords, which
// costs the same LDS time as the 16-byte stores did.  Column pass: lane = (strip vs, group vq) reads
// 16 bytes at 3 vs pairs + 4 vq dwords; 3 * 56 = 40 mod 64 puts lane 10 vs + vq on 16-byte slot
// (lane mod 16), distinct within each of the instruction's 16-lane groups (at 44 dwords: three turns).
constexpr int DF_RAW_ROWS = 48, DF_PITCH = 84;     // staged level rows -21..26 around the keypoint, bytes per row (64 used)
constexpr int DF_HP_PAIRS = 23, DF_HP_PITCH = 56;  // row pass of pairs of staged rows: dwords per pair (40 used)
constexpr int DF_B_PITCH = 48;                     // bytes per row of the blurred patch (rows and columns -18..18 used)
constexpr int DF_WAVE_BYTES = D_PAD + DF_RAW_ROWS * DF_PITCH + DF_HP_PAIRS * DF_HP_PITCH * 4;
constexpr int D_WAVE_BYTES = D_PAD + (D_RAW_ROWS + D_BLR_ROWS) * D_PITCH;

// Steering tables, filled by upload_orient_tables():
//  d_patf[pair]      = (x0, x1, y0, y1) of the rBRIEF pair as floats;
//  d_odw[row][half]  = weights of the 16 staged bytes a lane sums for the intensity centroid:
//                      4 dwords of |u| (0 outside the disc) and 4 dwords of 0/1.
__device__ float4 d_patf[256];
__device__ uint32_t d_odw[32][2][8];
//  d_hitems[a][i]    = item i of the fused blur's row pass (three rounds of 64 lanes) for a patch whose first
//                      column lies a bytes into its dword: the staged byte offset of the item's dwords (pair of
//                      rows, group of four columns; + 4, so that it is not negative) and, in the upper half, the
//                      byte offset of its results.  Only the items that a sampled pixel can depend on are listed:
//                      the rotated pattern stays in a disc, and the corners of the 37 x 37 patch are never read.
constexpr int DF_ITEMS = 192;
__device__ uint32_t d_hitems[4][DF_ITEMS];
//  d_vitems[a][lane] = the same for the column pass: a lane's block of 6 rows x 4 columns of the blurred patch (byte
//                      offset of the first of the six row pairs it reads; byte offset of its first output).
__device__ uint32_t d_vitems[4][64];

// cos and sin of x in [0, 2*pi + small]: quadrant reduction and the classic double precision
// minimax kernels on [-pi/4, pi/4] (error about 1e-16, so rounding the result to float gives
// the correctly rounded float except in astronomically rare ties).
__device__ __forceinline__ void sincos_small(double x, double& sn, double& cs) {
  const double k = __builtin_rint(x * 6.36619772367581382433e-01);
  double r = __builtin_fma(-k, 1.57079632673412561417e+00, x);
  r = __builtin_fma(-k, 6.07710050650619224932e-11, r);
  const double z = r * r;
  double ps = 1.58969099521155010221e-10;
  ps = __builtin_fma(ps, z, -2.50507602534068634195e-08);
  ps = __builtin_fma(ps, z, 2.75573137070700676789e-06);
  ps = __builtin_fma(ps, z, -1.98412698298579493134e-04);
  ps = __builtin_fma(ps, z, 8.33333333332248946124e-03);
  ps = __builtin_fma(ps, z, -1.66666666666666324348e-01);
  const double s = __builtin_fma(r * z, ps, r);
  double pc = -1.13596475577881948265e-11;
  pc = __builtin_fma(pc, z, 2.08757232129817482790e-09);
  pc = __builtin_fma(pc, z, -2.75573143513906633035e-07);
  pc = __builtin_fma(pc, z, 2.48015872894767294178e-05);
  pc = __builtin_fma(pc, z, -1.38888888888741095749e-03);
  pc = __builtin_fma(pc, z, 4.16666666666666019037e-02);
  const double c = __builtin_fma(z * z, pc, __builtin_fma(-0.5, z, 1.0));
  const int q = (int)k & 3;
  sn = q == 0 ? s : q == 1 ? c : q == 2 ? -s : -c;
  cs = q == 0 ? c : q == 1 ? -s : q == 2 ? -c : s;
}

typedef float od_f2 __attribute__((ext_vector_type(2)));

// The 16-byte pieces of a keypoint's two patches that one lane moves (see the kernel).
struct Staged { uint4 a0, a1, b0, b1, b2; };
__device__ __forceinline__ Staged od_issue(const uint8_t* __restrict__ pyr, const uint8_t* __restrict__ blur, uint32_t lvl,
                                           int cx, int pitch, int rs, int seg) {
  const int al = (cx + kXOff - 15) & 15, al2 = (cx + kXOff - 18) & 15;
  const size_t step = (size_t)16 * pitch;
  const uint8_t* g = pyr + lvl - 15 - al + (ptrdiff_t)(rs - 15) * pitch + 16 * seg;
  const uint8_t* gb = blur + lvl - 18 - al2 + (ptrdiff_t)(rs - 18) * pitch + 16 * seg;
  Staged s;
  s.a0 = *reinterpret_cast<const uint4*>(g);
  s.a1 = *reinterpret_cast<const uint4*>(g + step);
  s.b0 = *reinterpret_cast<const uint4*>(gb);
  s.b1 = *reinterpret_cast<const uint4*>(gb + step);
  s.b2 = *reinterpret_cast<const uint4*>(gb + 2 * step);
  return s;
}

// the fused form's three pieces: rows rs, rs + 16, rs + 32 of the level patch (b0, b1 unused)
__device__ __forceinline__ Staged od_issue_fused(const uint8_t* __restrict__ pyr, uint32_t lvl, int cx, int pitch, int rs,
                                                 int seg) {
  const int al = (cx + kXOff - 21) & 15;
  // a base that is the same for all lanes plus a 32-bit offset of the lane
  const uint8_t* g = pyr + (lvl - 21 - al - 21 * (uint32_t)pitch);
  const uint32_t off = __umul24(rs, pitch) + 16 * seg, step = 16u * (uint32_t)pitch;
  Staged s;
  s.a0 = *reinterpret_cast<const uint4*>(g + off);
  s.a1 = *reinterpret_cast<const uint4*>(g + (off + step));
  s.b0 = *reinterpret_cast<const uint4*>(g + (off + 2 * step));
  return s;
}

// orders a wavefront's LDS stores before the loads of its other lanes that follow
__device__ __forceinline__ void od_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// MODE 0: the descriptor samples the blurred level that blur_kernel wrote.
// MODE 1 / 2: the blurred patch is made here, in LDS, from the level's pixels (2: taps that sum
// to 257, saturated results) - the same fixed-point arithmetic as blur_kernel, whose row and
// column passes are exact, so only the final rounding matters and it is the same.
template <int KPW, int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void orient_desc_kernel(const uint8_t* __restrict__ pyr,
                                                                const uint8_t* __restrict__ blur,
                                                                const uint32_t* __restrict__ sel_all,
                                                                const int* __restrict__ sel_cnt, Plan P,
                                                                OrbKeyPoint* __restrict__ out_kps,
                                                                uint8_t* __restrict__ out_desc,
                                                                int* __restrict__ n_out, int max_kp, int nframes,
                                                                int nchunks, int t0, int t1, int t2, int t3) {
  constexpr bool FUSED = MODE != 0;
  __shared__ __attribute__((aligned(16))) uint8_t s_patch[THREADS / 64][FUSED ? DF_WAVE_BYTES : D_WAVE_BYTES];
  // Workgroups are handed to the 8 XCDs round-robin by linear index.  Give every XCD whole
  // frames, one after the other, so that the patches its wavefronts gather stay in its L2.
  const int w = blockIdx.x;
  const int xcd = w & 7, slot = w >> 3;
  const int f = __builtin_amdgcn_readfirstlane((slot / nchunks) * 8 + xcd);
  if (f >= nframes) return;
  const int chunk = slot - (slot / nchunks) * nchunks;
  const int lane = threadIdx.x & 63;
  const uint8_t* const fpyr = pyr + (size_t)f * P.slab;   // the frame's slab
  const uint8_t* const fblur = blur + (size_t)f * P.slab;
  // A wavefront takes KPW consecutive slots of the per-frame selected array, so that the
  // steering tables (constant per lane) are fetched once for all of them.
  const int k0 = __builtin_amdgcn_readfirstlane((chunk * (THREADS / 64) + (int)(threadIdx.x >> 6)) * KPW);
  if (k0 >= P.sel_total) return;
  // one round trip for everything needed to find the keypoints: the per-level counts
  // (lane l holds level l) and the selected records (lane j holds slot k0 + j)
  const int mycnt = lane < P.nlevels ? sel_cnt[f * P.nlevels + lane] : 0;
  const uint32_t myv = (lane < KPW && k0 + lane < P.sel_total) ? sel_all[(size_t)f * P.sel_total + k0 + lane] : 0u;
  float4 pat[4];  // pattern pairs lane + 64*j
#pragma unroll
  for (int j = 0; j < 4; j++) pat[j] = d_patf[lane + 64 * j];
  const int row = min(lane >> 1, 31), half = lane & 1;
  const uint4 wm = *reinterpret_cast<const uint4*>(&d_odw[row][half][0]);
  const uint4 w1 = *reinterpret_cast<const uint4*>(&d_odw[row][half][4]);
  uint8_t* s_raw = s_patch[threadIdx.x >> 6] + D_PAD;
  uint8_t* s_blr = s_raw + D_RAW_ROWS * D_PITCH;                                        // MODE 0
  uint8_t* s_hp = s_raw + DF_RAW_ROWS * DF_PITCH;                                       // fused: row-pass results
  // fused form, row pass: item = (pair of staged rows 0..21, group of 4 columns 0..9); of the 220
  // items the 192 (or fewer) listed in d_hitems, in 3 rounds of 64 lanes; column pass: lane =
  // (group of 4 columns 0..9, strip 0..5 of the 7 rows from row 6 * strip on: a strip starts on an
  // even row, its last row is the next one's first)
  uint32_t HA0 = 0, HB0 = 0, HA1 = 0, HB1 = 0, HC1 = 0, HA2 = 0, HB2 = 0, HC2 = 0, HC3 = 0;
  uint32_t T1 = 0, TPe0 = 0, TPe1 = 0, TPe2 = 0, TPe3 = 0, TPo0 = 0, TPo1 = 0, TPo2 = 0, TPo3 = 0;
  if constexpr (FUSED) {
    // Row pass: pixel j of a group of 4 uses bytes j + 1 .. j + 7 of three dwords; instead of
    // shifting the bytes into place, each pixel has its own copies of the taps (t0 t1 t2 t3 t2 t1
    // t0), shifted: HA* go with the first dword, HB* / T1 with the second, HC* with the third.
    auto pack4 = [](int b0, int b1, int b2, int b3) {
      return (uint32_t)b0 | ((uint32_t)b1 << 8) | ((uint32_t)b2 << 16) | ((uint32_t)b3 << 24);
    };
    HA0 = pack4(0, t0, t1, t2); HB0 = pack4(t3, t2, t1, t0);
    HA1 = pack4(0, 0, t0, t1);  HB1 = pack4(t2, t3, t2, t1); HC1 = pack4(t0, 0, 0, 0);
    HA2 = pack4(0, 0, 0, t0);   HB2 = pack4(t1, t2, t3, t2); HC2 = pack4(t1, t0, 0, 0);
    T1 = pack4(t0, t1, t2, t3); HC3 = pack4(t2, t1, t0, 0);
    TPe0 = (uint32_t)t0 | ((uint32_t)t1 << 16); TPe1 = (uint32_t)t2 | ((uint32_t)t3 << 16);
    TPe2 = (uint32_t)t2 | ((uint32_t)t1 << 16); TPe3 = (uint32_t)t0;
    TPo0 = (uint32_t)t0 << 16; TPo1 = (uint32_t)t1 | ((uint32_t)t2 << 16);
    TPo2 = (uint32_t)t3 | ((uint32_t)t2 << 16); TPo3 = (uint32_t)t1 | ((uint32_t)t0 << 16);
  }

  // Where the wavefront's slots live.  Lane j works out slot k0 + j once: its level (the last one
  // whose first slot is not beyond it), its row of the frame's output (the counts of the levels
  // before it, plus its index in its own) and whether it holds a keypoint at all; the loop below
  // takes keypoint kk's answer from lane kk.
  uint32_t loc_v;   // level | output row << 4 | valid << 31
  uint32_t lvl_v;   // offset of the keypoint's pixel in the frame's slab
  int pitch_v;      // row pitch of its level
  float scale_v, size_v;  // mvScaleFactor of its level, keypoint size
  {
    // lane j: level j's entries of the plan; a keypoint's lane takes those of its level with a lane permute
    const LevelGeom& G = P.lv[min(lane, kMaxLevels - 1)];
    const int t_first = G.sel_off, t_pitch = G.pitch, t_off = G.off;
    const float t_scale = G.scale, t_size = G.kp_size;
    const int k = k0 + lane;
    int l = 0;  // first slots grow with the level: the level is the number of those not beyond k
#pragma unroll
    for (int j = 1; j < kMaxLevels; j++) l += (j < P.nlevels && k >= P.lv[j].sel_off) ? 1 : 0;
    // lane j holds level j's count (lanes from nlevels on: 0); running sums along the first 16 lanes
    int incl = mycnt;
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, true);  // row_shr 1
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, true);  // row_shr 2
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, true);  // row_shr 4
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, true);  // row_shr 8
    const int cnt_l = __builtin_amdgcn_ds_bpermute(4 * l, mycnt);
    const int before = __builtin_amdgcn_ds_bpermute(4 * l, incl) - cnt_l;
    const int first = __builtin_amdgcn_ds_bpermute(4 * l, t_first);
    const int pitch = __builtin_amdgcn_ds_bpermute(4 * l, t_pitch);
    const int off = __builtin_amdgcn_ds_bpermute(4 * l, t_off);
    scale_v = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * l, __builtin_bit_cast(int, t_scale)));
    size_v = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * l, __builtin_bit_cast(int, t_size)));
    if (k == 0) n_out[f] = __builtin_amdgcn_readlane(incl, 15);
    const int i = k - first, o = before + i;
    const bool ok = lane < KPW && k < P.sel_total && i < cnt_l && o < max_kp;
    loc_v = (uint32_t)l | ((uint32_t)o << 4) | (ok ? 0x80000000u : 0u);
    const int cx = (int)(myv & kCoordMask) + kBorder, cy = (int)((myv >> kCoordBits) & kCoordMask) + kBorder;
    pitch_v = pitch;
    lvl_v = (uint32_t)off + (uint32_t)(cy + kEdge) * (uint32_t)pitch + (uint32_t)(cx + kXOff);
  }
  struct Loc { bool ok; int l, o, cx, cy, pitch; uint32_t v; uint32_t lvl; };
  auto locate = [&](int kk) {
    Loc q;
    q.ok = false; q.l = 0; q.o = 0; q.cx = 0; q.cy = 0; q.pitch = 0; q.v = 0; q.lvl = 0;
    if (kk >= KPW) return q;
    const uint32_t p = __builtin_amdgcn_readlane(loc_v, kk);
    if (!(p >> 31)) return q;
    q.ok = true;
    q.v = __builtin_amdgcn_readlane(myv, kk);
    q.l = (int)(p & 15u); q.o = (int)((p >> 4) & 0x7FFFFFFu);
    q.cx = (int)(q.v & kCoordMask) + kBorder; q.cy = (int)((q.v >> kCoordBits) & kCoordMask) + kBorder;
    q.pitch = __builtin_amdgcn_readlane(pitch_v, kk);
    q.lvl = __builtin_amdgcn_readlane(lvl_v, kk);
    return q;
  };
  // sum over the wavefront: within rows of 16 lanes by DPP, then the four row totals
  auto wave_sum = [](int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
    x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    x += __builtin_amdgcn_update_dpp(0, x, 0x124, 0xF, 0xF, false);  // row_ror 4
    x += __builtin_amdgcn_update_dpp(0, x, 0x128, 0xF, 0xF, false);  // row_ror 8
    return __builtin_amdgcn_readlane(x, 0) + __builtin_amdgcn_readlane(x, 16) + __builtin_amdgcn_readlane(x, 32) +
           __builtin_amdgcn_readlane(x, 48);
  };
  // angle (degrees) and the rotation's cosine and sine from the patch moments
  auto steer = [](int m01, int m10, float& angle, float& a, float& b) {
    angle = fast_atan2_deg((float)m01, (float)m10);
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    const float rad = __fmul_rn(angle, factorPI);
    double sn, cs;
    sincos_small((double)rad, sn, cs);
    a = (float)cs; b = (float)sn;
  };
  // With several keypoints per wavefront the orientations are made first, for all of them: a lane
  // takes the 16 bytes of its half row of the disc straight from the level (a dword-aligned
  // 16-byte load and the dword after it, shifted into place), the moments of keypoint j end up in
  // lane j, and the angle, cosine and sine - the same arithmetic in every lane, whatever it holds -
  // are evaluated once for the wavefront's keypoints instead of once per keypoint.  The loop
  // below then finds a keypoint's angle and rotation in lane j of these registers.
  const int rs = lane >> 2, seg = lane & 3;
  Staged st;
  // (an empty slot - pitch 0, column 0 - reads near the start of the slab, and its pieces are not used: the loads
  // are issued without a branch, so that the registers they fill are not handed on through copies)
  auto issue = [&](const Loc& q) {
    const uint32_t lvl = q.ok ? q.lvl : 64u;
    if constexpr (FUSED) return od_issue_fused(fpyr, lvl, q.cx, q.pitch, rs, seg);
    else return od_issue(fpyr, fblur, lvl, q.cx, q.pitch, rs, seg);
  };
  // this lane's row-pass items of the fused blur (see d_hitems): requested along with a keypoint's
  // patch, one keypoint ahead
  uint32_t hit[4] = {0, 0, 0, 0};
  auto issue_items = [&](const Loc& q) {
    if constexpr (FUSED) {
      const int a = (q.cx + kXOff - 21 + 3) & 3;
      const uint32_t* tab = &d_hitems[a][lane];
      hit[0] = tab[0]; hit[1] = tab[64]; hit[2] = tab[128];
      hit[3] = d_vitems[a][lane];
    }
  };
  Loc cur;
  float ang_v = 0.f, cos_v = 1.f, sin_v = 0.f;
  {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4), aligned(4)));
    u32x4 c4[KPW];
    uint32_t c1[KPW], csh[KPW];
#pragma unroll
    for (int kk = 0; kk < KPW; kk++) {
      const Loc q = locate(kk);
      // first byte: column -16 (left half, weight 0) or 0 (right half) of row `row` - 15, as the
      // slab pointer plus a 32-bit offset; the address is cut to a dword boundary and the bytes are
      // shifted into place afterwards.  An empty slot reads the start of the slab (and is not used).
      const uint32_t lv = q.ok ? q.lvl - 15u * (uint32_t)q.pitch - 16u : 0u;
      const uint32_t d = ((uint32_t)reinterpret_cast<uintptr_t>(fpyr) + lv) & 3u;
      const uint32_t t = __umul24(row, q.pitch) + (16u * half + d);
      const uint8_t* g = fpyr + ((lv - d) + (t & ~3u));
      csh[kk] = t & 3u;
      c4[kk] = *reinterpret_cast<const u32x4*>(g);
      c1[kk] = *reinterpret_cast<const uint32_t*>(g + 16);
      if (kk == 0) {  // the first keypoint's patches are on their way while the moments are summed
        cur = q;
        st = issue(q);
        issue_items(q);
      }
    }
    int m10_v = 0, m01_v = 0;
    int part[2 * KPW];  // per lane: its share of keypoint kk's m10 (entry 2 kk) and m01 (2 kk + 1)
#pragma unroll
    for (int kk = 0; kk < KPW; kk++) {
      const uint32_t sh = csh[kk];
      const uint32_t y0 = __builtin_amdgcn_alignbyte(c4[kk].y, c4[kk].x, sh), y1 = __builtin_amdgcn_alignbyte(c4[kk].z, c4[kk].y, sh);
      const uint32_t y2 = __builtin_amdgcn_alignbyte(c4[kk].w, c4[kk].z, sh), y3 = __builtin_amdgcn_alignbyte(c1[kk], c4[kk].w, sh);
      uint32_t su = __builtin_amdgcn_udot4(y0, wm.x, 0u, false);
      su = __builtin_amdgcn_udot4(y1, wm.y, su, false);
      su = __builtin_amdgcn_udot4(y2, wm.z, su, false);
      su = __builtin_amdgcn_udot4(y3, wm.w, su, false);
      uint32_t s1 = __builtin_amdgcn_udot4(y0, w1.x, 0u, false);
      s1 = __builtin_amdgcn_udot4(y1, w1.y, s1, false);
      s1 = __builtin_amdgcn_udot4(y2, w1.z, s1, false);
      s1 = __builtin_amdgcn_udot4(y3, w1.w, s1, false);
      part[2 * kk] = half ? (int)su : -(int)su;
      part[2 * kk + 1] = __mul24(row - 15, (int)s1);
    }
    if constexpr (KPW == 8) {
      // Sixteen sums over the wavefront at once.  Each step adds lanes a fixed pattern apart and
      // halves the number of registers: the sum that a register carries on is kept only in the
      // lanes that the step's pattern leaves to it (two registers share one afterwards).  The
      // first two steps act on the four groups of four lanes of a row of 16 and write under a
      // group mask; the two within a group select by lane; the rows are added last.
#define ORB_DPP_ADD_BANKS(R, Y, CTRL, BANKS) \
  asm volatile("s_nop 1\n\tv_add_u32_dpp %0, %1, %1 " CTRL " row_mask:0xf bank_mask:" BANKS : "+v"(R) : "v"(Y))
      int r8[8], r4[4], r2[2];
#pragma unroll
      for (int i = 0; i < 8; i++) {  // lane i + lane 15 - i: groups 0, 1 keep entry 2 i, groups 2, 3 entry 2 i + 1
        r8[i] = part[2 * i] + __builtin_amdgcn_update_dpp(0, part[2 * i], 0x140, 0xF, 0xF, false);  // row_mirror
        ORB_DPP_ADD_BANKS(r8[i], part[2 * i + 1], "row_mirror", "0xc");
      }
#pragma unroll
      for (int i = 0; i < 4; i++) {  // lane i + lane 7 - i of each half row: groups 1, 3 keep r8[2 i], groups 0, 2 r8[2 i + 1]
        r4[i] = r8[2 * i] + __builtin_amdgcn_update_dpp(0, r8[2 * i], 0x141, 0xF, 0xF, false);  // row_half_mirror
        ORB_DPP_ADD_BANKS(r4[i], r8[2 * i + 1], "row_half_mirror", "0x5");
      }
#undef ORB_DPP_ADD_BANKS
      const bool up2 = (lane & 2) != 0, up1 = (lane & 1) != 0;
#pragma unroll
      for (int i = 0; i < 2; i++) {  // lanes two apart: lanes 0, 1 of a group keep r4[2 i], lanes 2, 3 r4[2 i + 1]
        const int keep = up2 ? r4[2 * i + 1] : r4[2 * i], give = up2 ? r4[2 * i] : r4[2 * i + 1];
        r2[i] = keep + __builtin_amdgcn_update_dpp(0, give, 0x4E, 0xF, 0xF, false);  // quad_perm [2,3,0,1]
      }
      const int keep = up1 ? r2[1] : r2[0], give = up1 ? r2[0] : r2[1];
      int tot = keep + __builtin_amdgcn_update_dpp(0, give, 0xB1, 0xF, 0xF, false);  // quad_perm [1,0,3,2]
      tot += __builtin_amdgcn_ds_bpermute(4 * (lane ^ 16), tot);
      tot += __builtin_amdgcn_ds_bpermute(4 * (lane ^ 32), tot);
      // Lane 4 b + 2 q1 + q0 of a row now holds entry 8 q0 + 4 q1 + {2, 0, 3, 1}[b]; lane kk takes
      // entries 2 kk and 2 kk + 1 (the two tables list the lanes that hold them, a nibble per kk).
      const uint32_t sa = (0x37152604u >> (4 * (lane & 7))) & 15u, sb = (0xBF9DAE8Cu >> (4 * (lane & 7))) & 15u;
      m10_v = __builtin_amdgcn_ds_bpermute(4 * (int)sa, tot);
      m01_v = __builtin_amdgcn_ds_bpermute(4 * (int)sb, tot);
    } else {
#pragma unroll
      for (int kk = 0; kk < KPW; kk++) {
        const int m10 = wave_sum(part[2 * kk]), m01 = wave_sum(part[2 * kk + 1]);
        if (lane == kk) m10_v = m10;
        if (lane == kk) m01_v = m01;
      }
    }
    steer(m01_v, m10_v, ang_v, cos_v, sin_v);
    // the keypoint records (epilogue of operator(): level coordinates scaled to the image), lane j's own
    if (loc_v >> 31) {
      const int l = (int)(loc_v & 15u), o = (int)((loc_v >> 4) & 0x7FFFFFFu);
      const int cx = (int)(myv & kCoordMask) + kBorder, cy = (int)((myv >> kCoordBits) & kCoordMask) + kBorder;
      OrbKeyPoint kp;
      kp.x = l ? __fmul_rn((float)cx, scale_v) : (float)cx;
      kp.y = l ? __fmul_rn((float)cy, scale_v) : (float)cy;
      kp.size = size_v;
      kp.angle = ang_v;
      kp.response = (float)(myv >> 24);
      kp.octave = l;
      out_kps[(size_t)f * max_kp + o] = kp;
    }
  }
  // The two patches of a keypoint are staged in the wave's LDS region: rows -15..15 of the
  // level (orientation disc) and rows -18..18 of the blurred level (a rotated pattern point
  // reaches at most round(13*sqrt(2)) = 18 px), 64 bytes per row from a 16-byte aligned
  // start.  A lane moves 16 bytes, a load instruction 16 rows, so a keypoint takes five of
  // them; the rows staged beyond the patches (up to 29 below the centre) still lie in the
  // slab's frame.  The loads of the next keypoint are issued before the current one is computed.
#pragma unroll
  for (int kk = 0; kk < KPW; kk++) {
  const Loc nxt = locate(kk + 1);
  const int o = cur.o, cx = cur.cx;
  const int al = (cx + kXOff - (FUSED ? 21 : 15)) & 15, al2 = (cx + kXOff - 18) & 15;
  const uint32_t hit0 = hit[0], hit1 = hit[1], hit2 = hit[2], hit3 = hit[3];  // (hit[] goes on to the next keypoint's)
  if (cur.ok) {
    if constexpr (FUSED) {
      uint32_t* dr = reinterpret_cast<uint32_t*>(&s_raw[rs * DF_PITCH + 16 * seg]);
      auto put = [&](int rows, const uint4& v) {
        uint32_t* d = dr + rows * (DF_PITCH / 4);
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      };
      put(0, st.a0); put(16, st.a1); put(32, st.b0);
    } else {
      uint4* dr = reinterpret_cast<uint4*>(&s_raw[rs * D_PITCH + 16 * seg]);
      uint4* db = reinterpret_cast<uint4*>(&s_blr[rs * D_PITCH + 16 * seg]);
      dr[0] = st.a0; dr[16 * (D_PITCH / 16)] = st.a1;
      db[0] = st.b0; db[16 * (D_PITCH / 16)] = st.b1; db[32 * (D_PITCH / 16)] = st.b2;
    }
  }
  if (kk + 1 < KPW) {  // (known at compile time: the loop is unrolled)
    st = issue(nxt);
    issue_items(nxt);
  }
  if (!cur.ok) {
    cur = nxt;
    continue;
  }
  od_wave_sync();

  const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cos_v), kk));
  const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sin_v), kk));

  // steered BRIEF on the blurred patch (two sample points per packed operation)
  const od_f2 a2 = {a, a}, b2 = {b, b};
  if constexpr (FUSED) {
    // Row pass: the group G0 + g holds the outputs at staged bytes 4 (G0 + g) .. + 3; the patch's
    // first column (-18) is staged byte al + 3.  Pixel j of a group uses bytes j + 1 .. j + 7 of
    // the three dwords from the one before the group on; the results of two rows share a dword.
    const int G0 = (al + 3) >> 2;
    uint32_t wr[3][2][3];  // every round's dwords are read before the first results are stored
#pragma unroll
    for (int rd = 0; rd < 3; rd++) {
      const uint32_t hv = rd == 0 ? hit0 : rd == 1 ? hit1 : hit2;
      const uint8_t* src = s_raw + (hv & 0xFFFFu) + (4 * G0 - 4);
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const uint32_t* w = reinterpret_cast<const uint32_t*>(src + k * DF_PITCH);
        wr[rd][k][0] = w[0]; wr[rd][k][1] = w[1]; wr[rd][k][2] = w[2];
      }
    }
#pragma unroll
    for (int rd = 0; rd < 3; rd++) {
      {
        uint32_t hh[2][4];
#pragma unroll
        for (int k = 0; k < 2; k++) {
          const uint32_t w0 = wr[rd][k][0], w1 = wr[rd][k][1], w2 = wr[rd][k][2];
          // (the taps are shifted instead of the pixels)
          hh[k][0] = __builtin_amdgcn_udot4(w1, HB0, __builtin_amdgcn_udot4(w0, HA0, 0u, false), false);
          hh[k][1] = __builtin_amdgcn_udot4(w2, HC1, __builtin_amdgcn_udot4(w1, HB1, __builtin_amdgcn_udot4(w0, HA1, 0u, false), false), false);
          hh[k][2] = __builtin_amdgcn_udot4(w2, HC2, __builtin_amdgcn_udot4(w1, HB2, __builtin_amdgcn_udot4(w0, HA2, 0u, false), false), false);
          hh[k][3] = __builtin_amdgcn_udot4(w2, HC3, __builtin_amdgcn_udot4(w1, T1, 0u, false), false);
        }
        uint4 o;
        // (results are below 2^16: the low halves of the two rows' sums make the dword)
        o.x = __builtin_amdgcn_perm(hh[1][0], hh[0][0], 0x05040100u); o.y = __builtin_amdgcn_perm(hh[1][1], hh[0][1], 0x05040100u);
        o.z = __builtin_amdgcn_perm(hh[1][2], hh[0][2], 0x05040100u); o.w = __builtin_amdgcn_perm(hh[1][3], hh[0][3], 0x05040100u);
        *reinterpret_cast<uint4*>(s_hp + ((rd == 0 ? hit0 : rd == 1 ? hit1 : hit2) >> 16)) = o;
      }
    }
    od_wave_sync();
    // Column pass: 6 output rows x 4 columns per lane from 6 pairs of rows (the lane's block: see
    // d_vitems); output row i (patch row i - 18) uses staged rows i .. i + 6.  The blurred patch
    // overwrites the staged level.
    {
      const uint4* hp = reinterpret_cast<const uint4*>(s_hp + (hit3 & 0xFFFFu));
      const uint32_t v_dst = hit3 >> 16;
      uint4 v[6];
#pragma unroll
      for (int k = 0; k < 6; k++) v[k] = hp[k * (DF_HP_PITCH / 4)];
#pragma unroll
      for (int j = 0; j < 6; j++) {
        const int p = j >> 1;
        const uint32_t tp0 = (j & 1) ? TPo0 : TPe0, tp1 = (j & 1) ? TPo1 : TPe1, tp2 = (j & 1) ? TPo2 : TPe2, tp3 = (j & 1) ? TPo3 : TPe3;
        uint32_t c0 = blur_dot2(v[p + 3].x, tp3, blur_dot2(v[p + 2].x, tp2, blur_dot2(v[p + 1].x, tp1, blur_dot2(v[p].x, tp0, 32768u))));
        uint32_t c1 = blur_dot2(v[p + 3].y, tp3, blur_dot2(v[p + 2].y, tp2, blur_dot2(v[p + 1].y, tp1, blur_dot2(v[p].y, tp0, 32768u))));
        uint32_t c2 = blur_dot2(v[p + 3].z, tp3, blur_dot2(v[p + 2].z, tp2, blur_dot2(v[p + 1].z, tp1, blur_dot2(v[p].z, tp0, 32768u))));
        uint32_t c3 = blur_dot2(v[p + 3].w, tp3, blur_dot2(v[p + 2].w, tp2, blur_dot2(v[p + 1].w, tp1, blur_dot2(v[p].w, tp0, 32768u))));
        if (MODE == 2) { c0 = min(c0, 0xFFFFFFu); c1 = min(c1, 0xFFFFFFu); c2 = min(c2, 0xFFFFFFu); c3 = min(c3, 0xFFFFFFu); }
        // byte 2 of every sum is the pixel
        const uint32_t lo = __builtin_amdgcn_perm(c1, c0, 0x0c0c0602u), hi = __builtin_amdgcn_perm(c3, c2, 0x06020c0cu);
        *reinterpret_cast<uint32_t*>(s_raw + v_dst + j * DF_B_PITCH) = lo | hi;
      }
    }
    od_wave_sync();
  }
  constexpr int CP = FUSED ? DF_B_PITCH : D_PITCH;
  // LDS address of the blurred patch's centre, less what the rounding trick below leaves in the sums
  typedef __attribute__((address_space(3))) const uint8_t* lds_bytes;
  const uint32_t ctr = (uint32_t)(uintptr_t)(FUSED ? (lds_bytes)&s_raw[18 * DF_B_PITCH + 18 + ((al + 3) & 3)]
                                                   : (lds_bytes)&s_blr[18 * D_PITCH + 18 + al2]) -
                       (0x400000u * CP + 0x4B400000u);
  int t0[4], t1[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const od_f2 X = {pat[j].x, pat[j].y}, Y = {pat[j].z, pat[j].w};
    const od_f2 r = X * b2 + Y * a2;  // row offsets of the two points
    const od_f2 q = X * a2 - Y * b2;  // column offsets
    // cvRound by the float adder: x + 1.5 * 2^23 rounds x to the nearest integer, ties to even,
    // and leaves 0x4B400000 + x in the bits (|x| < 2^22) - one packed add for two points instead
    // of a rounding and a conversion for each.  The low 24 bits of that, 0x400000 + x, go into a
    // 24-bit multiply-add as they are; the constants are taken off the base address.
    const od_f2 magic = {12582912.f, 12582912.f};
    const od_f2 rm = r + magic, qm = q + magic;
    const uint32_t i0 = (uint32_t)(__mul24(__float_as_int(rm.x), CP) + __float_as_int(qm.x)) + ctr;
    const uint32_t i1 = (uint32_t)(__mul24(__float_as_int(rm.y), CP) + __float_as_int(qm.y)) + ctr;
    t0[j] = *(lds_bytes)(uintptr_t)i0;
    t1[j] = *(lds_bytes)(uintptr_t)i1;
  }
  unsigned long long bits[4];
#pragma unroll
  for (int j = 0; j < 4; j++) bits[j] = __builtin_amdgcn_ballot_w64(t0[j] < t1[j]);
  if (lane < 4) {
    const unsigned long long w = lane == 0 ? bits[0] : lane == 1 ? bits[1] : lane == 2 ? bits[2] : bits[3];
    reinterpret_cast<unsigned long long*>(out_desc + ((size_t)f * max_kp + o) * 32)[lane] = w;
  }
  cur = nxt;
  }  // keypoints of this wavefront
}

// Fills the steering tables on the current device.
hipError_t upload_orient_tables() {
  static const int8_t pattern[256 * 4] = {
#include "orb_pattern.inc"
  };
  // half-widths of the rows of the radius-15 orientation disc (reference
  // src/ORBextractor.cc:544-572 evaluates to these values; checked on the host at start-up)
  static const int umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
  std::vector<float4> pf(256);
  for (int i = 0; i < 256; i++)
    pf[i] = make_float4((float)pattern[4 * i], (float)pattern[4 * i + 2], (float)pattern[4 * i + 1], (float)pattern[4 * i + 3]);
  std::vector<uint32_t> wt(32 * 2 * 8, 0u);
  for (int row = 0; row < 31; row++) {
    const int d = umax[row < 15 ? 15 - row : row - 15];
    for (int half = 0; half < 2; half++)
      for (int bt = 0; bt < 16; bt++) {
        const int au = half ? bt : 16 - bt;  // |u| of staged byte bt: u = bt - 16 (left half) or bt (right half)
        if (au > d) continue;
        wt[(row * 2 + half) * 8 + bt / 4] |= (uint32_t)au << (8 * (bt & 3));
        wt[(row * 2 + half) * 8 + 4 + bt / 4] |= 1u << (8 * (bt & 3));
      }
  }
  // Row-pass items of the fused blur.  A sampled pixel is (round(x b + y a), round(x a - y b)) for a
  // pattern point (x, y), |x|, |y| <= 13, and a rotation (a, b): it lies within half a pixel, per
  // coordinate, of a point of the disc of radius 13 sqrt(2) (+ a margin for the float arithmetic).
  // The blurred pixel at (r, q) needs the row pass at rows r - 3 .. r + 3, column q.  Item (pr, g)
  // holds staged rows -21 + 2 pr and the next one, columns 4 g - a - 18 .. + 3.
  std::vector<uint32_t> hit(4 * DF_ITEMS);
  for (int a = 0; a < 4; a++) {
    int n = 0;
    uint32_t last = 0;
    for (int pr = 0; pr < 22; pr++)
      for (int g = 0; g < 10; g++) {
        bool need = false;
        for (int sr = -21 + 2 * pr; sr <= -20 + 2 * pr; sr++)
          for (int j = 0; j < 4; j++) {
            const int q = 4 * g + j - a - 18;
            for (int r = std::max(sr - 3, -18); r <= std::min(sr + 3, 18); r++) {
              const double dr = std::max(std::abs(r) - 0.5, 0.0), dq = std::max(std::abs(q) - 0.5, 0.0);
              if (dr * dr + dq * dq <= 338.5) need = true;
            }
          }
        if (!need) continue;
        if (n >= DF_ITEMS) return hipErrorInvalidValue;  // (cannot happen: 189 or 190 items)
        last = (uint32_t)(2 * pr * DF_PITCH + 4 * g) | ((uint32_t)(pr * (DF_HP_PITCH * 4) + 16 * g) << 16);
        hit[a * DF_ITEMS + n++] = last;
      }
    if (n == 0) return hipErrorInvalidValue;
    for (; n < DF_ITEMS; n++) hit[a * DF_ITEMS + n] = last;  // idle lanes repeat the last item
  }
  // Column-pass blocks: strips of six output rows from patch row -18 on (the seventh strip would hold row 18
  // alone: it is rows 14 .. 19 instead, which start on an even staged row like the others and need no row
  // pair beyond those the row pass makes; rows 14 .. 17 are then written twice, with the same values)
  std::vector<uint32_t> vit(4 * 64);
  for (int a = 0; a < 4; a++) {
    int n = 0;
    uint32_t last = 0;
    for (int st = 0; st < 7; st++)
      for (int g = 0; g < 10; g++) {
        const int r0 = st < 6 ? -18 + 6 * st : 14;
        bool need = false;
        for (int r = st < 6 ? r0 : 18; r <= std::min(r0 + 5, 18); r++)
          for (int j = 0; j < 4; j++) {
            const int q = 4 * g + j - a - 18;
            const double dr = std::max(std::abs(r) - 0.5, 0.0), dq = std::max(std::abs(q) - 0.5, 0.0);
            if (dr * dr + dq * dq <= 338.5) need = true;
          }
        if (!need) continue;
        if (n >= 64) return hipErrorInvalidValue;  // (cannot happen: 59 or 60 blocks)
        last = (uint32_t)((r0 + 18) / 2 * (DF_HP_PITCH * 4) + 16 * g) | ((uint32_t)((r0 + 18) * DF_B_PITCH + 4 * g) << 16);
        vit[a * 64 + n++] = last;
      }
    for (; n < 64; n++) vit[a * 64 + n] = last;
  }
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(d_hitems), hit.data(), sizeof(uint32_t) * hit.size());
  if (e != hipSuccess) return e;
  e = hipMemcpyToSymbol(HIP_SYMBOL(d_vitems), vit.data(), sizeof(uint32_t) * vit.size());
  if (e != hipSuccess) return e;
  e = hipMemcpyToSymbol(HIP_SYMBOL(d_patf), pf.data(), sizeof(float4) * 256);
  if (e != hipSuccess) return e;
  return hipMemcpyToSymbol(HIP_SYMBOL(d_odw), wt.data(), sizeof(uint32_t) * wt.size());
}

// ---- launchers --------------------------------------------------------------
void launch_pyr_level0(const uint8_t* img, size_t stride, size_t frame_stride, uint8_t* pyr, const Plan& P,
                       int* cand_cnt, int nframes, hipStream_t st) {
  const LevelGeom& L = P.lv[0];
  const int nchunk = L.pitch / 16, nrows = L.h + 2 * kEdge;
  // interior pieces: 16 c - kXOff >= 4 and 16 c - kXOff + 20 <= w
  L0Map M{};
  M.ca = (kXOff + 4 + 15) / 16;
  const int cb = (L.w + kXOff - 20) / 16 + 1;
  M.ni = cb - M.ca >= 2 ? cb - M.ca : 0;
  M.nb = nchunk - M.ni;
  M.ni_m = M.ni ? (uint32_t)((1ull << 32) / (unsigned)M.ni + 1) : 0;
  M.nb_m = (uint32_t)((1ull << 32) / (unsigned)M.nb + 1);
  M.n_int = M.ni * nrows;
  M.total = nchunk * nrows;
  const int per_thread = (M.total + L0_PIECES - 1) / L0_PIECES;
  dim3 grid(nframes, (per_thread + 255) / 256);
  hipLaunchKernelGGL(pyr_level0_kernel, grid, dim3(256), 0, st, img, stride, frame_stride, pyr, P.slab, L, M, cand_cnt,
                     P.nlevels);
}
void launch_pyr_level0_color(const uint8_t* img, size_t stride, size_t frame_stride, int channels, bool rgb_order,
                             const int* gray /* weight R, G, B, shift */, uint8_t* pyr, const Plan& P, int* cand_cnt,
                             int nframes, hipStream_t st) {
  const LevelGeom& L = P.lv[0];
  const int pieces = (L.pitch / 16) * (L.h + 2 * kEdge);
  dim3 grid(nframes, (pieces + 255) / 256);
  const int kr = gray[0], kg = gray[1], kb = gray[2], shift = gray[3], rnd = 1 << (shift - 1);
  const int k0 = rgb_order ? kr : kb, k2 = rgb_order ? kb : kr;
  if (channels == 3)
    hipLaunchKernelGGL(pyr_level0_color_kernel<3>, grid, dim3(256), 0, st, img, stride, frame_stride, pyr, P.slab, L, k0,
                       kg, k2, rnd, shift, cand_cnt, P.nlevels);
  else
    hipLaunchKernelGGL(pyr_level0_color_kernel<4>, grid, dim3(256), 0, st, img, stride, frame_stride, pyr, P.slab, L, k0,
                       kg, k2, rnd, shift, cand_cnt, P.nlevels);
}
void launch_pyr_resize(uint8_t* pyr, const Plan& P, int level, const XTab* xt, const YTab* yt, int nframes,
                       hipStream_t st) {
  const LevelGeom& Ld = P.lv[level];
  dim3 grid((Ld.pitch / 4 + 255) / 256, Ld.h + 2 * kEdge, nframes);
  hipLaunchKernelGGL(pyr_resize_kernel, grid, dim3(256), 0, st, pyr, P.slab, P.lv[level - 1], Ld, xt + Ld.xtab_off,
                     yt + Ld.ytab_off);
}
void launch_pyr_resize_tile(uint8_t* pyr, const Plan& P, int level, const uint32_t* xtp, const YTab* yt,
                            const int2* xrange, int ngx, const int2* yrange, int ngy, int tile_rows, int src_rows,
                            int src_pitch, int nframes, hipStream_t st) {
  hipLaunchKernelGGL(pyr_resize_tile_kernel, dim3(nframes, ngy, ngx), dim3(PR_THREADS), (size_t)src_rows * src_pitch, st,
                     pyr, P.slab, P.lv[level - 1], P.lv[level], xtp, yt + P.lv[level].ytab_off, xrange, yrange,
                     src_pitch, tile_rows);
}
void launch_pyr_resize_dual(uint8_t* pyr, const Plan& P, int level, const uint32_t* xtp1, const uint32_t* xtp2,
                            const YTab* yt, const int2* xrange1, int ngx1, const int2* yrange1, int ngy1,
                            const int2* yrange2, int ngy2, int tile_rows, int spitch1, int spitch0, int pitchB, int offB,
                            int lds_bytes, int nframes, hipStream_t st) {
  hipLaunchKernelGGL(pyr_resize_dual_kernel, dim3(nframes, ngy1 + ngy2, ngx1), dim3(PR_THREADS), (size_t)lds_bytes, st, pyr,
                     P.slab, P.lv[level - 1], P.lv[level], P.lv[level + 1], xtp1, xtp2, yt, xrange1, yrange1, yrange2,
                     spitch1, tile_rows, ngy1, spitch0, pitchB, offB);
}
void launch_pyr_level01(const uint8_t* img, size_t stride, size_t frame_stride, uint8_t* pyr, const Plan& P,
                        const uint32_t* xtp, const YTab* yt, const int2* xrange, int ngx, const int2* yrange, int ngy,
                        const int2* own_rows, int tile_rows, int src_rows, int src_pitch, int* cand_cnt, int nframes,
                        hipStream_t st) {
  hipLaunchKernelGGL(pyr_level01_kernel, dim3(nframes, ngy, ngx), dim3(PR_THREADS), (size_t)src_rows * src_pitch, st, img,
                     stride, frame_stride, pyr, P.slab, P.lv[0], P.lv[1], xtp, yt + P.lv[1].ytab_off, xrange, yrange,
                     own_rows, src_pitch, tile_rows, cand_cnt, P.nlevels);
}
int pyr_tile_cols() { return PR_TC; }
int pyr_tile_rows() { return PR_TR; }
int pyr_tile_rows_small() { return PR_TR_SMALL; }
int pyr_tile_max_src_rows() { return PR_SROWS; }
int pyr_tile_max_src_bytes() { return PR_MAXPITCH; }
// The tallest cell view with which six workgroups of fast_kernel fit a CU's LDS (160 KB, handed
// out in 512-byte units on gfx950; five fit with the tallest view the kernel takes, and the sixth
// is worth a few percent of the kernel's time): at the wide pitch, and at the narrow one.
static long fast_lds_budget() {
  static const long budget = [] {
    hipFuncAttributes a{};
    if (hipFuncGetAttributes(&a, reinterpret_cast<const void*>(fast_kernel)) != hipSuccess) return 0L;
    return (long)((160 * 1024 / 6) / 512 * 512) - (long)a.sharedSizeBytes;  // dynamic LDS a workgroup may take
  }();
  return budget;
}
static int fast_rows_for(long budget, int pitch) {
  const long r = (budget / pitch + 4) / 2;
  return budget < 10 * pitch ? 0 : (int)(r < F_MAXH ? r : F_MAXH);
}
int fast_rows_dense() { return fast_rows_for(fast_lds_budget(), F_PITCH); }
int fast_rows_narrow() { return fast_rows_for(fast_lds_budget(), F_PITCH_NARROW); }
int fast_use_width_narrow() { return F_USEW_NARROW; }
// items [0, ntall) have views taller than fast_rows_dense() rows and are cut for the narrow pitch
// (see fast_kernel; they come first: the longest workgroups should not be the last to start), the
// others follow.  ntall < 0: one launch, everything at the wide pitch with room for the tallest view
// (launches of a few frames: occupancy does not matter).
void launch_fast(const uint8_t* pyr, const Plan& P, const FastWork* items, int nitems, int ntall, uint32_t* cand,
                 int* cand_cnt, int iniTh, int minTh, int nframes, hipStream_t st) {
  const FastPlan FPn{P.slab, P.cand_total, P.nlevels};
  int rows = 7;
  for (int l = 0; l < P.nlevels; l++) rows = max(rows, P.lv[l].hCell + 6);
  rows = min(rows, F_MAXH);
  const int dense = fast_rows_dense();
  if (ntall >= 0 && rows > dense && rows <= fast_rows_narrow()) {
    // one launch, two kinds of workgroup, the same LDS footprint
    const size_t lds = max((2 * (size_t)dense - 4) * F_PITCH, (2 * (size_t)rows - 4) * F_PITCH_NARROW);
    hipLaunchKernelGGL(fast_kernel, dim3(nframes, nitems), dim3(F_THREADS), lds, st, pyr, FPn, items, cand, cand_cnt, iniTh,
                       minTh, dense, rows);
    return;
  }
  if (ntall > 0 && ntall < nitems && rows > dense) {
    // views too tall even for the narrow pitch: those with room for the tallest, then the others
    // with the small footprint
    hipLaunchKernelGGL(fast_kernel, dim3(nframes, ntall), dim3(F_THREADS), (2 * (size_t)rows - 4) * F_PITCH, st, pyr, FPn, items,
                       cand, cand_cnt, iniTh, minTh, rows, rows);
    hipLaunchKernelGGL(fast_kernel, dim3(nframes, nitems - ntall), dim3(F_THREADS), (2 * (size_t)dense - 4) * F_PITCH, st, pyr,
                       FPn, items + ntall, cand, cand_cnt, iniTh, minTh, dense, dense);
    return;
  }
  if (nitems > 0)
    hipLaunchKernelGGL(fast_kernel, dim3(nframes, nitems), dim3(F_THREADS), (2 * (size_t)rows - 4) * F_PITCH, st, pyr, FPn, items,
                       cand, cand_cnt, iniTh, minTh, rows, rows);
}
template <int MAXS, int THREADS>
static hipError_t launch_octree_t(const uint32_t* cand, const int* cand_cnt, uint32_t* pnode, uint32_t* sel,
                                  int* sel_cnt, const Plan& P, int nframes, hipStream_t st) {
  const size_t lds = sizeof(OtShared<MAXS>);
  if (lds > 64 * 1024) {
    // the attribute belongs to the function (per device): raise it to this size before the launch
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&octree_kernel<MAXS, THREADS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((octree_kernel<MAXS, THREADS>), dim3(nframes, P.nlevels), dim3(THREADS), lds, st, cand, cand_cnt,
                     pnode, sel, sel_cnt, P);
  return hipSuccess;
}
// slots: enough for the largest level (quota + 3 nodes, dead root slots, slack).  Frames of
// more than about 1.5 megapixels have tens of thousands of candidates on their first levels:
// the point loops then dominate and want the larger workgroup.
hipError_t launch_octree(const uint32_t* cand, const int* cand_cnt, uint32_t* pnode, uint32_t* sel, int* sel_cnt,
                         const Plan& P, int max_slots_needed, int nframes, hipStream_t st) {
  const bool big = (long long)P.lv[0].w * P.lv[0].h > 1500000;
  if (max_slots_needed <= 1024) {
    if (big) return launch_octree_t<1024, 1024>(cand, cand_cnt, pnode, sel, sel_cnt, P, nframes, st);
    return launch_octree_t<1024, 512>(cand, cand_cnt, pnode, sel, sel_cnt, P, nframes, st);
  }
  if (max_slots_needed <= 2048) {
    if (big) return launch_octree_t<2048, 1024>(cand, cand_cnt, pnode, sel, sel_cnt, P, nframes, st);
    return launch_octree_t<2048, 512>(cand, cand_cnt, pnode, sel, sel_cnt, P, nframes, st);
  }
  // up to about 3000 keypoints on one level: the node tables take nearly all of a compute unit's LDS
  // (one workgroup per CU)
  static_assert(sizeof(OtShared<3072>) <= 160 * 1024, "the largest node table must fit the LDS");
  return launch_octree_t<3072, 1024>(cand, cand_cnt, pnode, sel, sel_cnt, P, nframes, st);
}
void launch_blur(const uint8_t* pyr, uint8_t* blur, const Plan& P, const BlurItem* items, int nitems, const int* taps,
                 int nframes, hipStream_t st) {
  // With the frame alone as the fastest block index, the workgroups in flight work on the same
  // tile of every frame of the batch, one slab apart; beyond 128 frames that costs this
  // bandwidth-bound kernel a quarter of its speed (0.324 instead of 0.242 ms per 256 frames).
  // Large batches are therefore dispatched 8 frames (one per XCD) at a time.  The other kernels
  // gain nothing from it, and FAST loses: its items are ordered longest first over the batch.
  int fchunk = nframes > 128 ? 8 : nframes;
  while ((nframes + fchunk - 1) / fchunk > 65535) fchunk *= 2;  // grid.z limit
  fchunk = min(fchunk, nframes);
  const int nchunks = (nframes + fchunk - 1) / fchunk;
  if (taps[3] + 2 * (taps[0] + taps[1] + taps[2]) > 256)
    hipLaunchKernelGGL(blur_kernel<true>, dim3(fchunk, nitems, nchunks), dim3(B_THREADS), 0, st, pyr, blur, P, items,
                       nitems, fchunk, nframes, taps[0], taps[1], taps[2], taps[3]);
  else
    hipLaunchKernelGGL(blur_kernel<false>, dim3(fchunk, nitems, nchunks), dim3(B_THREADS), 0, st, pyr, blur, P, items,
                       nitems, fchunk, nframes, taps[0], taps[1], taps[2], taps[3]);
}
// taps == nullptr: the descriptors sample the blurred levels in `blur`; otherwise `blur` is not
// read and the kernel blurs each keypoint's patch itself with these taps
void launch_orient_desc(const uint8_t* pyr, const uint8_t* blur, const uint32_t* sel, const int* sel_cnt, const Plan& P,
                        OrbKeyPoint* kps, uint8_t* desc, int* n_out, int max_kp, int nframes, const int* taps,
                        hipStream_t st) {
  const bool small = nframes <= 8;
  const int threads = small ? D_THREADS_SMALL : D_THREADS;
  const int per_wg = (threads / 64) * (small ? D_KPW_SMALL : D_KPW);
  const int nb = (P.sel_total + per_wg - 1) / per_wg;
  const int groups = (nframes + 7) / 8;  // frames are dealt to the 8 XCDs in groups
  const dim3 grid((unsigned)groups * 8u * (unsigned)nb), block(threads);
  const int mode = !taps ? 0 : taps[3] + 2 * (taps[0] + taps[1] + taps[2]) > 256 ? 2 : 1;
  const int t0 = taps ? taps[0] : 0, t1 = taps ? taps[1] : 0, t2 = taps ? taps[2] : 0, t3 = taps ? taps[3] : 0;
#define ORB_OD_LAUNCH(K, M, T)                                                                                        \
  hipLaunchKernelGGL((orient_desc_kernel<K, M, T>), grid, block, 0, st, pyr, blur, sel, sel_cnt, P, kps, desc, n_out, \
                     max_kp, nframes, nb, t0, t1, t2, t3)
  if (small) {
    if (mode == 0) ORB_OD_LAUNCH(D_KPW_SMALL, 0, D_THREADS_SMALL); else if (mode == 1) ORB_OD_LAUNCH(D_KPW_SMALL, 1, D_THREADS_SMALL); else ORB_OD_LAUNCH(D_KPW_SMALL, 2, D_THREADS_SMALL);
  } else {
    if (mode == 0) ORB_OD_LAUNCH(D_KPW, 0, D_THREADS); else if (mode == 1) ORB_OD_LAUNCH(D_KPW, 1, D_THREADS); else ORB_OD_LAUNCH(D_KPW, 2, D_THREADS);
  }
#undef ORB_OD_LAUNCH
}

int fast_use_width() { return F_USEW; }
int fast_max_cells() { return F_MAXCELLS; }
int fast_max_view_h() { return F_MAXH; }
int octree_max_slots() { return 3072; }
int blur_tile_w() { return B_TW; }
int blur_tile_h() { return B_TH; }

}  // namespace orbhip
