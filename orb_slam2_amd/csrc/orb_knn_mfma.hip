// Brute-force Hamming kNN-2 on the matrix cores of gfx950.
//
// With the bits of a database row as 0 / 2 and the bits of a query as 0 / 1 (int8), the dot
// product of the two is 2 * (#bits set in both); started from -(#bits set in the row) instead of
// from zero, the accumulator ends at  acc = 2 * both - row,  and
//   hamming = query + row - 2 * both = (#bits set in the query) - acc,
// an exact integer.  So the 1M x 1M distance table is an int8 GEMM with K = 256, which
// v_mfma_i32_32x32x32_i8 does at 1024 multiply-adds per cycle and SIMD - the popcount form needs
// 16 vector instructions per pair.  (Both operands as -1 / +1 give 256 - 2 * hamming with no
// start values at all; but the matrix cores' clock depends on how many operand bits change from
// one instruction to the next and on how many products are zero: tools/ubench/mfma_rate.hip
// sustains 3703 TOP/s on random -1 / +1 operands, 3928 with one operand as 0 / 1 and 4562 with
// 0 / 2 against 0 / 1; this kernel went from 0.1746 to 0.169 s with the first change.)
// Results are the same integers as the popcount kernel's (orb_match.hip: knn2_kernel),
// including the tie rule (first minimum = lowest database row).
//
// Work decomposition: a wavefront owns KM_QT tiles of 32 queries as the B operand, expanded
// once into registers (KM_QT * 8 K-steps * 16 bytes per lane).  Database rows stream through
// LDS 32 at a time: the workgroup expands the packed 32 x 32 bytes to 32 x 256 int8 (through a
// byte -> 8 bytes table in LDS; three buffers in turn, one barrier per tile) in the fragment order of
// the A operand, every wavefront reads the eight A fragments (one ds_read_b128 each; the first
// four are requested before the next tile is expanded, so that the MFMAs can start right after
// the barrier) and issues 8 * KM_QT MFMAs.  In the
// 32x32 result a lane holds one query (column) and 16 database rows; it keeps the two
// smallest (distance, row) keys of its rows, lanes l and l + 32 (same query, other rows) are
// merged at the end.  Rows arrive in ascending order, so after the first few thousand a tile
// almost never changes any lane's two best: the update is skipped for the whole wavefront
// unless some accumulator exceeds the lane's second-best dot product.
#include "orb_common.h"

namespace orbhip {
namespace {

typedef int km_v4i __attribute__((ext_vector_type(4)));
typedef int km_v16i __attribute__((ext_vector_type(16)));

constexpr int KM_THREADS = 256;        // 4 wavefronts
constexpr int KM_QT = 4;               // query tiles (of 32) per wavefront
constexpr int KM_QPW = 32 * KM_QT;     // queries per wavefront
constexpr int KM_QPB = KM_QPW * (KM_THREADS / 64);
constexpr int KM_EARLY = 4;            // A fragments requested before the expansion of the next tile
constexpr int KM_ROWBITS = 21;         // key = (256 - dot) << 21 | row; 256 - dot <= 512
constexpr int KM_CHUNK = 1 << KM_ROWBITS;

// 16 descriptor bits -> 16 bytes of SET / CLEAR, byte i of dword d = bit 4d + i.
// The multiplication puts bit i of a nibble at bit 2 of byte i, i.e. a v_perm selector byte of
// 4 (first byte of SET) or 0 (first byte of CLEAR); rows past the end of the
// database get selector 0x0c everywhere (constant 0): their dot products are 0 and they are
// skipped by index.
template <uint32_t SET, uint32_t CLEAR>  // four bytes each: what a set and a clear bit become
__device__ __forceinline__ km_v4i expand16(uint32_t bits, bool valid) {
  const uint32_t off = valid ? 0u : 0x0c0c0c0cu;
  km_v4i r;
#pragma unroll
  for (int d = 0; d < 4; d++) {
    const uint32_t x = (bits >> (4 * d)) & 15u;
    const uint32_t sel = ((x * 0x00810204u) & 0x04040404u) | off;
    r[d] = (int)__builtin_amdgcn_perm(SET, CLEAR, sel);
  }
  return r;
}

__global__ __launch_bounds__(KM_THREADS) void knn2_mfma_kernel(const uint16_t* __restrict__ q16, int nq,
                                                              const uint16_t* __restrict__ db16, int ndb, int index_base,
                                                              float ratio, int* __restrict__ out_best,
                                                              int* __restrict__ out_second, int* __restrict__ out_idx,
                                                              uint8_t* __restrict__ out_accept) {
  // A fragments of one 32-row tile: [group g = 2 * kstep + half][row][16 bytes], three buffers
  // used in turn (tile % 3)
  __shared__ __attribute__((aligned(16))) km_v4i s_a[3][16 * 32];
  // 8 descriptor bits -> 8 bytes of 2 / 0; entry 256 (all zero) stands for rows past the end
  __shared__ __attribute__((aligned(8))) uint2 s_lut[257];
  __shared__ int s_pq[KM_THREADS / 64][KM_QT][32];
  // minus the set bits of the rows of a tile: the accumulators' start values.  Four sets in turn: the
  // set of tile i + 2 is cleared during iteration i (last read in iteration i - 2), filled during
  // iteration i + 1 and read in iteration i + 2.
  __shared__ __attribute__((aligned(16))) int s_nrow[4][32];
  // running (best, second, index) of the wave's queries over the chunks (the registers are full)
  __shared__ int s_run[KM_THREADS / 64][KM_QT][3][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int qbase = blockIdx.x * KM_QPB + wave * KM_QPW;

  for (int i = tid; i < 257; i += KM_THREADS) {
    const km_v4i e = expand16<0x02020202u, 0u>((uint32_t)i, i < 256);
    s_lut[i] = make_uint2((uint32_t)e[0], (uint32_t)e[1]);
  }
  __syncthreads();
  // set bits of the wave's queries (used when a chunk's keys are turned into distances; kept in LDS,
  // the registers are full)
#pragma unroll
  for (int t = 0; t < KM_QT; t++) {
    const uint4* qp = reinterpret_cast<const uint4*>(q16 + (size_t)min(qbase + 32 * t + r, nq - 1) * 16);
    const uint4 u = qp[h];
    const int half = __popc(u.x) + __popc(u.y) + __popc(u.z) + __popc(u.w);
    const int pq = half + __shfl_xor(half, 32, 64);
    if (h == 0) s_pq[wave][t][r] = pq;
  }
  // B operand: the wave's queries, expanded once.  Lane (r, h) holds for tile t and K-step s
  // the 16 bits 32 s + 16 h .. + 15 of query qbase + 32 t + r - the same bit -> k order as the
  // A fragments below, which is all the dot product needs.
  km_v4i b[KM_QT][8];
#pragma unroll
  for (int t = 0; t < KM_QT; t++) {
    const int qi = min(qbase + 32 * t + r, nq - 1);
#pragma unroll
    for (int s = 0; s < 8; s++) b[t][s] = expand16<0x01010101u, 0u>(q16[(size_t)qi * 16 + 2 * s + h], true);
  }
#pragma unroll
  for (int t = 0; t < KM_QT; t++) { s_run[wave][t][0][lane] = 256; s_run[wave][t][1][lane] = 256; s_run[wave][t][2][lane] = -1; }

  // the two work items (row, group) of this thread in the expansion of a tile
  const int er = tid & 31, eg0 = tid >> 5;  // groups eg0 and eg0 + 8

  for (int cbase = 0; cbase < ndb; cbase += KM_CHUNK) {
    const int cn = min(KM_CHUNK, ndb - cbase);
    const int ntiles = (cn + 31) >> 5;
    uint32_t k1[KM_QT], k2[KM_QT];
    int thr[KM_QT];  // an accumulator must exceed this to change (k1, k2)
#pragma unroll
    for (int t = 0; t < KM_QT; t++) { k1[t] = 0xFFFFFFFFu; k2[t] = 0xFFFFFFFFu; thr[t] = -100000; }
    uint32_t pk0 = 0, pk1 = 0;
    auto fetch = [&](int tile) {
      const int row = tile * 32 + er;
      const size_t o = (size_t)(cbase + (row < cn ? row : 0)) * 16;
      pk0 = db16[o + eg0];
      pk1 = db16[o + eg0 + 8];
    };
    // the packed rows in (pk0, pk1) go to buffer `tile % 3` as A fragments
    auto expand = [&](int tile, int slot) {
      const bool v = tile * 32 + er < cn;
      km_v4i* buf = s_a[slot];
      const uint2 a0 = s_lut[v ? (pk0 & 255u) : 256u], a1 = s_lut[v ? (pk0 >> 8) : 256u];
      const uint2 c0 = s_lut[v ? (pk1 & 255u) : 256u], c1 = s_lut[v ? (pk1 >> 8) : 256u];
      buf[eg0 * 32 + er] = km_v4i{(int)a0.x, (int)a0.y, (int)a1.x, (int)a1.y};
      buf[(eg0 + 8) * 32 + er] = km_v4i{(int)c0.x, (int)c0.y, (int)c1.x, (int)c1.y};
      // this thread's 32 bits of the row, counted into the row's start value
      if (v) atomicSub(&s_nrow[tile & 3][er], __popc(pk0 & 0xFFFFu) + __popc(pk1 & 0xFFFFu));
      // the set after the next one is cleared for its turn
      if (tid < 32) s_nrow[(tile + 1) & 3][tid] = 0;
    };
    // In iteration i the first fragments of tile i (complete since the barrier of iteration
    // i - 1) are requested, tile i + 1 is expanded into the next buffer while they travel, and
    // the remaining fragments of tile i are read after the barrier.  A wavefront that runs ahead
    // into iteration i + 1 writes tile i + 2, i.e. buffer (i + 2) % 3, which no read of tile i or
    // i + 1 touches; it cannot get further ahead than that, because the barrier of iteration
    // i + 1 waits for every wavefront to have finished iteration i.
    if (tid < 64) s_nrow[tid >> 5][tid & 31] = 0;  // the sets of tiles 0 and 1
    fetch(0);
    __syncthreads();
    expand(0, 0);
    if (ntiles > 1) fetch(1);
    __syncthreads();
    int slot = 0;  // tile % 3
    for (int tile = 0; tile < ntiles; tile++) {
      const km_v4i* buf = s_a[slot];
      slot = slot == 2 ? 0 : slot + 1;
      km_v4i a[8];  // the first KM_EARLY fragments before the barrier, the others behind the first MFMAs
#pragma unroll
      for (int s = 0; s < KM_EARLY; s++) a[s] = buf[(2 * s + h) * 32 + r];
      // start values (complete since the barrier of the iteration before, like the first fragments):
      // accumulator entry j belongs to row (j & 3) + 8 * (j >> 2) + 4 * h of the tile
      km_v16i start;
      {
        const km_v4i* nr = reinterpret_cast<const km_v4i*>(&s_nrow[tile & 3][4 * h]);
        const km_v4i n0 = nr[0], n1 = nr[2], n2 = nr[4], n3 = nr[6];
        start = km_v16i{n0[0], n0[1], n0[2], n0[3], n1[0], n1[1], n1[2], n1[3],
                        n2[0], n2[1], n2[2], n2[3], n3[0], n3[1], n3[2], n3[3]};
      }
      if (tile + 1 < ntiles) {
        expand(tile + 1, slot);
        if (tile + 2 < ntiles) fetch(tile + 2);  // in flight during the MFMAs
      }
      __syncthreads();
      {
#pragma unroll
        for (int s = KM_EARLY; s < 8; s++) a[s] = buf[(2 * s + h) * 32 + r];
        km_v16i acc[KM_QT];
#pragma unroll
        for (int s = 0; s < 8; s++) {
#pragma unroll
          for (int t = 0; t < KM_QT; t++)
            acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s], b[t][s], s == 0 ? start : acc[t], 0, 0, 0);
        }
        // acc[t][j] = dot(query of this lane in tile t, database row row0 + (j & 3) + 8 * (j >> 2))
        const int row0 = tile * 32 + 4 * h;
#pragma unroll
        for (int t = 0; t < KM_QT; t++) {
          int m = acc[t][0];
#pragma unroll
          for (int j = 1; j < 16; j++) m = max(m, acc[t][j]);
          if (__builtin_amdgcn_ballot_w64(m > thr[t]) != 0) {
            int rbase = row0;
            asm volatile("" : "+v"(rbase));  // keeps the 16 row indices out of the registers of the common path
#pragma unroll
            for (int j = 0; j < 16; j++) {
              const int row = rbase + (j & 3) + 8 * (j >> 2);
              const uint32_t key = row < cn ? ((uint32_t)(256 - acc[t][j]) << KM_ROWBITS) | (uint32_t)row : 0xFFFFFFFFu;
              // two smallest keys: k1 <= k2 always, so the new second is the median of the three
              const uint32_t lo = min(k1[t], key);
              k2[t] = max(min(k2[t], key), min(max(k1[t], key), k2[t]));
              k1[t] = lo;
            }
            thr[t] = k2[t] == 0xFFFFFFFFu ? -100000 : 256 - (int)(k2[t] >> KM_ROWBITS);
          }
        }
      }
    }
    __syncthreads();  // the next chunk starts writing buffer 0 again, which the last tiles may still be read from
    // lanes l and l + 32 hold the same queries over different rows
#pragma unroll
    for (int t = 0; t < KM_QT; t++) {
      const uint32_t o1 = (uint32_t)__shfl_xor((int)k1[t], 32, 64), o2 = (uint32_t)__shfl_xor((int)k2[t], 32, 64);
      const uint32_t m1 = min(k1[t], o1);
      const uint32_t m2 = min(max(k1[t], o1), min(k2[t], o2));
      // fold the chunk into the running (best, second, index): an earlier chunk keeps ties
      // keys hold 256 - acc; distance = (set bits of the query) - acc
      const int pq = s_pq[wave][t][r];
      const int cb = m1 == 0xFFFFFFFFu ? 256 : (int)(m1 >> KM_ROWBITS) - (256 - pq);
      const int cs = m2 == 0xFFFFFFFFu ? 256 : (int)(m2 >> KM_ROWBITS) - (256 - pq);
      const int ci = (int)(m1 & (KM_CHUNK - 1)) + cbase;
      int bd = s_run[wave][t][0][lane], sd = s_run[wave][t][1][lane], bi = s_run[wave][t][2][lane];
      if (cb < bd) { sd = min(bd, cs); bd = cb; bi = ci; }
      else sd = min(sd, cb);
      s_run[wave][t][0][lane] = bd; s_run[wave][t][1][lane] = sd; s_run[wave][t][2][lane] = bi;
    }
  }
  if (h == 0) {
#pragma unroll
    for (int t = 0; t < KM_QT; t++) {
      const int qi = qbase + 32 * t + r;
      if (qi < nq) {
        const int bd = s_run[wave][t][0][lane], sd = s_run[wave][t][1][lane], bi = s_run[wave][t][2][lane];
        out_best[qi] = bd;
        out_second[qi] = sd;
        out_idx[qi] = bi < 0 ? -1 : bi + index_base;
        if (out_accept) out_accept[qi] = (uint8_t)(bi >= 0 && (float)bd < __fmul_rn(ratio, (float)sd));
      }
    }
  }
}

}  // namespace

hipError_t launch_knn2_mfma(const uint8_t* d_q, int n_q, const uint8_t* d_db, int n_db, int index_base, float ratio,
                            int* d_best, int* d_second, int* d_idx, uint8_t* d_accept, hipStream_t st) {
  hipLaunchKernelGGL(knn2_mfma_kernel, dim3((n_q + KM_QPB - 1) / KM_QPB), dim3(KM_THREADS), 0, st,
                     reinterpret_cast<const uint16_t*>(d_q), n_q, reinterpret_cast<const uint16_t*>(d_db), n_db,
                     index_base, ratio, d_best, d_second, d_idx, d_accept);
  return hipGetLastError();
}
int knn2_mfma_queries_per_block() { return KM_QPB; }

}  // namespace orbhip
