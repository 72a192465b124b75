// Micro-benchmark: sustained rate of v_mfma_i32_32x32x32_i8 on gfx950, and the clock the chip
// holds while the matrix cores are busy.  Every wavefront issues `iters` rounds of NACC
// independent accumulator chains; 1, 2 or 4 wavefronts per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 -o mfma_rate mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k(int* out, int seed, int iters, long long* clocks) {
  v4i a = {seed, seed + 1, seed + 2, (int)threadIdx.x}, b = {seed * 3, 5, 7, (int)threadIdx.x * 9};
  v16i acc[NACC];
  for (int i = 0; i < NACC; i++)
    for (int j = 0; j < 16; j++) acc[i][j] = 0;
  const long long c0 = __builtin_readcyclecounter();  // s_memtime: constant 100 MHz counter
  const long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++)
#pragma unroll
      for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[i], 0, 0, 0);
  }
  const long long t1 = clock64();
  int s = 0;
  for (int i = 0; i < NACC; i++)
    for (int j = 0; j < 16; j++) s += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clocks[0] = t1 - t0; clocks[1] = __builtin_readcyclecounter() - c0; }
}

// The shape of a kNN-2 tile: 8 K-steps x 4 accumulator chains that start from zero, then vector
// instructions that read the results.  DISTINCT_B 1: 32 different B operands as in the real kernel;
// 2: operands of random +1 / -1 bytes that keep changing (the chip clocks lower on such data).
template <int DISTINCT_B, int READS>
__global__ __launch_bounds__(256) void ktile(int* out, int seed, int iters, long long* clocks) {
  v4i a = {seed, seed + 1, seed + 2, (int)threadIdx.x};
  v4i b[4][8];
  for (int t = 0; t < 4; t++)
    for (int s = 0; s < 8; s++) b[t][s] = v4i{seed * (t + 1), s * 5 + seed, 7 + t, (int)threadIdx.x * 9 + s};
  // 2: A and B of random +1 / -1 bytes; 3: A of random 0 / 1 bytes, B of +1 / -1 (the kNN-2 kernel);
  // 4: A of random 0 / 2 bytes, B of 0 / 1
  constexpr unsigned A_LO = DISTINCT_B == 2 ? 0xFFu : 0x00u, A_HI = DISTINCT_B == 4 ? 0x02u : 0x01u;
  constexpr unsigned B_LO = DISTINCT_B == 4 ? 0x00u : 0xFFu, B_HI = 0x01u;
  constexpr unsigned A_X = (A_LO ^ A_HI) * 0x01010101u;
  if (DISTINCT_B >= 2) {
    unsigned x = seed * 2654435761u + threadIdx.x * 40503u + blockIdx.x * 69069u;
    auto pm = [&](unsigned lo, unsigned hi) {
      unsigned r = 0;
      for (int i = 0; i < 4; i++) { x = x * 1664525u + 1013904223u; r |= ((x >> 16) & 1 ? hi : lo) << (8 * i); }
      return (int)r;
    };
    for (int t = 0; t < 4; t++)
      for (int s = 0; s < 8; s++) b[t][s] = v4i{pm(B_LO, B_HI), pm(B_LO, B_HI), pm(B_LO, B_HI), pm(B_LO, B_HI)};
    a = v4i{pm(A_LO, A_HI), pm(A_LO, A_HI), pm(A_LO, A_HI), pm(A_LO, A_HI)};
  }
  const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  int sum = 0;
  for (int it = 0; it < iters; it++) {
    v16i acc[4];
#pragma unroll
    for (int s = 0; s < 8; s++)
#pragma unroll
      for (int t = 0; t < 4; t++)
        acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, DISTINCT_B ? b[t][s] : b[0][0], s == 0 ? zero : acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 4; t++) {
      if (READS == 1) sum += acc[t][0];
      if (READS == 16) {
        int m = acc[t][0];
#pragma unroll
        for (int j = 1; j < 16; j++) m = max(m, acc[t][j]);
        sum += m;
      }
    }
    if (DISTINCT_B >= 2) { a[0] ^= A_X; a[1] ^= (sum & 1) ? A_X : 0u; a[2] ^= A_X & 0xFF00FF00u; }  // the bytes keep changing
    else a[0] += sum & 1;
  }
  out[blockIdx.x * 256 + threadIdx.x] = sum;
}

template <int DISTINCT_B, int READS>
int run_tile(const char* name, int waves_per_simd, int iters) {
  int* out; long long* clk;
  const int blocks = 256 * waves_per_simd;
  CHECK(hipMalloc(&out, sizeof(int) * 256 * blocks));
  CHECK(hipMalloc(&clk, 16));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((ktile<DISTINCT_B, READS>), dim3(blocks), dim3(256), 0, 0, out, 1, 100, clk);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((ktile<DISTINCT_B, READS>), dim3(blocks), dim3(256), 0, 0, out, 1, iters, clk);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double n = (double)blocks * 4 * iters * 32;
  printf("%-44s %d waves/SIMD: %.3f ms, %.3e MFMA/s, %.1f TOP/s int8\n", name, waves_per_simd, ms, n / (ms * 1e-3),
         2 * n * 32768 / (ms * 1e-3) / 1e12);
  hipFree(out); hipFree(clk);
  return 0;
}

template <int NACC>
int run(const char* name, int waves_per_simd, int iters) {
  int* out; long long* clk;
  const int blocks = 256 * waves_per_simd;  // a 256-thread block = one wavefront per SIMD of a CU
  CHECK(hipMalloc(&out, sizeof(int) * 256 * blocks));
  CHECK(hipMalloc(&clk, 16));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, 1, 100, clk);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, 1, iters, clk);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  long long h[2]; CHECK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
  const double n = (double)blocks * 4 * iters * 8 * NACC;  // MFMAs
  const double macs = n * 32768;
  printf("%-28s %d waves/SIMD: %.3f ms, %.3e MFMA/s, %.1f TOP/s int8, wave clock ticks per MFMA and wave %.1f\n", name,
         waves_per_simd, ms, n / (ms * 1e-3), 2 * macs / (ms * 1e-3) / 1e12, (double)h[0] / ((double)iters * 8 * NACC));
  hipFree(out); hipFree(clk);
  return 0;
}

int main() {
  for (int w = 1; w <= 2; w++) {
    if (run_tile<1, 1>("tile: 32 B operands, 1 result read per chain", w, 20000)) return 1;
    if (run_tile<1, 16>("tile: 32 B operands, max of the 16 results", w, 20000)) return 1;
  }
  if (run_tile<1, 16>("tile: 32 B operands, max of 16, long run", 2, 200000)) return 1;
  if (run_tile<2, 16>("tile: random +1/-1 operands, max of 16, long run", 2, 200000)) return 1;
  if (run_tile<3, 16>("tile: A random 0/1, B random +1/-1, max of 16, long run", 2, 200000)) return 1;
  if (run_tile<4, 16>("tile: A random 0/2, B random 0/1, max of 16, long run", 2, 200000)) return 1;
  if (run<4>("4 accumulators, long run", 2, 240000)) return 1;
  for (int w = 1; w <= 4; w *= 2) {
    if (run<4>("4 independent accumulators", w, 20000)) return 1;
    if (run<2>("2 independent accumulators", w, 40000)) return 1;
    if (run<1>("1 accumulator (dependent)", w, 80000)) return 1;
  }
  return 0;
}
