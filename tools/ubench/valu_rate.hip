// Micro-benchmark: issue rate of a few integer VALU instructions on gfx950.
// Each thread runs N iterations of 8 independent chains of one instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned* out, unsigned seed, int iters) {
  unsigned a[8];
  for (int i = 0; i < 8; i++) a[i] = seed * (threadIdx.x + 1) + i;
  unsigned b = seed ^ 0x9e3779b9u;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (OP == 0) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == 1) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
        if (OP == 2) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
        if (OP == 3) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == 4) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b));
        if (OP == 5) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == 6) asm volatile("v_dot4_u32_u8 %0, %1, %1, %0" : "+v"(a[i]) : "v"(b));
        if (OP == 7) asm volatile("v_sad_u8 %0, %1, %1, %0" : "+v"(a[i]) : "v"(b));
        if (OP == 8) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
        if (OP == 9) asm volatile("v_mad_u32_u24 %0, %1, %1, %0" : "+v"(a[i]) : "v"(b));
        if (OP == 10) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == 11) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(a[i]) : "v"(b));
        if (OP == 12) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == 13) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == 14) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      }
    }
  }
  unsigned s = 0;
  for (int i = 0; i < 8; i++) s += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// 64-bit operands and conversions
template <int OP>
__global__ __launch_bounds__(256) void k64(unsigned* out, unsigned seed, int iters) {
  double a[8];
  for (int i = 0; i < 8; i++) a[i] = 1.0 + 1e-3 * (double)(seed % 7 + threadIdx.x + i);
  double b = 1.0 + 1e-9 * (double)(seed % 5);
  float fa[8];
  for (int i = 0; i < 8; i++) fa[i] = 1.0f + 1e-3f * (float)(threadIdx.x + i);
  float fb = 1.0f + 1e-6f * (float)(seed % 5);
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (OP == 0) asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(a[i]) : "v"(b));
        if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == 3) asm volatile("v_rndne_f64 %0, %0" : "+v"(a[i]));
        if (OP == 4) asm volatile("v_cvt_f32_f64 %0, %1" : "+v"(fa[i]) : "v"(a[i]));
        if (OP == 5) asm volatile("v_cvt_f64_f32 %0, %1" : "+v"(a[i]) : "v"(fa[i]));
        if (OP == 6) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(fa[i]));
        if (OP == 7) asm volatile("v_rcp_f32 %0, %0" : "+v"(fa[i]));
        if (OP == 8) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == 9) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(fa[i]) : "v"(fb));
        if (OP == 10) asm volatile("v_rndne_f32 %0, %0" : "+v"(fa[i]));
        if (OP == 11) asm volatile("v_add_u32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(fa[i]) : "v"(fb));
        if (OP == 12) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(fa[i]) : "v"(fb));
        if (OP == 13) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(fa[i]));
      }
    }
  }
  double s = 0; float fs = 0;
  for (int i = 0; i < 8; i++) { s += a[i]; fs += fa[i]; }
  out[blockIdx.x * 256 + threadIdx.x] = (unsigned)(long long)s + (unsigned)fs;
}

template <int OP, bool WIDE = false>
int run(const char* name, unsigned* d) {
  const int iters = 6000, blocks = 256 * 8;  // 8 blocks of 4 waves per CU = 8 waves per SIMD
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  auto kern = WIDE ? k64<OP> : k<OP>;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 12345u, 10);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 12345u, iters);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double laneops = (double)blocks * 256 * iters * 64;
  printf("%-16s %8.3f ms  %.3e lane-ops/s  (%.1f%% of 256CU*128lanes*2.4GHz)\n", name, ms, laneops / (ms * 1e-3),
         100.0 * laneops / (ms * 1e-3) / (256.0 * 128 * 2.4e9));
  return 0;
}

int main() {
  unsigned* d;
  CHECK(hipMalloc(&d, 256 * 8 * 256 * 4));
  run<9>("warm-up mad24", d); run<1>("v_bcnt_u32_b32", d); run<0>("v_xor_b32", d); run<10>("v_add_u32", d); run<11>("v_fma_f32", d);
  run<12>("v_and_b32", d); run<13>("v_max_i32", d); run<14>("v_sub_u32", d);
  run<0>("v_xor_b32", d); run<1>("v_bcnt_u32_b32", d); run<2>("v_add3_u32", d); run<3>("v_min_u32", d);
  run<4>("v_lshl_or_b32", d); run<5>("v_pk_min_i16", d); run<6>("v_dot4_u32_u8", d); run<7>("v_sad_u8", d);
  run<8>("v_perm_b32", d); run<9>("v_mad_u32_u24", d);
  run<0, true>("v_fma_f64", d); run<1, true>("v_mul_f64", d); run<2, true>("v_add_f64", d); run<3, true>("v_rndne_f64", d);
  run<4, true>("v_cvt_f32_f64", d); run<5, true>("v_cvt_f64_f32", d); run<6, true>("v_cvt_i32_f32", d);
  run<7, true>("v_rcp_f32", d); run<8, true>("v_pk_mul_f32", d); run<9, true>("v_mul_lo_u32", d);
  run<10, true>("v_rndne_f32", d); run<11, true>("v_add_u32_dpp", d); run<12, true>("v_mul_f32", d);
  run<13, true>("v_cvt_f32_i32", d);
  return 0;
}
