// Stand-alone check of hipMemcpy2D device -> pageable host memory after work on a non-blocking
// stream, the call sequence that once stood behind orb_get_pyramid_level:
//   kernels on a hipStreamNonBlocking stream fill a pitched device buffer
//   hipDeviceSynchronize()
//   hipMemcpy2D(dst (pageable, its own stride), src (pitched), width, height, DeviceToHost)
// and every byte of dst is compared with what the kernel wrote.  Run it plain and under
// `rocprofv3 --kernel-trace -- ./memcpy2d_repro`: a copy that comes back incomplete shows up as
// mismatching rows.  Shapes are those of the pyramid levels of a 376 x 1241 frame (source offset
// inside a larger allocation, odd widths, pitches that are multiples of 64).
// As a program it runs on the HIP runtime under /opt/rocm; built as a library and called from
// tools/memcpy2d_repro.py it runs on the HIP runtime that PyTorch-ROCm bundles, which is what the
// library's Python tests run on (an older release than the profiler's own libraries).
//   build: hipcc --offload-arch=gfx950 -O2 -o memcpy2d_repro memcpy2d_repro.hip
//          hipcc --offload-arch=gfx950 -O2 -shared -fPIC -DMEMCPY2D_REPRO_LIBRARY -o libmemcpy2d_repro.so memcpy2d_repro.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__global__ void fill(uint8_t* p, size_t n, uint32_t seed) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = (uint8_t)((i * 2654435761u + seed) >> 13);
}

extern "C" __attribute__((visibility("default"))) int memcpy2d_repro_run(int rounds) {
  const int W[8] = {1241, 1034, 862, 718, 598, 499, 416, 346}, H[8] = {376, 313, 261, 218, 181, 151, 126, 105};
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  size_t off[8], total = 0;
  int pitch[8];
  for (int l = 0; l < 8; l++) {
    pitch[l] = (W[l] + 32 + 19 + 63) / 64 * 64;
    off[l] = total;
    total += (size_t)pitch[l] * (H[l] + 38);
    total = (total + 255) / 256 * 256;
  }
  const int frames = 4;
  uint8_t* d;
  CK(hipMalloc(&d, total * frames));
  long bad_rows = 0, checked = 0, untouched_rows = 0, late_rows = 0;
  for (int it = 0; it < rounds; it++) {
    const uint32_t seed = 1000u + (uint32_t)it;
    hipLaunchKernelGGL(fill, dim3((unsigned)((total * frames + 255) / 256)), dim3(256), 0, st, d, total * frames, seed);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    for (int l = 0; l < 8; l++)
      for (int padded = 0; padded < 2; padded++) {
        const int f = (it + l) % frames;
        const int w = padded ? W[l] + 38 : W[l], h = padded ? H[l] + 38 : H[l];
        const size_t so = (size_t)f * total + off[l] + (padded ? 32 - 19 : (size_t)19 * pitch[l] + 32);
        const size_t dstride = (size_t)w + (size_t)(it % 3) * 5;  // pageable destination, sometimes with a stride of its own
        uint8_t* dst = (uint8_t*)malloc(dstride * h);
        memset(dst, 0xEE, dstride * h);
        CK(hipMemcpy2D(dst, dstride, d + so, pitch[l], w, h, hipMemcpyDeviceToHost));
        auto wrong_rows = [&](long* untouched) {
          long n = 0;
          for (int r = 0; r < h; r++) {
            bool ok = true, fill_only = true;
            for (int c = 0; c < w; c++) {
              const size_t i = so + (size_t)r * pitch[l] + c;
              const uint8_t got = dst[(size_t)r * dstride + c];
              ok = ok && got == (uint8_t)((i * 2654435761u + seed) >> 13);
              fill_only = fill_only && got == 0xEE;
            }
            n += !ok;
            if (untouched && fill_only) ++*untouched;
          }
          return n;
        };
        const long wrong = wrong_rows(&untouched_rows);
        bad_rows += wrong;
        checked += h;
        if (wrong) {
          // does the data arrive later (the call returned early), or never?
          CK(hipDeviceSynchronize());
          late_rows += wrong - wrong_rows(nullptr);
        }
        free(dst);
      }
  }
  printf("hipMemcpy2D device -> pageable host after hipDeviceSynchronize: %ld rows checked, %ld rows wrong "
         "(%ld of them still hold the destination's fill pattern, %ld were right after another hipDeviceSynchronize)\n",
         checked, bad_rows, untouched_rows, late_rows);
  CK(hipFree(d));
  CK(hipStreamDestroy(st));
  return bad_rows ? 1 : 0;
}

#ifndef MEMCPY2D_REPRO_LIBRARY
int main(int argc, char** argv) { return memcpy2d_repro_run(argc > 1 ? atoi(argv[1]) : 20); }
#endif
