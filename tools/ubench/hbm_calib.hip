// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the
// ORB kernels use: each kernel streams a buffer of known size once with 1, 4 or 16 bytes
// per lane.  Run under `rocprofv3 --pmc FETCH_SIZE` (and WRITE_SIZE in a second pass) and
// compare the counter with the byte count printed here.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <typename T> __device__ unsigned fold(T v);
template <> __device__ unsigned fold<uint8_t>(uint8_t v) { return v; }
template <> __device__ unsigned fold<uint32_t>(uint32_t v) { return v; }
template <> __device__ unsigned fold<uint4>(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

template <typename T> __global__ __launch_bounds__(256) void read_kernel(const T* __restrict__ src, size_t n, unsigned* __restrict__ out) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += fold<T>(src[i]);
  if (acc == 0x12345678u) out[blockIdx.x] = acc;  // practically never true; keeps the loads alive
}
template <typename T> __device__ T splat(unsigned v);
template <> __device__ uint8_t splat<uint8_t>(unsigned v) { return (uint8_t)v; }
template <> __device__ uint32_t splat<uint32_t>(unsigned v) { return v; }
template <> __device__ uint4 splat<uint4>(unsigned v) { return make_uint4(v, v, v, v); }
template <typename T> __global__ __launch_bounds__(256) void write_kernel(T* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = splat<T>((unsigned)i);
}

int main() {
  const size_t bytes = (size_t)512 << 20;
  void *buf; unsigned* out;
  CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 65536 * 4));
  CK(hipMemset(buf, 1, bytes)); CK(hipDeviceSynchronize());
  const int grid = 8192;
  read_kernel<uint8_t><<<grid, 256>>>((const uint8_t*)buf, bytes / 4, out);   // 128 MiB by bytes
  read_kernel<uint32_t><<<grid, 256>>>((const uint32_t*)buf, bytes / 4, out);
  read_kernel<uint4><<<grid, 256>>>((const uint4*)buf, bytes / 16, out);
  write_kernel<uint8_t><<<grid, 256>>>((uint8_t*)buf, bytes / 4);
  write_kernel<uint32_t><<<grid, 256>>>((uint32_t*)buf, bytes / 4);
  write_kernel<uint4><<<grid, 256>>>((uint4*)buf, bytes / 16);
  CK(hipDeviceSynchronize());
  printf("read u8: %zu B  read u32: %zu B  read uint4: %zu B\n", bytes / 4, bytes, bytes);
  printf("write u8: %zu B  write u32: %zu B  write uint4: %zu B\n", bytes / 4, bytes, bytes);
  return 0;
}
