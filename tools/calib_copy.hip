// PMC calibration (MI355X_MICROARCH.md §HBM: "calibrate on a known byte count in your own access pattern"):
// a dword-per-lane copy (the access shape of the quad arrays) and a float4-per-lane copy of a known size.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void copy_dword(const float* __restrict__ a, float* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void copy_x4(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
int main() {
  const size_t bytes = 1ull << 30;  // 1 GiB each way: far beyond the 256 MiB Infinity Cache
  float *a, *b;
  (void)hipMalloc(&a, bytes); (void)hipMalloc(&b, bytes);
  (void)hipMemset(a, 1, bytes); (void)hipMemset(b, 0, bytes);
  for (int i = 0; i < 3; ++i) {
    hipLaunchKernelGGL(copy_dword, dim3(4096), dim3(256), 0, 0, a, b, bytes / 4);
    hipLaunchKernelGGL(copy_x4, dim3(4096), dim3(256), 0, 0, (const float4*)a, (float4*)b, bytes / 16);
  }
  (void)hipDeviceSynchronize();
  printf("copied %zu bytes per kernel per direction\n", bytes);
  return 0;
}
