// Diagnostic: achievable f32-input MFMA rate on this box (one wave per SIMD, 8 independent accumulators, no memory).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  f32x4 acc[8];
  for (int t = 0; t < 8; ++t) acc[t] = f32x4{0, 0, 0, 0};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a + t, b + j, acc[t], 0, 0, 0);
  }
  float s = 0;
  for (int t = 0; t < 8; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks : {256, 512, 1024}) {
    const int iters = 2000;
    k<<<blocks, 256>>>(out, 10, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<<<blocks, 256>>>(out, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * iters * 32 * 2048.0;
    printf("blocks %d: %.3f ms  %.1f TFLOP/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz, 1 wave/SIMD equiv)\n", blocks, ms, flop / ms / 1e9,
           ms * 1e-3 * 2.4e9 / (iters * 32.0 * (blocks / 256.0)));
  }
  for (int iters : {22, 44, 88, 352}) {  // short kernels, launched back to back (704 MFMAs per wave = layer 0 of the actor)
    for (int i = 0; i < 20; ++i) k<<<256, 256>>>(out, iters, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 200; ++i) k<<<256, 256>>>(out, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("iters %d (%d MFMAs/wave): %.2f us per launch; MFMA-bound %.2f us at 2.4 GHz\n", iters, iters * 32, ms * 1e3 / 200, iters * 32 * 32 / 2400.0);
  }
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("CUs %d clock %d kHz\n", p.multiProcessorCount, p.clockRate);
  return 0;
}
