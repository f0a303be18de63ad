// Diagnostic: what s_memtime counts and what the shader clock is while a long / a short kernel runs (MI355X).
//   hipcc --offload-arch=gfx950 -O2 -o clock_probe clock_probe.hip && ./clock_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void chain(float* out, unsigned long long* t, int n) {
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  unsigned long long c0 = wall_clock64();
  float x = out[threadIdx.x];
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) x = __builtin_fmaf(x, 1.0000001f, 1e-9f);  // dependent chain: one VALU op per issue slot
  }
  unsigned long long c1 = wall_clock64();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  out[threadIdx.x] = x;
  if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = t1 - t0; t[1] = c1 - c0; }
}

int main() {
  float* d; unsigned long long* t; unsigned long long h[2];
  hipMalloc(&d, 64 * 4 * 4096); hipMemset(d, 0, 64 * 4 * 4096); hipMalloc(&t, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int wc = 0; hipDeviceGetAttribute(&wc, hipDeviceAttributeWallClockRate, 0);
  int cr = 0; hipDeviceGetAttribute(&cr, hipDeviceAttributeClockRate, 0);
  printf("wall_clock rate %d kHz, reported shader clock %d kHz\n", wc, cr);
  for (int blocks : {1, 1024}) for (int n : {1000, 100000, 1000000}) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0); hipLaunchKernelGGL(chain, dim3(blocks), dim3(64), 0, 0, d, t, n); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
      const double us = (double)h[1] / (wc / 1000.0);
      printf("blocks %4d n %7d: event %9.1f us | wall_clock %9.1f us | s_memtime ticks %12llu (%.1f MHz) | %.2f cycles-if-2.4GHz per FMA | ns per FMA %.3f\n", blocks, n,
             ms * 1e3, us, h[0], h[0] / us, us * 2400.0 / (16.0 * n), us * 1e3 / (16.0 * n));
    }
  }
  return 0;
}
