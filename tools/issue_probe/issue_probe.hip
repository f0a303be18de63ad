// Diagnostic: VALU issue cost of ONE wave on a CU (MI355X, gfx950), in s_memtime ticks (shader clocks) per instruction:
// dependent vs independent chains of v_fma_f32 and of the packed v_pk_fma_f32 (2 f32 FMAs per lane per instruction).
//   hipcc --offload-arch=gfx950 -O2 -o issue_probe issue_probe.hip && ./issue_probe
// What the step kernel's physics can gain from (a) interleaving independent chains, (b) packed f32 math.
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f2 __attribute__((ext_vector_type(2)));

#define STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")
#define REP16(S) S S S S S S S S S S S S S S S S

template <int CHAINS>
__global__ void fma_chains(float* out, unsigned long long* t, int n) {
  float x0 = out[threadIdx.x], x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
  const float a = 1.0000001f, b = 1e-9f;
  unsigned long long t0, t1;
  STAMP(t0);
  for (int i = 0; i < n; ++i) {
    if (CHAINS == 1) asm volatile(REP16("v_fma_f32 %0, %0, %1, %2\n\t") : "+v"(x0) : "v"(a), "v"(b));
    if (CHAINS == 2) asm volatile(REP16("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3\n\t") : "+v"(x0), "+v"(x1) : "v"(a), "v"(b));
    if (CHAINS == 4)
      asm volatile(REP16("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5\n\t")
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));
  }
  STAMP(t1);
  out[threadIdx.x] = x0 + x1 + x2 + x3;
  if (threadIdx.x == 0) t[0] = t1 - t0;
}

template <int CHAINS>
__global__ void pk_chains(float* out, unsigned long long* t, int n) {
  f2 x0 = {out[threadIdx.x], 1.f}, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
  const f2 a = {1.0000001f, 1.0000002f}, b = {1e-9f, 2e-9f};
  unsigned long long t0, t1;
  STAMP(t0);
  for (int i = 0; i < n; ++i) {
    if (CHAINS == 1) asm volatile(REP16("v_pk_fma_f32 %0, %0, %1, %2\n\t") : "+v"(x0) : "v"(a), "v"(b));
    if (CHAINS == 2) asm volatile(REP16("v_pk_fma_f32 %0, %0, %2, %3\n\tv_pk_fma_f32 %1, %1, %2, %3\n\t") : "+v"(x0), "+v"(x1) : "v"(a), "v"(b));
    if (CHAINS == 4)
      asm volatile(REP16("v_pk_fma_f32 %0, %0, %4, %5\n\tv_pk_fma_f32 %1, %1, %4, %5\n\tv_pk_fma_f32 %2, %2, %4, %5\n\tv_pk_fma_f32 %3, %3, %4, %5\n\t")
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));
  }
  STAMP(t1);
  const f2 s = x0 + x1 + x2 + x3;
  out[threadIdx.x] = s.x + s.y;
  if (threadIdx.x == 0) t[0] = t1 - t0;
}

template <typename K>
static void run(const char* name, K kernel, int chains, float* d, unsigned long long* t) {
  const int n = 4000;
  unsigned long long h = 0, best = ~0ull;
  for (int rep = 0; rep < 5; ++rep) {
    hipLaunchKernelGGL(kernel, dim3(1), dim3(64), 0, 0, d, t, n);
    hipDeviceSynchronize();
    hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    if (h < best) best = h;
  }
  printf("%-44s %6.2f ticks per instruction\n", name, (double)best / (16.0 * chains * n));
}

int main() {
  float* d; unsigned long long* t;
  hipMalloc(&d, 64 * 4); hipMemset(d, 0, 64 * 4); hipMalloc(&t, 16);
  run("v_fma_f32, 1 dependent chain", fma_chains<1>, 1, d, t);
  run("v_fma_f32, 2 independent chains interleaved", fma_chains<2>, 2, d, t);
  run("v_fma_f32, 4 independent chains interleaved", fma_chains<4>, 4, d, t);
  run("v_pk_fma_f32, 1 dependent chain", pk_chains<1>, 1, d, t);
  run("v_pk_fma_f32, 2 independent chains interleaved", pk_chains<2>, 2, d, t);
  run("v_pk_fma_f32, 4 independent chains interleaved", pk_chains<4>, 4, d, t);
  return 0;
}
