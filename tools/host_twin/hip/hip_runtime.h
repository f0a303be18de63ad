// Host stand-in for <hip/hip_runtime.h> used ONLY by tools/host_twin (CPU build of the kernels' physics source).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __constant__ static const
static inline uint32_t __umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32); }
static inline float __int_as_float(int x) { float f; std::memcpy(&f, &x, 4); return f; }
static inline int __float_as_int(float f) { int x; std::memcpy(&x, &f, 4); return x; }
