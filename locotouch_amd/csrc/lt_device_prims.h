// Hardware primitives of the env kernels: the only place that names gfx950 intrinsics.  Everything above (lt_device_math.h,
// lt_physics_crba.h) is plain C++ over these, so tools/host_twin can run the very same physics source on the CPU (its own
// lt_device_prims.h emulates the quad exchange with four lock-stepped threads) and compare formulations in seconds.
#pragma once

#include <hip/hip_runtime.h>

namespace lt {

// quad DPP: value of another lane of the caller's quad (quad_perm control word CTRL)
template <int CTRL>
__device__ __forceinline__ float dpp(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int dppi(int x) {
  return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, true);
}
// v_sqrt_f32 / v_rsq_f32 as they are (1 ulp).  `sqrtf` expands to ~16 VALU operations (denormal pre-scaling + a correctly-rounded
// fix-up) - 12 of them sat in every physics substep; nothing on this path is near the denormal range or needs the last bit.
__device__ __forceinline__ float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float frsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
// v_sin_f32 / v_cos_f32: |err| ~1e-6 on |x| < 4.6 rad (joint angles, reset Euler angles)
__device__ __forceinline__ float fsin(float x) { return __sinf(x); }
__device__ __forceinline__ float fcos(float x) { return __cosf(x); }

// a value the optimiser cannot tie to its source: what is recomputed from it is recomputed, not kept live in registers
__device__ __forceinline__ float opaque(float x) { asm volatile("" : "+v"(x)); return x; }

}  // namespace lt
