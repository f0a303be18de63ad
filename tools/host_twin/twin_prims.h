// Host emulation of lt_device_prims.h: the four lanes of a quad are four threads in lock step; a DPP quad_perm read is an
// exchange through a shared slot array bracketed by two barriers.  Every lane must execute the same sequence of dpp() calls
// (true of the kernels: quad exchanges never sit under lane-divergent control flow).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

namespace lt {
struct TwinBarrier {  // spinning sense-reversal barrier of the quad's four threads
  std::atomic<int> count{0}, sense{0};
  void arrive_and_wait() {
    const int s = sense.load(std::memory_order_acquire);
    if (count.fetch_add(1, std::memory_order_acq_rel) == 3) { count.store(0, std::memory_order_relaxed); sense.store(s ^ 1, std::memory_order_release); }
    else while (sense.load(std::memory_order_acquire) == s) { }
  }
};
struct TwinQuad {
  TwinBarrier bar;
  float slot[4];
};
inline thread_local TwinQuad* t_quad = nullptr;
inline thread_local int t_lane = 0;

template <int CTRL>
inline float dpp(float x) {
  TwinQuad& q = *t_quad;
  q.slot[t_lane] = x;
  q.bar.arrive_and_wait();
  const float y = q.slot[(CTRL >> (2 * t_lane)) & 3];
  q.bar.arrive_and_wait();
  return y;
}
template <int CTRL>
inline int dppi(int x) { return __float_as_int(dpp<CTRL>(__int_as_float(x))); }
inline float fsqrt(float x) { return std::sqrt(x); }
inline float frsqrt(float x) { return 1.0f / std::sqrt(x); }
inline float fsin(float x) { return std::sin(x); }
inline float fcos(float x) { return std::cos(x); }
inline float opaque(float x) { return x; }
}  // namespace lt
