// Standalone (no torch) driver of the C ABI: isolates runtime/toolchain issues from kernel issues.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../include/lt_env.h"
int main(int argc, char** argv) {
  int n = argc > 1 ? atoi(argv[1]) : 64;
  lt_cfg cfg;
  lt_cfg_default(LT_TASK_TRANSPORT_TEACHER, &cfg);
  cfg.num_envs = n;
  lt_env* env = nullptr;
  printf("create %d\n", lt_env_create(&cfg, &env));
  size_t bytes = 0;
  lt_env_state_bytes(&cfg, &bytes);
  void* arena = nullptr;
  printf("malloc %d (%zu bytes)\n", (int)hipMalloc(&arena, bytes), bytes);
  printf("bind %d\n", lt_env_bind(env, arena, bytes));
  printf("reset %d %s\n", lt_env_reset_all(env, nullptr), lt_last_error());
  printf("sync %d\n", (int)hipDeviceSynchronize());
  float* act = nullptr;
  hipMalloc(&act, sizeof(float) * 12 * n);
  hipMemset(act, 0, sizeof(float) * 12 * n);
  for (int t = 0; t < 5; ++t) printf("step %d -> %d\n", t, lt_env_step(env, act, nullptr));
  printf("sync %d %s\n", (int)hipDeviceSynchronize(), hipGetErrorString(hipGetLastError()));
  lt_view v;
  lt_env_get_view(env, LT_F_REWARD, &v);
  std::vector<float> r(n);
  hipMemcpy(r.data(), v.ptr, sizeof(float) * n, hipMemcpyDeviceToHost);
  printf("reward[0..3] %f %f %f %f\n", r[0], r[1], r[2], r[3]);
  lt_env_get_view(env, LT_F_OBS_POLICY, &v);
  std::vector<float> o(348);
  hipMemcpy(o.data(), v.ptr, sizeof(float) * 348, hipMemcpyDeviceToHost);
  printf("obs[0][0..8] %f %f %f %f %f %f %f %f %f\n", o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], o[8]);
  return 0;
}
