// Internal declarations shared by the host side (lt_env_host.cpp) and the HIP kernels (lt_env.hip).
#pragma once

#include <cstdint>

#include "../../include/lt_env.h"
#include "../../include/lt_layout.h"

struct lt_env {
  lt_cfg cfg;
  lt_layout layout;
  lt_dev_args dev_args;  // host image of the device-resident (cfg, layout) block
  void* arena = nullptr;   // caller-owned device memory
  size_t arena_bytes = 0;
  void* ev_start = nullptr;  // hipEvent_t pair for lt_env_step_profiled (created lazily)
  void* ev_stop = nullptr;
  int defer_gate = 0;        // lt_env_defer_gate mode: 0 pass behind every step, 1 the caller's lt_env_gate_update, 2 chained (next step launch)
  mutable int pending_steps = 0;  // steps launched whose common_step_counter bump is outstanding (modes 1, 2)
  mutable int gate_pending = 0;   // a population pass is outstanding
  int rows_bf16 = 0;         // lt_env_set_row_format: the row pointers of lt_env_step_rows / _rollout are bf16 rows
  mutable int test_chain_skew = 0;  // lt_env_defer_gate mode 3 (test hook): the next chained launch announces a wrong step id
};

// implemented in lt_env.hip -------------------------------------------------------------------------
// Every launcher enqueues on `stream` and returns a hipError_t value as int (0 = hipSuccess).
int lt_check_layout(const lt_layout* L);  // 1 if the kernels' compile-time quad-field offsets match this layout
int lt_launch_reset_all(const lt_env* env, void* stream);
int lt_launch_step(const lt_env* env, const float* actions, void* stream);
int lt_launch_step_rows(const lt_env* env, const float* actions, const float* const prev[2], float* const next[2], const float* values,
                        float gamma, float* st_rewards, unsigned char* st_dones, void* stream);
int lt_launch_step_profiled(lt_env* env, const float* actions, const float* const* prev, float* const* next, void* stream, float* ms);
int lt_launch_eval_terms(const lt_env* env, void* stream);
void lt_release_events(lt_env* env);
int lt_launch_curriculum(const lt_env* env, const float* records, void* stream);
int lt_launch_gate_decide(const lt_env* env, int bump_counter, void* stream);  // the one-wave global half of the curriculum pass
int lt_launch_set_command_ranges(const lt_env* env, const float ranges[6], int zero_steps, float rel_standing, void* stream);
int lt_launch_curriculum_apply_global(const lt_env* env, const float* ring_sums, int nsteps, long long n_total, void* stream);
int lt_launch_tactile(const lt_env* env, void* stream);  // lt_tactile.hip
int lt_launch_check(const lt_env* env, void* stream, long long* count);
const char* lt_hip_error_string(int err);

void lt_set_error(const char* msg);
