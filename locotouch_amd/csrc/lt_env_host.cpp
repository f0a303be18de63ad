// Host side of the C ABI (include/lt_env.h): handle management, arena binding and zero-copy views.
// No device memory is allocated here - the caller owns the arena (torch allocates it in the Python host
// layer, any hipMalloc'd block works) - and nothing in this file synchronises with the device.
#include <cstdio>
#include <cstring>
#include <new>
#include <cstdint>
#include <string>

#include "lt_internal.h"

namespace {
thread_local std::string g_last_error;

bool cfg_valid(const lt_cfg* c) {
  if (!c) return false;
  if (c->num_envs <= 0 || c->num_envs > (1 << 24)) return false;
  if (c->task != LT_TASK_LOCOMOTION && c->task != LT_TASK_TRANSPORT_TEACHER) return false;
  if (c->decimation < 1 || c->decimation > 64 || c->phys_substeps < 1 || c->phys_substeps > 16) return false;
  if (c->obs_history != 6) return false;  // the kernels are specialised for the reference's history length
  if (!(c->sim_dt > 0.f) || c->max_episode_length <= 0) return false;
  if (c->tactile_enabled && (c->task != LT_TASK_TRANSPORT_TEACHER || !(c->tactile_update_period >= c->sim_dt))) return false;
  if (c->tactile_enabled && (c->tactile_format < LT_TACTILE_BINARY || c->tactile_format > LT_TACTILE_ORIGINAL || (c->tactile_aux_groups & ~3) ||
                             !(c->tactile_maximal_force > 0.f) || c->tactile_total_levels < 1 || c->tactile_force_noise < 0.f ||
                             c->tactile_level_noise < 0.f))
    return false;
  return true;
}
}  // namespace

void lt_set_error(const char* msg) { g_last_error = msg ? msg : ""; }

extern "C" {

const char* lt_last_error(void) { return g_last_error.c_str(); }

int lt_env_state_bytes(const lt_cfg* cfg, size_t* bytes) {
  if (!cfg_valid(cfg) || !bytes) { lt_set_error("lt_env_state_bytes: invalid cfg"); return LT_EINVAL; }
  lt_layout L;
  lt_layout_init(&L, cfg->num_envs, lt_cfg_obs_dim(cfg), cfg->tactile_enabled);
  *bytes = (size_t)L.total_bytes;
  return LT_OK;
}

int lt_env_create(const lt_cfg* cfg, lt_env** out) {
  if (!out) return LT_EINVAL;
  *out = nullptr;
  if (!cfg_valid(cfg)) { lt_set_error("lt_env_create: invalid cfg"); return LT_EINVAL; }
  lt_env* e = new (std::nothrow) lt_env();
  if (!e) return LT_ENOMEM;
  e->cfg = *cfg;
  lt_layout_init(&e->layout, cfg->num_envs, lt_cfg_obs_dim(cfg), cfg->tactile_enabled);
  if (!lt_check_layout(&e->layout)) {
    delete e;
    lt_set_error("lt_env_create: lt_layout_init and the kernels' compile-time field offsets disagree (library built from mixed sources)");
    return LT_EINVAL;
  }
  std::memset(&e->dev_args, 0, sizeof(e->dev_args));
  e->dev_args.cfg = e->cfg;
  e->dev_args.layout = e->layout;
  *out = e;
  return LT_OK;
}

int lt_env_destroy(lt_env* env) {
  if (env) lt_release_events(env);
  delete env;
  return LT_OK;
}

int lt_env_bind(lt_env* env, void* device_arena, size_t bytes) {
  if (!env || !device_arena) return LT_EINVAL;
  if (bytes < (size_t)env->layout.total_bytes) { lt_set_error("lt_env_bind: arena too small"); return LT_EFAULT; }
  if (((uintptr_t)device_arena & 255) != 0) { lt_set_error("lt_env_bind: arena must be 256-byte aligned"); return LT_EINVAL; }
  env->arena = device_arena;
  env->arena_bytes = bytes;
  return LT_OK;
}

int lt_env_get_view(lt_env* env, int field, lt_view* v) {
  if (!env || !v) return LT_EINVAL;
  const lt_layout& L = env->layout;
  char* base = (char*)env->arena;  // may be null: offsets are then relative to 0 (layout queries before bind)
  std::memset(v, 0, sizeof(*v));
  if ((field == LT_F_PLATE_SAMPLES || field == LT_F_OBS_TACTILE || field == LT_F_OBS_TACTILE_ORIGINAL || field == LT_F_OBS_TACTILE_PROCESSED) && !L.tactile) {
    lt_set_error("lt_env_get_view: the tactile fields exist only with cfg.tactile_enabled");
    return LT_EINVAL;
  }
  if (field >= 0 && field < LT_NUM_QUAD_FIELDS) {
    const int q = lt_field_quads(field);
    v->ptr = base + L.quad_off[field];
    v->dtype = (field == LT_F_GAIT_FLAGS) ? 3 : 0;
    v->ndim = 3;
    v->shape[0] = L.n; v->shape[1] = q; v->shape[2] = 4;
    v->stride[0] = 4; v->stride[1] = L.npad * 4; v->stride[2] = 1;
    return LT_OK;
  }
  auto plain = [&](int64_t off, int dtype, int64_t d0, int64_t d1) {
    v->ptr = base + off;
    v->dtype = dtype;
    v->ndim = d1 > 0 ? 2 : 1;
    v->shape[0] = d0; v->stride[0] = d1 > 0 ? d1 : 1;
    if (d1 > 0) { v->shape[1] = d1; v->stride[1] = 1; }
  };
  switch (field) {
    case LT_F_EP_LEN: plain(L.off_ep_len, 1, L.n, 0); break;
    case LT_F_OBS_POLICY: plain(L.off_obs_policy, 0, L.n, L.obs_dim); break;
    case LT_F_OBS_CRITIC: plain(L.off_obs_critic, 0, L.n, L.obs_dim); break;
    case LT_F_REWARD: plain(L.off_reward, 0, L.n, 0); break;
    case LT_F_DONES: plain(L.off_dones, 1, L.n, 0); break;
    case LT_F_TERMINATED: plain(L.off_terminated, 2, L.n, 0); break;
    case LT_F_TIME_OUT: plain(L.off_time_out, 2, L.n, 0); break;
    case LT_F_TERM_BITS: plain(L.off_term_bits, 3, L.n, 0); break;
    case LT_F_CMD_PARAMS: plain(L.off_cmd_params, 0, LT_CMD_PARAMS_LEN, 0); break;
    case LT_F_COUNTERS: plain(L.off_counters, 1, 4, 0); break;
    case LT_F_OBJ_SIZES: plain(L.off_obj_sizes, 0, L.n, 2); break;
    case LT_F_GATE_RING: plain(L.off_gate_ring, 0, LT_GATE_RING, LT_PARTIAL_FLOATS); break;
    case LT_F_OBS_TACTILE: plain(L.off_obs_tactile, 0, L.n, lt_cfg_tactile_dim(&env->cfg)); break;
    case LT_F_OBS_TACTILE_ORIGINAL:
    case LT_F_OBS_TACTILE_PROCESSED: {
      const int k = field == LT_F_OBS_TACTILE_ORIGINAL ? 1 : 2;
      if (!(env->cfg.tactile_aux_groups & k)) { lt_set_error("lt_env_get_view: this tactile group is not enabled (cfg.tactile_aux_groups)"); return LT_EINVAL; }
      plain(L.off_obs_tactile + (int64_t)k * L.npad * LT_TACTILE_WIDE_DIM * 4, 0, L.n, LT_TACTILE_WIDE_DIM);
      break;
    }
    case LT_F_OBS_OBJECT_STATE:  // the object-state term block (13 x 6) closes the policy rows of the transport tasks
      if (env->cfg.task != LT_TASK_TRANSPORT_TEACHER) { lt_set_error("lt_env_get_view: no object in this task"); return LT_EINVAL; }
      plain(L.off_obs_policy + (int64_t)(L.obs_dim - 13 * env->cfg.obs_history) * 4, 0, L.n, 13 * env->cfg.obs_history);
      v->stride[0] = L.obs_dim;
      break;
    default: lt_set_error("lt_env_get_view: unknown field"); return LT_EINVAL;
  }
  return LT_OK;
}

static int finish(int hip_err, const char* what) {
  if (hip_err == 0) return LT_OK;
  std::string m = std::string(what) + ": " + lt_hip_error_string(hip_err);
  lt_set_error(m.c_str());
  return LT_EHIP;
}

int lt_env_reset_all(lt_env* env, void* stream) {
  if (!env) return LT_EINVAL;
  if (!env->arena) { lt_set_error("lt_env_reset_all: arena not bound"); return LT_EFAULT; }
  const int rc = finish(lt_launch_reset_all(env, stream), "lt_env_reset_all");
  if (rc != LT_OK || !env->cfg.tactile_enabled) return rc;
  return finish(lt_launch_tactile(env, stream), "lt_env_reset_all (tactile)");
}

int lt_env_curriculum_apply_global(lt_env* env, const float* ring_sums, int nsteps, int64_t n_total, void* stream) {
  if (!env || !ring_sums || nsteps < 1 || nsteps > LT_GATE_RING || n_total < 1) return LT_EINVAL;
  if (!env->arena) { lt_set_error("lt_env_curriculum_apply_global: arena not bound"); return LT_EFAULT; }
  if (!env->cfg.cur_gate_external) {
    lt_set_error("lt_env_curriculum_apply_global: cfg.cur_gate_external is 0 (the step kernel decides by itself)");
    return LT_EINVAL;
  }
  return finish(lt_launch_curriculum_apply_global(env, ring_sums, nsteps, (long long)n_total, stream), "lt_env_curriculum_apply_global");
}

int lt_env_tactile_update(lt_env* env, void* stream) {
  if (!env) return LT_EINVAL;
  if (!env->cfg.tactile_enabled) return LT_OK;
  if (!env->arena) { lt_set_error("lt_env_tactile_update: arena not bound"); return LT_EFAULT; }
  return finish(lt_launch_tactile(env, stream), "lt_env_tactile_update");
}

int lt_env_step(lt_env* env, const float* actions, void* stream) {
  if (!env || !actions) return LT_EINVAL;
  if (!env->arena) { lt_set_error("lt_env_step: arena not bound"); return LT_EFAULT; }
  const int rc = finish(lt_launch_step(env, actions, stream), "lt_env_step");
  if (rc != LT_OK || !env->cfg.tactile_enabled) return rc;
  return finish(lt_launch_tactile(env, stream), "lt_env_step (tactile)");
}

int lt_env_step_rows(lt_env* env, const float* actions, const float* prev_policy, const float* prev_critic, float* next_policy,
                     float* next_critic, void* stream) {
  if (!env || !actions) return LT_EINVAL;
  if (!env->arena) { lt_set_error("lt_env_step_rows: arena not bound"); return LT_EFAULT; }
  if (((uintptr_t)prev_policy | (uintptr_t)prev_critic | (uintptr_t)next_policy | (uintptr_t)next_critic) & 15) {
    lt_set_error("lt_env_step_rows: observation rows must be 16-byte aligned");
    return LT_EINVAL;
  }
  const float* prev[2] = {prev_policy, prev_critic};
  float* next[2] = {next_policy, next_critic};
  return finish(lt_launch_step_rows(env, actions, prev, next, nullptr, 0.f, nullptr, nullptr, stream), "lt_env_step_rows");
}

int lt_env_step_rollout(lt_env* env, const float* actions, const float* prev_policy, const float* prev_critic, float* next_policy,
                        float* next_critic, const float* values, float gamma, float* st_rewards, uint8_t* st_dones, void* stream) {
  if (!env || !actions || !values || !st_rewards || !st_dones) return LT_EINVAL;
  if (!env->arena) { lt_set_error("lt_env_step_rollout: arena not bound"); return LT_EFAULT; }
  if (((uintptr_t)prev_policy | (uintptr_t)prev_critic | (uintptr_t)next_policy | (uintptr_t)next_critic) & 15) {
    lt_set_error("lt_env_step_rollout: observation rows must be 16-byte aligned");
    return LT_EINVAL;
  }
  const float* prev[2] = {prev_policy, prev_critic};
  float* next[2] = {next_policy, next_critic};
  return finish(lt_launch_step_rows(env, actions, prev, next, values, gamma, st_rewards, st_dones, stream), "lt_env_step_rollout");
}

int lt_env_defer_gate(lt_env* env, int mode) {
  if (!env || mode < 0 || mode > 3) return LT_EINVAL;
  if (env->gate_pending && mode == 0) { lt_set_error("lt_env_defer_gate: a population pass is outstanding - call lt_env_gate_update first"); return LT_EINVAL; }
  if (mode == 3) { env->defer_gate = 2; env->test_chain_skew = 1; return LT_OK; }  // test hook (include/lt_env.h)
  env->defer_gate = mode;
  return LT_OK;
}

int lt_env_set_row_format(lt_env* env, int format) {
  if (!env || (format != LT_ROWS_F32 && format != LT_ROWS_BF16)) return LT_EINVAL;
  if (format == LT_ROWS_BF16 && env->cfg.tactile_enabled) { lt_set_error("lt_env_set_row_format: the tactile tasks keep f32 rows"); return LT_EINVAL; }
  env->rows_bf16 = format == LT_ROWS_BF16;
  return LT_OK;
}

int lt_env_check(lt_env* env, void* stream) {
  if (!env) return LT_EINVAL;
  if (!env->arena) { lt_set_error("lt_env_check: arena not bound"); return LT_EFAULT; }
  long long lost = 0;
  const int rc = finish(lt_launch_check(env, stream, &lost), "lt_env_check");
  if (rc != LT_OK) return rc;
  if (lost != 0) {
    lt_set_error("lt_env_check: a chained step launch never saw its population pass announced (lt_env_defer_gate mode 2: host bookkeeping of "
                 "step_offset / decide_first and the device's step counter disagree); the command block of that step may be stale");
    return LT_EHIP;
  }
  return LT_OK;
}

int lt_env_gate_update(lt_env* env, void* stream) {
  if (!env) return LT_EINVAL;
  if (!env->arena) { lt_set_error("lt_env_gate_update: arena not bound"); return LT_EFAULT; }
  if (!env->gate_pending) return LT_OK;
  return finish(lt_launch_gate_decide(env, -1, stream), "lt_env_gate_update");
}

int lt_env_step_profiled(lt_env* env, const float* actions, void* stream, float* step_kernel_ms) {
  if (!env || !actions || !step_kernel_ms) return LT_EINVAL;
  if (!env->arena) { lt_set_error("lt_env_step_profiled: arena not bound"); return LT_EFAULT; }
  return finish(lt_launch_step_profiled(env, actions, nullptr, nullptr, stream, step_kernel_ms), "lt_env_step_profiled");
}

int lt_env_step_rows_profiled(lt_env* env, const float* actions, const float* prev_policy, const float* prev_critic, float* next_policy,
                              float* next_critic, void* stream, float* step_kernel_ms) {
  if (!env || !actions || !step_kernel_ms) return LT_EINVAL;
  if (!env->arena) { lt_set_error("lt_env_step_rows_profiled: arena not bound"); return LT_EFAULT; }
  const float* prev[2] = {prev_policy, prev_critic};
  float* next[2] = {next_policy, next_critic};
  return finish(lt_launch_step_profiled(env, actions, prev, next, stream, step_kernel_ms), "lt_env_step_rows_profiled");
}

int lt_env_eval_terms(lt_env* env, void* stream) {
  if (!env) return LT_EINVAL;
  if (!env->arena) { lt_set_error("lt_env_eval_terms: arena not bound"); return LT_EFAULT; }
  return finish(lt_launch_eval_terms(env, stream), "lt_env_eval_terms");
}

int lt_env_curriculum_update(lt_env* env, const float* records, void* stream) {
  if (!env || !records) return LT_EINVAL;
  if (!env->arena) { lt_set_error("lt_env_curriculum_update: arena not bound"); return LT_EFAULT; }
  return finish(lt_launch_curriculum(env, records, stream), "lt_env_curriculum_update");
}

int lt_env_set_command_ranges(lt_env* env, const float ranges[6], int zero_steps, float rel_standing, void* stream) {
  if (!env || !ranges) return LT_EINVAL;
  if (!env->arena) { lt_set_error("lt_env_set_command_ranges: arena not bound"); return LT_EFAULT; }
  return finish(lt_launch_set_command_ranges(env, ranges, zero_steps, rel_standing, stream), "lt_env_set_command_ranges");
}

const char* lt_env_kernel_name(int which) {
  switch (which) {
    case 0: return "lt_step_kernel";
    case 1: return "lt_post_kernel";
    case 2: return "lt_reset_all_kernel";
    default: return nullptr;
  }
}

}  // extern "C"
