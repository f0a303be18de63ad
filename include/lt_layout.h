/*
 * lt_layout.h - HBM layout of the environment state arena (plumbing shared by the HIP library and the
 * CPU oracle so that a test can hand the same bytes to both).
 *
 * Everything per-env lives in "quad arrays": float/int32 [N][4], element (env e, lane l) at e*4+l.
 * With 4 lanes per env (one per leg: FR, FL, RR, RL) and 16 envs per wave64, lane (e,l) touches element
 * e*4+l, so every wave-wide load/store of a quad array is one fully coalesced 256-byte access.
 *   - per-leg data (joint pos/vel of link type k, foot timers, ...): lane = leg.
 *   - per-env vectors (root pose, command, ...): component c lives in quad array c/4, lane c%4, and is
 *     broadcast inside the quad with a DPP quad_perm after the load.
 * Observation rows and the trainer-facing outputs are plain row-major arrays ([N][obs_dim], [N]).
 */
#ifndef LT_LAYOUT_H
#define LT_LAYOUT_H

#include "lt_env.h"

#ifdef __cplusplus
extern "C" {
#endif

/* number of quad arrays per quad field */
static inline int lt_field_quads(int f) {
  switch (f) {
    case LT_F_JOINT_POS: case LT_F_JOINT_VEL: case LT_F_JOINT_ACC: case LT_F_APPLIED_TORQUE:
    case LT_F_ACT_RAW: case LT_F_ACT_PREV_RAW: case LT_F_ACT_PREV_PREV_RAW:
    case LT_F_FOOT_POS_W: case LT_F_FOOT_VEL_W:
      return 3;
    case LT_F_FORCE_HIST: return 12;
    case LT_F_EPISODE_SUMS: case LT_F_LAST_EPISODE_SUMS: case LT_F_REWARD_TERMS: return 7;
    case LT_F_CURRICULUM: case LT_F_PLATE_SAMPLES: return 3;
    default: return 1;
  }
}

/* Device-resident copy of (lt_cfg, lt_layout) at the arena tail: the kernels read their configuration from HBM/L2
 * through one pointer instead of a ~1.7 KB kernarg block (kernargs live in host-visible memory: every scalar-cache
 * miss on them is a fabric round trip - measured 20 us of s_waitcnt per launch). */
#define LT_DEV_ARGS_BYTES 4096

/* Per-wave partial results of the curriculum / population-gate reduction at the tail of the step kernel (one slot per
 * 16-env wave): (any non-zero command, any reset, lin: not-all-reset, sum ep_len, sum reward, ang: the same three). */
#define LT_PARTIAL_FLOATS 8

typedef struct lt_layout {
  int64_t n;                         /* envs (padded to a multiple of 16 for the arena) */
  int64_t npad;
  int32_t obs_dim;
  int64_t quad_off[LT_NUM_QUAD_FIELDS]; /* byte offsets */
  int64_t off_ep_len, off_obs_policy, off_obs_critic, off_reward, off_dones, off_terminated, off_time_out,
      off_term_bits, off_cmd_params, off_counters, off_gate_ring, off_partials, off_obs_tactile, off_obj_sizes, off_dev_args;
  int64_t total_bytes;
  int32_t tactile, reserved;
} lt_layout;

static inline int64_t lt_align256(int64_t x) { return (x + 255) & ~(int64_t)255; }

/* `tactile`: cfg.tactile_enabled - the plate-sample field and the tactile rows take no space otherwise */
static inline void lt_layout_init(lt_layout* L, int64_t num_envs, int32_t obs_dim, int32_t tactile) {
  int64_t off = 0;
  L->n = num_envs;
  L->npad = (num_envs + 15) / 16 * 16;
  L->obs_dim = obs_dim;
  for (int f = 0; f < LT_NUM_QUAD_FIELDS; ++f) {
    L->quad_off[f] = off;
    off = lt_align256(off + (int64_t)((f == LT_F_PLATE_SAMPLES && !tactile) ? 0 : lt_field_quads(f)) * L->npad * 4 * 4);
  }
  L->off_ep_len = off;      off = lt_align256(off + L->npad * 8);
  L->off_obs_policy = off;  off = lt_align256(off + L->npad * (int64_t)obs_dim * 4);
  L->off_obs_critic = off;  off = lt_align256(off + L->npad * (int64_t)obs_dim * 4);
  L->off_reward = off;      off = lt_align256(off + L->npad * 4);
  L->off_dones = off;       off = lt_align256(off + L->npad * 8);
  L->off_terminated = off;  off = lt_align256(off + L->npad);
  L->off_time_out = off;    off = lt_align256(off + L->npad);
  L->off_term_bits = off;   off = lt_align256(off + L->npad * 4);
  L->off_cmd_params = off;  off = lt_align256(off + LT_CMD_PARAMS_LEN * 4);
  L->off_counters = off;    off = lt_align256(off + 4 * 8);
  L->off_gate_ring = off;   off = lt_align256(off + LT_GATE_RING * LT_PARTIAL_FLOATS * 4);
  L->off_partials = off;    off = lt_align256(off + 2 * (L->npad / 16) * LT_PARTIAL_FLOATS * 4); /* two slot sets, by step parity */
  L->off_obs_tactile = off; off = lt_align256(off + (tactile ? 3 * L->npad * (int64_t)LT_TACTILE_WIDE_DIM * 4 : 0)); /* tactile | original | processed */
  L->off_obj_sizes = off;   off = lt_align256(off + L->npad * 2 * 4); /* this and what follows survive lt_env_reset_all */
  L->off_dev_args = off;    off = lt_align256(off + LT_DEV_ARGS_BYTES);
  L->total_bytes = off;
  L->tactile = tactile; L->reserved = 0;
}

/* float* of quad array `q` of quad field `f` */
static inline float* lt_quad(void* arena, const lt_layout* L, int f, int q) {
  return (float*)((char*)arena + L->quad_off[f]) + (int64_t)q * L->npad * 4;
}

/* device-resident (cfg, layout) block; 16-byte sized so that the kernels stage it with whole 16-byte loads */
typedef struct __attribute__((aligned(16))) lt_dev_args {
  lt_cfg cfg;
  lt_layout layout;
} lt_dev_args;

#ifdef __cplusplus
}
#endif
#endif /* LT_LAYOUT_H */
