/*
 * ORACLE - TEST INFRASTRUCTURE ONLY (see oracle/lt_oracle.c).  Host pointers everywhere; the arena uses
 * the same layout as the device arena (include/lt_layout.h) so a test can hand identical bytes to both.
 */
#ifndef LT_ORACLE_H
#define LT_ORACLE_H

#include <stdint.h>

#include "../include/lt_env.h"
#include "../include/lt_layout.h"

#ifdef __cplusplus
extern "C" {
#endif

#define LT_ORACLE_MODE_STEP 0
#define LT_ORACLE_MODE_TERMS 1
#define LT_ORACLE_MODE_RESET_ALL 2

/* B3-style record of one env, as the reference's terms read it; [leg][link type]. */
typedef struct lt_term_in {
  float root_pos[3], root_quat[4], root_lin[3], root_ang[3];
  float q[4][3], qd[4][3], qdd[4][3], tau[4][3];
  float act_raw[4][3], act_prev[4][3];
  float fhist[3][4][4]; /* [slot][type hip,thigh,calf,foot][leg] : |F| history, newest first */
  float trunk_fhist[3];
  float foot_pos[4][3], foot_vel[4][3];
  float obj_pos[3], obj_quat[4], obj_lin[3], obj_ang[3];
  float obj_timers[4]; /* cur_air, cur_contact, last_air, last_contact */
  float cmd[3];
  int32_t terminated;
} lt_term_in;

/* state + inputs of AdaptiveSymmetricGaitReward(withObject); arrays in the class's foot column order
 * [FR, RL, FL, RR] (reference locotouch/mdp/rewards.py:89-92) */
typedef struct lt_gait_io {
  /* inputs */
  float cur_air[4], cur_con[4], sensor_last_air[4];
  float cmd[3];
  float lin_err, ang_err;
  float obj_xy_yaw[2];
  int32_t any_nonzero_cmd;
  /* state (rewards.py:96-105) */
  float last_step_air[4], last_step_con[4], valid_last_air[4];
  int32_t swinging_in_zero_cmd[4], valid_prev_contact[4];
  float last_cmd[3];
  float step_from_change;
} lt_gait_io;

int lt_oracle_obs_dim(const lt_cfg* cfg);
int64_t lt_oracle_state_bytes(const lt_cfg* cfg);
int lt_oracle_reset_all(const lt_cfg* cfg, void* arena);
int lt_oracle_step(const lt_cfg* cfg, void* arena, const float* actions, int nthreads);
int lt_oracle_eval_terms(const lt_cfg* cfg, void* arena);

/* term-level entry points (golden-vector tests) */
void lt_oracle_process_action(const lt_cfg* cfg, const float a[12], float raw[12], float prev[12], float prev2[12]);
float lt_oracle_gait(const lt_cfg* cfg, lt_gait_io* G, float step_dt);
void lt_oracle_rewards(const lt_cfg* cfg, const lt_term_in* in, lt_gait_io* G, float step_dt, float* terms);
int lt_oracle_terminations(const lt_cfg* cfg, const lt_term_in* in, int64_t ep_len, int64_t max_len);
void lt_oracle_object_state_obs(const lt_cfg* cfg, const lt_term_in* in, const float* noise16, float out[13]);
void lt_oracle_command_update(int64_t ep_len, int zero_steps, const float buf[3], int standing, float cmd[3]);
/* explicit-uniform forms of the RNG-consuming terms (tests/golden/mdp_replay.npz replays recorded uniforms through the
 * reference's own code; the oracle's step path calls these very functions with Philox uniforms) */
void lt_oracle_command_resample_u(const lt_cfg* cfg, const float* P, const float ub[3], const float uv[3], float ustand, float utime,
                                  float cmd[3], float cmd_buf[3], float* standing, float* time_left);
/* K10 tactile (student tasks): taxel forces [221] from the four plate samples; BinaryTactileSignals [442] from taxel forces
 * and explicit uniforms (reference mdp/observations.py:121-126,154-184,307-308) */
void lt_oracle_taxel_forces(const float x[4], const float y[4], const float f[4], float* out);
void lt_oracle_tactile_signals_u(const lt_cfg* cfg, const float* forces, const float* u_thr, const float* u_drop, const float* u_add,
                                 float* out);
/* all TactileSignals classes (observations.py:154-429) with explicit per-taxel uniforms: the four channel maps, and one term in
 * its class's layout (format = LT_TACTILE_*; u8 = {u_thr, u_drop, u_dropf, u_add, u_addf, u_noise, u_small, u_level}) */
void lt_oracle_tactile_channels_u(const lt_cfg* cfg, int original, const float* forces, const float* u_thr, const float* u_drop,
                                  const float* u_dropf, const float* u_add, const float* u_addf, const float* u_noise,
                                  const float* u_small, const float* u_level, float* contact_out, float* norm_out, float* minmax_out,
                                  float* disc_out);
void lt_oracle_tactile_format_u(const lt_cfg* cfg, int format, const float* forces, const float* u8[8], float* out);
void lt_oracle_material_u(const float range_static[2], const float range_dynamic[2], const float range_restitution[2],
                          const float u[3], float out[3]);
void lt_oracle_reset_object_u(const lt_cfg* cfg, const float root_pos[3], const float root_quat[4], const float root_lin[3],
                              const float root_ang[3], float obj_length, const float u_pose[6], float pos[3], float quat[4],
                              float lin[3], float ang[3]);
void lt_oracle_cmd_params_init(const lt_cfg* cfg, float* P);
void lt_oracle_curriculum(const lt_cfg* cfg, float* P, int64_t n, const float* rec, float* trk);
/* arena twin of lt_env_curriculum_update: one curriculum pass on caller-supplied records [n][4] (trackers lag by one pass) */
int lt_oracle_curriculum_update(const lt_cfg* cfg, void* arena, const float* records);
/* multi-rank gate: the decision sequence on population sums (cross-checked against lt_oracle_curriculum in tests), and the
 * arena twin of lt_env_curriculum_apply_global */
void lt_oracle_gate_on_sums(const lt_cfg* cfg, float* P, const float r[8], float inv_n, int allow_lin, int allow_ang, int out[5]);
int lt_oracle_curriculum_apply_global(const lt_cfg* cfg, void* arena, const float* ring_sums, int nsteps, int64_t n_total);
void lt_oracle_obs_push(const int* term_dims, int nterms, int hist, const float* frame, int fill, float* row);

#ifdef __cplusplus
}
#endif
#endif
