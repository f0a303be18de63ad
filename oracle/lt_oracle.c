/*
 * ORACLE - TEST INFRASTRUCTURE ONLY.  Never imported, linked or executed by the product path
 * (locotouch_amd/); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * CPU restatement (plain C, one env at a time, fp32 by default) of the reference's hot path: one
 * `ManagerBasedRLEnv.step()` of the Go1 (+ carried cylinder) scene, in the stage order of SURVEY.md §3.3.
 * Each function cites the reference file:line it follows.
 *
 * Pinning status:
 *   - locotouch/mdp terms (rewards incl. the symmetric-gait class, object-state observation, terminations,
 *     action term, zero-command window, velocity curriculum): PINNED against tests/golden/mdp_*.npz, which
 *     tools/gen_golden.py produced by running the reference's own source.
 *   - IsaacLab manager/sensor/actuator semantics and the physics (PhysX in the
 *     reference, an ABA + implicit-penalty contact model here): PARITY UNPINNED - IsaacLab / IsaacSim /
 *     PhysX are absent from the build image and cannot be fetched.  The physics section is the executable
 *     specification that the HIP kernel must match; it is deliberately written differently from the kernel
 *     (generic 13-body tree, dense 6x6 spatial algebra, body loops) so that agreement is a real check.
 */
#include <stdlib.h>

#include "lt_oracle.h"
#include "lt_oracle_math.h"
#include "../include/lt_go1_model.h"

/* ------------------------------------------------------------------------------------------------ */
/* model tables                                                                                      */
/* ------------------------------------------------------------------------------------------------ */
static const float k_link_mass[4][3] = LT_LINK_MASS_INIT;
static const float k_link_com[4][3][3] = LT_LINK_COM_INIT;
static const float k_link_icom[4][3][6] = LT_LINK_ICOM_INIT;
static const float k_joint_off[4][3][3] = LT_JOINT_OFFSET_INIT;
static const int k_joint_axis[3] = LT_JOINT_AXIS_INIT;
static const float k_joint_lo[3] = LT_JOINT_LOWER_INIT;
static const float k_joint_hi[3] = LT_JOINT_UPPER_INIT;
static const float k_joint_default[4][3] = LT_JOINT_DEFAULT_INIT;
static const float k_trunk_com[3] = LT_TRUNK_COM_INIT;
static const float k_trunk_icom[6] = LT_TRUNK_ICOM_INIT;
static const float k_hip_cyl_y[4] = LT_HIP_CYL_Y_INIT;
static const float k_trunk_half[3] = LT_TRUNK_BOX_HALF_INIT;

#define NB 13 /* dynamic bodies: trunk + 4 x (hip, thigh, calf+foot) */
#define BODY(leg, k) (1 + (leg)*3 + (k))

/* contact spheres: per leg 6 (SURVEY.md §2.2 K2 collision primitives, reduced to spheres) */
enum { SP_FOOT = 0, SP_CALF, SP_KNEE, SP_HIP, SP_TRUNK_LO, SP_TRUNK_HI, SP_PER_LEG };
/* sensor body type each sphere reports to: 0 hip, 1 thigh, 2 calf, 3 foot, 4 trunk */
static const int k_sphere_sensor[SP_PER_LEG] = {3, 2, 1, 0, 4, 4};

static void sphere_def(int leg, int s, int* body, real r[3], real* rho) {
  switch (s) {
    case SP_FOOT: *body = BODY(leg, 2); v3_set(r, 0, 0, -0.213f); *rho = LT_FOOT_RADIUS; break;
    case SP_CALF: *body = BODY(leg, 2); v3_set(r, 0, 0, -0.1065f); *rho = 0.012f; break;
    case SP_KNEE: *body = BODY(leg, 1); v3_set(r, 0, 0, -0.213f); *rho = 0.022f; break;
    case SP_HIP: *body = BODY(leg, 0); v3_set(r, 0, k_hip_cyl_y[leg], 0); *rho = LT_HIP_CYL_RADIUS; break;
    case SP_TRUNK_LO:
      *body = 0;
      v3_set(r, (leg < 2 ? 1 : -1) * k_trunk_half[0], ((leg & 1) ? 1 : -1) * k_trunk_half[1], -k_trunk_half[2]);
      *rho = 0;
      break;
    default:
      *body = 0;
      v3_set(r, (leg < 2 ? 1 : -1) * k_trunk_half[0], ((leg & 1) ? 1 : -1) * LT_BACK_HALF_Y, LT_BACK_TOP_Z);
      *rho = 0;
      break;
  }
}

/* ------------------------------------------------------------------------------------------------ */
/* per-env working state (gathered from / scattered to the SoA arena)                                */
/* ------------------------------------------------------------------------------------------------ */
typedef struct {
  real root_pos[3], root_quat[4], root_lin[3], root_ang[3]; /* world */
  real q[4][3], qd[4][3], qdd[4][3], tau[4][3];             /* [leg][link type] */
  real act_raw[4][3], act_prev[4][3], act_prev2[4][3];
  real fhist[3][4][4];   /* [slot][type hip,thigh,calf,foot][leg] */
  real trunk_fhist[3];
  real foot_cur_air[4], foot_cur_con[4], foot_last_air[4], foot_last_con[4];
  real foot_pos[4][3], foot_vel[4][3];
  real foot_mu[4];
  real obj_pos[3], obj_quat[4], obj_lin[3], obj_ang[3];
  real obj_timers[4];    /* cur_air, cur_con, last_air, last_con */
  real obj_radius, obj_length, obj_mass, obj_mu;
  real trunk_mass_add, trunk_mu, trunk_rest, obj_rest;
  real gait_last_air[4], gait_last_con[4], gait_valid_last_air[4]; /* leg order */
  int32_t gait_flags[4]; /* bit0 swinging_in_zero_cmd, bit1 valid_previous_contact */
  real gait_cmd[3], gait_step_from_change;
  real cmd[3], cmd_time_left, cmd_buf[3], cmd_standing;
  real push_robot_left, push_obj_left;
  real m_exy, m_eyaw, m_airvar, last_m[4]; /* command-term metrics (commands.py:392-396) now / at the last reset + its step id */
  real sums[LT_REWARD_SLOTS], last_sums[LT_REWARD_SLOTS];
  real episodes_finished, last_ep_len, last_term_bits;
  real cur_step[4], cur_track[8];
  real terms[LT_REWARD_SLOTS];
  real plate[4][3];      /* tactile tasks: plate samples of the last sensor refresh (x, y in the trunk frame, normal force) */
  int64_t ep_len;
} env_t;

static void gather(env_t* E, void* arena, const lt_layout* L, int64_t e) {
#define Q(f, q, l) lt_quad(arena, L, f, q)[e * 4 + (l)]
  for (int c = 0; c < 3; ++c) {
    E->root_pos[c] = Q(LT_F_ROOT_POS, 0, c);
    E->root_lin[c] = Q(LT_F_ROOT_LIN_VEL_W, 0, c);
    E->root_ang[c] = Q(LT_F_ROOT_ANG_VEL_W, 0, c);
    E->obj_pos[c] = Q(LT_F_OBJ_POS, 0, c);
    E->obj_lin[c] = Q(LT_F_OBJ_LIN_VEL_W, 0, c);
    E->obj_ang[c] = Q(LT_F_OBJ_ANG_VEL_W, 0, c);
    E->gait_cmd[c] = Q(LT_F_GAIT_CMD, 0, c);
    E->cmd[c] = Q(LT_F_CMD, 0, c);
    E->cmd_buf[c] = Q(LT_F_CMD_BUF, 0, c);
    E->trunk_fhist[c] = Q(LT_F_TRUNK_FORCE_HIST, 0, c);
  }
  for (int c = 0; c < 4; ++c) {
    E->root_quat[c] = Q(LT_F_ROOT_QUAT, 0, c);
    E->obj_quat[c] = Q(LT_F_OBJ_QUAT, 0, c);
    E->obj_timers[c] = Q(LT_F_OBJ_TIMERS, 0, c);
    E->cur_step[c] = Q(LT_F_CURRICULUM, 0, c);
    E->cur_track[c] = Q(LT_F_CURRICULUM, 1, c);
    E->cur_track[4 + c] = Q(LT_F_CURRICULUM, 2, c);
  }
  E->gait_step_from_change = Q(LT_F_GAIT_CMD, 0, 3);
  E->cmd_time_left = Q(LT_F_CMD, 0, 3);
  E->cmd_standing = Q(LT_F_CMD_BUF, 0, 3);
  E->m_exy = Q(LT_F_EVENT_TIMERS, 0, 2); E->m_eyaw = Q(LT_F_EVENT_TIMERS, 0, 3); E->m_airvar = Q(LT_F_TRUNK_FORCE_HIST, 0, 3);
  for (int c = 0; c < 4; ++c) E->last_m[c] = Q(LT_F_LAST_CMD_METRICS, 0, c);
  E->push_robot_left = Q(LT_F_EVENT_TIMERS, 0, 0);
  E->push_obj_left = Q(LT_F_EVENT_TIMERS, 0, 1);
  E->obj_radius = Q(LT_F_OBJ_PARAMS, 0, 0); E->obj_length = Q(LT_F_OBJ_PARAMS, 0, 1);
  E->obj_mass = Q(LT_F_OBJ_PARAMS, 0, 2);   E->obj_mu = Q(LT_F_OBJ_PARAMS, 0, 3);
  E->trunk_mass_add = Q(LT_F_ENV_PARAMS, 0, 0); E->trunk_mu = Q(LT_F_ENV_PARAMS, 0, 1);
  E->trunk_rest = Q(LT_F_ENV_PARAMS, 0, 2);     E->obj_rest = Q(LT_F_ENV_PARAMS, 0, 3);
  E->episodes_finished = Q(LT_F_LAST_EPISODE_INFO, 0, 0);
  E->last_ep_len = Q(LT_F_LAST_EPISODE_INFO, 0, 1);
  E->last_term_bits = Q(LT_F_LAST_EPISODE_INFO, 0, 2);
  for (int l = 0; l < 4; ++l) {
    for (int k = 0; k < 3; ++k) {
      E->q[l][k] = Q(LT_F_JOINT_POS, k, l);       E->qd[l][k] = Q(LT_F_JOINT_VEL, k, l);
      E->qdd[l][k] = Q(LT_F_JOINT_ACC, k, l);     E->tau[l][k] = Q(LT_F_APPLIED_TORQUE, k, l);
      E->act_raw[l][k] = Q(LT_F_ACT_RAW, k, l);   E->act_prev[l][k] = Q(LT_F_ACT_PREV_RAW, k, l);
      E->act_prev2[l][k] = Q(LT_F_ACT_PREV_PREV_RAW, k, l);
      E->foot_pos[l][k] = Q(LT_F_FOOT_POS_W, k, l); E->foot_vel[l][k] = Q(LT_F_FOOT_VEL_W, k, l);
    }
    for (int s = 0; s < 3; ++s)
      for (int t = 0; t < 4; ++t) E->fhist[s][t][l] = Q(LT_F_FORCE_HIST, s * 4 + t, l);
    E->foot_cur_air[l] = Q(LT_F_FOOT_CUR_AIR, 0, l);   E->foot_cur_con[l] = Q(LT_F_FOOT_CUR_CONTACT, 0, l);
    E->foot_last_air[l] = Q(LT_F_FOOT_LAST_AIR, 0, l); E->foot_last_con[l] = Q(LT_F_FOOT_LAST_CONTACT, 0, l);
    E->foot_mu[l] = Q(LT_F_FOOT_FRICTION, 0, l);
    E->gait_last_air[l] = Q(LT_F_GAIT_LAST_AIR, 0, l); E->gait_last_con[l] = Q(LT_F_GAIT_LAST_CONTACT, 0, l);
    E->gait_valid_last_air[l] = Q(LT_F_GAIT_VALID_LAST_AIR, 0, l);
    memcpy(&E->gait_flags[l], &Q(LT_F_GAIT_FLAGS, 0, l), 4);
  }
  for (int i = 0; i < LT_REWARD_SLOTS; ++i) {
    E->sums[i] = Q(LT_F_EPISODE_SUMS, i / 4, i % 4);
    E->last_sums[i] = Q(LT_F_LAST_EPISODE_SUMS, i / 4, i % 4);
    E->terms[i] = Q(LT_F_REWARD_TERMS, i / 4, i % 4);
  }
  for (int k = 0; k < 4; ++k)
    for (int c = 0; c < 3; ++c) E->plate[k][c] = L->tactile ? Q(LT_F_PLATE_SAMPLES, c, k) : 0;
  E->ep_len = ((int64_t*)((char*)arena + L->off_ep_len))[e];
}

static void scatter(const env_t* E, void* arena, const lt_layout* L, int64_t e) {
  for (int c = 0; c < 3; ++c) {
    Q(LT_F_ROOT_POS, 0, c) = E->root_pos[c];
    Q(LT_F_ROOT_LIN_VEL_W, 0, c) = E->root_lin[c];
    Q(LT_F_ROOT_ANG_VEL_W, 0, c) = E->root_ang[c];
    Q(LT_F_OBJ_POS, 0, c) = E->obj_pos[c];
    Q(LT_F_OBJ_LIN_VEL_W, 0, c) = E->obj_lin[c];
    Q(LT_F_OBJ_ANG_VEL_W, 0, c) = E->obj_ang[c];
    Q(LT_F_GAIT_CMD, 0, c) = E->gait_cmd[c];
    Q(LT_F_CMD, 0, c) = E->cmd[c];
    Q(LT_F_CMD_BUF, 0, c) = E->cmd_buf[c];
    Q(LT_F_TRUNK_FORCE_HIST, 0, c) = E->trunk_fhist[c];
  }
  for (int c = 0; c < 4; ++c) {
    Q(LT_F_ROOT_QUAT, 0, c) = E->root_quat[c];
    Q(LT_F_OBJ_QUAT, 0, c) = E->obj_quat[c];
    Q(LT_F_OBJ_TIMERS, 0, c) = E->obj_timers[c];
    Q(LT_F_CURRICULUM, 0, c) = E->cur_step[c];
    Q(LT_F_CURRICULUM, 1, c) = E->cur_track[c];
    Q(LT_F_CURRICULUM, 2, c) = E->cur_track[4 + c];
  }
  Q(LT_F_GAIT_CMD, 0, 3) = E->gait_step_from_change;
  Q(LT_F_CMD, 0, 3) = E->cmd_time_left;
  Q(LT_F_CMD_BUF, 0, 3) = E->cmd_standing;
  Q(LT_F_EVENT_TIMERS, 0, 2) = E->m_exy; Q(LT_F_EVENT_TIMERS, 0, 3) = E->m_eyaw; Q(LT_F_TRUNK_FORCE_HIST, 0, 3) = E->m_airvar;
  for (int c = 0; c < 4; ++c) Q(LT_F_LAST_CMD_METRICS, 0, c) = E->last_m[c];
  Q(LT_F_EVENT_TIMERS, 0, 0) = E->push_robot_left;
  Q(LT_F_EVENT_TIMERS, 0, 1) = E->push_obj_left;
  Q(LT_F_OBJ_PARAMS, 0, 0) = E->obj_radius; Q(LT_F_OBJ_PARAMS, 0, 1) = E->obj_length;
  Q(LT_F_OBJ_PARAMS, 0, 2) = E->obj_mass;   Q(LT_F_OBJ_PARAMS, 0, 3) = E->obj_mu;
  Q(LT_F_ENV_PARAMS, 0, 0) = E->trunk_mass_add; Q(LT_F_ENV_PARAMS, 0, 1) = E->trunk_mu;
  Q(LT_F_ENV_PARAMS, 0, 2) = E->trunk_rest;     Q(LT_F_ENV_PARAMS, 0, 3) = E->obj_rest;
  Q(LT_F_LAST_EPISODE_INFO, 0, 0) = E->episodes_finished;
  Q(LT_F_LAST_EPISODE_INFO, 0, 1) = E->last_ep_len;
  Q(LT_F_LAST_EPISODE_INFO, 0, 2) = E->last_term_bits;
  for (int l = 0; l < 4; ++l) {
    for (int k = 0; k < 3; ++k) {
      Q(LT_F_JOINT_POS, k, l) = E->q[l][k];       Q(LT_F_JOINT_VEL, k, l) = E->qd[l][k];
      Q(LT_F_JOINT_ACC, k, l) = E->qdd[l][k];     Q(LT_F_APPLIED_TORQUE, k, l) = E->tau[l][k];
      Q(LT_F_ACT_RAW, k, l) = E->act_raw[l][k];   Q(LT_F_ACT_PREV_RAW, k, l) = E->act_prev[l][k];
      Q(LT_F_ACT_PREV_PREV_RAW, k, l) = E->act_prev2[l][k];
      Q(LT_F_FOOT_POS_W, k, l) = E->foot_pos[l][k]; Q(LT_F_FOOT_VEL_W, k, l) = E->foot_vel[l][k];
    }
    for (int s = 0; s < 3; ++s)
      for (int t = 0; t < 4; ++t) Q(LT_F_FORCE_HIST, s * 4 + t, l) = E->fhist[s][t][l];
    Q(LT_F_FOOT_CUR_AIR, 0, l) = E->foot_cur_air[l];   Q(LT_F_FOOT_CUR_CONTACT, 0, l) = E->foot_cur_con[l];
    Q(LT_F_FOOT_LAST_AIR, 0, l) = E->foot_last_air[l]; Q(LT_F_FOOT_LAST_CONTACT, 0, l) = E->foot_last_con[l];
    Q(LT_F_FOOT_FRICTION, 0, l) = E->foot_mu[l];
    Q(LT_F_GAIT_LAST_AIR, 0, l) = E->gait_last_air[l]; Q(LT_F_GAIT_LAST_CONTACT, 0, l) = E->gait_last_con[l];
    Q(LT_F_GAIT_VALID_LAST_AIR, 0, l) = E->gait_valid_last_air[l];
    memcpy(&Q(LT_F_GAIT_FLAGS, 0, l), &E->gait_flags[l], 4);
  }
  for (int i = 0; i < LT_REWARD_SLOTS; ++i) {
    Q(LT_F_EPISODE_SUMS, i / 4, i % 4) = E->sums[i];
    Q(LT_F_LAST_EPISODE_SUMS, i / 4, i % 4) = E->last_sums[i];
    Q(LT_F_REWARD_TERMS, i / 4, i % 4) = E->terms[i];
  }
  if (L->tactile)
    for (int k = 0; k < 4; ++k)
      for (int c = 0; c < 3; ++c) Q(LT_F_PLATE_SAMPLES, c, k) = E->plate[k][c];
  ((int64_t*)((char*)arena + L->off_ep_len))[e] = E->ep_len;
#undef Q
}

/* ------------------------------------------------------------------------------------------------ */
/* K1: action term + DC-motor PD                                                                     */
/* ------------------------------------------------------------------------------------------------ */
/* reference locotouch/mdp/actions.py:30-44 (shift prev<-raw, clip +-clip, scale) */
void lt_oracle_process_action(const lt_cfg* cfg, const float a[12], float raw[12], float prev[12], float prev2[12]) {
  for (int j = 0; j < 12; ++j) {
    prev2[j] = prev[j];
    prev[j] = raw[j];
    float x = a[j];
    x = x < -cfg->action_clip ? -cfg->action_clip : (x > cfg->action_clip ? cfg->action_clip : x);
    raw[j] = x * cfg->action_scale;
  }
}

/* DCMotor explicit PD with torque-speed clipping: reference assets/go1.py:41-49 + IsaacLab DCMotor [DEP] */
static real dc_motor(const lt_cfg* cfg, real q_des, real q, real qd) {
  real tau = cfg->kp * (q_des - q) + cfg->kd * (0 - qd);
  real hi = cfg->saturation_effort * (1 - qd / cfg->velocity_limit);
  hi = hi < 0 ? 0 : (hi > cfg->effort_limit ? cfg->effort_limit : hi);
  real lo = cfg->saturation_effort * (-1 - qd / cfg->velocity_limit);
  lo = lo < -cfg->effort_limit ? -cfg->effort_limit : (lo > 0 ? 0 : lo);
  return tau < lo ? lo : (tau > hi ? hi : tau);
}

/* ------------------------------------------------------------------------------------------------ */
/* K2: physics - Featherstone ABA (floating base, 12 revolute joints) with implicit penalty contacts  */
/* ------------------------------------------------------------------------------------------------ */
typedef real mat6[36];
typedef real vec6[6];

static void m6_zero(mat6 m) { memset(m, 0, sizeof(mat6)); }
static void m6_mulv(vec6 o, const mat6 m, const vec6 v) {
  vec6 t;
  for (int i = 0; i < 6; ++i) {
    real s = 0;
    for (int j = 0; j < 6; ++j) s += m[i * 6 + j] * v[j];
    t[i] = s;
  }
  memcpy(o, t, sizeof(vec6));
}
static void m6_tmulv(vec6 o, const mat6 m, const vec6 v) {
  vec6 t;
  for (int i = 0; i < 6; ++i) {
    real s = 0;
    for (int j = 0; j < 6; ++j) s += m[j * 6 + i] * v[j];
    t[i] = s;
  }
  memcpy(o, t, sizeof(vec6));
}
/* spatial inertia about the body origin from (m, com, Icom sym6) */
static void spatial_inertia(mat6 I, real m, const real c[3], const real ic[6], real iscale) {
  real C[9], CC[9];
  m3_skew(C, c);
  m3_mul(CC, C, C); /* c~ c~ = c c^T - |c|^2 1 */
  real Ic[9] = {ic[0], ic[1], ic[2], ic[1], ic[3], ic[4], ic[2], ic[4], ic[5]};
  m6_zero(I);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      I[i * 6 + j] = Ic[i * 3 + j] * iscale - m * CC[i * 3 + j];
      I[i * 6 + 3 + j] = m * C[i * 3 + j];
      I[(3 + i) * 6 + j] = -m * C[i * 3 + j]; /* (m c~)^T */
    }
  for (int i = 0; i < 3; ++i) I[(3 + i) * 6 + 3 + i] = m;
}
/* motion transform parent->child: E = R^T (R: child->parent), r = child origin in parent coords */
static void xform(mat6 X, const real R[9], const real r[3]) {
  real E[9], rs[9], Er[9];
  m3_transpose(E, R);
  m3_skew(rs, r);
  m3_mul(Er, E, rs);
  m6_zero(X);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      X[i * 6 + j] = E[i * 3 + j];
      X[(3 + i) * 6 + 3 + j] = E[i * 3 + j];
      X[(3 + i) * 6 + j] = -Er[i * 3 + j];
    }
}
static void rot_axis(real R[9], int axis, real q) {
  real c = (real)cos(q), s = (real)sin(q);
  if (axis == 0) { real t[9] = {1, 0, 0, 0, c, -s, 0, s, c}; memcpy(R, t, sizeof(t)); }
  else { real t[9] = {c, 0, s, 0, 1, 0, -s, 0, c}; memcpy(R, t, sizeof(t)); }
}
/* spatial cross products: v x m (motion), v x* f (force) */
static void crm(vec6 o, const vec6 v, const vec6 m) {
  real a[3], b[3], c[3];
  v3_cross(a, v, m);
  v3_cross(b, v, m + 3);
  v3_cross(c, v + 3, m);
  o[0] = a[0]; o[1] = a[1]; o[2] = a[2];
  o[3] = b[0] + c[0]; o[4] = b[1] + c[1]; o[5] = b[2] + c[2];
}
static void crf(vec6 o, const vec6 v, const vec6 f) {
  real a[3], b[3], c[3];
  v3_cross(a, v, f);
  v3_cross(b, v + 3, f + 3);
  v3_cross(c, v, f + 3);
  o[0] = a[0] + b[0]; o[1] = a[1] + b[1]; o[2] = a[2] + b[2];
  o[3] = c[0]; o[4] = c[1]; o[5] = c[2];
}
/* solve A x = b for symmetric positive definite 6x6 (Cholesky) */
static void spd6_solve(const mat6 A, const vec6 b, vec6 x) {
  real Lm[36];
  memset(Lm, 0, sizeof(Lm));
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j <= i; ++j) {
      real s = A[i * 6 + j];
      for (int k = 0; k < j; ++k) s -= Lm[i * 6 + k] * Lm[j * 6 + k];
      Lm[i * 6 + j] = (i == j) ? (real)sqrt((double)s) : s / Lm[j * 6 + j];
    }
  vec6 y;
  for (int i = 0; i < 6; ++i) {
    real s = b[i];
    for (int k = 0; k < i; ++k) s -= Lm[i * 6 + k] * y[k];
    y[i] = s / Lm[i * 6 + i];
  }
  for (int i = 5; i >= 0; --i) {
    real s = y[i];
    for (int k = i + 1; k < 6; ++k) s -= Lm[k * 6 + i] * x[k];
    x[i] = s / Lm[i * 6 + i];
  }
}

/* accumulate the implicit penalty contact of one point into (IA, fext) of its body.
 * point rc (body coords), world normal = +z of frame `Rn` (ground: identity; plate: trunk rotation),
 * vrel = contact point velocity relative to the other surface, expressed in the normal frame.
 * Law (DESIGN.md "contact model"):  Bn = kn h + cn ramp(d),  f0n = kn d - Bn vn,  active iff f0n > 0,
 *   ct_eff = min(ct, mu f0n / max(|vt|, 1e-6)),  F = F0 - h B (a_point),  B = diag(ct_eff, ct_eff, Bn). */
typedef struct { int active; real f0[3]; real B[3]; } contact_law;
static contact_law contact_eval(real d, const real vrel_n[3], real kn, real cn, real ct, real mu, real ramp_depth, real h) {
  contact_law c;
  memset(&c, 0, sizeof(c));
  if (!(d > 0)) return c;
  real ramp = d / ramp_depth; if (ramp > 1) ramp = 1;
  real Bn = kn * h + cn * ramp;
  real f0n = kn * d - Bn * vrel_n[2];
  if (!(f0n > 0)) return c;
  real vt = (real)sqrt((double)(vrel_n[0] * vrel_n[0] + vrel_n[1] * vrel_n[1]));
  real cte = mu * f0n / (vt > (real)1e-6 ? vt : (real)1e-6);
  if (cte > ct) cte = ct;
  c.active = 1;
  c.f0[0] = -cte * vrel_n[0]; c.f0[1] = -cte * vrel_n[1]; c.f0[2] = f0n;
  c.B[0] = cte; c.B[1] = cte; c.B[2] = Bn;
  return c;
}
/* add h * J^T (Rb B Rb^T) J to a 6x6 (J = [-r~ 1]) with Rb: normal frame -> coords of the 6x6 */
static void add_contact_inertia(mat6 IA, const real r[3], const real Rb[9], const real B[3], real h) {
  real Bb[9], D[9] = {B[0], 0, 0, 0, B[1], 0, 0, 0, B[2]}, Rt[9], T[9];
  m3_transpose(Rt, Rb);
  m3_mul(T, Rb, D);
  m3_mul(Bb, T, Rt);
  real rs[9], K[9], KR[9];
  m3_skew(rs, r);
  m3_mul(K, rs, Bb);     /* r~ B */
  m3_mul(KR, K, rs);     /* r~ B r~ */
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      IA[i * 6 + j] += -h * KR[i * 3 + j];
      IA[i * 6 + 3 + j] += h * K[i * 3 + j];
      IA[(3 + i) * 6 + j] += h * K[j * 3 + i]; /* (r~ B)^T = -B r~ */
      IA[(3 + i) * 6 + 3 + j] += h * Bb[i * 3 + j];
    }
}

typedef struct {
  real obj_force[3];            /* net contact force on the object (world) */
  real body_force[5][4][3];     /* [sensor type hip,thigh,calf,foot,trunk][leg] net contact force (world) */
  real trunk_force[3];
  real plate[4][3];             /* tactile tasks: plate sample k -> contact point (x, y) in the trunk frame, normal force on the plate */
} contact_report;

/* one integrator substep of length h; tau held constant */
static void physics_substep(const lt_cfg* cfg, env_t* E, real h, int has_object, contact_report* rep) {
  const real g = cfg->gravity;
  mat6 I[NB], X[NB], IA[NB];
  vec6 v[NB], cb[NB], pA[NB], a[NB], U[NB];
  real D[NB], uu[NB];
  real Rw[NB][9], pw[NB][3];
  int parent[NB], axis[NB];
  /* --- model + kinematics --- */
  real mt = LT_TRUNK_MASS + E->trunk_mass_add;
  spatial_inertia(I[0], mt, k_trunk_com, k_trunk_icom, mt / LT_TRUNK_MASS);
  quat_to_mat(Rw[0], E->root_quat);
  v3_copy(pw[0], E->root_pos);
  m3_tmulv(v[0], Rw[0], E->root_ang);      /* omega_b */
  m3_tmulv(v[0] + 3, Rw[0], E->root_lin);  /* v_b */
  parent[0] = -1; axis[0] = -1;
  for (int l = 0; l < 4; ++l)
    for (int k = 0; k < 3; ++k) {
      int b = BODY(l, k), p = k == 0 ? 0 : BODY(l, k - 1);
      real rk[3] = {k_link_com[l][k][0], k_link_com[l][k][1], k_link_com[l][k][2]};
      real ik[6];
      for (int i = 0; i < 6; ++i) ik[i] = k_link_icom[l][k][i];
      spatial_inertia(I[b], k_link_mass[l][k], rk, ik, 1);
      parent[b] = p; axis[b] = k_joint_axis[k];
      real R[9], off[3] = {k_joint_off[l][k][0], k_joint_off[l][k][1], k_joint_off[l][k][2]}, t[3];
      rot_axis(R, axis[b], E->q[l][k]);
      xform(X[b], R, off);
      m3_mul(Rw[b], Rw[p], R);
      m3_mulv(t, Rw[p], off);
      v3_add(pw[b], pw[p], t);
      m6_mulv(v[b], X[b], v[p]);
      vec6 vj = {0, 0, 0, 0, 0, 0};
      vj[axis[b]] = E->qd[l][k];
      for (int i = 0; i < 6; ++i) v[b][i] += vj[i];
      crm(cb[b], v[b], vj);
    }
  /* --- bias forces, gravity as an explicit spatial force on every body --- */
  for (int b = 0; b < NB; ++b) {
    vec6 Iv, gb = {0, 0, 0, 0, 0, 0}, fg;
    memcpy(IA[b], I[b], sizeof(mat6));
    m6_mulv(Iv, I[b], v[b]);
    crf(pA[b], v[b], Iv);
    real gw[3] = {0, 0, -g};
    m3_tmulv(gb + 3, Rw[b], gw);
    m6_mulv(fg, I[b], gb);
    for (int i = 0; i < 6; ++i) pA[b][i] -= fg[i];
  }
  memset(rep, 0, sizeof(*rep));
  /* --- object: free rigid body with implicit contacts against the carrying plate and the ground --- */
  real obj_a[6] = {0, 0, 0, 0, 0, 0};
  typedef struct { int active; real rho[3]; real F0[3]; real Bw[9]; real P[3]; } ocontact;
  ocontact oc[6];
  memset(oc, 0, sizeof(oc));
  real Ro[9];
  if (has_object) {
    quat_to_mat(Ro, E->obj_quat);
    real ay_w[3] = {Ro[1], Ro[4], Ro[7]}; /* cylinder axis (local y) in world */
    real rad = E->obj_radius, half = (real)0.5 * E->obj_length;
    real mu_plate = (real)0.5 * (E->trunk_mu + E->obj_mu);
    /* plate frame = trunk frame */
    real ct_[3], at_[3], d0[3];
    v3_sub(d0, E->obj_pos, pw[0]);
    m3_tmulv(ct_, Rw[0], d0);
    m3_tmulv(at_, Rw[0], ay_w);
    const real hx = LT_BACK_HALF_X, hy = LT_RAIL_Y + LT_RAIL_RADIUS, zp = LT_BACK_TOP_Z;
    real s0 = -half, s1 = half;
    int ok = 1;
    const real lim[2] = {hx, hy};
    for (int ax = 0; ax < 2 && ok; ++ax) { /* slab clipping of the axis segment against |x|<=hx, |y|<=hy */
      real c0 = ct_[ax], da = at_[ax];
      if (fabs((double)da) < 1e-9) { if (fabs((double)c0) > lim[ax]) ok = 0; }
      else {
        real ta = (-lim[ax] - c0) / da, tb = (lim[ax] - c0) / da;
        if (ta > tb) { real t = ta; ta = tb; tb = t; }
        if (ta > s0) s0 = ta;
        if (tb < s1) s1 = tb;
        if (s0 > s1) ok = 0;
      }
    }
    int nc = 0;
    if (ok) {
      real nz_a = at_[2]; /* n . a with n = +z of the plate */
      real up[3] = {-nz_a * at_[0], -nz_a * at_[1], 1 - nz_a * at_[2]}; /* n - (n.a) a */
      real un = v3_norm(up);
      real inv = 1 / (un > (real)1e-6 ? un : (real)1e-6);
      for (int k = 0; k < 4; ++k) {
        real s = s0 + (s1 - s0) * (real)k / 3;
        real Pt[3] = {ct_[0] + s * at_[0] - rad * up[0] * inv, ct_[1] + s * at_[1] - rad * up[1] * inv,
                      ct_[2] + s * at_[2] - rad * up[2] * inv};
        real d = zp - Pt[2];
        rep->plate[k][0] = Pt[0]; rep->plate[k][1] = Pt[1];
        ocontact* c = &oc[nc++];
        real Pw[3], t[3], vo[3], vtk[3], vrel[3], vrel_n[3], rt[3];
        m3_mulv(t, Rw[0], Pt);
        v3_add(Pw, pw[0], t);
        v3_copy(c->P, Pw);
        v3_sub(c->rho, Pw, E->obj_pos);
        v3_cross(vo, E->obj_ang, c->rho); v3_add(vo, vo, E->obj_lin);
        v3_sub(rt, Pw, pw[0]);
        v3_cross(vtk, E->root_ang, rt); v3_add(vtk, vtk, E->root_lin);
        v3_sub(vrel, vo, vtk);
        m3_tmulv(vrel_n, Rw[0], vrel);
        contact_law cl = contact_eval(d, vrel_n, cfg->plate_kn / 4, cfg->plate_cn / 4, cfg->plate_ct / 4, mu_plate,
                                      cfg->contact_ramp, h);
        c->active = cl.active;
        if (cl.active) {
          m3_mulv(c->F0, Rw[0], cl.f0);
          real Dm[9] = {cl.B[0], 0, 0, 0, cl.B[1], 0, 0, 0, cl.B[2]}, T[9], Rt[9];
          m3_transpose(Rt, Rw[0]);
          m3_mul(T, Rw[0], Dm);
          m3_mul(c->Bw, T, Rt);
        }
      }
    }
    nc = 4;
    /* ground: the two rim points under the axis ends */
    {
      real nz_a = ay_w[2];
      real up[3] = {-nz_a * ay_w[0], -nz_a * ay_w[1], 1 - nz_a * ay_w[2]};
      real un = v3_norm(up);
      real inv = 1 / (un > (real)1e-6 ? un : (real)1e-6);
      for (int k = 0; k < 2; ++k) {
        real s = k == 0 ? -half : half;
        ocontact* c = &oc[nc++];
        real Pw[3] = {E->obj_pos[0] + s * ay_w[0] - rad * up[0] * inv, E->obj_pos[1] + s * ay_w[1] - rad * up[1] * inv,
                      E->obj_pos[2] + s * ay_w[2] - rad * up[2] * inv};
        v3_copy(c->P, Pw);
        v3_sub(c->rho, Pw, E->obj_pos);
        real vo[3];
        v3_cross(vo, E->obj_ang, c->rho); v3_add(vo, vo, E->obj_lin);
        contact_law cl = contact_eval(-Pw[2], vo, cfg->ground_kn, cfg->ground_cn, cfg->ground_ct, E->obj_mu * cfg->ground_mu,
                                      cfg->contact_ramp, h);
        c->active = cl.active;
        if (cl.active) {
          v3_copy(c->F0, cl.f0);
          real Dm[9] = {cl.B[0], 0, 0, 0, cl.B[1], 0, 0, 0, cl.B[2]};
          memcpy(c->Bw, Dm, sizeof(Dm));
        }
      }
    }
    /* 6x6 solve in world axes about the object COM: [I_w 0; 0 m] + h sum J^T B J */
    real m = E->obj_mass;
    real Iyy = (real)0.5 * m * rad * rad, Ixx = m * (3 * rad * rad + E->obj_length * E->obj_length) / 12;
    real Dm[9] = {Ixx, 0, 0, 0, Iyy, 0, 0, 0, Ixx}, T[9], Rt[9], Iw[9];
    m3_transpose(Rt, Ro);
    m3_mul(T, Ro, Dm);
    m3_mul(Iw, T, Rt);
    mat6 M;
    m6_zero(M);
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) M[i * 6 + j] = Iw[i * 3 + j];
      M[(3 + i) * 6 + 3 + i] = m;
    }
    real Iom[3], gyro[3];
    m3_mulv(Iom, Iw, E->obj_ang);
    v3_cross(gyro, E->obj_ang, Iom);
    vec6 rhs = {-gyro[0], -gyro[1], -gyro[2], 0, 0, -m * g};
    for (int k = 0; k < 6; ++k)
      if (oc[k].active) {
        real tq[3];
        v3_cross(tq, oc[k].rho, oc[k].F0);
        for (int i = 0; i < 3; ++i) { rhs[i] += tq[i]; rhs[3 + i] += oc[k].F0[i]; }
        /* B is already in world axes: pass it through add_contact_inertia with an identity frame by
         * decomposing: h J^T Bw J */
        real rs[9], K[9], KR[9];
        m3_skew(rs, oc[k].rho);
        m3_mul(K, rs, oc[k].Bw);
        m3_mul(KR, K, rs);
        for (int i = 0; i < 3; ++i)
          for (int j = 0; j < 3; ++j) {
            M[i * 6 + j] += -h * KR[i * 3 + j];
            M[i * 6 + 3 + j] += h * K[i * 3 + j];
            M[(3 + i) * 6 + j] += h * K[j * 3 + i];
            M[(3 + i) * 6 + 3 + j] += h * oc[k].Bw[i * 3 + j];
          }
      }
    spd6_solve(M, rhs, obj_a);
    /* final contact forces; reaction on the trunk (explicit) */
    for (int k = 0; k < 6; ++k)
      if (oc[k].active) {
        real ap[3], t[3], F[3];
        v3_cross(ap, obj_a, oc[k].rho);
        v3_add(ap, ap, obj_a + 3);
        m3_mulv(t, oc[k].Bw, ap);
        for (int i = 0; i < 3; ++i) F[i] = oc[k].F0[i] - h * t[i];
        v3_add(rep->obj_force, rep->obj_force, F);
        if (k < 4) { /* plate contact: -F on the trunk at P */
          rep->plate[k][2] = F[0] * Rw[0][2] + F[1] * Rw[0][5] + F[2] * Rw[0][8]; /* plate normal = trunk z axis in world */
          real rb[3], fb[3], nb[3], dpos[3], Fn[3] = {-F[0], -F[1], -F[2]};
          v3_sub(dpos, oc[k].P, pw[0]);
          m3_tmulv(rb, Rw[0], dpos);
          m3_tmulv(fb, Rw[0], Fn);
          v3_cross(nb, rb, fb);
          for (int i = 0; i < 3; ++i) { pA[0][i] -= nb[i]; pA[0][3 + i] -= fb[i]; }
          v3_add(rep->trunk_force, rep->trunk_force, Fn);
        }
      }
  }
  /* --- robot ground contacts (implicit in the link acceleration) --- */
  typedef struct { int active, body; real rc[3]; real f0b[3]; real Bb[9]; } rcontact;
  rcontact rc_[4][SP_PER_LEG];
  memset(rc_, 0, sizeof(rc_));
  for (int l = 0; l < 4; ++l)
    for (int s = 0; s < SP_PER_LEG; ++s) {
      int b; real r[3], rho;
      sphere_def(l, s, &b, r, &rho);
      rcontact* c = &rc_[l][s];
      c->body = b;
      real zb[3] = {Rw[b][6], Rw[b][7], Rw[b][8]}; /* R^T z = third row of R */
      for (int i = 0; i < 3; ++i) c->rc[i] = r[i] - rho * zb[i];
      real Pc[3], t[3], vb[3], vwld[3];
      m3_mulv(t, Rw[b], c->rc);
      v3_add(Pc, pw[b], t);
      v3_cross(vb, v[b], c->rc);
      v3_add(vb, vb, v[b] + 3);
      m3_mulv(vwld, Rw[b], vb);
      real mu = (s == SP_FOOT ? E->foot_mu[l] : (real)1.0) * cfg->ground_mu;
      contact_law cl = contact_eval(-Pc[2], vwld, cfg->ground_kn, cfg->ground_cn, cfg->ground_ct, mu, cfg->contact_ramp, h);
      c->active = cl.active;
      if (s == SP_FOOT) {
        real ctr[3], rr[3] = {r[0], r[1], r[2]}, vc[3];
        m3_mulv(t, Rw[b], rr); v3_add(ctr, pw[b], t);
        v3_cross(vc, v[b], rr); v3_add(vc, vc, v[b] + 3);
        m3_mulv(E->foot_vel[l], Rw[b], vc);
        v3_copy(E->foot_pos[l], ctr);
      }
      if (!cl.active) continue;
      m3_tmulv(c->f0b, Rw[b], cl.f0);
      real Dm[9] = {cl.B[0], 0, 0, 0, cl.B[1], 0, 0, 0, cl.B[2]}, T[9], Rt[9];
      m3_transpose(Rt, Rw[b]);
      m3_mul(T, Rt, Dm);
      m3_mul(c->Bb, T, Rw[b]);
      real Rn2b[9];
      m3_transpose(Rn2b, Rw[b]); /* world(normal frame) -> body */
      add_contact_inertia(IA[b], c->rc, Rn2b, cl.B, h);
      real nb[3];
      v3_cross(nb, c->rc, c->f0b);
      for (int i = 0; i < 3; ++i) { pA[b][i] -= nb[i]; pA[b][3 + i] -= c->f0b[i]; }
    }
  /* --- ABA backward pass --- */
  for (int b = NB - 1; b >= 1; --b) {
    int l = (b - 1) / 3, k = (b - 1) % 3, ax = axis[b];
    for (int i = 0; i < 6; ++i) U[b][i] = IA[b][i * 6 + ax];
    D[b] = U[b][ax];
    uu[b] = E->tau[l][k] - pA[b][ax];
    { /* joint limit: unilateral implicit spring-damper on the joint coordinate (lt_cfg.joint_limit_*): beyond a limit by d the
       * torque along the inward direction s is  k d - B s qd_new,  B = k h + c  ->  (D + h B) qdd = u + s f0 - U^T a_p */
      real Bl = cfg->joint_limit_kp * h + cfg->joint_limit_kd;
      real dlo = k_joint_lo[k] - E->q[l][k], dhi = E->q[l][k] - k_joint_hi[k];
      real sg = dlo > dhi ? (real)1 : (real)-1, d = dlo > dhi ? dlo : dhi; /* the nearer limit */
      real f0 = cfg->joint_limit_kp * d - Bl * sg * E->qd[l][k];
      /* active when the joint would be beyond the limit at the end of the step at its present velocity; only pushes */
      if (d - h * sg * E->qd[l][k] > 0 && f0 > 0) { D[b] += h * Bl; uu[b] += sg * f0; }
    }
    mat6 Ia;
    vec6 pa, Iac;
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) Ia[i * 6 + j] = IA[b][i * 6 + j] - U[b][i] * U[b][j] / D[b];
    m6_mulv(Iac, Ia, cb[b]);
    for (int i = 0; i < 6; ++i) pa[i] = pA[b][i] + Iac[i] + U[b][i] * uu[b] / D[b];
    /* IA_p += X^T Ia X ; pA_p += X^T pa */
    mat6 T;
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) {
        real s = 0;
        for (int m = 0; m < 6; ++m) s += Ia[i * 6 + m] * X[b][m * 6 + j];
        T[i * 6 + j] = s;
      }
    int p = parent[b];
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) {
        real s = 0;
        for (int m = 0; m < 6; ++m) s += X[b][m * 6 + i] * T[m * 6 + j];
        IA[p][i * 6 + j] += s;
      }
    vec6 xp;
    m6_tmulv(xp, X[b], pa);
    for (int i = 0; i < 6; ++i) pA[p][i] += xp[i];
  }
  /* --- base acceleration and forward pass --- */
  vec6 nb_;
  for (int i = 0; i < 6; ++i) nb_[i] = -pA[0][i];
  spd6_solve(IA[0], nb_, a[0]);
  for (int b = 1; b < NB; ++b) {
    int l = (b - 1) / 3, k = (b - 1) % 3, ax = axis[b];
    vec6 ap;
    m6_mulv(ap, X[b], a[parent[b]]);
    for (int i = 0; i < 6; ++i) ap[i] += cb[b][i];
    real ud = 0;
    for (int i = 0; i < 6; ++i) ud += U[b][i] * ap[i];
    real qdd = (uu[b] - ud) / D[b];
    memcpy(a[b], ap, sizeof(vec6));
    a[b][ax] += qdd;
    E->qdd[l][k] = qdd; /* overwritten by the sim-step finite difference below (B3 joint_acc) */
  }
  /* --- final robot contact forces for the sensors --- */
  for (int l = 0; l < 4; ++l)
    for (int s = 0; s < SP_PER_LEG; ++s) {
      rcontact* c = &rc_[l][s];
      if (!c->active) continue;
      int b = c->body;
      real ap[3], t[3], fb[3], Fw[3];
      v3_cross(ap, a[b], c->rc);
      v3_add(ap, ap, a[b] + 3);
      m3_mulv(t, c->Bb, ap);
      for (int i = 0; i < 3; ++i) fb[i] = c->f0b[i] - h * t[i];
      m3_mulv(Fw, Rw[b], fb);
      int st = k_sphere_sensor[s];
      if (st == 4) v3_add(rep->trunk_force, rep->trunk_force, Fw);
      else v3_add(rep->body_force[st][l], rep->body_force[st][l], Fw);
    }
  /* --- semi-implicit Euler --- */
  for (int l = 0; l < 4; ++l)
    for (int k = 0; k < 3; ++k) {
      real qd = E->qd[l][k] + h * E->qdd[l][k];
      E->q[l][k] += h * qd; E->qd[l][k] = qd; /* (joint limits act inside the solve: no clamp) */
    }
  {
    real wxv[3], acl[3], aw[3], alw[3];
    v3_cross(wxv, v[0], v[0] + 3);
    v3_add(acl, a[0] + 3, wxv); /* classical acceleration of the base origin, body coords */
    m3_mulv(alw, Rw[0], acl);
    m3_mulv(aw, Rw[0], a[0]);
    v3_axpy(E->root_lin, h, alw);
    v3_axpy(E->root_ang, h, aw);
    v3_axpy(E->root_pos, h, E->root_lin);
    real dq[4] = {0, E->root_ang[0], E->root_ang[1], E->root_ang[2]}, t[4];
    quat_mul(t, dq, E->root_quat);
    for (int i = 0; i < 4; ++i) E->root_quat[i] += (real)0.5 * h * t[i];
    quat_normalize(E->root_quat);
  }
  if (has_object) {
    v3_axpy(E->obj_ang, h, obj_a);
    v3_axpy(E->obj_lin, h, obj_a + 3);
    v3_axpy(E->obj_pos, h, E->obj_lin);
    real dq[4] = {0, E->obj_ang[0], E->obj_ang[1], E->obj_ang[2]}, t[4];
    quat_mul(t, dq, E->obj_quat);
    for (int i = 0; i < 4; ++i) E->obj_quat[i] += (real)0.5 * h * t[i];
    quat_normalize(E->obj_quat);
  }
}

/* foot kinematics for the post-physics state (feet positions/velocities read by rewards R4/R5) */
static void foot_kinematics(env_t* E) {
  real R0[9], wb[3], vb[3];
  quat_to_mat(R0, E->root_quat);
  m3_tmulv(wb, R0, E->root_ang);
  m3_tmulv(vb, R0, E->root_lin);
  for (int l = 0; l < 4; ++l) {
    real Rp[9], pp[3], w[3], vv[3];
    memcpy(Rp, R0, sizeof(Rp));
    v3_copy(pp, E->root_pos); v3_copy(w, wb); v3_copy(vv, vb);
    for (int k = 0; k < 3; ++k) {
      real R[9], off[3] = {k_joint_off[l][k][0], k_joint_off[l][k][1], k_joint_off[l][k][2]}, t[3], Rn[9], wn[3], vn[3], c[3];
      rot_axis(R, k_joint_axis[k], E->q[l][k]);
      m3_mulv(t, Rp, off); v3_add(pp, pp, t);
      m3_mul(Rn, Rp, R);
      /* body-coords velocity propagation: w' = E w + S qd ; v' = E (v + w x r) */
      v3_cross(c, w, off); v3_add(c, c, vv);
      m3_tmulv(vn, R, c);
      m3_tmulv(wn, R, w);
      wn[k_joint_axis[k]] += E->qd[l][k];
      memcpy(Rp, Rn, sizeof(Rp)); v3_copy(w, wn); v3_copy(vv, vn);
    }
    real rf[3] = {0, 0, -0.213f}, t[3], c[3];
    m3_mulv(t, Rp, rf); v3_add(E->foot_pos[l], pp, t);
    v3_cross(c, w, rf); v3_add(c, c, vv);
    m3_mulv(E->foot_vel[l], Rp, c);
  }
}

/* ------------------------------------------------------------------------------------------------ */
/* K3: contact-sensor bookkeeping (IsaacLab ContactSensor [DEP])               */
/* ------------------------------------------------------------------------------------------------ */
static void timers_update(real* cur_air, real* cur_con, real* last_air, real* last_con, int contact, real dt) {
  int first_contact = (*cur_air > 0) && contact;
  int first_detach = (*cur_con > 0) && !contact;
  if (first_contact) *last_air = *cur_air + dt;
  if (first_detach) *last_con = *cur_con + dt;
  *cur_air = contact ? 0 : *cur_air + dt;
  *cur_con = contact ? *cur_con + dt : 0;
}
static void sensors_update(const lt_cfg* cfg, env_t* E, const contact_report* rep, int has_object) {
  real dt = cfg->sim_dt, thr = cfg->contact_force_threshold;
  for (int l = 0; l < 4; ++l) {
    for (int t = 0; t < 4; ++t) {
      E->fhist[2][t][l] = E->fhist[1][t][l];
      E->fhist[1][t][l] = E->fhist[0][t][l];
      E->fhist[0][t][l] = v3_norm(rep->body_force[t][l]);
    }
    timers_update(&E->foot_cur_air[l], &E->foot_cur_con[l], &E->foot_last_air[l], &E->foot_last_con[l],
                  E->fhist[0][3][l] > thr, dt);
  }
  E->trunk_fhist[2] = E->trunk_fhist[1]; E->trunk_fhist[1] = E->trunk_fhist[0];
  E->trunk_fhist[0] = v3_norm(rep->trunk_force);
  if (has_object)
    timers_update(&E->obj_timers[0], &E->obj_timers[1], &E->obj_timers[2], &E->obj_timers[3], v3_norm(rep->obj_force) > thr, dt);
}

/* ------------------------------------------------------------------------------------------------ */
/* K5: rewards (reference locotouch/mdp/rewards.py, whole file) and K4 terminations                  */
/* ------------------------------------------------------------------------------------------------ */
static const int k_gait_order[4] = {0, 3, 1, 2}; /* gait foot columns [FR, RL, FL, RR] (rewards.py:89-92) as leg ids */

/* rewards.py:158-200 : _update_valid_last_air_contact_time.  Arrays are in gait column order. */
static void gait_update(const lt_cfg* cfg, lt_gait_io* G) {
  const real judge = cfg->gait_judge_time;
  real cn = (real)sqrt((double)(G->cmd[0] * G->cmd[0] + G->cmd[1] * G->cmd[1] + G->cmd[2] * G->cmd[2]));
  int nonzero = cn > 0;
  if (!nonzero) for (int f = 0; f < 4; ++f) G->valid_last_air[f] = 0;                         /* :166-168 */
  for (int f = 0; f < 4; ++f) {
    int new_swing = (G->last_step_air[f] < judge) && (G->cur_air[f] > judge);                   /* :171 */
    if (new_swing && nonzero) G->swinging_in_zero_cmd[f] = 0;                                   /* :172-173 */
    if ((G->cur_air[f] > judge) && !nonzero) G->swinging_in_zero_cmd[f] = 1;                    /* :176-178 */
  }
  G->step_from_change += 1;                                                                     /* :181 */
  int changing = 0;
  for (int c = 0; c < 3; ++c) if (fabs((double)(G->cmd[c] - G->last_cmd[c])) > 1.0e-3) changing = 1; /* :183 */
  if (changing) {                                                                               /* :184-187 */
    for (int c = 0; c < 3; ++c) G->last_cmd[c] = G->cmd[c];
    G->step_from_change = 0;
    for (int f = 0; f < 4; ++f) { G->swinging_in_zero_cmd[f] = 1; G->valid_last_air[f] = 0; }
  }
  if (G->any_nonzero_cmd)                                                                       /* :190 (global gate) */
    for (int f = 0; f < 4; ++f) {
      int landing = (G->last_step_con[f] < judge) && (G->cur_con[f] > judge);                   /* :191 */
      if (landing && G->valid_prev_contact[f] && !G->swinging_in_zero_cmd[f]) G->valid_last_air[f] = G->sensor_last_air[f];
    }
  for (int f = 0; f < 4; ++f) {
    G->last_step_air[f] = G->cur_air[f];                                                        /* :196-197 */
    G->last_step_con[f] = G->cur_con[f];
    if (G->cur_con[f] > judge) G->valid_prev_contact[f] = 1;                                    /* :200 */
  }
}
static real clampr(real x, real lo, real hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* rewards.py:243-346 : _swinging_bonus for the pair (f0, f1) given in gait columns */
static real gait_swing_bonus(const lt_cfg* cfg, const lt_gait_io* G, int f0, int f1, real step_dt) {
  const real judge = cfg->gait_judge_time, ub = cfg->gait_rwd_upper, lb = cfg->gait_rwd_lower, tol = cfg->gait_tolerance_proportion;
  const real scale = ub / ((real)1.0 / (cfg->gait_soft_min_frequency * (real)2.0));            /* :72,:78 */
  real tbar = (G->cur_air[f0] + G->cur_air[f1]) / 2;
  int both_air = (G->cur_air[f0] > judge) && (G->cur_air[f1] > judge);                          /* :247 (no upper bound) */
  int t0 = (f0 < 2) ? 0 : 2, o0 = (f0 < 2) ? 2 : 0;                                              /* :250-254 */
  real vt0 = G->valid_last_air[t0], vt1 = G->valid_last_air[t0 + 1], vo0 = G->valid_last_air[o0], vo1 = G->valid_last_air[o0 + 1];
  real mean_t = (vt0 + vt1) / 2, mean_o = (vo0 + vo1) / 2;
  int valid_t = (vt0 > judge) && (vt1 > judge) && (vt0 > 2 * step_dt) && (vt1 > 2 * step_dt);  /* :253,:260 */
  int valid_o = (vo0 > judge) && (vo1 > judge) && (vo0 > 2 * step_dt) && (vo1 > 2 * step_dt);  /* :257,:261 */
  int with_ref = both_air && (valid_t || valid_o);                                              /* :264-265 */
  if (!with_ref) return 0;                                                                      /* :341-344 */
  real Tref = mean_o;                                                                           /* :266-269 */
  real Ttol = Tref + tol * Tref;                                                                /* :272 */
  real diff = mean_t - mean_o;                                                                  /* :273-276 */
  real Text = clampr(Ttol - diff, Tref, Ttol);                                                  /* :277-278 */
  int within = (tbar <= Text) || (diff < 0);                                                    /* :281,:285-286 */
  int between = (tbar > Text) && (tbar <= Ttol);                                                /* :282 */
  real r_within = scale * tbar; if (r_within > ub) r_within = ub;                               /* :289 */
  real r_ref = scale * Tref; if (r_ref > ub) r_ref = ub;
  real r_ext = scale * Text; if (r_ext > ub) r_ext = ub;
  real r_tol = scale * Ttol; if (r_tol > ub) r_tol = ub;
  if (within) return r_within;
  int ext_lt_tol = Text < Ttol;                                                                 /* :295 */
  if (between) {                                                                                /* :296-307 */
    if (!ext_lt_tol) return r_ext;
    real a = -r_ext / (Ttol - Text);
    return a * tbar + (-a * Ttol);
  }
  int ext_gt_ref = Text > Tref;                                                                 /* :310 */
  real lower = ext_lt_tol ? (diff / (tol * Tref)) * lb : r_tol;                                 /* :319-322 */
  if (!valid_o) lower = lb;                                                                     /* :323-324 */
  lower = clampr(lower, lb, ub);                                                                /* :325 */
  real beyond;
  if (ext_gt_ref) {                                                                             /* :311-318,:326-329 */
    real a = -r_ref / (Text - Tref);
    beyond = a * tbar + (-a * Ttol);
  } else beyond = lower;
  if (beyond < lower) beyond = lower;                                                           /* :330 */
  return beyond;
}

/* rewards.py:116-156 (+ :202-241, :348-368, :371-392).  Returns the gait term and updates the state. */
float lt_oracle_gait(const lt_cfg* cfg, lt_gait_io* G, float step_dt) {
  const real judge = cfg->gait_judge_time, ab = cfg->gait_air_bound, cb_ = cfg->gait_contact_bound;
  const real async_judge = judge + cfg->gait_async_tolerance;
  gait_update(cfg, G);
  real cn = (real)sqrt((double)(G->cmd[0] * G->cmd[0] + G->cmd[1] * G->cmd[1] + G->cmd[2] * G->cmd[2]));
  int nonzero = cn > 0;
  /* task performance score :202-213, :372-392 */
  real e_lin = nonzero ? G->lin_err : 0, e_ang = nonzero ? G->ang_err : 0;
  real vel_score = ((real)exp(-(double)(e_lin / cfg->gait_vel_sigma)) + (real)exp(-(double)(e_ang / cfg->gait_vel_sigma))) / 2;
  real score = vel_score;
  if (cfg->gait_with_object) {
    real sx = clampr(1 - (real)fabs((double)G->obj_xy_yaw[0]) / cfg->danger_x_max, 0, 1);
    real sy = clampr(1 - (real)fabs((double)G->obj_xy_yaw[1]) / cfg->danger_y_max, 0, 1);
    score = clampr((vel_score * 2 + (sx + sy) / 2) / 3, 0, 1);
  }
  real sync[2];
  for (int p = 0; p < 2; ++p) {                                                                 /* :218-241 */
    int f0 = 2 * p, f1 = 2 * p + 1;
    int both_air = (G->cur_air[f0] > judge && G->cur_air[f0] < ab) && (G->cur_air[f1] > judge && G->cur_air[f1] < ab);
    int both_con = (G->cur_con[f0] > judge && G->cur_con[f0] < cb_) && (G->cur_con[f1] > judge && G->cur_con[f1] < cb_);
    real bonus = gait_swing_bonus(cfg, G, f0, f1, step_dt);
    real rs = 1 - cfg->gait_task_ratio + cfg->gait_task_ratio * score;
    if (bonus > 0) bonus *= rs;                                                                 /* :237 */
    bonus += 1;
    sync[p] = both_air ? bonus : (both_con ? (real)1 : (real)0);
  }
  static const int pairs[4][2] = {{0, 2}, {1, 3}, {0, 3}, {2, 1}};                               /* :144-147 */
  real asum = 0;
  for (int p = 0; p < 4; ++p) {                                                                 /* :348-363 */
    int f0 = pairs[p][0], f1 = pairs[p][1];
    int c0t = G->cur_con[f0] > judge && G->cur_con[f0] <= async_judge, c1t = G->cur_con[f1] > judge && G->cur_con[f1] <= async_judge;
    int a0 = G->cur_air[f0] > judge && G->cur_air[f0] < ab, a1 = G->cur_air[f1] > judge && G->cur_air[f1] < ab;
    int c0 = G->cur_con[f0] > judge && G->cur_con[f0] < cb_, c1 = G->cur_con[f1] > judge && G->cur_con[f1] < cb_;
    asum += ((c0t && c1t) || (a0 && c1) || (c0 && a1)) ? 1 : 0;
  }
  real stepping = ((sync[0] + sync[1]) / 2 + asum / 4) / 2;                                     /* :141-151 */
  int all_stance = 1;
  for (int f = 0; f < 4; ++f) all_stance &= G->cur_con[f] > judge;                              /* :365-368 */
  real stance = (all_stance ? 1 : 0) * cfg->gait_stance_scale;
  return nonzero ? stepping : stance;                                                           /* :153-154 */
}

/* All reward terms of one env from a B3-style record; zero-weight terms are skipped (IsaacLab [DEP]). */
void lt_oracle_rewards(const lt_cfg* cfg, const lt_term_in* in, lt_gait_io* G, float step_dt, float* terms) {
  const real* cmd = in->cmd;
  real R0[9];
  quat_to_mat(R0, in->root_quat);
  real vb[3], wb[3], gb[3], gz[3] = {0, 0, -1};
  m3_tmulv(vb, R0, in->root_lin);
  m3_tmulv(wb, R0, in->root_ang);
  m3_tmulv(gb, R0, gz);
  real cn = v3_norm(cmd);
  real lin_err = (real)sqrt((double)((cmd[0] - vb[0]) * (cmd[0] - vb[0]) + (cmd[1] - vb[1]) * (cmd[1] - vb[1])));
  real ang_err = (real)fabs((double)(cmd[2] - wb[2]));
  const float* w = cfg->reward_weight;
  for (int i = 0; i < LT_REWARD_SLOTS; ++i) terms[i] = 0;
  if (w[LT_R_ALIVE] != 0) terms[LT_R_ALIVE] = in->terminated ? 0 : 1;                           /* is_alive [DEP] */
  if (w[LT_R_TRACK_LIN_VEL_XY] != 0) terms[LT_R_TRACK_LIN_VEL_XY] = (real)exp(-(double)(lin_err / cfg->track_sigma)); /* :15-20 */
  if (w[LT_R_TRACK_ANG_VEL_Z] != 0) terms[LT_R_TRACK_ANG_VEL_Z] = (real)exp(-(double)(ang_err / cfg->track_sigma));   /* :22-27 */
  if (w[LT_R_FOOT_SLIP] != 0) {                                                                 /* :31-42 */
    real s = 0;
    for (int l = 0; l < 4; ++l) {
      real mx = in->fhist[0][3][l];
      if (in->fhist[1][3][l] > mx) mx = in->fhist[1][3][l];
      if (in->fhist[2][3][l] > mx) mx = in->fhist[2][3][l];
      real pv = (real)sqrt((double)(in->foot_vel[l][0] * in->foot_vel[l][0] + in->foot_vel[l][1] * in->foot_vel[l][1]));
      if (mx > cfg->foot_slip_threshold) s += pv;
    }
    terms[LT_R_FOOT_SLIP] = s;
  }
  if (w[LT_R_FOOT_DRAGGING] != 0) {                                                             /* :44-56 */
    real s = 0;
    for (int l = 0; l < 4; ++l) {
      real pv = (real)sqrt((double)(in->foot_vel[l][0] * in->foot_vel[l][0] + in->foot_vel[l][1] * in->foot_vel[l][1]));
      if (in->foot_pos[l][2] <= cfg->foot_drag_height && pv > cfg->foot_drag_vel) s += 1;
    }
    terms[LT_R_FOOT_DRAGGING] = s;
  }
  if (w[LT_R_GAIT] != 0) {
    G->lin_err = lin_err; G->ang_err = ang_err;
    for (int c = 0; c < 3; ++c) G->cmd[c] = cmd[c];
    if (cfg->gait_with_object) {                                                                /* :372-385 yaw-only frame */
      real yaw = quat_yaw_2pi(in->root_quat), qy[4], d[3], o[3];
      quat_from_euler(qy, 0, 0, yaw);
      v3_sub(d, in->obj_pos, in->root_pos);
      quat_apply_inv(o, qy, d);
      G->obj_xy_yaw[0] = o[0]; G->obj_xy_yaw[1] = o[1];
    }
    terms[LT_R_GAIT] = lt_oracle_gait(cfg, G, step_dt);
  }
  if (w[LT_R_TRACK_BASE_HEIGHT] != 0) { real d = in->root_pos[2] - cfg->base_height_target; terms[LT_R_TRACK_BASE_HEIGHT] = d * d; } /* :398-402 */
  if (w[LT_R_BASE_Z_VELOCITY] != 0) terms[LT_R_BASE_Z_VELOCITY] = vb[2] * vb[2];                /* :404-408 */
  if (w[LT_R_BASE_ROLL_PITCH_ANGLE] != 0) terms[LT_R_BASE_ROLL_PITCH_ANGLE] = gb[0] * gb[0] + gb[1] * gb[1]; /* :416-420 */
  if (w[LT_R_BASE_ROLL_PITCH_VELOCITY] != 0) terms[LT_R_BASE_ROLL_PITCH_VELOCITY] = (real)(fabs((double)wb[0]) + fabs((double)wb[1])); /* :410-414 */
  real s_lim = 0, s_pos = 0, s_acc = 0, s_vel = 0, s_tau = 0, s_act = 0;
  for (int l = 0; l < 4; ++l)
    for (int k = 0; k < 3; ++k) {
      real mid = (k_joint_lo[k] + k_joint_hi[k]) / 2, rng = k_joint_hi[k] - k_joint_lo[k];
      real lo = mid - (real)0.5 * rng * LT_SOFT_LIMIT_FACTOR, hi = mid + (real)0.5 * rng * LT_SOFT_LIMIT_FACTOR;
      real q = in->q[l][k];
      real a = q - lo; if (a > 0) a = 0;
      real b = q - hi; if (b < 0) b = 0;
      s_lim += -a + b;                                                                          /* :423-427 */
      real dq = q - k_joint_default[l][k];
      s_pos += dq * dq; s_acc += in->qdd[l][k] * in->qdd[l][k]; s_vel += in->qd[l][k] * in->qd[l][k];
      s_tau += in->tau[l][k] * in->tau[l][k];
      real da = in->act_raw[l][k] - in->act_prev[l][k];
      s_act += da * da;
    }
  if (w[LT_R_JOINT_POSITION_LIMIT] != 0) terms[LT_R_JOINT_POSITION_LIMIT] = s_lim;
  if (w[LT_R_JOINT_POSITION] != 0) {                                                            /* :429-440 */
    real bv = (real)sqrt((double)(vb[0] * vb[0] + vb[1] * vb[1]));
    real r = (real)sqrt((double)s_pos);
    terms[LT_R_JOINT_POSITION] = (cn > 0 || bv > cfg->joint_pos_vel_threshold) ? r : cfg->joint_pos_stand_scale * r;
  }
  if (w[LT_R_JOINT_ACCELERATION] != 0) terms[LT_R_JOINT_ACCELERATION] = (real)sqrt((double)s_acc); /* :446-448 */
  if (w[LT_R_JOINT_VELOCITY] != 0) terms[LT_R_JOINT_VELOCITY] = (real)sqrt((double)s_vel);      /* :442-444 */
  if (w[LT_R_JOINT_TORQUE] != 0) terms[LT_R_JOINT_TORQUE] = (real)sqrt((double)s_tau);          /* :450-452 */
  if (w[LT_R_ACTION_RATE] != 0) terms[LT_R_ACTION_RATE] = s_act;                                /* :454-456 */
  if (w[LT_R_THIGH_CALF_COLLISION] != 0) {                                                      /* :459-466 */
    real s = 0;
    for (int l = 0; l < 4; ++l)
      for (int t = 1; t <= 2; ++t) {
        real mx = in->fhist[0][t][l];
        if (in->fhist[1][t][l] > mx) mx = in->fhist[1][t][l];
        if (in->fhist[2][t][l] > mx) mx = in->fhist[2][t][l];
        if (mx > cfg->thigh_calf_threshold) s += 1;
      }
    terms[LT_R_THIGH_CALF_COLLISION] = s;
  }
  if (cfg->task == LT_TASK_LOCOMOTION) return;
  real dpos[3], dlin[3], dang[3], pr[3], lr[3], ar[3];
  v3_sub(dpos, in->obj_pos, in->root_pos);
  v3_sub(dlin, in->obj_lin, in->root_lin);
  v3_sub(dang, in->obj_ang, in->root_ang);
  quat_apply_inv(pr, in->root_quat, dpos);
  quat_apply_inv(lr, in->root_quat, dlin);
  quat_apply_inv(ar, in->root_quat, dang);
  if (w[LT_R_OBJECT_XY_POSITION] != 0)                                                          /* :469-481 world frame, x[cmd>0] */
    terms[LT_R_OBJECT_XY_POSITION] = (real)sqrt((double)(dpos[0] * dpos[0] + dpos[1] * dpos[1])) * (cn > 0 ? 1 : 0);
  if (w[LT_R_OBJECT_XY_VELOCITY] != 0) terms[LT_R_OBJECT_XY_VELOCITY] = lr[0] * lr[0] + lr[1] * lr[1]; /* :483-491 */
  if (w[LT_R_OBJECT_Z_CONTACT] != 0)                                                            /* :596-604 */
    terms[LT_R_OBJECT_Z_CONTACT] = (in->obj_timers[3] > 0 && in->obj_timers[0] > 0) ? 1 : 0;
  if (w[LT_R_OBJECT_Z_VELOCITY] != 0) terms[LT_R_OBJECT_Z_VELOCITY] = lr[2] * lr[2];            /* :493-501 */
  if (w[LT_R_OBJECT_ROLL_PITCH_ANGLE] != 0 || w[LT_R_OBJECT_YAW_ALIGNMENT] != 0) {
    /* :524-533 : world gravity direction seen from the object, re-expressed in the robot frame */
    real gob[3], gw[3], gr[3];
    quat_apply_inv(gob, in->obj_quat, gz);
    quat_apply(gw, in->obj_quat, gob);
    quat_apply_inv(gr, in->root_quat, gw);
    if (w[LT_R_OBJECT_ROLL_PITCH_ANGLE] != 0) terms[LT_R_OBJECT_ROLL_PITCH_ANGLE] = gr[1] * gr[1];
  }
  if (w[LT_R_OBJECT_ROLL_PITCH_VELOCITY] != 0) terms[LT_R_OBJECT_ROLL_PITCH_VELOCITY] = ar[0] * ar[0]; /* :535-543 */
  if (w[LT_R_OBJECT_YAW_ALIGNMENT] != 0) {                                                      /* :545-567 */
    real ry = quat_yaw_2pi(in->root_quat), oy = quat_yaw_2pi(in->obj_quat), qr[4], qo[4], qi[4], qd_[4];
    quat_from_euler(qr, 0, 0, ry);
    quat_from_euler(qo, 0, 0, oy);
    quat_conj(qi, qr);
    quat_mul(qd_, qi, qo);
    real yd = quat_yaw_2pi(qd_);
    const real pi = (real)LT_PI;
    if (yd > pi) yd -= 2 * pi;
    if (yd > (real)0.5 * pi) yd -= pi;
    if (yd <= (real)-0.5 * pi) yd += pi;
    terms[LT_R_OBJECT_YAW_ALIGNMENT] = yd * yd * (cn > 0 ? 1 : 0);
  }
  if (w[LT_R_OBJECT_DANGEROUS_STATE] != 0) {                                                    /* :569-594 (roll_pitch_max=None) */
    int danger = (fabs((double)pr[0]) > cfg->danger_x_max) || (fabs((double)pr[1]) > cfg->danger_y_max) || (pr[2] < cfg->danger_z_min);
    danger |= sqrt((double)(lr[0] * lr[0] + lr[1] * lr[1])) > cfg->danger_vel_xy_max;
    terms[LT_R_OBJECT_DANGEROUS_STATE] = danger ? 1 : 0;
  }
}

/* terminations: stock terms [DEP] (cfg locomotion_base_env_cfg.py:296-313) + reference mdp/terminations.py:10-23 */
int lt_oracle_terminations(const lt_cfg* cfg, const lt_term_in* in, int64_t ep_len, int64_t max_len) {
  int bits = 0;
  real gz[3] = {0, 0, -1}, gb[3];
  quat_apply_inv(gb, in->root_quat, gz);
  if (cfg->term_enabled[LT_T_TIME_OUT] && ep_len >= max_len) bits |= 1 << LT_T_TIME_OUT;
  if (cfg->term_enabled[LT_T_BASE_ORIENTATION] && (real)acos((double)(-gb[2])) > cfg->term_orientation_limit) bits |= 1 << LT_T_BASE_ORIENTATION;
  if (cfg->term_enabled[LT_T_BASE_HEIGHT] && in->root_pos[2] < cfg->term_min_height) bits |= 1 << LT_T_BASE_HEIGHT;
  if (cfg->term_enabled[LT_T_BASE_CONTACT]) {
    real mx = in->trunk_fhist[0];
    if (in->trunk_fhist[1] > mx) mx = in->trunk_fhist[1];
    if (in->trunk_fhist[2] > mx) mx = in->trunk_fhist[2];
    if (mx > cfg->term_contact_threshold) bits |= 1 << LT_T_BASE_CONTACT;
  }
  if (cfg->term_enabled[LT_T_HIP_CONTACT])
    for (int l = 0; l < 4; ++l) {
      real mx = in->fhist[0][0][l];
      if (in->fhist[1][0][l] > mx) mx = in->fhist[1][0][l];
      if (in->fhist[2][0][l] > mx) mx = in->fhist[2][0][l];
      if (mx > cfg->term_contact_threshold) bits |= 1 << LT_T_HIP_CONTACT;
    }
  if (cfg->task != LT_TASK_LOCOMOTION) {
    if (cfg->term_enabled[LT_T_OBJECT_BELOW_ROBOT] && in->obj_pos[2] < in->root_pos[2]) bits |= 1 << LT_T_OBJECT_BELOW_ROBOT; /* terminations.py:10-17 */
    if (cfg->term_enabled[LT_T_OBJECT_BAD_ROLL]) {                                              /* terminations.py:19-23 */
      real go[3];
      quat_apply_inv(go, in->obj_quat, gz);
      if ((real)fabs(asin((double)go[1])) > cfg->term_object_roll_limit) bits |= 1 << LT_T_OBJECT_BAD_ROLL;
    }
  }
  return bits;
}

/* object_state_in_robot_frame, reference locotouch/mdp/observations.py:38-91.  `noise16` = 16 uniforms
 * (13 additive + 3 euler) or NULL for the critic flavour. */
void lt_oracle_object_state_obs(const lt_cfg* cfg, const lt_term_in* in, const float* noise16, float out[13]) {
  real dpos[3], dlin[3], dang[3], s[13], qi[4];
  v3_sub(dpos, in->obj_pos, in->root_pos);
  v3_sub(dlin, in->obj_lin, in->root_lin);
  v3_sub(dang, in->obj_ang, in->root_ang);
  quat_apply_inv(s, in->root_quat, dpos);                                                       /* :55 */
  quat_apply_inv(s + 3, in->root_quat, dlin);                                                   /* :56 */
  quat_conj(qi, in->root_quat);
  quat_mul(s + 6, qi, in->obj_quat);                                                            /* :57 */
  quat_apply_inv(s + 10, in->root_quat, dang);                                                  /* :58 */
  int non_contact = (in->obj_timers[3] < cfg->obj_contact_time_threshold) && (in->obj_timers[1] < cfg->obj_contact_time_threshold); /* :65 */
  if (non_contact) { for (int i = 0; i < 13; ++i) s[i] = 0; s[6] = 1; }                         /* :68-70,:89 */
  if (noise16) {                                                                                /* :71-83 */
    static const int nidx[13] = {0, 1, 2, 3, 4, 5, -1, -1, -1, -1, 9, 10, 11};
    for (int i = 0; i < 13; ++i)
      if (nidx[i] >= 0) { real n = cfg->obj_noise[nidx[i]]; s[i] += (real)noise16[i] * (2 * n) - n; }
    real e[3], qn[4], qq[4];
    for (int i = 0; i < 3; ++i) { real n = cfg->obj_noise[6 + i]; e[i] = (real)noise16[13 + i] * (2 * n) - n; }
    quat_from_euler(qn, e[0], e[1], e[2]);
    quat_mul(qq, s + 6, qn);
    memcpy(s + 6, qq, sizeof(qq));
  }
  for (int i = 0; i < 13; ++i) out[i] = s[i] * cfg->obj_scale[i];                               /* :85-89 */
}

/* ------------------------------------------------------------------------------------------------ */
/* K7: command term (reference locotouch/mdp/commands.py:517-576 + UniformVelocityCommand [DEP])     */
/* ------------------------------------------------------------------------------------------------ */
/* explicit-uniform form (pinned by tests/golden/mdp_replay.npz, which replays recorded uniforms through the reference's own
 * _resample_command): ub = bin draws (torch.multinomial by inverse CDF over (p, 1-2p, p)), uv = value draws, ustand = the
 * standing draw, utime = the resampling-period draw [DEP CommandTerm]. */
void lt_oracle_command_resample_u(const lt_cfg* cfg, const float* P, const float ub[3], const float uv[3], float ustand, float utime,
                                  float cmd[3], float cmd_buf[3], float* standing, float* time_left) {
  for (int d = 0; d < 3; ++d) {
    real lo = P[2 * d], hi = P[2 * d + 1];
    if (cfg->cmd_multi_sampling && P[12 + d] == 0) {                                            /* commands.py:530-553 */
      real plo = P[6 + 2 * d], phi = P[6 + 2 * d + 1];
      real p = cfg->cmd_new_probs;
      if (ub[d] < p) { hi = plo; }
      else if (ub[d] < 1 - p) { lo = plo; hi = phi; }
      else { lo = phi; }
    }
    cmd[d] = lo + (real)uv[d] * (hi - lo);
  }
  if (cfg->cmd_binary_maximal) {  /* :518-521: one of the 8 sign combinations (index 4 i + 2 j + k over [-1, 1]) times the upper bounds */
    int idx = (int)(ub[0] * 8.0f);
    if (idx > 7) idx = 7;
    cmd[0] = (idx & 4) ? P[1] : -P[1]; cmd[1] = (idx & 2) ? P[3] : -P[3]; cmd[2] = (idx & 1) ? P[5] : -P[5];
  } else {
    *standing = (ustand <= P[16]) ? 1 : 0;                                                      /* :555 */
  }
  *time_left = cfg->cmd_resample_time[0] + (real)utime * (cfg->cmd_resample_time[1] - cfg->cmd_resample_time[0]);
  for (int d = 0; d < 3; ++d) cmd_buf[d] = cmd[d];                                              /* :558 */
}
static void command_resample(const lt_cfg* cfg, const float* P, uint64_t seed, uint32_t env, uint64_t step, uint32_t stream,
                             real cmd[3], real cmd_buf[3], real* standing, real* time_left) {
  float u0[4], u1[4];
  lt_rng4(seed, env, step, stream, u0);
  lt_rng4(seed, env, step, stream + 1, u1);
  const float ub[3] = {u0[0], u0[2], u1[0]}, uv[3] = {u0[1], u0[3], u1[1]};
  lt_oracle_command_resample_u(cfg, P, ub, uv, u1[2], u1[3], cmd, cmd_buf, standing, time_left);
}
/* commands.py:561-576 + base class standing zero */
void lt_oracle_command_update(int64_t ep_len, int zero_steps, const float buf[3], int standing, float cmd[3]) {
  if (ep_len < zero_steps) for (int d = 0; d < 3; ++d) cmd[d] = buf[d] * 0.0f;
  if (ep_len == zero_steps) for (int d = 0; d < 3; ++d) cmd[d] = buf[d];
  if (standing) for (int d = 0; d < 3; ++d) cmd[d] = 0.0f;
}

/* ------------------------------------------------------------------------------------------------ */
/* U1: velocity curriculum (reference locotouch/mdp/curriculums.py:184-275, commands.py:471-505)     */
/* ------------------------------------------------------------------------------------------------ */
void lt_oracle_cmd_params_init(const lt_cfg* cfg, float* P) {
  memset(P, 0, sizeof(float) * LT_CMD_PARAMS_LEN);
  for (int d = 0; d < 3; ++d) {
    P[2 * d] = cfg->cmd_range_init[d][0]; P[2 * d + 1] = cfg->cmd_range_init[d][1];
    P[6 + 2 * d] = P[2 * d]; P[6 + 2 * d + 1] = P[2 * d + 1];
    P[12 + d] = 1;
    P[21 + d] = cfg->cur_enabled ? (cfg->cmd_range_max[d] - cfg->cmd_range_init[d][1]) / (float)cfg->cur_bins[d] : 0; /* :191-193 */
  }
  P[15] = (float)cfg->cmd_zero_steps;
  P[16] = cfg->cmd_rel_standing;
}
static void set_range(const lt_cfg* cfg, float* P, int d, float lo, float hi) {                  /* commands.py:471-491 */
  P[6 + 2 * d] = P[2 * d]; P[6 + 2 * d + 1] = P[2 * d + 1];
  P[2 * d] = lo; P[2 * d + 1] = hi;
  P[12 + d] = (P[6 + 2 * d] == P[2 * d] && P[6 + 2 * d + 1] == P[2 * d + 1]) ? 1 : 0;
  (void)cfg;
}
static void set_ranges_done(const lt_cfg* cfg, float* P) {                                       /* commands.py:497-503 */
  if (P[12] != 0 && P[13] != 0 && P[14] != 0) {
    P[15] = (float)cfg->cmd_zero_steps_final;
    P[16] = cfg->cmd_rel_standing_final;
  }
}
static float clipf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* One curriculum call for a reset batch.  rec = per-env (reset_flag, ep_len, sum_lin, sum_ang) of this step;
 * trk = per-env trackers (reset_lin, len_lin, sum_lin, reset_ang, len_ang, sum_ang, _, _).
 * ops (may be NULL) receives what the call did to the trackers: [0] merged the records into the lin trackers,
 * [1] ... into the ang trackers, [2] cleared the lin trackers, [3] cleared the ang trackers. */
static void curriculum_core(const lt_cfg* cfg, float* P, int64_t n, const float* rec, float* trk, int ops[4]) {
  int dummy[4];
  if (!ops) ops = dummy;
  ops[0] = ops[1] = ops[2] = ops[3] = 0;
  if (!cfg->cur_enabled) return;
  int any = 0;
  for (int64_t e = 0; e < n; ++e) any |= rec[e * 4] != 0;
  if (!any) return; /* _reset_idx (and with it the curriculum) only runs when some env resets */
  const float* mx = cfg->cmd_range_max;
  int lin_open = (P[1] != mx[0] || P[12] == 0 || P[3] != mx[1] || P[13] == 0) && (P[17] - P[18] <= (float)cfg->cur_max_distance_bins); /* :218-220 */
  if (lin_open) {
    int all = 1; double sl = 0, sr = 0;
    ops[0] = 1;
    for (int64_t e = 0; e < n; ++e) {
      if (rec[e * 4] != 0) { trk[e * 8 + 0] = 1; trk[e * 8 + 1] = rec[e * 4 + 1]; trk[e * 8 + 2] = rec[e * 4 + 2]; } /* :221-223 */
      all &= trk[e * 8 + 0] != 0; sl += trk[e * 8 + 1]; sr += trk[e * 8 + 2];
    }
    if (all && (float)(sl / (double)n) > cfg->cur_len_threshold && (float)(sr / (double)n) > cfg->cur_reward_threshold[0]) { /* :224 */
      P[19] += 1;
      if ((int)P[19] == cfg->cur_repeat_times[0]) {                                             /* :226-235 */
        float lx = clipf(P[0] - P[21], -mx[0], 0.f), ly = clipf(P[2] - P[22], -mx[1], 0.f);
        set_range(cfg, P, 0, lx, -lx);
        set_range(cfg, P, 1, ly, -ly);
        set_ranges_done(cfg, P);
        P[19] = 0; P[17] += 1;
      }
      ops[2] = 1;
      for (int64_t e = 0; e < n; ++e) { trk[e * 8 + 0] = 0; trk[e * 8 + 1] = 0; trk[e * 8 + 2] = 0; } /* :236-238 */
    }
  }
  int ang_open = (P[5] != mx[2] || P[14] == 0) && (P[18] - P[17] <= (float)cfg->cur_max_distance_bins); /* :239-240 */
  if (ang_open) {
    int all = 1; double sl = 0, sr = 0;
    ops[1] = 1;
    for (int64_t e = 0; e < n; ++e) {
      if (rec[e * 4] != 0) { trk[e * 8 + 3] = 1; trk[e * 8 + 4] = rec[e * 4 + 1]; trk[e * 8 + 5] = rec[e * 4 + 3]; }
      all &= trk[e * 8 + 3] != 0; sl += trk[e * 8 + 4]; sr += trk[e * 8 + 5];
    }
    if (all && (float)(sl / (double)n) > cfg->cur_len_threshold && (float)(sr / (double)n) > cfg->cur_reward_threshold[1]) {
      P[20] += 1;
      if ((int)P[20] == cfg->cur_repeat_times[1]) {
        float lz = clipf(P[4] - P[23], -mx[2], 0.f);
        set_range(cfg, P, 2, lz, -lz);
        set_ranges_done(cfg, P);
        P[20] = 0; P[18] += 1;
      }
      ops[3] = 1;
      for (int64_t e = 0; e < n; ++e) { trk[e * 8 + 3] = 0; trk[e * 8 + 4] = 0; trk[e * 8 + 5] = 0; }
    }
  }
  P[24] = (float)lin_open; P[25] = (float)ang_open;
}
void lt_oracle_curriculum(const lt_cfg* cfg, float* P, int64_t n, const float* rec, float* trk) { curriculum_core(cfg, P, n, rec, trk, NULL); }

/* Arena form of the curriculum state (shared with the HIP library, whose pass runs at the tail of the step kernel where
 * no wave may touch another wave's envs): the trackers in the arena lag the reference's by one pass, and
 * LT_F_CMD_PARAMS[27..30] holds the operations that pass decided - (merge lin, merge ang, clear lin, clear ang) - to be
 * applied, on the record that pass saw, by the next one.  The decisions themselves are curriculum_core's, i.e. the
 * reference's (pinned by tests/golden/mdp_curriculum.npz). */
static void curriculum_apply_pending(void* arena, const lt_layout* L) {
  const float* P = (const float*)((char*)arena + L->off_cmd_params);
  const float* rec = lt_quad(arena, L, LT_F_CURRICULUM, 0);
  float* t1 = lt_quad(arena, L, LT_F_CURRICULUM, 1);
  float* t2 = lt_quad(arena, L, LT_F_CURRICULUM, 2);
  for (int64_t e = 0; e < L->npad; ++e) {  /* padded tail envs never reset: their records stay zero */
    const float* r = rec + e * 4;
    const int had = r[0] != 0;
    if (P[27] != 0 && had) { t1[e * 4 + 0] = 1; t1[e * 4 + 1] = r[1]; t1[e * 4 + 2] = r[2]; }
    if (P[29] != 0) { t1[e * 4 + 0] = 0; t1[e * 4 + 1] = 0; t1[e * 4 + 2] = 0; }
    if (P[28] != 0 && had) { t1[e * 4 + 3] = 1; t2[e * 4 + 0] = r[1]; t2[e * 4 + 1] = r[3]; }
    if (P[30] != 0) { t1[e * 4 + 3] = 0; t2[e * 4 + 0] = 0; t2[e * 4 + 1] = 0; }
  }
}
/* The decision sequence of curriculum_core on POPULATION SUMS instead of per-env arrays (what a multi-rank run has after its
 * all-reduce): r = (groups with a non-zero command, groups with a reset, groups whose lin trackers are not all reset, sum ep_len
 * lin, sum reward lin, same three for ang), trackers already merged.  tests/test_oracle_gate.py checks it against
 * curriculum_core on the same population.  out = (run, lin_open, lin_pass, ang_open, ang_pass). */
void lt_oracle_gate_on_sums(const lt_cfg* cfg, float* P, const float r[8], float inv_n, int allow_lin, int allow_ang, int out[5]) {
  const float* mx = cfg->cmd_range_max;
  out[0] = cfg->cur_enabled != 0 && r[1] > 0;
  out[1] = out[0] && (P[1] != mx[0] || P[12] == 0 || P[3] != mx[1] || P[13] == 0) && (P[17] - P[18] <= (float)cfg->cur_max_distance_bins);
  out[2] = 0;
  if (out[1] && allow_lin) {
    out[2] = r[2] == 0 && r[3] * inv_n > cfg->cur_len_threshold && r[4] * inv_n > cfg->cur_reward_threshold[0];
    if (out[2]) {
      P[19] += 1;
      if ((int)P[19] == cfg->cur_repeat_times[0]) {
        float lx = clipf(P[0] - P[21], -mx[0], 0.f), ly = clipf(P[2] - P[22], -mx[1], 0.f);
        set_range(cfg, P, 0, lx, -lx);
        set_range(cfg, P, 1, ly, -ly);
        set_ranges_done(cfg, P);
        P[19] = 0; P[17] += 1;
      }
    }
  }
  out[3] = out[0] && (P[5] != mx[2] || P[14] == 0) && (P[18] - P[17] <= (float)cfg->cur_max_distance_bins);
  out[4] = 0;
  if (out[3] && allow_ang) {
    out[4] = r[5] == 0 && r[6] * inv_n > cfg->cur_len_threshold && r[7] * inv_n > cfg->cur_reward_threshold[1];
    if (out[4]) {
      P[20] += 1;
      if ((int)P[20] == cfg->cur_repeat_times[1]) {
        float lz = clipf(P[4] - P[23], -mx[2], 0.f);
        set_range(cfg, P, 2, lz, -lz);
        set_ranges_done(cfg, P);
        P[20] = 0; P[18] += 1;
      }
    }
  }
}

/* population sums of one pass, per 16-env group as the HIP kernel's waves form them (trackers with this step's records merged) */
static void curriculum_sums(void* arena, const lt_layout* L, float r[8]) {
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const float* rec = lt_quad(arena, L, LT_F_CURRICULUM, 0);
  const float* t1 = lt_quad(arena, L, LT_F_CURRICULUM, 1);
  const float* t2 = lt_quad(arena, L, LT_F_CURRICULUM, 2);
  for (int64_t g0 = 0; g0 < L->npad; g0 += 16) {
    int nz = 0, any = 0, nl = 0, na = 0;
    for (int64_t e = g0; e < g0 + 16 && e < L->n; ++e) {
      const float* cm = lt_quad(arena, L, LT_F_CMD, 0) + e * 4;
      const int reset = rec[e * 4] != 0;
      nz |= cm[0] != 0 || cm[1] != 0 || cm[2] != 0;
      any |= reset;
      nl |= !(reset || t1[e * 4 + 0] != 0);
      na |= !(reset || t1[e * 4 + 3] != 0);
      acc[3] += reset ? rec[e * 4 + 1] : t1[e * 4 + 1];
      acc[4] += reset ? rec[e * 4 + 2] : t1[e * 4 + 2];
      acc[6] += reset ? rec[e * 4 + 1] : t2[e * 4 + 0];
      acc[7] += reset ? rec[e * 4 + 3] : t2[e * 4 + 1];
    }
    acc[0] += nz; acc[1] += any; acc[2] += nl; acc[5] += na;
  }
  for (int i = 0; i < 8; ++i) r[i] = (float)acc[i];
}

/* the pass proper on the (now up-to-date) trackers and this step's records: decisions into P, tracker operations deferred */
static void curriculum_decide(const lt_cfg* cfg, void* arena, const lt_layout* L) {
  float* P = (float*)((char*)arena + L->off_cmd_params);
  int64_t n = L->n;
  float* rec = (float*)malloc(sizeof(float) * 4 * (size_t)n);
  float* trk = (float*)malloc(sizeof(float) * 8 * (size_t)n);
  for (int64_t e = 0; e < n; ++e)
    for (int c = 0; c < 4; ++c) {
      rec[e * 4 + c] = lt_quad(arena, L, LT_F_CURRICULUM, 0)[e * 4 + c];
      /* tracker memory order in the arena: (reset_lin,len_lin,sum_lin,reset_ang),(len_ang,sum_ang,_,_) */
      trk[e * 8 + c] = lt_quad(arena, L, LT_F_CURRICULUM, 1)[e * 4 + c];
      trk[e * 8 + 4 + c] = lt_quad(arena, L, LT_F_CURRICULUM, 2)[e * 4 + c];
    }
  int ops[4];
  if (cfg->cur_gate_external) {  /* multi-rank: trackers merge as usual, the success test waits for the cross-rank sums */
    lt_cfg never = *cfg;
    never.cur_len_threshold = INFINITY;
    curriculum_core(&never, P, n, rec, trk, ops);
  } else {
    curriculum_core(cfg, P, n, rec, trk, ops);  /* works on the copy: the arena's trackers are updated by the next pass */
  }
  P[27] = (float)ops[0]; P[28] = (float)ops[1]; P[29] = (float)ops[2]; P[30] = (float)ops[3];
  free(rec); free(trk);
  {
    int64_t* cnt = (int64_t*)((char*)arena + L->off_counters);
    curriculum_sums(arena, L, (float*)((char*)arena + L->off_gate_ring) + (cnt[3] % LT_GATE_RING) * LT_PARTIAL_FLOATS);
    cnt[3] += 1;
  }
}
/* multi-rank gate (HIP twin: lt_env_curriculum_apply_global): the decision sequence over the last `nsteps` passes on sums
 * all-reduced over the ranks; a group is closed after its first success in the window and its tracker clear is scheduled */
int lt_oracle_curriculum_apply_global(const lt_cfg* cfg, void* arena, const float* ring_sums, int nsteps, int64_t n_total) {
  lt_layout L;
  lt_layout_init(&L, cfg->num_envs, (cfg->task == LT_TASK_LOCOMOTION ? 45 : 58) * cfg->obs_history, cfg->tactile_enabled);
  float* P = (float*)((char*)arena + L.off_cmd_params);
  const int64_t pos = ((int64_t*)((char*)arena + L.off_counters))[3];
  const float inv_n = 1.0f / (float)n_total;
  int lin_done = 0, ang_done = 0;
  for (int k = 0; k < nsteps; ++k) {
    const int64_t pass = pos - nsteps + k;
    if (pass < 0) continue;
    int out[5];
    lt_oracle_gate_on_sums(cfg, P, ring_sums + (pass % LT_GATE_RING) * LT_PARTIAL_FLOATS, inv_n, !lin_done, !ang_done, out);
    lin_done |= out[2];
    ang_done |= out[4];
  }
  if (lin_done) P[29] = 1;
  if (ang_done) P[30] = 1;
  return 0;
}
/* HIP-hook twin (lt_env_curriculum_update): one pass on caller-supplied records, no step-counter increment */
int lt_oracle_curriculum_update(const lt_cfg* cfg, void* arena, const float* records) {
  lt_layout L;
  lt_layout_init(&L, cfg->num_envs, (cfg->task == LT_TASK_LOCOMOTION ? 45 : 58) * cfg->obs_history, cfg->tactile_enabled);
  curriculum_apply_pending(arena, &L);
  float* rec = lt_quad(arena, &L, LT_F_CURRICULUM, 0);
  for (int64_t e = 0; e < L.n; ++e)
    for (int c = 0; c < 4; ++c) rec[e * 4 + c] = records[e * 4] != 0 ? (c == 0 ? 1.0f : records[e * 4 + c]) : 0.0f;
  curriculum_decide(cfg, arena, &L);
  int nz = 0;
  for (int64_t e = 0; e < L.n; ++e) {
    const float* cm = lt_quad(arena, &L, LT_F_CMD, 0) + e * 4;
    nz |= cm[0] != 0 || cm[1] != 0 || cm[2] != 0;
  }
  ((float*)((char*)arena + L.off_cmd_params))[26] = nz ? 1.0f : 0.0f;
  return 0;
}

/* ------------------------------------------------------------------------------------------------ */
/* K6: reset path (events in cfg declaration order)                           */
/* ------------------------------------------------------------------------------------------------ */
enum {
  RS_NOISE_JPOS = 0x100, RS_NOISE_JVEL = 0x110, RS_NOISE_BASE = 0x120, RS_NOISE_OBJ = 0x130,
  RS_RESET_ROOT = 0x200, RS_RESET_JOINT = 0x210, RS_RESET_MAT = 0x220, RS_RESET_OBJ = 0x221, RS_RESET_EVENT = 0x223,
  RS_CMD_RESET = 0x230, RS_CMD_TIMER = 0x240, RS_PUSH_ROBOT = 0x250, RS_PUSH_OBJ = 0x260, RS_STARTUP = 0x300,
  RS_BUCKET_FEET = 0x310, RS_BUCKET_OBJ = 0x311, /* material pools: key = bucket index (not an env), startup stream */
  RS_TACTILE_THR = 0x400, /* + 0x40 term + taxel / 4 (startup stream): per-(env, taxel) threshold offsets, drawn once */
  RS_TACTILE = 0x500      /* + 0x200 term + 2 taxel + {0, 1} (step stream): see tactile_pass */
};

/* RNG stream key of env e: its global index over all ranks (cfg.env_index_offset = index of this shard's env 0) */
#define EKEY(cfg, e) ((uint32_t)(e) + (uint32_t)(cfg)->env_index_offset)

static void startup_env(const lt_cfg* cfg, env_t* E, uint32_t env, const float* sizes) {
  /* startup events: trunk mass (locomotion_base_env_cfg.py:224-232), foot material (:233-244 + teacher override),
   * and the per-env cylinder size (rand_cylinder_transport_teacher_env_cfg.py:21-27; seeded here, quirk Q2) */
  float u[4];
  const uint64_t st = ~(uint64_t)0;
  lt_rng4(cfg->seed, env, st, RS_STARTUP, u);
  E->trunk_mass_add = lt_lerp(cfg->trunk_mass_add, u[0]);
  E->obj_radius = lt_lerp(cfg->obj_radius, u[1]);
  E->obj_length = lt_lerp(cfg->obj_length, u[2]);
  if (cfg->obj_size_explicit && sizes) { E->obj_radius = sizes[0]; E->obj_length = sizes[1]; }
  for (int l = 0; l < 4; ++l) {
    lt_rng4(cfg->seed, env, st, RS_STARTUP + 0x10 + l, u);
    if (cfg->foot_material_buckets > 0) { /* bucketed materials [DEP randomize_rigid_body_material]: pool entry b = the draw keyed by b */
      int b = (int)(u[0] * (float)cfg->foot_material_buckets);
      if (b > cfg->foot_material_buckets - 1) b = cfg->foot_material_buckets - 1;
      lt_rng4(cfg->seed, (uint32_t)b, st, RS_BUCKET_FEET, u);
    }
    real ms = lt_lerp(cfg->foot_friction, u[0]), md = lt_lerp(cfg->foot_friction, u[1]);
    E->foot_mu[l] = md < ms ? md : ms; /* make_consistent: dynamic = min(static, dynamic); the contact law uses it */
  }
  E->obj_mass = 1.0f; E->obj_mu = 1.0f; E->trunk_mu = 1.0f; E->trunk_rest = 0; E->obj_rest = 0;
}

/* E3 randomize_friction_restitution.__call__ (events.py:160-196): (static, dynamic, restitution) = lo + u * (hi - lo), then
 * make_consistent => dynamic = min(static, dynamic).  Explicit-uniform form, pinned by tests/golden/mdp_replay.npz. */
void lt_oracle_material_u(const float range_static[2], const float range_dynamic[2], const float range_restitution[2],
                          const float u[3], float out[3]) {
  out[0] = lt_lerp(range_static, u[0]);
  out[1] = lt_lerp(range_dynamic, u[1]);
  out[2] = lt_lerp(range_restitution, u[2]);
  if (out[0] < out[1]) out[1] = out[0];
}
/* E6 ResetObjectStateUniform.__call__ (events.py:85-109): offset added in WORLD axes (:98), + height/2 (:99), orientation
 * = robot quat (x) euler(roll, pitch, yaw) (:100-101), velocity = robot root velocity (+ zero-range samples) (:104-105).
 * cfg->obj_reset_robot_frame selects the function variant reset_object_state_uniform (:13-53): offset rotated by the robot quat.
 * Explicit-uniform form, pinned by tests/golden/mdp_replay.npz. */
void lt_oracle_reset_object_u(const lt_cfg* cfg, const float root_pos[3], const float root_quat[4], const float root_lin[3],
                              const float root_ang[3], float obj_length, const float u_pose[6], float pos[3], float quat[4],
                              float lin[3], float ang[3]) {
  real d[3], dw[3];
  for (int c = 0; c < 3; ++c) d[c] = lt_lerp(cfg->obj_reset_pos[c], u_pose[c]);
  d[2] += obj_length / 2;
  if (cfg->obj_reset_robot_frame) quat_apply(dw, root_quat, d);  /* function variant, events.py:43-44 */
  else v3_copy(dw, d);
  for (int c = 0; c < 3; ++c) pos[c] = root_pos[c] + dw[c];
  real dq[4];
  quat_from_euler(dq, lt_lerp(cfg->obj_reset_rpy[0], u_pose[3]), lt_lerp(cfg->obj_reset_rpy[1], u_pose[4]), lt_lerp(cfg->obj_reset_rpy[2], u_pose[5]));
  quat_mul(quat, root_quat, dq);
  for (int c = 0; c < 3; ++c) { lin[c] = root_lin[c]; ang[c] = root_ang[c]; }
}

static void reset_env(const lt_cfg* cfg, const float* P, env_t* E, uint32_t env, uint64_t step, int has_object) {
  float u[4], w[4];
  /* E4 reset_root_state_uniform [DEP], params locomotion_base_env_cfg.py:249-267 / teacher :144-160 */
  lt_rng4(cfg->seed, env, step, RS_RESET_ROOT, u);
  E->root_pos[0] = lt_lerp(cfg->reset_root_pos[0], u[0]);
  E->root_pos[1] = lt_lerp(cfg->reset_root_pos[1], u[1]);
  E->root_pos[2] = LT_ROOT_INIT_HEIGHT + lt_lerp(cfg->reset_root_pos[2], u[2]);
  lt_rng4(cfg->seed, env, step, RS_RESET_ROOT + 1, u);
  quat_from_euler(E->root_quat, lt_lerp(cfg->reset_root_rpy[0], u[0]), lt_lerp(cfg->reset_root_rpy[1], u[1]), lt_lerp(cfg->reset_root_rpy[2], u[2]));
  lt_rng4(cfg->seed, env, step, RS_RESET_ROOT + 2, u);
  lt_rng4(cfg->seed, env, step, RS_RESET_ROOT + 3, w);
  for (int c = 0; c < 3; ++c) { E->root_lin[c] = lt_lerp(cfg->reset_root_vel[c], u[c]); E->root_ang[c] = lt_lerp(cfg->reset_root_vel[3 + c], w[c]); }
  /* E5 reset_joints_by_offset [DEP] :269-276, clamped to the soft limits */
  for (int l = 0; l < 4; ++l) {
    lt_rng4(cfg->seed, env, step, RS_RESET_JOINT + l, u);
    lt_rng4(cfg->seed, env, step, RS_RESET_JOINT + 4 + l, w);
    for (int k = 0; k < 3; ++k) {
      real mid = (k_joint_lo[k] + k_joint_hi[k]) / 2, rng = k_joint_hi[k] - k_joint_lo[k];
      real lo = mid - (real)0.5 * rng * LT_SOFT_LIMIT_FACTOR, hi = mid + (real)0.5 * rng * LT_SOFT_LIMIT_FACTOR;
      E->q[l][k] = clampr(k_joint_default[l][k] + lt_lerp(cfg->reset_joint_pos, u[k]), lo, hi);
      E->qd[l][k] = clampr(lt_lerp(cfg->reset_joint_vel, w[k]), -cfg->velocity_limit, cfg->velocity_limit);
      E->qdd[l][k] = 0; E->tau[l][k] = 0;
      E->act_raw[l][k] = 0; E->act_prev[l][k] = 0; E->act_prev2[l][k] = 0;                       /* actions.py:46-52 */
    }
    for (int s = 0; s < 3; ++s) for (int t = 0; t < 4; ++t) E->fhist[s][t][l] = 0;
    E->foot_cur_air[l] = E->foot_cur_con[l] = E->foot_last_air[l] = E->foot_last_con[l] = 0;
    E->gait_last_air[l] = E->gait_last_con[l] = E->gait_valid_last_air[l] = 0; E->gait_flags[l] = 0; /* rewards.py:107-114 */
  }
  for (int c = 0; c < 3; ++c) { E->trunk_fhist[c] = 0; E->gait_cmd[c] = 0; }
  E->gait_step_from_change = 0;
  if (has_object) {
    /* E3 randomize_friction_restitution (events.py:160-196, make_consistent) and E2 object material */
    lt_rng4(cfg->seed, env, step, RS_RESET_MAT, u);
    if (cfg->obj_material_buckets > 0) { /* E2: the object's (friction, restitution) come from a pool of obj_material_buckets entries */
      float ub[4];
      int b = (int)(u[2] * (float)cfg->obj_material_buckets);
      if (b > cfg->obj_material_buckets - 1) b = cfg->obj_material_buckets - 1;
      lt_rng4(cfg->seed, (uint32_t)b, ~(uint64_t)0, RS_BUCKET_OBJ, ub);
      u[2] = ub[0]; u[3] = ub[1];
    }
    {
      /* the dynamic-friction range of both cfgs is (1, 1): its draw does not matter (object_transport_teacher_env_cfg.py:121-143) */
      const float one[2] = {1.0f, 1.0f};
      const float ut[3] = {u[0], 0.0f, u[1]}, uo[3] = {u[2], 0.0f, u[3]};
      float m[3];
      lt_oracle_material_u(cfg->trunk_friction, one, cfg->trunk_restitution, ut, m);
      E->trunk_mu = m[1]; E->trunk_rest = m[2];       /* the contact law takes the consistent (dynamic) coefficient */
      lt_oracle_material_u(cfg->obj_friction, one, cfg->obj_restitution, uo, m);
      E->obj_mu = m[1]; E->obj_rest = m[2];
    }
    /* E6 ResetObjectStateUniform.__call__ (events.py:85-109): world-axis offset, + length/2, robot velocity */
    lt_rng4(cfg->seed, env, step, RS_RESET_OBJ, u);
    lt_rng4(cfg->seed, env, step, RS_RESET_OBJ + 1, w);
    {
      const float up[6] = {u[0], u[1], u[2], w[0], w[1], w[2]};
      lt_oracle_reset_object_u(cfg, E->root_pos, E->root_quat, E->root_lin, E->root_ang, E->obj_length, up, E->obj_pos, E->obj_quat,
                               E->obj_lin, E->obj_ang);
    }
    /* E1 object mass: default 1.0 + U (operation add on the default) */
    E->obj_mass = 1.0f + lt_lerp(cfg->obj_mass_add, w[3]);
    for (int c = 0; c < 4; ++c) E->obj_timers[c] = 0;
  }
  /* manager resets: reward sums, command, interval-event timers */
  for (int i = 0; i < LT_REWARD_SLOTS; ++i) E->sums[i] = 0;
  command_resample(cfg, P, cfg->seed, env, step, RS_CMD_RESET, E->cmd, E->cmd_buf, &E->cmd_standing, &E->cmd_time_left);
  lt_rng4(cfg->seed, env, step, RS_RESET_EVENT, u);
  E->push_robot_left = lt_lerp(cfg->push_robot_interval, u[0]);
  E->push_obj_left = lt_lerp(cfg->push_obj_interval, u[1]);
  E->ep_len = 0;
  foot_kinematics(E);
}

/* ------------------------------------------------------------------------------------------------ */
/* K9: observations                                                                                   */
/* ------------------------------------------------------------------------------------------------ */
/* newest frame of every term, policy (noisy) and critic flavours; term order locomotion_base_env_cfg.py:74-109
 * then object_state (object_transport_teacher_env_cfg.py:37-43).  Noise model: AdditiveUniformNoise [DEP]. */
static int obs_frame(const lt_cfg* cfg, const env_t* E, uint32_t env, uint64_t step, int has_object, float* pol, float* cri) {
  real R0[9], wb[3], gb[3], gz[3] = {0, 0, -1};
  quat_to_mat(R0, E->root_quat);
  m3_tmulv(wb, R0, E->root_ang);
  m3_tmulv(gb, R0, gz);
  int n = 0;
  float ua[4], ug[4], uj[4][4], uv[4][4];
  lt_rng4(cfg->seed, env, step, RS_NOISE_BASE, ua);
  lt_rng4(cfg->seed, env, step, RS_NOISE_BASE + 1, ug);
  for (int l = 0; l < 4; ++l) { lt_rng4(cfg->seed, env, step, RS_NOISE_JPOS + l, uj[l]); lt_rng4(cfg->seed, env, step, RS_NOISE_JVEL + l, uv[l]); }
  int noisy = cfg->enable_corruption;
#define NZ(u, amp) (noisy ? ((real)(u) * (2 * (amp)) - (amp)) : 0)
  for (int c = 0; c < 3; ++c) { pol[n] = E->cmd[c]; cri[n] = E->cmd[c]; ++n; }
  for (int c = 0; c < 3; ++c) { cri[n] = wb[c] * cfg->obs_scale_ang_vel; pol[n] = (wb[c] + NZ(ua[c], cfg->obs_noise_ang_vel)) * cfg->obs_scale_ang_vel; ++n; }
  for (int c = 0; c < 3; ++c) { cri[n] = gb[c]; pol[n] = gb[c] + NZ(ug[c], cfg->obs_noise_gravity); ++n; }
  for (int k = 0; k < 3; ++k) for (int l = 0; l < 4; ++l) { real x = E->q[l][k] - k_joint_default[l][k]; cri[n] = x; pol[n] = x + NZ(uj[l][k], cfg->obs_noise_joint_pos); ++n; }
  for (int k = 0; k < 3; ++k) for (int l = 0; l < 4; ++l) { real x = E->qd[l][k]; cri[n] = x * cfg->obs_scale_joint_vel; pol[n] = (x + NZ(uv[l][k], cfg->obs_noise_joint_vel)) * cfg->obs_scale_joint_vel; ++n; }
  for (int k = 0; k < 3; ++k) for (int l = 0; l < 4; ++l) { pol[n] = E->act_raw[l][k]; cri[n] = E->act_raw[l][k]; ++n; }
#undef NZ
  if (has_object) {
    lt_term_in in;
    memset(&in, 0, sizeof(in));
    memcpy(in.root_pos, E->root_pos, sizeof(in.root_pos)); memcpy(in.root_quat, E->root_quat, sizeof(in.root_quat));
    memcpy(in.root_lin, E->root_lin, sizeof(in.root_lin)); memcpy(in.root_ang, E->root_ang, sizeof(in.root_ang));
    memcpy(in.obj_pos, E->obj_pos, sizeof(in.obj_pos)); memcpy(in.obj_quat, E->obj_quat, sizeof(in.obj_quat));
    memcpy(in.obj_lin, E->obj_lin, sizeof(in.obj_lin)); memcpy(in.obj_ang, E->obj_ang, sizeof(in.obj_ang));
    memcpy(in.obj_timers, E->obj_timers, sizeof(in.obj_timers));
    float u16[16];
    for (int b = 0; b < 4; ++b) lt_rng4(cfg->seed, env, step, RS_NOISE_OBJ + b, u16 + 4 * b);
    lt_oracle_object_state_obs(cfg, &in, noisy ? u16 : NULL, pol + n);
    lt_oracle_object_state_obs(cfg, &in, NULL, cri + n);
    n += 13;
  }
  return n;
}
/* ObservationManager history [DEP, SURVEY.md Appendix C]: per term a 6-deep ring flattened oldest->newest,
 * terms concatenated term-major; the first push after a reset fills every slot. */
void lt_oracle_obs_push(const int* term_dims, int nterms, int hist, const float* frame, int fill, float* row) {
  int off = 0, fo = 0;
  for (int t = 0; t < nterms; ++t) {
    int d = term_dims[t];
    for (int s = 0; s < hist; ++s)
      for (int k = 0; k < d; ++k) {
        if (fill || s == hist - 1) row[off + s * d + k] = frame[fo + k];
        else row[off + s * d + k] = row[off + (s + 1) * d + k];
      }
    off += hist * d; fo += d;
  }
}

/* ------------------------------------------------------------------------------------------------ */
/* full env step                                                                                      */
/* ------------------------------------------------------------------------------------------------ */
static void term_in_from_env(const env_t* E, int terminated, lt_term_in* in) {
  memset(in, 0, sizeof(*in));
  memcpy(in->root_pos, E->root_pos, sizeof(in->root_pos)); memcpy(in->root_quat, E->root_quat, sizeof(in->root_quat));
  memcpy(in->root_lin, E->root_lin, sizeof(in->root_lin)); memcpy(in->root_ang, E->root_ang, sizeof(in->root_ang));
  memcpy(in->q, E->q, sizeof(in->q)); memcpy(in->qd, E->qd, sizeof(in->qd)); memcpy(in->qdd, E->qdd, sizeof(in->qdd));
  memcpy(in->tau, E->tau, sizeof(in->tau)); memcpy(in->act_raw, E->act_raw, sizeof(in->act_raw));
  memcpy(in->act_prev, E->act_prev, sizeof(in->act_prev)); memcpy(in->fhist, E->fhist, sizeof(in->fhist));
  memcpy(in->trunk_fhist, E->trunk_fhist, sizeof(in->trunk_fhist));
  memcpy(in->foot_pos, E->foot_pos, sizeof(in->foot_pos)); memcpy(in->foot_vel, E->foot_vel, sizeof(in->foot_vel));
  memcpy(in->obj_pos, E->obj_pos, sizeof(in->obj_pos)); memcpy(in->obj_quat, E->obj_quat, sizeof(in->obj_quat));
  memcpy(in->obj_lin, E->obj_lin, sizeof(in->obj_lin)); memcpy(in->obj_ang, E->obj_ang, sizeof(in->obj_ang));
  memcpy(in->obj_timers, E->obj_timers, sizeof(in->obj_timers)); memcpy(in->cmd, E->cmd, sizeof(in->cmd));
  in->terminated = terminated;
}
static void gait_io_from_env(const env_t* E, lt_gait_io* G) {
  memset(G, 0, sizeof(*G));
  for (int f = 0; f < 4; ++f) {
    int l = k_gait_order[f];
    G->cur_air[f] = E->foot_cur_air[l]; G->cur_con[f] = E->foot_cur_con[l]; G->sensor_last_air[f] = E->foot_last_air[l];
    G->last_step_air[f] = E->gait_last_air[l]; G->last_step_con[f] = E->gait_last_con[l];
    G->valid_last_air[f] = E->gait_valid_last_air[l];
    G->swinging_in_zero_cmd[f] = E->gait_flags[l] & 1; G->valid_prev_contact[f] = (E->gait_flags[l] >> 1) & 1;
  }
  for (int c = 0; c < 3; ++c) G->last_cmd[c] = E->gait_cmd[c];
  G->step_from_change = E->gait_step_from_change;
}
static void gait_io_to_env(const lt_gait_io* G, env_t* E) {
  for (int f = 0; f < 4; ++f) {
    int l = k_gait_order[f];
    E->gait_last_air[l] = G->last_step_air[f]; E->gait_last_con[l] = G->last_step_con[f];
    E->gait_valid_last_air[l] = G->valid_last_air[f];
    E->gait_flags[l] = (G->swinging_in_zero_cmd[f] ? 1 : 0) | (G->valid_prev_contact[f] ? 2 : 0);
  }
  for (int c = 0; c < 3; ++c) E->gait_cmd[c] = G->last_cmd[c];
  E->gait_step_from_change = G->step_from_change;
}

static int task_term_dims(const lt_cfg* cfg, int* dims) {
  static const int base[6] = {3, 3, 3, 12, 12, 12};
  memcpy(dims, base, sizeof(base));
  if (cfg->task == LT_TASK_LOCOMOTION) return 6;
  dims[6] = 13;
  return 7;
}

static void step_one(const lt_cfg* cfg, void* arena, const lt_layout* L, const float* actions, int64_t e, uint64_t step,
                     int any_nonzero_cmd, int mode) {
  const float* P = (const float*)((char*)arena + L->off_cmd_params);
  const int has_object = cfg->task != LT_TASK_LOCOMOTION;
  const real step_dt = cfg->sim_dt * (real)cfg->decimation;
  const int64_t max_len = cfg->max_episode_length;
  env_t E;
  gather(&E, arena, L, e);
  int terminated = 0, time_out = 0, bits = 0, reset = 0;
  if (mode == LT_ORACLE_MODE_STEP) {
    /* 1. action term (A1) */
    float raw[12], prev[12], prev2[12];
    for (int l = 0; l < 4; ++l) for (int k = 0; k < 3; ++k) { raw[k * 4 + l] = E.act_raw[l][k]; prev[k * 4 + l] = E.act_prev[l][k]; prev2[k * 4 + l] = E.act_prev2[l][k]; }
    lt_oracle_process_action(cfg, actions + e * 12, raw, prev, prev2);
    for (int l = 0; l < 4; ++l) for (int k = 0; k < 3; ++k) { E.act_raw[l][k] = raw[k * 4 + l]; E.act_prev[l][k] = prev[k * 4 + l]; E.act_prev2[l][k] = prev2[k * 4 + l]; }
    /* 2. decimation x (PD -> physics -> sensors) */
    /* tactile ContactSensor cadence [DEP] (update_period 0.025 s, object_transport_student_env_cfg.py:195-201): the taxel
     * forces refresh at the first sim step after a reset and then whenever the period has elapsed - sim-step indices
     * 0, 5, 10, ... counted from the reset (the episode step counter supplies the index) */
    const int tac_every = cfg->tactile_enabled ? (int)(cfg->tactile_update_period / cfg->sim_dt + 0.5f) : 1;
    int tac_phase = cfg->tactile_enabled ? (int)((E.ep_len * (int64_t)cfg->decimation) % (int64_t)(tac_every > 0 ? tac_every : 1)) : 0;
    for (int d = 0; d < cfg->decimation; ++d) {
      real qd0[4][3];
      memcpy(qd0, E.qd, sizeof(qd0));
      for (int l = 0; l < 4; ++l) for (int k = 0; k < 3; ++k) E.tau[l][k] = dc_motor(cfg, k_joint_default[l][k] + E.act_raw[l][k], E.q[l][k], E.qd[l][k]);
      contact_report rep;
      real h = cfg->sim_dt / (real)cfg->phys_substeps;
      for (int s = 0; s < cfg->phys_substeps; ++s) physics_substep(cfg, &E, h, has_object, &rep);
      for (int l = 0; l < 4; ++l) for (int k = 0; k < 3; ++k) E.qdd[l][k] = (E.qd[l][k] - qd0[l][k]) / cfg->sim_dt; /* Articulation.data.joint_acc [DEP] */
      sensors_update(cfg, &E, &rep, has_object);
      if (cfg->tactile_enabled) {
        if (tac_phase == 0) memcpy(E.plate, rep.plate, sizeof(E.plate));
        tac_phase = tac_phase + 1 >= tac_every ? 0 : tac_phase + 1;
      }
    }
    foot_kinematics(&E);
    /* 3. counters */
    E.ep_len += 1;
  }
  if (mode != LT_ORACLE_MODE_RESET_ALL) {
    /* 4. terminations */
    lt_term_in in;
    term_in_from_env(&E, 0, &in);
    bits = lt_oracle_terminations(cfg, &in, E.ep_len, max_len);
    /* a termination the caller requested on the state the previous step left (include/lt_env.h, LT_T_USER) */
    if (mode == LT_ORACLE_MODE_STEP && ((((const int32_t*)((char*)arena + L->off_term_bits))[e] >> LT_TERM_REQUEST_BIT) & 1)) bits |= 1 << LT_T_USER;
    if (mode == LT_ORACLE_MODE_STEP && ((((const int32_t*)((char*)arena + L->off_term_bits))[e] >> LT_TIMEOUT_REQUEST_BIT) & 1)) bits |= 1 << LT_T_USER_TIME_OUT;
    time_out = (bits & ((1 << LT_T_TIME_OUT) | (1 << LT_T_USER_TIME_OUT))) != 0;
    terminated = (bits & ~((1 << LT_T_TIME_OUT) | (1 << LT_T_USER_TIME_OUT))) != 0;
    ((int32_t*)((char*)arena + L->off_term_bits))[e] = bits;
    /* the terms-only hook takes `terminated` (for the alive term) from the arena */
    in.terminated = mode == LT_ORACLE_MODE_TERMS ? ((uint8_t*)arena + L->off_terminated)[e] : terminated;
    /* 5. rewards */
    lt_gait_io G;
    gait_io_from_env(&E, &G);
    G.any_nonzero_cmd = any_nonzero_cmd;
    float terms[LT_REWARD_SLOTS];
    lt_oracle_rewards(cfg, &in, &G, step_dt, terms);
    gait_io_to_env(&G, &E);
    real rew = 0;
    for (int i = 0; i < LT_NUM_REWARD_TERMS; ++i) {
      real v = terms[i] * cfg->reward_weight[i] * step_dt;
      rew += v;
      E.sums[i] += v;
      if (cfg->debug_terms) E.terms[i] = terms[i];
    }
    ((float*)((char*)arena + L->off_reward))[e] = rew;
  }
  if (mode == LT_ORACLE_MODE_STEP) {
    ((uint8_t*)arena + L->off_terminated)[e] = (uint8_t)terminated;
    ((uint8_t*)arena + L->off_time_out)[e] = (uint8_t)time_out;
    ((int64_t*)((char*)arena + L->off_dones))[e] = (terminated || time_out) ? 1 : 0;
    reset = terminated || time_out;
    /* 6. reset */
    E.cur_step[0] = reset ? 1 : 0;
    E.cur_step[1] = reset ? (real)E.ep_len : 0;
    E.cur_step[2] = reset ? E.sums[LT_R_TRACK_LIN_VEL_XY] : 0;
    E.cur_step[3] = reset ? E.sums[LT_R_TRACK_ANG_VEL_Z] : 0;
    if (reset) {
      for (int i = 0; i < LT_REWARD_SLOTS; ++i) E.last_sums[i] = E.sums[i];
      E.episodes_finished += 1; E.last_ep_len = (real)E.ep_len; E.last_term_bits = (real)bits;
      /* CommandTerm.reset [DEP]: what this env contributes to its step's reset batch is what the last compute() left */
      E.last_m[0] = E.m_exy; E.last_m[1] = E.m_eyaw; E.last_m[2] = E.m_airvar; E.last_m[3] = (real)(step & 0xFFFFFFull);
      reset_env(cfg, P, &E, EKEY(cfg, e), step, has_object);
      memset(E.plate, 0, sizeof(E.plate)); /* ContactSensor.reset [DEP]: the reset envs' net forces are zeroed */
    }
    /* 7. command term compute: _update_metrics first (commands.py:392-396), on the state the resets left and the command as it
     *    stands before this call's resample */
    {
      real vb[3], wb[3];
      quat_apply_inv(vb, E.root_quat, E.root_lin);
      quat_apply_inv(wb, E.root_quat, E.root_ang);
      const real ex = E.cmd[0] - vb[0], ey = E.cmd[1] - vb[1];
      E.m_exy = (real)sqrtf((float)(ex * ex + ey * ey));
      E.m_eyaw = (real)fabsf((float)(E.cmd[2] - wb[2]));
      const real mean = (real)0.25 * (((E.foot_last_air[0] + E.foot_last_air[1]) + E.foot_last_air[2]) + E.foot_last_air[3]);
      real v = 0;
      for (int l = 0; l < 4; ++l) v += (E.foot_last_air[l] - mean) * (E.foot_last_air[l] - mean);
      E.m_airvar = v * (real)(1.0 / 3.0); /* torch.var: unbiased */
    }
    E.cmd_time_left -= step_dt;
    if (E.cmd_time_left <= 0)
      command_resample(cfg, P, cfg->seed, EKEY(cfg, e), step, RS_CMD_TIMER, E.cmd, E.cmd_buf, &E.cmd_standing, &E.cmd_time_left);
    {
      float c[3] = {E.cmd[0], E.cmd[1], E.cmd[2]}, b[3] = {E.cmd_buf[0], E.cmd_buf[1], E.cmd_buf[2]};
      if (cfg->cmd_multi_sampling) lt_oracle_command_update(E.ep_len, (int)P[15], b, E.cmd_standing != 0, c);
      else if (E.cmd_standing != 0) c[0] = c[1] = c[2] = 0;
      E.cmd[0] = c[0]; E.cmd[1] = c[1]; E.cmd[2] = c[2];
    }
    /* 8. interval events: push_by_setting_velocity [DEP] */
    float u[4], w[4];
    E.push_robot_left -= step_dt;
    if (E.push_robot_left < (real)1e-6) {
      lt_rng4(cfg->seed, EKEY(cfg, e), step, RS_PUSH_ROBOT, u);
      lt_rng4(cfg->seed, EKEY(cfg, e), step, RS_PUSH_ROBOT + 1, w);
      E.push_robot_left = lt_lerp(cfg->push_robot_interval, u[3]);
      for (int c = 0; c < 3; ++c) { E.root_lin[c] += lt_lerp(cfg->push_robot_vel[c], u[c]); E.root_ang[c] += lt_lerp(cfg->push_robot_vel[3 + c], w[c]); }
    }
    if (has_object) {
      E.push_obj_left -= step_dt;
      if (E.push_obj_left < (real)1e-6) {
        lt_rng4(cfg->seed, EKEY(cfg, e), step, RS_PUSH_OBJ, u);
        lt_rng4(cfg->seed, EKEY(cfg, e), step, RS_PUSH_OBJ + 1, w);
        E.push_obj_left = lt_lerp(cfg->push_obj_interval, u[3]);
        for (int c = 0; c < 3; ++c) { E.obj_lin[c] += lt_lerp(cfg->push_obj_vel[c], u[c]); E.obj_ang[c] += lt_lerp(cfg->push_obj_vel[3 + c], w[c]); }
      }
    }
  }
  /* 9. observations (the terms-only hook and reset-all fill every history slot with the current frame) */
  float pol[64], cri[64];
  int dims[8];
  int nt = task_term_dims(cfg, dims);
  obs_frame(cfg, &E, EKEY(cfg, e), step, has_object, pol, cri);
  float* rp = (float*)((char*)arena + L->off_obs_policy) + e * L->obs_dim;
  float* rc = (float*)((char*)arena + L->off_obs_critic) + e * L->obs_dim;
  int fill = reset || mode != LT_ORACLE_MODE_STEP;
  lt_oracle_obs_push(dims, nt, cfg->obs_history, pol, fill, rp);
  lt_oracle_obs_push(dims, nt, cfg->obs_history, cri, fill, rc);
  scatter(&E, arena, L, e);
}

int lt_oracle_obs_dim(const lt_cfg* cfg) { return (cfg->task == LT_TASK_LOCOMOTION ? 45 : 58) * cfg->obs_history; }

static int64_t* counters(void* arena, const lt_layout* L) { return (int64_t*)((char*)arena + L->off_counters); }

static int any_nonzero(void* arena, const lt_layout* L) {
  /* rewards.py:190 `if torch.any(non_zero_cmd_env)` is a population-level gate */
  for (int64_t e = 0; e < L->n; ++e) {
    float* c = lt_quad(arena, L, LT_F_CMD, 0) + e * 4;
    if (c[0] != 0 || c[1] != 0 || c[2] != 0) return 1;
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------------ */
/* K10: tactile observation (student tasks)                                                          */
/* ------------------------------------------------------------------------------------------------ */
/* cumulative integral of the piecewise-linear line pressure from the first sample to arc length s */
static float pressure_integral(const float p[4], float dl, float s) {
  float acc = 0;
  for (int k = 0; k < 3; ++k) {
    float t = s - (float)k * dl;
    t = t < 0 ? 0 : (t > dl ? dl : t);
    acc += t * (p[k] + (p[k + 1] - p[k]) * t / (2 * dl));
  }
  return acc;
}

/* Taxel normal forces [17 x 13] from the four plate samples (x[4], y[4], f[4]).  The reference reads one net contact force
 * per taxel body from PhysX [DEP] (mdp/observations.py:154-158); the restated engine carries the cylinder-on-plate contact
 * as a 4-sample line, so a taxel's force is the integral, over the part of the contact line inside the taxel's collision box
 * (generate_locotouch_urdf.py:4-8, locotouch.urdf:816-821), of the piecewise-linear line pressure whose cell integrals are
 * the sample forces (end cells are half cells).  Engine restatement: parity unpinned (no PhysX taxel-force fixture exists). */
void lt_oracle_taxel_forces(const float x[4], const float y[4], const float f_in[4], float* out) {
  float f[4];
  for (int k = 0; k < 4; ++k) f[k] = f_in[k] > 0 ? f_in[k] : 0;
  const float dx = x[3] - x[0], dy = y[3] - y[0];
  const float len = sqrtf(dx * dx + dy * dy);
  const float ftot = f[0] + f[1] + f[2] + f[3];
  for (int t = 0; t < LT_TAXEL_ROWS * LT_TAXEL_COLS; ++t) {
    const int row = t / LT_TAXEL_COLS, col = t - row * LT_TAXEL_COLS;
    const float cx = LT_TAXEL_X0 - LT_TAXEL_DX * (float)row, cy = LT_TAXEL_Y0 - LT_TAXEL_DY * (float)col;
    out[t] = 0;
    if (!(ftot > 0)) continue;
    if (len < 1e-6f) {
      if (fabsf(x[0] - cx) <= LT_TAXEL_HALF_X && fabsf(y[0] - cy) <= LT_TAXEL_HALF_Y) out[t] = ftot;
      continue;
    }
    float t0 = 0, t1 = 1;
    const float pp[2] = {x[0] - cx, y[0] - cy}, dd[2] = {dx, dy}, hh[2] = {LT_TAXEL_HALF_X, LT_TAXEL_HALF_Y};
    int miss = 0;
    for (int ax = 0; ax < 2 && !miss; ++ax) {
      if (fabsf(dd[ax]) < 1e-9f) { if (fabsf(pp[ax]) > hh[ax]) miss = 1; }
      else {
        float ta = (-hh[ax] - pp[ax]) / dd[ax], tb = (hh[ax] - pp[ax]) / dd[ax];
        if (ta > tb) { float tt = ta; ta = tb; tb = tt; }
        if (ta > t0) t0 = ta;
        if (tb < t1) t1 = tb;
      }
    }
    if (miss || !(t1 > t0)) continue;
    const float dl = len / 3;
    const float p[4] = {f[0] / (0.5f * dl), f[1] / dl, f[2] / dl, f[3] / (0.5f * dl)};
    out[t] = pressure_integral(p, dl, t1 * len) - pressure_integral(p, dl, t0 * len);
  }
}

/* BinaryTactileSignals with explicit uniforms (reference mdp/observations.py:121-126 thresholds, :154-158 contact map,
 * :166-184 dropout then addition, :307-308 two identical channels).  forces / u_*: [221]; out: [442]. */
void lt_oracle_tactile_signals_u(const lt_cfg* cfg, const float* forces, const float* u_thr, const float* u_drop, const float* u_add,
                                 float* out) {
  const int nt = LT_TAXEL_ROWS * LT_TAXEL_COLS;
  for (int t = 0; t < nt; ++t) {
    const float n_min = -cfg->tactile_threshold_noise, n_max = cfg->tactile_threshold_noise;
    const float thr = cfg->tactile_threshold + (u_thr[t] * (n_max - n_min) + n_min);   /* :126 */
    int contact = forces[t] > thr;                                                       /* :158 */
    if (cfg->tactile_dropout_prob > 0 && contact && u_drop[t] < cfg->tactile_dropout_prob) contact = 0;   /* :170-175 */
    if (cfg->tactile_addition_prob > 0 && !contact && u_add[t] < cfg->tactile_addition_prob) contact = 1; /* :179-184 */
    out[t] = out[nt + t] = contact ? 1.0f : 0.0f;
  }
}

/* Every TactileSignals class with explicit uniforms (reference mdp/observations.py:154-246): forces [221] -> the four channel maps
 * (contact, normalised force, per-env min-max normalised, discretised), each [221].  `original` != 0: TactileSignals.__call__
 * (:248-279, the raw reading: thresholds and level noise only); otherwise the processed pipeline of get_normal_forces (:164-197):
 * dropout (force <- u * thr), addition (force <- thr (1 + 0.2 u)), force noise on contact taxels with the too-small repair, then
 * get_normalized_forces (:199-203), compute_min_max_normalized_signals (:205-222), compute_discretized_signals (:224-235).
 * Uniforms, each [221]: u_thr (construction), u_drop, u_dropf, u_add, u_addf, u_noise, u_small, u_level.  The reference draws the
 * masked ones only for the selected taxels; here they are indexed by taxel (the golden generator scatters its tape accordingly). */
void lt_oracle_tactile_channels_u(const lt_cfg* cfg, int original, const float* forces, const float* u_thr, const float* u_drop,
                                  const float* u_dropf, const float* u_add, const float* u_addf, const float* u_noise,
                                  const float* u_small, const float* u_level, float* contact_out, float* norm_out, float* minmax_out,
                                  float* disc_out) {
  const int nt = LT_TAXEL_ROWS * LT_TAXEL_COLS;
  float valid[LT_TAXEL_ROWS * LT_TAXEL_COLS];
  float mn = 2.0f, mx = -1.0f;
  for (int t = 0; t < nt; ++t) {
    const float n_min = -cfg->tactile_threshold_noise, n_max = cfg->tactile_threshold_noise;
    const float thr = cfg->tactile_threshold + (u_thr[t] * (n_max - n_min) + n_min);   /* :126 */
    int contact = forces[t] > thr;                                                       /* :158 */
    float f = forces[t];
    if (!original) {
      if (cfg->tactile_dropout_prob > 0 && contact && u_drop[t] < cfg->tactile_dropout_prob) { f = u_dropf[t] * thr; contact = 0; }
      if (cfg->tactile_addition_prob > 0 && !contact && u_add[t] < cfg->tactile_addition_prob) { f = thr * (1.0f + 0.2f * u_addf[t]); contact = 1; }
      if (cfg->tactile_force_noise > 0) {
        const float p_min = -cfg->tactile_force_noise, p_max = cfg->tactile_force_noise;
        if (contact) f *= 1.0f + (u_noise[t] * (p_max - p_min) + p_min);
        f = f > 0 ? f : 0;
        if (contact && f < thr) f = thr * (1.0f + 0.2f * u_small[t]);
      }
    }
    float nrm = f / cfg->tactile_maximal_force;
    nrm = nrm < 0 ? 0 : (nrm > 1 ? 1 : nrm);
    contact_out[t] = contact ? 1.0f : 0.0f;
    norm_out[t] = nrm;
    valid[t] = contact ? nrm : 0.0f;
    if (valid[t] < mn) mn = valid[t];
    if (valid[t] > mx) mx = valid[t];
  }
  const float range = (mx - mn) > 0 ? (mx - mn) : 1.0f;
  const float bin = 1.0f / (float)cfg->tactile_total_levels;
  for (int t = 0; t < nt; ++t) {
    float mm = (valid[t] - mn) / range;
    mm = mm < 0 ? 0 : (mm > 1 ? 1 : mm);
    minmax_out[t] = mm;
    float d = rintf(mm / bin); /* torch.round: half to even */
    if (cfg->tactile_level_noise > 0) d += u_level[t] * (cfg->tactile_level_noise - (-cfg->tactile_level_noise)) + (-cfg->tactile_level_noise);
    d *= bin;
    d = d < 0 ? 0 : (d > 1 ? 1 : d);
    disc_out[t] = contact_out[t] != 0 ? d : 0.0f;
  }
}

/* One term in its class's channel layout (:279, :308, :332, :357, :383, :425-429): out [442] or [884] */
void lt_oracle_tactile_format_u(const lt_cfg* cfg, int format, const float* forces, const float* u8[8], float* out) {
  const int nt = LT_TAXEL_ROWS * LT_TAXEL_COLS;
  float ch[4][LT_TAXEL_ROWS * LT_TAXEL_COLS];
  lt_oracle_tactile_channels_u(cfg, format == LT_TACTILE_ORIGINAL, forces, u8[0], u8[1], u8[2], u8[3], u8[4], u8[5], u8[6], u8[7], ch[0], ch[1],
                               ch[2], ch[3]);
  if (format == LT_TACTILE_PROCESSED || format == LT_TACTILE_ORIGINAL) {
    for (int k = 0; k < 4; ++k) memcpy(out + k * nt, ch[k], sizeof(float) * nt);
    return;
  }
  const int second = format == LT_TACTILE_BINARY ? 0 : (format == LT_TACTILE_NORMALIZED ? 2 : (format == LT_TACTILE_DISCRETE ? 3 : 1));
  memcpy(out, ch[0], sizeof(float) * nt);
  memcpy(out + nt, ch[second], sizeof(float) * nt);
}

static void tactile_pass(const lt_cfg* cfg, void* arena, const lt_layout* L) {
  enum { NT = LT_TAXEL_ROWS * LT_TAXEL_COLS };
  const uint64_t step = (uint64_t)((int64_t*)((char*)arena + L->off_counters))[0];
  float* const base = (float*)((char*)arena + L->off_obs_tactile);
  const int64_t blk = L->npad * (int64_t)LT_TACTILE_WIDE_DIM;
  const int dim = (cfg->tactile_format == LT_TACTILE_PROCESSED || cfg->tactile_format == LT_TACTILE_ORIGINAL) ? LT_TACTILE_WIDE_DIM : LT_TACTILE_DIM;
  for (int64_t e = 0; e < L->n; ++e) {
    float x[4], y[4], f[4], forces[NT], ut[NT + 4], U[7][NT], u[4];
    for (int k = 0; k < 4; ++k) {
      x[k] = lt_quad(arena, L, LT_F_PLATE_SAMPLES, 0)[e * 4 + k];
      y[k] = lt_quad(arena, L, LT_F_PLATE_SAMPLES, 1)[e * 4 + k];
      f[k] = lt_quad(arena, L, LT_F_PLATE_SAMPLES, 2)[e * 4 + k];
    }
    lt_oracle_taxel_forces(x, y, f, forces);
    for (int term = 0; term < 3; ++term) {
      if (term > 0 && !(cfg->tactile_aux_groups & term)) continue;
      /* stream ids: thresholds RS_TACTILE_THR + 0x40 term + taxel / 4 (startup key); per step RS_TACTILE + 0x200 term + 2 taxel
       * -> (drop, add, drop force, add force), + 1 -> (force noise, too-small repair, level noise, -) */
      for (int t = 0; t < NT; t += 4) { lt_rng4(cfg->seed, EKEY(cfg, e), ~(uint64_t)0, RS_TACTILE_THR + 0x40u * (uint32_t)term + (uint32_t)(t >> 2), u); memcpy(ut + t, u, sizeof(u)); }
      for (int t = 0; t < NT; ++t) {
        lt_rng4(cfg->seed, EKEY(cfg, e), step, RS_TACTILE + 0x200u * (uint32_t)term + 2u * (uint32_t)t, u);
        U[0][t] = u[0]; U[2][t] = u[1]; U[1][t] = u[2]; U[3][t] = u[3];
        lt_rng4(cfg->seed, EKEY(cfg, e), step, RS_TACTILE + 0x200u * (uint32_t)term + 2u * (uint32_t)t + 1u, u);
        U[4][t] = u[0]; U[5][t] = u[1]; U[6][t] = u[2];
      }
      const float* u8[8] = {ut, U[0], U[1], U[2], U[3], U[4], U[5], U[6]};
      if (term == 0) lt_oracle_tactile_format_u(cfg, cfg->tactile_format, forces, u8, base + e * dim);
      else lt_oracle_tactile_format_u(cfg, term == 1 ? LT_TACTILE_ORIGINAL : LT_TACTILE_PROCESSED, forces, u8, base + term * blk + e * LT_TACTILE_WIDE_DIM);
    }
  }
}

static void post_step(const lt_cfg* cfg, void* arena, const lt_layout* L) {
  float* P = (float*)((char*)arena + L->off_cmd_params);
  curriculum_decide(cfg, arena, L);
  P[26] = any_nonzero(arena, L) ? 1.0f : 0.0f; /* population gate of rewards.py:190 for the next step */
  counters(arena, L)[0] += 1;
}

int lt_oracle_reset_all(const lt_cfg* cfg, void* arena) {
  lt_layout L;
  lt_layout_init(&L, cfg->num_envs, lt_oracle_obs_dim(cfg), cfg->tactile_enabled);
  memset(arena, 0, (size_t)L.off_obj_sizes);  /* LT_F_OBJ_SIZES (host input) survives */
  float* P = (float*)((char*)arena + L.off_cmd_params);
  lt_oracle_cmd_params_init(cfg, P);
  const int has_object = cfg->task != LT_TASK_LOCOMOTION;
  for (int64_t e = 0; e < L.n; ++e) {
    env_t E;
    gather(&E, arena, &L, e);
    startup_env(cfg, &E, EKEY(cfg, e), (const float*)((char*)arena + L.off_obj_sizes) + e * 2);
    reset_env(cfg, P, &E, EKEY(cfg, e), 0, has_object);
    if (!has_object) E.obj_quat[0] = 1;
    /* ManagerBasedRLEnv.reset() does not run command_manager.compute(); only the zero-command window that
     * _resample_command applies itself (commands.py:559) is visible in the first observation */
    if (cfg->cmd_multi_sampling && 0 < (int)P[15]) E.cmd[0] = E.cmd[1] = E.cmd[2] = 0;
    scatter(&E, arena, &L, e);
    step_one(cfg, arena, &L, NULL, e, 0, 1, LT_ORACLE_MODE_RESET_ALL);
  }
  if (cfg->cur_enabled) { P[24] = 1; P[25] = 1; }
  P[26] = 1;
  counters(arena, &L)[0] = 1;
  if (cfg->tactile_enabled) tactile_pass(cfg, arena, &L);
  {
    lt_dev_args da;  /* byte-identical to what the HIP library uploads (tests copy whole arenas between the two) */
    memset(&da, 0, sizeof(da));
    da.cfg = *cfg;
    da.layout = L;
    memcpy((char*)arena + L.off_dev_args, &da, sizeof(da));
  }
  return 0;
}

int lt_oracle_step(const lt_cfg* cfg, void* arena, const float* actions, int nthreads) {
  lt_layout L;
  lt_layout_init(&L, cfg->num_envs, lt_oracle_obs_dim(cfg), cfg->tactile_enabled);
  uint64_t step = (uint64_t)counters(arena, &L)[0];
  int nz = ((const float*)((char*)arena + L.off_cmd_params))[26] != 0;
  int64_t n = L.n;
  (void)nthreads;
  curriculum_apply_pending(arena, &L);  /* before the env loop overwrites the records the pending operations refer to */
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(static)
#endif
  for (int64_t e = 0; e < n; ++e) step_one(cfg, arena, &L, actions, e, step, nz, LT_ORACLE_MODE_STEP);
  post_step(cfg, arena, &L);
  if (cfg->tactile_enabled) tactile_pass(cfg, arena, &L);
  return 0;
}

int lt_oracle_eval_terms(const lt_cfg* cfg, void* arena) {
  lt_layout L;
  lt_layout_init(&L, cfg->num_envs, lt_oracle_obs_dim(cfg), cfg->tactile_enabled);
  uint64_t step = (uint64_t)counters(arena, &L)[0];
  int nz = any_nonzero(arena, &L);
  for (int64_t e = 0; e < L.n; ++e) step_one(cfg, arena, &L, NULL, e, step, nz, LT_ORACLE_MODE_TERMS);
  return 0;
}

int64_t lt_oracle_state_bytes(const lt_cfg* cfg) {
  lt_layout L;
  lt_layout_init(&L, cfg->num_envs, lt_oracle_obs_dim(cfg), cfg->tactile_enabled);
  return L.total_bytes;
}
