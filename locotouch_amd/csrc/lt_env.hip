// lt_env.hip - the environment-step kernels for MI355X (gfx950 / CDNA4).
//
// One launch of lt_step_kernel = one ManagerBasedRLEnv.step() for every env on this GPU:
//   action term -> 4 x (DC-motor PD -> ABA forward dynamics with implicit penalty contacts -> contact sensors)
//   -> terminations -> 23 reward terms (incl. the adaptive symmetric-gait class) -> masked reset (events, RNG)
//   -> command term -> interval pushes -> observation frame + 6-deep history -> outputs.
// The per-tile half of the velocity curriculum / population gate runs at the kernel's tail, the one-wave global half behind it or
// chained into the next launch (lt_post.h) - no host synchronisation, hipGraph-capturable.
//
// Lane mapping: 4 lanes per env (lane&3 = leg FR/FL/RR/RL, each lane owns the hip-thigh-calf chain of its leg), 16 envs per wave64.
// Tree-level coupling (floating base, carried cylinder) and all per-env reductions go through DPP quad butterflies.  Grids of at
// most one 16-env tile per CU (4096 envs) run FOUR waves per tile - leg dynamics, CRBA + bias, carried cylinder, random numbers -
// that meet at two LDS-only barriers per physics substep; larger grids run one wave per tile (template parameter HELPERS).
// All persistent state is SoA "quad arrays" float[N][4] (include/lt_layout.h), so every state access of a wave is one coalesced
// 256-byte transaction.  The roofline the kernel is REPORTED against is HBM (6720 algorithmic bytes per env-step);
// what bounds it is VALU issue on one wave per SIMD - DESIGN.md §4 has both.
//
// Reference citations use paths relative to the reference repo (locotouch/...).
#include <hip/hip_runtime.h>
#include <cstdlib>

#include "lt_device_math.h"
#include "lt_internal.h"
#include "lt_post.h"
#include "../../include/lt_go1_model.h"

using namespace lt;

namespace {

constexpr int MODE_STEP = 0, MODE_TERMS = 1, MODE_RESET_ALL = 2;

// Kernel arguments: two pointers.  Configuration + layout are read from the device-resident lt_dev_args block at the
// arena tail (include/lt_layout.h) - a 1.7 KB by-value kernarg cost ~20 us of scalar-load stalls per launch.
struct KArgs {
  const lt_dev_args* __restrict__ d;
  char* arena;
  const float* actions;
  const long long* ep_len;  // == arena + layout.off_ep_len (the tactile refresh phase is needed before the layout block is staged)
  long long npad;  // == d->layout.npad: lets the state loads be issued before the (cfg, layout) block has been staged
  // observation rows [npad][OBS] of the two groups: previous rows (read) and new rows (written).  Equal pointers =
  // in-place update of the arena rows (lt_env_step); distinct = rollout-storage slots t / t+1 (lt_env_step_rows).
  const float* obs_prev[2];
  float* obs_next[2];
  // optional rollout-storage writes of this transition (lt_env_step_rollout): rewards with the time-out bootstrap, dones
  const float* rec_values;
  float rec_gamma;
  float* rec_rewards;
  unsigned char* rec_dones;
  // chained steps (lt_env_defer_gate mode 2; helper form): `step_offset` steps precede this one whose common_step_counter bump is
  // still outstanding (step id = counters[0] + step_offset); `decide_first`: the population pass of the previous step has not run
  // yet - workgroup 0's RNG wave runs it beside this step's first physics substep (lt_post.h, chained form)
  int step_offset;
  int decide_first;
};
static_assert(sizeof(lt_dev_args) <= LT_DEV_ARGS_BYTES, "lt_dev_args outgrew its arena slot");

// Quad-array index of every quad field (prefix sum of lt_field_quads): npad * 16 B is a multiple of 256, so the 256-byte
// alignment of lt_layout_init never pads and quad_off[f] == k_cumq.v[f] * npad * 16 (lt_check_layout verifies it on the
// host).  Field addresses are then compile-time multiples of one runtime stride: no offset-table loads in the kernels.
struct CumQ { int v[LT_NUM_QUAD_FIELDS + 1]; };
constexpr int field_quads_c(int f) {
  return (f == LT_F_JOINT_POS || f == LT_F_JOINT_VEL || f == LT_F_JOINT_ACC || f == LT_F_APPLIED_TORQUE || f == LT_F_ACT_RAW ||
          f == LT_F_ACT_PREV_RAW || f == LT_F_ACT_PREV_PREV_RAW || f == LT_F_FOOT_POS_W || f == LT_F_FOOT_VEL_W || f == LT_F_CURRICULUM || f == LT_F_PLATE_SAMPLES)
             ? 3
             : (f == LT_F_FORCE_HIST ? 12 : ((f == LT_F_EPISODE_SUMS || f == LT_F_LAST_EPISODE_SUMS || f == LT_F_REWARD_TERMS) ? 7 : 1));
}
constexpr CumQ make_cumq() {
  CumQ c{};
  int acc = 0;
  for (int f = 0; f < LT_NUM_QUAD_FIELDS; ++f) { c.v[f] = acc; acc += field_quads_c(f); }
  c.v[LT_NUM_QUAD_FIELDS] = acc;
  return c;
}
__device__ constexpr CumQ k_cumq = make_cumq();

// RNG stream ids - shared spec with oracle/lt_oracle.c
enum {
  RS_NOISE_JPOS = 0x100, RS_NOISE_JVEL = 0x110, RS_NOISE_BASE = 0x120, RS_NOISE_OBJ = 0x130,
  RS_RESET_ROOT = 0x200, RS_RESET_JOINT = 0x210, RS_RESET_MAT = 0x220, RS_RESET_OBJ = 0x221, RS_RESET_EVENT = 0x223,
  RS_CMD_RESET = 0x230, RS_CMD_TIMER = 0x240, RS_PUSH_ROBOT = 0x250, RS_PUSH_OBJ = 0x260, RS_STARTUP = 0x300,
  RS_BUCKET_FEET = 0x310, RS_BUCKET_OBJ = 0x311,  // material pools: key = bucket index (not an env), startup stream
  RS_TACTILE_THR = 0x400,  // + 0x40 term + taxel / 4 (startup stream): the per-(env, taxel) threshold offsets, drawn once
  RS_TACTILE = 0x500       // + 0x200 term + 2 taxel + {0, 1} (step stream): csrc/lt_tactile.hip
};
// kernel-template task index of the transport task WITH the tactile sensor (cfg.task stays LT_TASK_TRANSPORT_TEACHER, the
// student registrations derive from the teacher's env cfg: object_transport_student_env_cfg.py:161-168)
constexpr int K_TASK_TACTILE = 2;

// observation history tables: for output column `col` of a group row, where does the value come from?
//   src[col] >= 0 : old row, column src[col] (one slot newer);  src[col] < 0 : newest frame, element -src-1
//   frame[col]    : newest-frame element of that column's term (used when every slot is filled, i.e. after reset)
struct ObsTable { short src[352]; short frame[352]; };
constexpr ObsTable make_obs_table(int nterms) {
  ObsTable t{};
  const int dims[7] = {3, 3, 3, 12, 12, 12, 13};  // cmd, ang vel, gravity, joint pos, joint vel, last action, object state
  int off = 0, fo = 0;
  for (int i = 0; i < nterms; ++i) {
    const int d = dims[i];
    for (int s = 0; s < 6; ++s)
      for (int k = 0; k < d; ++k) {
        t.src[off + s * d + k] = (short)(s == 5 ? -(fo + k) - 1 : off + (s + 1) * d + k);
        t.frame[off + s * d + k] = (short)(fo + k);
      }
    off += 6 * d;
    fo += d;
  }
  return t;
}
__constant__ ObsTable k_obs_tab_loco = make_obs_table(6);
__constant__ ObsTable k_obs_tab_teacher = make_obs_table(7);

#include "lt_physics_crba.h"

// foot centre position / velocity (world) of this lane's leg for the current state
__device__ __forceinline__ void foot_kinematics(const float (&sgn)[4], const Base& B, Leg& G) {
  const LinkC LC[3] = {make_link<0>(sgn), make_link<1>(sgn), make_link<2>(sgn)};
  const M3 R0 = quat_to_mat(B.q.w, B.q.x, B.q.y, B.q.z);
  V3 om[3], vl[3], pw[3], ca, cl;
  M3 Rw[3];
  float s, c;
  s = fsin(G.q[0]); c = fcos(G.q[0]);
  joint_fk<0>(tmul(R0, B.w), tmul(R0, B.u), R0, B.p, LC[0].r, c, s, G.qd[0], om[0], vl[0], Rw[0], pw[0], ca, cl);
  s = fsin(G.q[1]); c = fcos(G.q[1]);
  joint_fk<1>(om[0], vl[0], Rw[0], pw[0], LC[1].r, c, s, G.qd[1], om[1], vl[1], Rw[1], pw[1], ca, cl);
  s = fsin(G.q[2]); c = fcos(G.q[2]);
  joint_fk<1>(om[1], vl[1], Rw[1], pw[1], LC[2].r, c, s, G.qd[2], om[2], vl[2], Rw[2], pw[2], ca, cl);
  const V3 rf = v3(0.f, 0.f, -0.213f);
  G.foot_p = pw[2] + mul(Rw[2], rf);
  G.foot_v = mul(Rw[2], vl[2] + cross(om[2], rf));
}

// ---- K1: DC-motor PD (reference assets/go1.py:41-49 + IsaacLab DCMotor [DEP]) ----
__device__ __forceinline__ float dc_motor(const lt_cfg& c, float q_des, float q, float qd) {
  const float tau = c.kp * (q_des - q) + c.kd * (0.f - qd);
  float hi = c.saturation_effort * (1.f - qd / c.velocity_limit);
  hi = hi < 0.f ? 0.f : (hi > c.effort_limit ? c.effort_limit : hi);
  float lo = c.saturation_effort * (-1.f - qd / c.velocity_limit);
  lo = lo < -c.effort_limit ? -c.effort_limit : (lo > 0.f ? 0.f : lo);
  return tau < lo ? lo : (tau > hi ? hi : tau);
}

// ---- K3: ContactSensor air/contact timers [DEP, SURVEY.md Appendix C] ----
__device__ __forceinline__ void timers_update(float& cur_air, float& cur_con, float& last_air, float& last_con, bool contact, float dt) {
  const bool first_contact = (cur_air > 0.f) && contact;
  const bool first_detach = (cur_con > 0.f) && !contact;
  if (first_contact) last_air = cur_air + dt;
  if (first_detach) last_con = cur_con + dt;
  cur_air = contact ? 0.f : cur_air + dt;
  cur_con = contact ? cur_con + dt : 0.f;
}

// ---- K5: adaptive symmetric gait reward (reference locotouch/mdp/rewards.py:60-392) --------------------
// Arrays are in the class's foot column order [FR, RL, FL, RR]; every lane of the quad evaluates the same code.
struct Gait {
  float cur_air[4], cur_con[4], sensor_last_air[4];
  float last_air[4], last_con[4], valid[4];
  int swing0[4], prevc[4];
  V3 last_cmd; float step_from_change;
};
__device__ __forceinline__ float gait_swing_bonus(const lt_cfg& c, const Gait& G, int f0, int f1, float step_dt) {
  const float judge = c.gait_judge_time, ub = c.gait_rwd_upper, lb = c.gait_rwd_lower, tol = c.gait_tolerance_proportion;
  const float scale = ub / (1.0f / (c.gait_soft_min_frequency * 2.0f));                                     // :72,:78
  const float tbar = (G.cur_air[f0] + G.cur_air[f1]) / 2.f;
  const bool both_air = (G.cur_air[f0] > judge) && (G.cur_air[f1] > judge);                                 // :247
  const int t0 = f0 < 2 ? 0 : 2, o0 = f0 < 2 ? 2 : 0;                                                        // :250-254
  const float vt0 = G.valid[t0], vt1 = G.valid[t0 + 1], vo0 = G.valid[o0], vo1 = G.valid[o0 + 1];
  const float mean_t = (vt0 + vt1) / 2.f, mean_o = (vo0 + vo1) / 2.f;
  const bool valid_t = (vt0 > judge) && (vt1 > judge) && (vt0 > 2.f * step_dt) && (vt1 > 2.f * step_dt);   // :253,:260
  const bool valid_o = (vo0 > judge) && (vo1 > judge) && (vo0 > 2.f * step_dt) && (vo1 > 2.f * step_dt);   // :257,:261
  if (!(both_air && (valid_t || valid_o))) return 0.f;                                                      // :264-265,:341-344
  const float Tref = mean_o, Ttol = Tref + tol * Tref, diff = mean_t - mean_o;                               // :266-276
  const float Text = clampf(Ttol - diff, Tref, Ttol);                                                       // :277-278
  float r_within = scale * tbar; r_within = r_within > ub ? ub : r_within;                                  // :289
  float r_ref = scale * Tref; r_ref = r_ref > ub ? ub : r_ref;
  float r_ext = scale * Text; r_ext = r_ext > ub ? ub : r_ext;
  float r_tol = scale * Ttol; r_tol = r_tol > ub ? ub : r_tol;
  if ((tbar <= Text) || (diff < 0.f)) return r_within;                                                      // :281,:285-286
  const bool ext_lt_tol = Text < Ttol;                                                                      // :295
  if ((tbar > Text) && (tbar <= Ttol)) {                                                                    // :282,:296-307
    if (!ext_lt_tol) return r_ext;
    const float a = -r_ext / (Ttol - Text);
    return a * tbar + (-a * Ttol);
  }
  float lower = lb;                                                                                         // :319-325
  if (valid_o) lower = clampf(ext_lt_tol ? (diff / (tol * Tref)) * lb : r_tol, lb, ub);
  float beyond = lower;                                                                                     // :310-330
  if (Text > Tref) {
    const float a = -r_ref / (Text - Tref);
    beyond = a * tbar + (-a * Ttol);
  }
  return beyond < lower ? lower : beyond;
}
__device__ __forceinline__ float gait_reward(const lt_cfg& c, Gait& G, V3 cmd, float lin_err, float ang_err, float ox, float oy,
                                             bool any_nonzero_cmd, float step_dt) {
  const float judge = c.gait_judge_time, ab = c.gait_air_bound, cb = c.gait_contact_bound;
  const float async_judge = judge + c.gait_async_tolerance;
  const bool nonzero = norm(cmd) > 0.f;
  // _update_valid_last_air_contact_time :158-200
  if (!nonzero) for (int f = 0; f < 4; ++f) G.valid[f] = 0.f;
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const bool new_swing = (G.last_air[f] < judge) && (G.cur_air[f] > judge);
    if (new_swing && nonzero) G.swing0[f] = 0;
    if ((G.cur_air[f] > judge) && !nonzero) G.swing0[f] = 1;
  }
  G.step_from_change += 1.f;
  const bool changing = (fabsf(cmd.x - G.last_cmd.x) > 1.0e-3f) || (fabsf(cmd.y - G.last_cmd.y) > 1.0e-3f) || (fabsf(cmd.z - G.last_cmd.z) > 1.0e-3f);
  if (changing) {
    G.last_cmd = cmd;
    G.step_from_change = 0.f;
#pragma unroll
    for (int f = 0; f < 4; ++f) { G.swing0[f] = 1; G.valid[f] = 0.f; }
  }
  if (any_nonzero_cmd) {
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const bool landing = (G.last_con[f] < judge) && (G.cur_con[f] > judge);
      if (landing && G.prevc[f] && !G.swing0[f]) G.valid[f] = G.sensor_last_air[f];
    }
  }
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    G.last_air[f] = G.cur_air[f];
    G.last_con[f] = G.cur_con[f];
    if (G.cur_con[f] > judge) G.prevc[f] = 1;
  }
  // task performance score :202-213, :372-392
  const float e_lin = nonzero ? lin_err : 0.f, e_ang = nonzero ? ang_err : 0.f;
  const float vel_score = (__expf(-(e_lin / c.gait_vel_sigma)) + __expf(-(e_ang / c.gait_vel_sigma))) / 2.f;  // (v_exp_f32: ~2e-7 relative)
  float score = vel_score;
  if (c.gait_with_object) {
    const float sx = clampf(1.f - fabsf(ox) / c.danger_x_max, 0.f, 1.f);
    const float sy = clampf(1.f - fabsf(oy) / c.danger_y_max, 0.f, 1.f);
    score = clampf((vel_score * 2.f + (sx + sy) / 2.f) / 3.f, 0.f, 1.f);
  }
  float sync[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {                                                                             // :218-241
    const int f0 = 2 * p, f1 = 2 * p + 1;
    const bool both_air = (G.cur_air[f0] > judge && G.cur_air[f0] < ab) && (G.cur_air[f1] > judge && G.cur_air[f1] < ab);
    const bool both_con = (G.cur_con[f0] > judge && G.cur_con[f0] < cb) && (G.cur_con[f1] > judge && G.cur_con[f1] < cb);
    float bonus = gait_swing_bonus(c, G, f0, f1, step_dt);
    const float rs = 1.f - c.gait_task_ratio + c.gait_task_ratio * score;
    if (bonus > 0.f) bonus *= rs;
    bonus += 1.f;
    sync[p] = both_air ? bonus : (both_con ? 1.f : 0.f);
  }
  float asum = 0.f;
  const int pf0[4] = {0, 1, 0, 2}, pf1[4] = {2, 3, 3, 1};                                                    // :144-147
#pragma unroll
  for (int p = 0; p < 4; ++p) {                                                                             // :348-363
    const int f0 = pf0[p], f1 = pf1[p];
    const bool c0t = G.cur_con[f0] > judge && G.cur_con[f0] <= async_judge, c1t = G.cur_con[f1] > judge && G.cur_con[f1] <= async_judge;
    const bool a0 = G.cur_air[f0] > judge && G.cur_air[f0] < ab, a1 = G.cur_air[f1] > judge && G.cur_air[f1] < ab;
    const bool c0 = G.cur_con[f0] > judge && G.cur_con[f0] < cb, c1 = G.cur_con[f1] > judge && G.cur_con[f1] < cb;
    asum += ((c0t && c1t) || (a0 && c1) || (c0 && a1)) ? 1.f : 0.f;
  }
  const float stepping = ((sync[0] + sync[1]) / 2.f + asum / 4.f) / 2.f;                                    // :141-151
  bool all_stance = true;
#pragma unroll
  for (int f = 0; f < 4; ++f) all_stance = all_stance && (G.cur_con[f] > judge);                            // :365-368
  const float stance = (all_stance ? 1.f : 0.f) * c.gait_stance_scale;
  return nonzero ? stepping : stance;                                                                       // :153-154
}

// The gait term of one step for the quad's four feet (reference rewards.py:60-392, AdaptiveSymmetricGaitReward[withObject]).
// `io`: the class state of this lane (per-foot last air / contact time, valid last air time, flags) and of the env (last command,
// steps since it changed), updated in place; unchanged and 0 when the term's weight is 0.
struct GaitIO { float last_air, last_con, valid; int flags; V3 cmd; float step; };
template <bool HAS_OBJ>
__device__ __forceinline__ float gait_term(const lt_cfg& c, int leg, float cur_air, float cur_con, float sensor_last_air, GaitIO& io, const V3& bp,
                                           const Q4& bq, const V3& op, const V3& cmd, float lin_err, float ang_err, bool any_nonzero_cmd,
                                           float step_dt) {
  if (c.reward_weight[LT_R_GAIT] == 0.f) return 0.f;
  // gather the four feet in class column order [FR, RL, FL, RR] = legs [0, 3, 1, 2]
  Gait GT;
  GT.cur_air[0] = qbcast<0>(cur_air); GT.cur_air[1] = qbcast<3>(cur_air); GT.cur_air[2] = qbcast<1>(cur_air); GT.cur_air[3] = qbcast<2>(cur_air);
  GT.cur_con[0] = qbcast<0>(cur_con); GT.cur_con[1] = qbcast<3>(cur_con); GT.cur_con[2] = qbcast<1>(cur_con); GT.cur_con[3] = qbcast<2>(cur_con);
  GT.sensor_last_air[0] = qbcast<0>(sensor_last_air); GT.sensor_last_air[1] = qbcast<3>(sensor_last_air); GT.sensor_last_air[2] = qbcast<1>(sensor_last_air); GT.sensor_last_air[3] = qbcast<2>(sensor_last_air);
  GT.last_air[0] = qbcast<0>(io.last_air); GT.last_air[1] = qbcast<3>(io.last_air); GT.last_air[2] = qbcast<1>(io.last_air); GT.last_air[3] = qbcast<2>(io.last_air);
  GT.last_con[0] = qbcast<0>(io.last_con); GT.last_con[1] = qbcast<3>(io.last_con); GT.last_con[2] = qbcast<1>(io.last_con); GT.last_con[3] = qbcast<2>(io.last_con);
  GT.valid[0] = qbcast<0>(io.valid); GT.valid[1] = qbcast<3>(io.valid); GT.valid[2] = qbcast<1>(io.valid); GT.valid[3] = qbcast<2>(io.valid);
  const int f0 = qbcasti<0>(io.flags), f1 = qbcasti<3>(io.flags), f2 = qbcasti<1>(io.flags), f3 = qbcasti<2>(io.flags);
  GT.swing0[0] = f0 & 1; GT.swing0[1] = f1 & 1; GT.swing0[2] = f2 & 1; GT.swing0[3] = f3 & 1;
  GT.prevc[0] = (f0 >> 1) & 1; GT.prevc[1] = (f1 >> 1) & 1; GT.prevc[2] = (f2 >> 1) & 1; GT.prevc[3] = (f3 >> 1) & 1;
  GT.last_cmd = io.cmd; GT.step_from_change = io.step;
  float ox = 0.f, oy = 0.f;
  if (HAS_OBJ && c.gait_with_object) {                                                                      // :372-385
    // quat_apply_inverse(yaw_quat(base), object - base): a planar rotation by -yaw (q_yaw_cs: no angle, no trigonometry)
    float cy, sy;
    q_yaw_cs(bq, cy, sy);
    const V3 d = op - bp;
    ox = cy * d.x + sy * d.y; oy = cy * d.y - sy * d.x;
  }
  const float term = gait_reward(c, GT, cmd, lin_err, ang_err, ox, oy, any_nonzero_cmd, step_dt);
  // scatter the class state back: lane (leg) owns column {0:0, 1:2, 2:3, 3:1}
  io.last_air = sel4(leg, GT.last_air[0], GT.last_air[2], GT.last_air[3], GT.last_air[1]);
  io.last_con = sel4(leg, GT.last_con[0], GT.last_con[2], GT.last_con[3], GT.last_con[1]);
  io.valid = sel4(leg, GT.valid[0], GT.valid[2], GT.valid[3], GT.valid[1]);
  const int fl0 = GT.swing0[0] | (GT.prevc[0] << 1), fl1 = GT.swing0[1] | (GT.prevc[1] << 1), fl2 = GT.swing0[2] | (GT.prevc[2] << 1), fl3 = GT.swing0[3] | (GT.prevc[3] << 1);
  io.flags = leg == 0 ? fl0 : (leg == 1 ? fl2 : (leg == 2 ? fl3 : fl1));
  io.cmd = GT.last_cmd; io.step = GT.step_from_change;
  return term;
}

// The eight object reward terms (reference rewards.py:469-604), in enum order from LT_R_OBJECT_XY_POSITION; a term whose weight
// is 0 stays exactly 0.  `cn`: norm of the env's command.
__device__ __forceinline__ void object_terms(const lt_cfg& c, const Base& B, const Obj& O, float cn, float (&t)[8]) {
  const float* w = c.reward_weight;
  auto on = [&](int i, float v) { return w[i] != 0.f ? v : 0.f; };
  const V3 dpos = O.p - B.p;
  const V3 pr = qapply_inv(B.q, dpos), lr = qapply_inv(B.q, O.u - B.u), ar = qapply_inv(B.q, O.w - B.w);
  t[LT_R_OBJECT_XY_POSITION - LT_R_OBJECT_XY_POSITION] = on(LT_R_OBJECT_XY_POSITION, fsqrt(dpos.x * dpos.x + dpos.y * dpos.y) * (cn > 0.f ? 1.f : 0.f));  // :469-481
  t[LT_R_OBJECT_XY_VELOCITY - LT_R_OBJECT_XY_POSITION] = on(LT_R_OBJECT_XY_VELOCITY, lr.x * lr.x + lr.y * lr.y);                  // :483-491
  t[LT_R_OBJECT_Z_CONTACT - LT_R_OBJECT_XY_POSITION] = on(LT_R_OBJECT_Z_CONTACT, (O.last_con > 0.f && O.cur_air > 0.f) ? 1.f : 0.f);  // :596-604
  t[LT_R_OBJECT_Z_VELOCITY - LT_R_OBJECT_XY_POSITION] = on(LT_R_OBJECT_Z_VELOCITY, lr.z * lr.z);                                  // :493-501
  {                                                                                                         // :524-533
    const V3 gz = v3(0.f, 0.f, -1.f);
    const V3 gr = qapply_inv(B.q, qapply(O.q, qapply_inv(O.q, gz)));
    t[LT_R_OBJECT_ROLL_PITCH_ANGLE - LT_R_OBJECT_XY_POSITION] = on(LT_R_OBJECT_ROLL_PITCH_ANGLE, gr.y * gr.y);
  }
  t[LT_R_OBJECT_ROLL_PITCH_VELOCITY - LT_R_OBJECT_XY_POSITION] = on(LT_R_OBJECT_ROLL_PITCH_VELOCITY, ar.x * ar.x);                // :535-543
  {                                                                                                         // :545-567
    // yaw of conj(yaw_quat(robot)) * yaw_quat(object), wrapped to (-pi, pi]: the angle between the two yaw directions, from
    // their (cos, sin) pairs - one atan2f instead of three, no sincos, no quaternion product
    float cr, sr, co, so;
    q_yaw_cs(B.q, cr, sr);
    q_yaw_cs(O.q, co, so);
    float yd = atan2f(so * cr - co * sr, co * cr + so * sr);
    const float pi = 3.14159265358979323846f;
    yd = yd > 0.5f * pi ? yd - pi : yd;
    yd = yd <= -0.5f * pi ? yd + pi : yd;
    t[LT_R_OBJECT_YAW_ALIGNMENT - LT_R_OBJECT_XY_POSITION] = on(LT_R_OBJECT_YAW_ALIGNMENT, yd * yd * (cn > 0.f ? 1.f : 0.f));
  }
  {                                                                                                         // :569-594
    const bool danger = (fabsf(pr.x) > c.danger_x_max) || (fabsf(pr.y) > c.danger_y_max) || (pr.z < c.danger_z_min) ||
                        (fsqrt(lr.x * lr.x + lr.y * lr.y) > c.danger_vel_xy_max);
    t[LT_R_OBJECT_DANGEROUS_STATE - LT_R_OBJECT_XY_POSITION] = on(LT_R_OBJECT_DANGEROUS_STATE, danger ? 1.f : 0.f);
  }
}

// object_state_in_robot_frame (reference locotouch/mdp/observations.py:38-91); u16 = 16 uniforms or nullptr-like flag
__device__ __forceinline__ void object_state_obs(const lt_cfg& c, const Base& B, const Obj& O, bool noisy, const float* u16, float* out) {
  float s[13];
  const V3 pr = qapply_inv(B.q, O.p - B.p), lr = qapply_inv(B.q, O.u - B.u), ar = qapply_inv(B.q, O.w - B.w);
  Q4 qr = qmul(qconj(B.q), O.q);
  s[0] = pr.x; s[1] = pr.y; s[2] = pr.z; s[3] = lr.x; s[4] = lr.y; s[5] = lr.z;
  s[10] = ar.x; s[11] = ar.y; s[12] = ar.z;
  const bool non_contact = (O.last_con < c.obj_contact_time_threshold) && (O.cur_con < c.obj_contact_time_threshold);  // :65
  if (non_contact) {
#pragma unroll
    for (int i = 0; i < 13; ++i) s[i] = 0.f;
    qr.w = 1.f; qr.x = qr.y = qr.z = 0.f;
  }
  if (noisy) {                                                                                              // :71-83
    const int nidx[13] = {0, 1, 2, 3, 4, 5, -1, -1, -1, -1, 9, 10, 11};
#pragma unroll
    for (int i = 0; i < 13; ++i)
      if (nidx[i] >= 0) { const float n = c.obj_noise[nidx[i]]; s[i] += u16[i] * (2.f * n) - n; }
    float e[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { const float n = c.obj_noise[6 + i]; e[i] = u16[13 + i] * (2.f * n) - n; }
    qr = qmul(qr, q_from_euler(e[0], e[1], e[2]));
  }
  s[6] = qr.w; s[7] = qr.x; s[8] = qr.y; s[9] = qr.z;
#pragma unroll
  for (int i = 0; i < 13; ++i) out[i] = s[i] * c.obj_scale[i];                                              // :85-89
}

// command resampling (reference locotouch/mdp/commands.py:517-559 + UniformVelocityCommand [DEP])
__device__ __forceinline__ void command_resample(const lt_cfg& c, const float* P, const U4& u0, const U4& u1, Misc& X) {
  const float ub[3] = {u0.a, u0.c, u1.a}, uv[3] = {u0.b, u0.d, u1.b};
  float cmd[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    float lo = P[2 * d], hi = P[2 * d + 1];
    if (c.cmd_multi_sampling && P[12 + d] == 0.f) {
      const float plo = P[6 + 2 * d], phi = P[6 + 2 * d + 1], p = c.cmd_new_probs;
      if (ub[d] < p) { hi = plo; }
      else if (ub[d] < 1.f - p) { lo = plo; hi = phi; }
      else { lo = phi; }
    }
    cmd[d] = lo + uv[d] * (hi - lo);
  }
  if (c.cmd_binary_maximal) {  // commands.py:518-521: maximal_command_sampling[randint(8)] * (upper range bounds); index = 4 i + 2 j + k over [-1, 1]
    const int idx = min((int)(u0.a * 8.f), 7);
    cmd[0] = (idx & 4) ? P[1] : -P[1]; cmd[1] = (idx & 2) ? P[3] : -P[3]; cmd[2] = (idx & 1) ? P[5] : -P[5];
  }
  X.cmd = v3(cmd[0], cmd[1], cmd[2]);
  X.cmd_buf = X.cmd;
  if (!c.cmd_binary_maximal) X.cmd_standing = (u1.c <= P[16]) ? 1.f : 0.f;  // (that branch leaves is_standing_env alone)
  X.cmd_time_left = c.cmd_resample_time[0] + u1.d * (c.cmd_resample_time[1] - c.cmd_resample_time[0]);
}
__device__ __forceinline__ void command_resample(const lt_cfg& c, const float* P, uint32_t env, uint64_t step, uint32_t stream, Misc& X) {
  command_resample(c, P, rng4(c.seed, env, step, stream), rng4(c.seed, env, step, stream + 1), X);
}

// stores of the observation rows (LT_STORE_MODE: 0 plain, 1 non-temporal - the default: the rows are read next by another kernel on other XCDs,
// measured 56.3 M against 54.1 M env-steps/s at 4096 envs -, 2 write-through sc0 sc1: slower)
#ifndef LT_STORE_MODE
#define LT_STORE_MODE 1
#endif
// state quad arrays at the end of a step.  Helper form (one tile per CU): non-temporal - the lines leave the L2 while the tile's
// other stores issue instead of in the write-back at the kernel's end (4096 envs: 62.4 -> 61.7 us per rollout step); the
// one-wave form of large grids keeps plain stores (no difference measured at 32768 envs).
#define ST_STATE(p, v) do { if (HELPERS) __builtin_nontemporal_store((v), (p)); else *(p) = (v); } while (0)
__device__ __forceinline__ void st_out(float* p, float v) {
#if LT_STORE_MODE == 1
  __builtin_nontemporal_store(v, p);
#elif LT_STORE_MODE == 2
  asm volatile("global_store_dword %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
#else
  *p = v;
#endif
}
__device__ __forceinline__ void st_out(unsigned* p, unsigned v) {
#if LT_STORE_MODE == 1
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}
__device__ __forceinline__ unsigned bf16_rne(float x) {  // round to nearest even (no NaNs on this path)
  const unsigned u = __float_as_uint(x);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
// =====================================================================================================
// the step kernel
// =====================================================================================================
#ifdef LT_STAMPS
#define LT_STAMP(i) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); stamps_[i] = t_; } while (0)
#else
#define LT_STAMP(i) do { } while (0)
#endif

// HELPERS (small grids, <= 2 workgroups per CU - at 4096 envs there is ONE 16-env tile per CU, and a lone wave per SIMD has
// nothing to hide latency behind): the workgroup is 4 waves, one per SIMD, that split one 16-env tile by TASK:
//   wave 0    the env step proper (everything below); it leaves the newest observation frame, the reset flags and the
//             curriculum record in LDS and meets the others at ONE barrier;
//   wave 1/2  the 6-deep history rows of the policy / critic group: the previous rows (16 x OBS floats) come in by LDS-DMA
//             (global_load_lds_dwordx4, no VGPRs) at kernel start, the shifted 5/6 of every row is stored back while the
//             physics runs, the newest frame (and whole rows of envs that reset) after the barrier;
//   wave 3    the curriculum / population-gate publish (lt_post.h) - its write-through stores and the ticket round trip
//             stall this wave, not the step.
// Large grids run the single-wave form (HELPERS = false): there every SIMD already has work, the history pass stages ONE
// group's old rows in LDS at a time (22 KB: four tiles per CU still fit) and the tail runs inline.
#ifndef LT_STEP_MIN_WAVES_LARGE
#define LT_STEP_MIN_WAVES_LARGE 1
#endif
// ---- helper form: the physics of one 16-env tile runs on three waves.  Wave 0 keeps the leg dynamics; per substep it
//      publishes (cos q, sin q, base state), wave 1 returns the CRBA part and wave 2 - which owns the object's state for the
//      whole step - the object part, through LDS mailboxes [slot][lane] (same lane = env * 4 + leg mapping in all waves, so
//      every slot row is one conflict-free 256-B line).  Two workgroup barriers per substep order the exchange:
//        A: wave 0's inputs are in LDS          (helpers start)
//        B: the helpers' results are in LDS     (wave 0 has meanwhile done FK / RNEA / contacts / the backward pass)
//      `s_waitcnt lgkmcnt(0); s_barrier` only: no vmcnt drain (the history waves' DMA and wave 0's stores stay in flight).
#ifndef LT_CHAIN_POLL_MAX
#define LT_CHAIN_POLL_MAX (1 << 20)  // polls of the chained command-block hand-off before it is declared lost (s_sleep 2 = ~128 clocks each)
#endif
constexpr int MB_IN = 22, MB_CRBA = 43, MB_OBJ = 15, MB_OFIN = 13;
__device__ __forceinline__ void wg_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#ifdef LT_STAMPS
#define LT_TIMED_BARRIER(acc) do { unsigned long long t0_, t1_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0_) :: "memory"); \
    wg_barrier_lds(); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1_) :: "memory"); *(acc) += t1_ - t0_; } while (0)
#else
#define LT_TIMED_BARRIER(acc) wg_barrier_lds()
#endif
struct HelperParts {
  static constexpr bool external = true;
  float (*in)[64]; float (*crba)[64]; float (*obj)[64];
  int lane;
  unsigned long long* wait;  // LT_STAMPS builds: cycles spent in barriers A ([0]) and B ([1]); [2] last mark, [3..10] phase sums
  __device__ __forceinline__ void mark(int i) const {
#ifdef LT_STAMPS
    unsigned long long t_;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");
    if (i > 0) wait[3 + i] += t_ - wait[2];
    wait[2] = t_;
#else
    (void)i;
#endif
  }
  __device__ __forceinline__ void publish(const float (&cq)[3], const float (&sq)[3], const float (&qd)[3], const Base& B) const {
    const float v[MB_IN] = {cq[0], cq[1], cq[2], sq[0], sq[1], sq[2], B.p.x, B.p.y, B.p.z, B.q.w, B.q.x, B.q.y, B.q.z,
                            B.u.x, B.u.y, B.u.z, B.w.x, B.w.y, B.w.z, qd[0], qd[1], qd[2]};
#pragma unroll
    for (int i = 0; i < MB_IN; ++i) in[i][lane] = v[i];
    LT_TIMED_BARRIER(wait);  // A
  }
  __device__ __forceinline__ void fetch(PhysExt& e) const {
    LT_TIMED_BARRIER(wait + 1);  // B
    float c[MB_CRBA], o[MB_OBJ];
#pragma unroll
    for (int i = 0; i < MB_CRBA; ++i) c[i] = crba[i][lane];
#pragma unroll
    for (int i = 0; i < MB_OBJ; ++i) o[i] = obj[i][lane];
    CrbaOut& r = e.crba;
    r.h00 = c[0]; r.h01 = c[1]; r.h02 = c[2]; r.h11 = c[3]; r.h12 = c[4]; r.h22 = c[5];
#pragma unroll
    for (int j = 0; j < 3; ++j) { r.bn[j] = v3(c[6 + 3 * j], c[7 + 3 * j], c[8 + 3 * j]); r.bl[j] = v3(c[15 + 3 * j], c[16 + 3 * j], c[17 + 3 * j]); }
    r.Io.xx = c[24]; r.Io.xy = c[25]; r.Io.xz = c[26]; r.Io.yy = c[27]; r.Io.yz = c[28]; r.Io.zz = c[29];
    r.mc = v3(c[30], c[31], c[32]); r.m = c[33];
    r.tb[0] = c[34]; r.tb[1] = c[35]; r.tb[2] = c[36];
    r.pbn = v3(c[37], c[38], c[39]); r.pbf = v3(c[40], c[41], c[42]);
    ObjOut& q = e.obj;
    q.pb_n = v3(o[0], o[1], o[2]); q.pb_f = v3(o[3], o[4], o[5]); q.obj_part = v3(o[6], o[7], o[8]);
    q.trunk_part = v3(o[9], o[10], o[11]); q.plate = v3(o[12], o[13], o[14]);
  }
};

// RB: the observation rows behind a.obs_prev / a.obs_next are bf16 (rollout-storage slots of BASELINE config 5), two columns per
// 32-bit word; the newest frame is rounded to nearest-even when it enters a row, older frames are carried bit for bit.
template <int TASK, int MODE, bool HELPERS, bool RB = false>
__global__ __launch_bounds__(HELPERS ? 256 : 64, HELPERS ? 1 : LT_STEP_MIN_WAVES_LARGE) void lt_step_kernel(const KArgs a) {
#ifdef LT_STAMPS
  unsigned long long stamps_[8], bar_wait_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 8; ++i) stamps_[i] = 0;
#endif
  LT_STAMP(0);
  constexpr bool HAS_OBJ = TASK != LT_TASK_LOCOMOTION;
  constexpr bool TAC = TASK == K_TASK_TACTILE;
  constexpr int FRAME = HAS_OBJ ? 58 : 45;
  constexpr int OBS = FRAME * 6;
  const int lane = threadIdx.x & 63;
  const int wave = HELPERS ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;  // scalar: role branches are s_cbranch
  const int leg = lane & 3;
  const long long gid = (long long)blockIdx.x * 64 + lane;  // == env*4 + leg
  const long long env = gid >> 2;
  const long long q4 = a.npad * 4;
  char* const arena = a.arena;
  auto F = [&](int field, int q) -> float* { return (float*)arena + (long long)(k_cumq.v[field] + q) * q4 + gid; };

  // ---- prologue: ONE memory round trip.  The (cfg, layout) block (two 16-B loads per lane, staged into LDS: ~130 scattered
  //      scalar loads - each its own L2 round trip for a lone wave - become cheap ds_reads) and the state the physics needs
  //      (one coalesced 256-B access per quad array) are issued together; vmcnt retires in order, so the 44-KB LDS-DMA of the
  //      old observation rows is issued only AFTER them - it then lands behind the physics instead of in front of it.
  //      Everything the physics never touches is loaded after the decimation loop (registers). ----
  __shared__ lt_dev_args s_d;
  static_assert(sizeof(lt_dev_args) % 16 == 0, "lt_dev_args must be a multiple of 16 bytes");
  constexpr int STG = (int)(sizeof(lt_dev_args) / 16);
  static_assert(STG <= 128, "staging assumes at most two 16-byte pieces per lane");
  const uint4* const stg_src = (const uint4*)a.d;
  uint4 stg0 = make_uint4(0, 0, 0, 0), stg1 = stg0;
  struct {
    float root_pos, root_quat, root_lin, root_ang, obj_pos, obj_quat, obj_lin, obj_ang, obj_timers, obj_params, env_params, trunk_fh;
    float q[3], qd[3], raw[3], fh[12], cur_air, cur_con, last_air, last_con, mu;
    long long ep0;
  } hot;
  hot.ep0 = 0;
  if (wave == 0) {
    stg0 = stg_src[lane < STG ? lane : 0]; stg1 = stg_src[lane + 64 < STG ? lane + 64 : 0];
    hot.root_pos = *F(LT_F_ROOT_POS, 0); hot.root_quat = *F(LT_F_ROOT_QUAT, 0);
    hot.root_lin = *F(LT_F_ROOT_LIN_VEL_W, 0); hot.root_ang = *F(LT_F_ROOT_ANG_VEL_W, 0);
    hot.obj_pos = *F(LT_F_OBJ_POS, 0); hot.obj_quat = *F(LT_F_OBJ_QUAT, 0);
    hot.obj_lin = *F(LT_F_OBJ_LIN_VEL_W, 0); hot.obj_ang = *F(LT_F_OBJ_ANG_VEL_W, 0);
    hot.obj_timers = *F(LT_F_OBJ_TIMERS, 0); hot.obj_params = *F(LT_F_OBJ_PARAMS, 0);
    hot.env_params = *F(LT_F_ENV_PARAMS, 0); hot.trunk_fh = *F(LT_F_TRUNK_FORCE_HIST, 0);
#pragma unroll
    for (int k = 0; k < 3; ++k) { hot.q[k] = *F(LT_F_JOINT_POS, k); hot.qd[k] = *F(LT_F_JOINT_VEL, k); hot.raw[k] = *F(LT_F_ACT_RAW, k); }
#pragma unroll
    for (int i = 0; i < 12; ++i) hot.fh[i] = *F(LT_F_FORCE_HIST, i);
    hot.cur_air = *F(LT_F_FOOT_CUR_AIR, 0); hot.cur_con = *F(LT_F_FOOT_CUR_CONTACT, 0);
    hot.last_air = *F(LT_F_FOOT_LAST_AIR, 0); hot.last_con = *F(LT_F_FOOT_LAST_CONTACT, 0);
    hot.mu = *F(LT_F_FOOT_FRICTION, 0);
    if (TAC && MODE == MODE_STEP) hot.ep0 = a.ep_len[env];
    uint4* dst = (uint4*)&s_d;
    if (lane < STG) dst[lane] = stg0;
    if (lane + 64 < STG) dst[lane + 64] = stg1;
  } else if (HELPERS && wave == 1) {  // the CRBA / bias helper adds the trunk's rigid body: its randomised mass
    hot.env_params = *F(LT_F_ENV_PARAMS, 0);
  } else if (HELPERS && HAS_OBJ && wave == 2) {  // the object helper owns the object's state during the physics
    hot.obj_pos = *F(LT_F_OBJ_POS, 0); hot.obj_quat = *F(LT_F_OBJ_QUAT, 0);
    hot.obj_lin = *F(LT_F_OBJ_LIN_VEL_W, 0); hot.obj_ang = *F(LT_F_OBJ_ANG_VEL_W, 0);
    hot.obj_params = *F(LT_F_OBJ_PARAMS, 0); hot.env_params = *F(LT_F_ENV_PARAMS, 0);
  }
  __shared__ float s_frame[2][16][64];
  // the 16 old rows of history group g (one contiguous chunk of 16*OBS floats) -> LDS by LDS-DMA: each wave-instruction moves
  // 64 lanes x 16 B = 1 KiB, lane-linear in LDS, no VGPRs
  constexpr int EB = RB ? 2 : 4;  // bytes per row element
  // (`prev`: the group's previous rows, a.obs_prev[g] picked by the caller - indexing the by-value kernel argument struct with a
  //  runtime g inside this lambda put the whole struct on the stack: 128 B of scratch in front of every launch)
  auto dma_old_rows = [&](const float* prev, float* dst) __attribute__((always_inline)) {  // (a lambda called twice is a real call otherwise: scratch, vmcnt(0))
    constexpr int CHUNK16 = 16 * OBS * EB / 16;  // 16-byte pieces per group (16 * OBS * EB is a multiple of 16 for both tasks)
    const char* gsrc = (const char*)prev + (long long)blockIdx.x * 16 * OBS * EB;
    for (int i = 0; i < (CHUNK16 + 63) / 64; ++i) {
      const int v = i * 64 + lane;
      if (v < CHUNK16)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + v * 16),
                                         (__attribute__((address_space(3))) void*)(dst + i * 256), 16, 0, 0);
    }
  };
  // bf16 rows: a lane owns column PAIRS p = lane, lane + 64, ... (columns 2p, 2p + 1: one 32-bit word) of every row
  constexpr int NPC = (OBS / 2 + 63) / 64;
  static_assert(OBS % 2 == 0, "bf16 rows are handled as column pairs");
  __shared__ int s_fill[16];
  __shared__ short s_tab[1][704];  // observation history tables (src[352] | frame[352]) of the one-wave form
  __shared__ __attribute__((aligned(16))) float s_old[HELPERS ? 2 * 16 * OBS : 16 * OBS];  // old history rows (one-wave form: one group at a time)
  __shared__ float s_cur[HELPERS ? 64 * 5 : 1];  // wave 0 -> wave 3: this step's curriculum record per lane
  __shared__ float s_mb_in[HELPERS ? MB_IN : 1][64], s_mb_crba[HELPERS ? MB_CRBA : 1][64], s_mb_obj[HELPERS ? MB_OBJ : 1][64],
      s_mb_ofin[HELPERS ? MB_OFIN : 1][64];  // physics mailboxes (HelperParts)
  __shared__ float s_mb_rng[HELPERS ? 16 : 1][64];  // wave 3 -> wave 0: this step's observation-noise uniforms
  // wave 3 -> wave 0: the step's EVENT draws (reset events, command resampling, pushes), computed beside the physics.  An env
  // uses them on few steps, but some tile of the launch does on every step and the launch ends with its slowest tile: 13 Philox
  // calls (~8 k cycles) sat on that tile's critical path.  Slot = one rng4 result of an owner lane: [slot][component][lane].
  // wave 0 <-> waves 1 / 2 after the physics: the robot's final base state, the tracking errors and this leg's sensor timers go out
  // (s_mb_fin), the gait term with its updated class state (wave 1) and the eight object terms (wave 2, the object's owner) come
  // back (s_mb_rew) - evaluated beside wave 0's terminations and remaining 15 terms, between two extra barriers (D, E)
  constexpr int MB_FIN = 24, MB_REW = 17;
  __shared__ float s_mb_fin[HELPERS ? MB_FIN : 1][64], s_mb_rew[HELPERS ? MB_REW : 1][64];
  constexpr int BANK_SLOTS = 7;
  __shared__ float s_bank[HELPERS ? BANK_SLOTS : 1][4][64];
  if (!HELPERS) {
    const ObsTable& tab = HAS_OBJ ? k_obs_tab_teacher : k_obs_tab_loco;
    for (int i = lane; i < 352; i += 64) { s_tab[0][i] = tab.src[i]; s_tab[0][352 + i] = tab.frame[i]; }
    dma_old_rows(a.obs_prev[0], s_old);  // the policy group's old history rows, behind the state loads (the critic group's follow at the end)
  }
#ifdef LT_STAMPS
  unsigned long long pro_a_, pro_b_;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pro_a_) :: "memory");
#endif
  if (HELPERS) __syncthreads();  // B0: the (cfg, layout) block is in LDS
  else wg_barrier_lds();       //     (one-wave form: no vmcnt drain - the row DMA stays in flight beside the physics)
#ifdef LT_STAMPS
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pro_b_) :: "memory");
#endif
  const lt_cfg& c = s_d.cfg;
  const lt_layout& L = s_d.layout;
  // the command block: a copy in LDS - helper form: wave 3 leaves it there beside the last physics substep; one-wave form: requested
  // here, stored behind the physics (below).  Read from global memory it cost the resetting / resampling waves of the one-wave form a
  // dozen DEPENDENT loads (commands.py:517-559 branches on the block's values), each behind an `s_waitcnt vmcnt(0)`.  (The terms-only
  // and reset-all launches read it where it is.)
  __shared__ float s_P[MODE == MODE_STEP ? 32 : 1];
  const float* P = MODE == MODE_STEP ? (const float*)s_P : (const float*)(arena + L.off_cmd_params);
  float p_own = 0.f;
  if (!HELPERS && MODE == MODE_STEP) p_own = ((const float*)(arena + L.off_cmd_params))[lane & 31];
  const uint64_t step = MODE == MODE_RESET_ALL ? 0ull : (uint64_t)(((const long long*)(arena + L.off_counters))[0] + (MODE == MODE_STEP ? a.step_offset : 0));
  const float step_dt = c.sim_dt * (float)c.decimation;
  const uint32_t ekey = (uint32_t)env + (uint32_t)c.env_index_offset;  // RNG stream key of this env (global index over all ranks)

  if (HELPERS && __builtin_expect(wave != 0, 0)) {  // (unlikely: keeps wave 0's path the fall-through behind B0 - see DESIGN.md "far jump")
    // waves 1 / 2: per-lane column routing of history group g (lane l owns columns l, l + 64, ... of EVERY row, so the table
    // lookups are done once per lane), and the group's rows
    const int g = wave == 2 ? 1 : 0;
    constexpr int NCH = (OBS + 63) / 64;
    int src[NCH], frm[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int col = i * 64 + lane;
      const int cc = col < OBS ? col : OBS - 1;
      const ObsTable& tabc = HAS_OBJ ? k_obs_tab_teacher : k_obs_tab_loco;
      src[i] = tabc.src[cc];    // >= 0: old column (one slot newer); < 0: newest frame element -src-1
      frm[i] = tabc.frame[cc];  // newest-frame element of this column's term (rows that were just reset)
    }
    float* const rows = (g ? a.obs_next[1] : a.obs_next[0]) + (long long)blockIdx.x * 16 * OBS * EB / 4;
    const float* const old = s_old + g * (16 * OBS * EB / 4);
    // bf16 rows: routing of this lane's column pairs (both halves), the rows as 32-bit words, the staged old rows as 16-bit elements
    constexpr int NPR = RB ? NPC : 1;  // (no registers, no scratch in the f32 instantiation)
    int psrc[NPR][2], pfrm[NPR][2];
    if constexpr (RB) {
#pragma unroll
      for (int i = 0; i < NPR; ++i) {
        const int pr = i * 64 + lane, c0 = 2 * (pr < OBS / 2 ? pr : OBS / 2 - 1);
        const ObsTable& tabc = HAS_OBJ ? k_obs_tab_teacher : k_obs_tab_loco;
        psrc[i][0] = tabc.src[c0]; psrc[i][1] = tabc.src[c0 + 1];
        pfrm[i][0] = tabc.frame[c0]; pfrm[i][1] = tabc.frame[c0 + 1];
      }
    }
    unsigned* const rows32 = (unsigned*)rows;
    const unsigned short* const old16 = (const unsigned short*)old;
    // ---- physics helpers: as many (A, B) barrier pairs as wave 0 runs substeps ----
    {
      const int nsub = c.decimation * c.phys_substeps;
      const float hs = c.sim_dt / (float)c.phys_substeps;
      const float sxh = leg < 2 ? 1.f : -1.f, syh = (leg & 1) ? 1.f : -1.f;
      const float sgnh[4] = {1.f, sxh, syh, sxh * syh};
      Obj Oh;
      float trunk_mu = 0.f;
      const float trunk_mass_add_h = wave == 1 ? qbcast<0>(hot.env_params) : 0.f;
      float bank_mat_c = 0.f;
      GaitIO gio = {0.f, 0.f, 0.f, 0, v3(0, 0, 0), 0.f};
      if (wave == 1 && MODE == MODE_STEP) {  // the gait class state of this lane / env: wave 1 evaluates the gait term (below)
        gio.last_air = *F(LT_F_GAIT_LAST_AIR, 0); gio.last_con = *F(LT_F_GAIT_LAST_CONTACT, 0); gio.valid = *F(LT_F_GAIT_VALID_LAST_AIR, 0);
        gio.flags = *(const int*)F(LT_F_GAIT_FLAGS, 0);
        const float t = *F(LT_F_GAIT_CMD, 0);
        gio.cmd = v3(qbcast<0>(t), qbcast<1>(t), qbcast<2>(t)); gio.step = qbcast<3>(t);
      }
      if (HAS_OBJ && wave == 2) {
        float t;
        t = hot.obj_pos; Oh.p = v3(qbcast<0>(t), qbcast<1>(t), qbcast<2>(t));
        t = hot.obj_quat; Oh.q.w = qbcast<0>(t); Oh.q.x = qbcast<1>(t); Oh.q.y = qbcast<2>(t); Oh.q.z = qbcast<3>(t);
        t = hot.obj_lin; Oh.u = v3(qbcast<0>(t), qbcast<1>(t), qbcast<2>(t));
        t = hot.obj_ang; Oh.w = v3(qbcast<0>(t), qbcast<1>(t), qbcast<2>(t));
        t = hot.obj_params; Oh.rad = qbcast<0>(t); Oh.len = qbcast<1>(t); Oh.mass = qbcast<2>(t); Oh.mu = qbcast<3>(t);
        Oh.cur_air = Oh.cur_con = Oh.last_air = Oh.last_con = 0.f;
        trunk_mu = qbcast<1>(hot.env_params);
      }
      if (wave == 1 || wave == 2) {
        // history rows, step 0 (waves 1 / 2): the group's 16 old rows = one contiguous chunk of 16*OBS floats, fetched by LDS-DMA -
        // each wave-instruction moves 64 lanes x 16 B = 1 KiB, lane-linear in LDS, no VGPRs - NOW, so that it lands beside the
        // physics (the physics barriers wait for LDS traffic only; B0, a full barrier, is behind us).
        // (the routing tables were read before: their loads must not queue behind the DMA - vmcnt retires in order)
        dma_old_rows(g ? a.obs_prev[1] : a.obs_prev[0], s_old + g * (16 * OBS * EB / 4));
      }
      // wave 3: the observation-noise uniforms of this step (Philox is ~1 k cycles per call).  Per lane: joint-pos and joint-vel noise of
      // its leg, object-noise block `leg`, base-noise block (legs 2, 3 -> RS_NOISE_BASE, + 1).  Same streams and keys as the
      // inline form: bit-identical draws.  Block i -> rows 4 i .. 4 i + 3 of s_mb_rng.
      auto noise_draw = [&](int i) __attribute__((always_inline)) {
        const uint32_t stream = i == 0 ? RS_NOISE_JPOS + leg : (i == 1 ? RS_NOISE_JVEL + leg : (i == 2 ? RS_NOISE_OBJ + leg : RS_NOISE_BASE + (leg == 3 ? 1 : 0)));
        const U4 u = rng4(c.seed, ekey, step, stream);
        s_mb_rng[4 * i + 0][lane] = u.a; s_mb_rng[4 * i + 1][lane] = u.b; s_mb_rng[4 * i + 2][lane] = u.c; s_mb_rng[4 * i + 3][lane] = u.d;
      };
      for (int it = 0; it < nsub; ++it) {
        wg_barrier_lds();  // A
        if (wave == 1) {
          const float cqh[3] = {s_mb_in[0][lane], s_mb_in[1][lane], s_mb_in[2][lane]};
          const float sqh[3] = {s_mb_in[3][lane], s_mb_in[4][lane], s_mb_in[5][lane]};
          const float qdh[3] = {s_mb_in[19][lane], s_mb_in[20][lane], s_mb_in[21][lane]};
          // the base's packed motion, as physics_substep forms it: (omega | 0), (v | +g z) in base coordinates
          const M3 R0h = quat_to_mat(s_mb_in[9][lane], s_mb_in[10][lane], s_mb_in[11][lane], s_mb_in[12][lane]);
          const P3 WVh = tmul(R0h, pair(v3(s_mb_in[16][lane], s_mb_in[17][lane], s_mb_in[18][lane]), v3(s_mb_in[13][lane], s_mb_in[14][lane], s_mb_in[15][lane])));
          const CrbaOut r = crba_part(sgnh, leg, cqh, sqh, qdh, pair(lo(WVh), v3(0, 0, 0)), pair(hi(WVh), c.gravity * row(R0h, 2)), trunk_mass_add_h);
          const float v[MB_CRBA] = {r.h00, r.h01, r.h02, r.h11, r.h12, r.h22,
                                    r.bn[0].x, r.bn[0].y, r.bn[0].z, r.bn[1].x, r.bn[1].y, r.bn[1].z, r.bn[2].x, r.bn[2].y, r.bn[2].z,
                                    r.bl[0].x, r.bl[0].y, r.bl[0].z, r.bl[1].x, r.bl[1].y, r.bl[1].z, r.bl[2].x, r.bl[2].y, r.bl[2].z,
                                    r.Io.xx, r.Io.xy, r.Io.xz, r.Io.yy, r.Io.yz, r.Io.zz, r.mc.x, r.mc.y, r.mc.z, r.m,
                                    r.tb[0], r.tb[1], r.tb[2], r.pbn.x, r.pbn.y, r.pbn.z, r.pbf.x, r.pbf.y, r.pbf.z};
#pragma unroll
          for (int i = 0; i < MB_CRBA; ++i) s_mb_crba[i][lane] = v[i];
        } else if (HAS_OBJ && wave == 2) {
          Base Bh;
          Bh.p = v3(s_mb_in[6][lane], s_mb_in[7][lane], s_mb_in[8][lane]);
          Bh.q.w = s_mb_in[9][lane]; Bh.q.x = s_mb_in[10][lane]; Bh.q.y = s_mb_in[11][lane]; Bh.q.z = s_mb_in[12][lane];
          Bh.u = v3(s_mb_in[13][lane], s_mb_in[14][lane], s_mb_in[15][lane]);
          Bh.w = v3(s_mb_in[16][lane], s_mb_in[17][lane], s_mb_in[18][lane]);
          const ObjOut o = object_part<TAC>(c, hs, leg, Bh, Oh, trunk_mu);
          const float v[MB_OBJ] = {o.pb_n.x, o.pb_n.y, o.pb_n.z, o.pb_f.x, o.pb_f.y, o.pb_f.z, o.obj_part.x, o.obj_part.y, o.obj_part.z,
                                   o.trunk_part.x, o.trunk_part.y, o.trunk_part.z, o.plate.x, o.plate.y, o.plate.z};
#pragma unroll
          for (int i = 0; i < MB_OBJ; ++i) s_mb_obj[i][lane] = v[i];
          if (it == nsub - 1) {  // the object's state after the last substep, for wave 0's terminations / rewards / observations
            const float w[MB_OFIN] = {Oh.p.x, Oh.p.y, Oh.p.z, Oh.q.w, Oh.q.x, Oh.q.y, Oh.q.z, Oh.u.x, Oh.u.y, Oh.u.z, Oh.w.x, Oh.w.y, Oh.w.z};
#pragma unroll
            for (int i = 0; i < MB_OFIN; ++i) s_mb_ofin[i][lane] = w[i];
          }
        } else if (wave == 3 && MODE == MODE_STEP) {
          // the step's event draws, at most two Philox calls per substep (~2 k of the substep's ~11 k cycles).  Same seeds, keys
          // and streams as the inline form: bit-identical uniforms.  Slots 0-2 are per lane (stream + leg); slots 3-5 hold four
          // per-env streams, one per lane of the quad; slot 6 is the object-material pool entry picked by RS_RESET_MAT's third draw.
          auto put = [&](int slot, const U4& u) { s_bank[slot][0][lane] = u.a; s_bank[slot][1][lane] = u.b; s_bank[slot][2][lane] = u.c; s_bank[slot][3][lane] = u.d; };
          const int last = nsub - 1;
          // chained steps: the previous step's population pass, once per launch, here - ~6 k cycles inside an 11 k-cycle substep
          // (decide_first == 3: test hook of lt_env_defer_gate mode 3 - the publisher announces a wrong step id once, so the consumers'
          //  bounded poll below times out and raises the error word)
          if (it == 0 && a.decide_first && blockIdx.x == 0) curriculum_decide(c, L, arena, 0, (long long)step + (a.decide_first == 3 ? 1 : 0), (int)((step - 1) & 1));
          if (it == last) {
            // the command block of THIS step -> LDS.  Chained: wait until the launch's decision is final (it has been for ~35 us)
            // and read it with sc1 loads; nobody in this launch read the block earlier, so no cache of this XCD holds an older line.
            const float* const Pg = (const float*)(arena + L.off_cmd_params);
            if (a.decide_first) {
              long long* const cnt = (long long*)(arena + L.off_counters);
              // bounded poll: the flag has normally been up for ~30 us when this wave looks.  A host-side bookkeeping slip (a launch
              // whose publisher announces another step id) must end as an ERROR, not as a hung GPU: after LT_CHAIN_POLL_MAX polls
              // (~50 ms) the wave raises counters[1] - lt_env_check turns it into LT_EHIP - and goes on with the block as it is.
              int polls = 0;
              while (__hip_atomic_load(cnt + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (long long)step && polls < LT_CHAIN_POLL_MAX) {
                __builtin_amdgcn_s_sleep(2);
                ++polls;
              }
              if (polls >= LT_CHAIN_POLL_MAX && lane == 0) __hip_atomic_fetch_add((unsigned long long*)cnt + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (lane < 31) s_P[lane] = __hip_atomic_load(Pg + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (lane < 31) {
              s_P[lane] = Pg[lane];
            }
          }
          if (it == 0) {
            put(0, rng4(c.seed, ekey, step, RS_RESET_ROOT + leg));
            put(1, rng4(c.seed, ekey, step, RS_RESET_JOINT + leg));
          }
          if (it == 1 || (it == last && last < 1)) {
            put(2, rng4(c.seed, ekey, step, RS_RESET_JOINT + 4 + leg));
            const U4 ua = rng4(c.seed, ekey, step, leg == 0 ? RS_RESET_MAT : (leg == 1 ? RS_RESET_OBJ : (leg == 2 ? RS_RESET_OBJ + 1 : RS_RESET_EVENT)));
            put(3, ua);
            bank_mat_c = qbcast<0>(ua.c);
          }
          if (it == 2 || (it == last && last < 2)) {
            put(4, rng4(c.seed, ekey, step, leg == 0 ? RS_CMD_RESET : (leg == 1 ? RS_CMD_RESET + 1 : (leg == 2 ? RS_CMD_TIMER : RS_CMD_TIMER + 1))));
            put(5, rng4(c.seed, ekey, step, leg == 0 ? RS_PUSH_ROBOT : (leg == 1 ? RS_PUSH_ROBOT + 1 : (leg == 2 ? RS_PUSH_OBJ : RS_PUSH_OBJ + 1))));
          }
          if ((it == 3 || (it == last && last < 3)) && HAS_OBJ && c.obj_material_buckets > 0) {
            const int b = min((int)(bank_mat_c * (float)c.obj_material_buckets), c.obj_material_buckets - 1);
            put(6, rng4(c.seed, (uint32_t)b, ~0ull, RS_BUCKET_OBJ));
          }
          // one of the step's four observation-noise draws per substep (the rest, if there are fewer substeps, behind the loop):
          // drawn between D and E they ended ~3 k cycles after wave 0 reached barrier C
          if (it < 4) noise_draw(it);
        }
        wg_barrier_lds();  // B
        if (wave == 1 || wave == 2) {
          // history rows, this substep's share: row(t) is built from row(t-1) - every term block shifts left by one frame and takes
          // the newest frame from wave 0 at the end.  The shifted 5/6 of the rows go out HERE, between B and the next A, while
          // wave 0 eliminates, solves and integrates (~2.4 k cycles in which the helpers have nothing else to do); as one pass
          // behind the last substep it kept wave 0 waiting ~3 k cycles at barrier C.  In-place safety (rows updated in the arena):
          // the DMA of the whole chunk has landed (vmcnt(0)) before this wave stores anything, and only this wave touches the
          // group's rows.  (the builtin, not an asm: the waitcnt pass then knows the row DMA has landed; behind an opaque wait it
          //  re-waits vmcnt(0) - for every store of the loop - before each LDS read of the staged rows.  0x0070 = vmcnt 0, lgkmcnt 0)
          if (it == 0) __builtin_amdgcn_s_waitcnt(0x0070);
          const int r0 = (16 * it) / nsub, r1 = (16 * (it + 1)) / nsub;
          for (int r = r0; r < r1; ++r) {
            if constexpr (RB) {  // pairs whose two columns both come from the old row; a pair that takes a newest-frame column goes out behind B1
#pragma unroll
              for (int i = 0; i < NPR; ++i) {
                const int pr = i * 64 + lane;
                if (pr < OBS / 2 && psrc[i][0] >= 0 && psrc[i][1] >= 0)
                  st_out(&rows32[r * (OBS / 2) + pr], (unsigned)old16[r * OBS + psrc[i][0]] | ((unsigned)old16[r * OBS + psrc[i][1]] << 16));
              }
            } else {
#pragma unroll
              for (int i = 0; i < NCH; ++i) {
                const int col = i * 64 + lane;
                if (col < OBS && src[i] >= 0) st_out(&rows[r * OBS + col], old[r * OBS + src[i]]);
              }
            }
          }
        }
      }
      if (MODE == MODE_STEP) {
        // ---- D .. E: the gait term (wave 1) and the object terms (wave 2) beside wave 0's own terms ----
        wg_barrier_lds();  // D: wave 0's hand-over (s_mb_fin) is in LDS
        if (wave == 1) {
          const V3 bp = v3(s_mb_fin[0][lane], s_mb_fin[1][lane], s_mb_fin[2][lane]);
          Q4 bq; bq.w = s_mb_fin[3][lane]; bq.x = s_mb_fin[4][lane]; bq.y = s_mb_fin[5][lane]; bq.z = s_mb_fin[6][lane];
          const V3 op = HAS_OBJ ? v3(s_mb_ofin[0][lane], s_mb_ofin[1][lane], s_mb_ofin[2][lane]) : v3(0, 0, 0);
          const V3 cmdh = v3(s_mb_fin[21][lane], s_mb_fin[22][lane], s_mb_fin[23][lane]);
          const float term = gait_term<HAS_OBJ>(c, leg, s_mb_fin[16][lane], s_mb_fin[17][lane], s_mb_fin[18][lane], gio, bp, bq, op, cmdh,
                                                s_mb_fin[13][lane], s_mb_fin[14][lane], s_P[26] != 0.f, step_dt);
          const float v[9] = {term, gio.last_air, gio.last_con, gio.valid, __int_as_float(gio.flags), gio.cmd.x, gio.cmd.y, gio.cmd.z, gio.step};
#pragma unroll
          for (int i = 0; i < 9; ++i) s_mb_rew[i][lane] = v[i];
        } else if (HAS_OBJ && wave == 2) {
          Base Bf;
          Bf.p = v3(s_mb_fin[0][lane], s_mb_fin[1][lane], s_mb_fin[2][lane]);
          Bf.q.w = s_mb_fin[3][lane]; Bf.q.x = s_mb_fin[4][lane]; Bf.q.y = s_mb_fin[5][lane]; Bf.q.z = s_mb_fin[6][lane];
          Bf.u = v3(s_mb_fin[7][lane], s_mb_fin[8][lane], s_mb_fin[9][lane]);
          Bf.w = v3(s_mb_fin[10][lane], s_mb_fin[11][lane], s_mb_fin[12][lane]);
          Oh.last_con = s_mb_fin[19][lane]; Oh.cur_air = s_mb_fin[20][lane];  // (the object's contact timers are kept by wave 0)
          float ot[8];
          object_terms(c, Bf, Oh, s_mb_fin[15][lane], ot);
#pragma unroll
          for (int i = 0; i < 8; ++i) s_mb_rew[9 + i][lane] = ot[i];
        } else if (wave == 3) {
          for (int i = nsub; i < 4; ++i) noise_draw(i);  // (fewer than four substeps: the draws the loop above did not reach)
        }
        wg_barrier_lds();  // E: the terms are in LDS
      }
    }
    if (wave == 3) {
      wg_barrier_lds();  // C: the uniforms are in LDS
      // ---- curriculum / population gate / step counter (lt_post.h) on the record wave 0 leaves in LDS ----
      wg_barrier_lds();  // B1 (LDS traffic only: nobody behind it reads what a wave stored to global memory before it)
      CurIn in;
      in.valid = env < L.n;
      in.reset = s_cur[lane * 5 + 0] != 0.f; in.ep_len = s_cur[lane * 5 + 1];
      in.sum_lin = s_cur[lane * 5 + 2]; in.sum_ang = s_cur[lane * 5 + 3]; in.cmd_nonzero = s_cur[lane * 5 + 4] != 0.f;
      curriculum_publish(L, arena, P, gid, leg, in, (int)(step & 1));  // the decision is lt_gate_decide_kernel's (or the next launch's, chained)
#ifdef LT_STAMPS
      LT_STAMP(1);
      if (lane == 0) ((float*)(arena + L.quad_off[LT_F_REWARD_TERMS]) + 3 * L.npad * 4 + (long long)blockIdx.x * 64)[3] = (float)(long long)(stamps_[1] - stamps_[0]);
#endif
      return;
    }
    // ---- waves 1, 2: history rows of group g.  Row(t) is built from row(t-1): every term block shifts left by one frame
    //      and takes the newest frame from wave 0.  Lane l owns columns l, l+64, ... of EVERY row, so the per-column routing
    //      (table lookups) is done once per lane.  In-place safety (rows updated in the arena): the DMA of the whole chunk
    //      has landed (vmcnt(0)) before this wave stores anything, and only this wave touches the group's rows. ----
    // (the shifted 5/6 of every row went out between the physics substeps, above)
    wg_barrier_lds();  // C (wave 3's noise uniforms -> wave 0)
    wg_barrier_lds();  // B1: newest frame + reset flags are in LDS (no vmcnt drain: the ~100 row stores above stay in flight)
    // after B1: the newest-frame columns of every row (1/6 of the columns: each lane owns about one of its NCH), and whole rows
    // for envs that were just reset.  The per-lane column routing is loop-invariant, so the row loop is one LDS read + one store
    // per owned newest column; reset rows (rare) take the full pass.
    unsigned fills = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) fills |= (s_fill[r] != 0 ? 1u : 0u) << r;  // first push after a reset fills all 6 slots
    if constexpr (RB) {
      // the newest frame of this group -> bf16 bit patterns, once (in place; only this wave reads the group's frame from here on)
      for (int idx = lane; idx < 16 * 64; idx += 64) s_frame[g][idx >> 6][idx & 63] = __uint_as_float(bf16_rne(s_frame[g][idx >> 6][idx & 63]));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < NPR; ++i) {
        const int pr = i * 64 + lane;
        if (pr < OBS / 2 && (psrc[i][0] < 0 || psrc[i][1] < 0)) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const unsigned lo16 = psrc[i][0] >= 0 ? (unsigned)old16[r * OBS + psrc[i][0]] : __float_as_uint(s_frame[g][r][(-psrc[i][0] - 1) & 63]);
            const unsigned hi16 = psrc[i][1] >= 0 ? (unsigned)old16[r * OBS + psrc[i][1]] : __float_as_uint(s_frame[g][r][(-psrc[i][1] - 1) & 63]);
            st_out(&rows32[r * (OBS / 2) + pr], lo16 | (hi16 << 16));
          }
        }
      }
      while (fills) {
        const int r = __builtin_ctz(fills);
        fills &= fills - 1;
#pragma unroll
        for (int i = 0; i < NPR; ++i) {
          const int pr = i * 64 + lane;
          if (pr < OBS / 2) st_out(&rows32[r * (OBS / 2) + pr], __float_as_uint(s_frame[g][r][pfrm[i][0] & 63]) | (__float_as_uint(s_frame[g][r][pfrm[i][1] & 63]) << 16));
        }
      }
    } else {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int col = i * 64 + lane;
      if (col < OBS && src[i] < 0) {
        const int fe = (-src[i] - 1) & 63;
#pragma unroll
        for (int r = 0; r < 16; ++r) st_out(&rows[r * OBS + col], s_frame[g][r][fe]);
      }
    }
    while (fills) {
      const int r = __builtin_ctz(fills);
      fills &= fills - 1;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int col = i * 64 + lane;
        if (col < OBS && src[i] >= 0) st_out(&rows[r * OBS + col], s_frame[g][r][frm[i] & 63]);
      }
    }
    }
#ifdef LT_STAMPS
    LT_STAMP(1);
    if (lane == 0) ((float*)(arena + L.quad_off[LT_F_REWARD_TERMS]) + 3 * L.npad * 4 + (long long)blockIdx.x * 64)[wave] = (float)(long long)(stamps_[1] - stamps_[0]);
#endif
    return;
  }

#ifdef LT_STAMPS
  unsigned long long pro_c_;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pro_c_) :: "memory");
#endif
  // ---- sign pattern of this lane's leg (mirror form of the model constants) ----
  const float sx = leg < 2 ? 1.f : -1.f, sy = (leg & 1) ? 1.f : -1.f;
  const float sgn[4] = {1.f, sx, sy, sx * sy};
  float qdef[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) qdef[k] = k_m_dq[k] * sgn[k_m_dq_pat[k]];

  // ---- unpack the hot state (per-env vectors are broadcast inside the quad) ----
  Base B; Obj O; Leg G; Misc X;
  {
    float t;
    t = hot.root_pos; B.p = v3(qbcast<0>(t), qbcast<1>(t), qbcast<2>(t));
    t = hot.root_quat; B.q.w = qbcast<0>(t); B.q.x = qbcast<1>(t); B.q.y = qbcast<2>(t); B.q.z = qbcast<3>(t);
    t = hot.root_lin; B.u = v3(qbcast<0>(t), qbcast<1>(t), qbcast<2>(t));
    t = hot.root_ang; B.w = v3(qbcast<0>(t), qbcast<1>(t), qbcast<2>(t));
    t = hot.obj_pos; O.p = v3(qbcast<0>(t), qbcast<1>(t), qbcast<2>(t));
    t = hot.obj_quat; O.q.w = qbcast<0>(t); O.q.x = qbcast<1>(t); O.q.y = qbcast<2>(t); O.q.z = qbcast<3>(t);
    t = hot.obj_lin; O.u = v3(qbcast<0>(t), qbcast<1>(t), qbcast<2>(t));
    t = hot.obj_ang; O.w = v3(qbcast<0>(t), qbcast<1>(t), qbcast<2>(t));
    t = hot.obj_timers; O.cur_air = qbcast<0>(t); O.cur_con = qbcast<1>(t); O.last_air = qbcast<2>(t); O.last_con = qbcast<3>(t);
    t = hot.obj_params; O.rad = qbcast<0>(t); O.len = qbcast<1>(t); O.mass = qbcast<2>(t); O.mu = qbcast<3>(t);
    t = hot.env_params; X.trunk_mass_add = qbcast<0>(t); X.trunk_mu = qbcast<1>(t); X.trunk_rest = qbcast<2>(t); X.obj_rest = qbcast<3>(t);
    t = hot.trunk_fh; X.trunk_fh[0] = qbcast<0>(t); X.trunk_fh[1] = qbcast<1>(t); X.trunk_fh[2] = qbcast<2>(t); X.m_airvar = qbcast<3>(t);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      G.q[k] = hot.q[k]; G.qd[k] = hot.qd[k];
      G.raw[k] = hot.raw[k];
      G.qdd[k] = 0.f; G.tau[k] = 0.f; G.prev[k] = 0.f; G.prev2[k] = 0.f;
    }
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int ty = 0; ty < 4; ++ty) G.fh[s][ty] = hot.fh[s * 4 + ty];
    G.cur_air = hot.cur_air; G.cur_con = hot.cur_con;
    G.last_air = hot.last_air; G.last_con = hot.last_con;
    G.mu = hot.mu;
  }

  LT_STAMP(1);
  // ---- startup events (reset-all only): reference locomotion_base_env_cfg.py:224-244, rand_cylinder_...:21-27 ----
  if (MODE == MODE_RESET_ALL) {
    const uint64_t st = ~0ull;
    const U4 u = rng4(c.seed, ekey, st, RS_STARTUP);
    X.trunk_mass_add = lerp2(c.trunk_mass_add, u.a);
    O.rad = lerp2(c.obj_radius, u.b);
    O.len = lerp2(c.obj_length, u.c);
    if (c.obj_size_explicit) {  // the cfg tree carried per-env cylinders (rand_cylinder_transport_teacher_env_cfg.py:26-38)
      const float* sz = (const float*)(arena + L.off_obj_sizes) + env * 2;
      O.rad = sz[0]; O.len = sz[1];
    }
    U4 uf = rng4(c.seed, ekey, st, RS_STARTUP + 0x10 + leg);
    if (c.foot_material_buckets > 0) {  // bucketed materials [DEP randomize_rigid_body_material]: pool entry b = the draw keyed by b
      const int b = min((int)(uf.a * (float)c.foot_material_buckets), c.foot_material_buckets - 1);
      uf = rng4(c.seed, (uint32_t)b, st, RS_BUCKET_FEET);
    }
    const float ms = lerp2(c.foot_friction, uf.a), md = lerp2(c.foot_friction, uf.b);
    G.mu = md < ms ? md : ms;
    O.mass = 1.0f; O.mu = 1.0f; X.trunk_mu = 1.0f; X.trunk_rest = 0.f; X.obj_rest = 0.f;
    if (!HAS_OBJ) { O.q.w = 1.f; O.q.x = O.q.y = O.q.z = 0.f; }
  }

  // =================================================================================================
  // stages 1-3: action term, decimation x (PD, physics, sensors), counters
  // =================================================================================================
  // tactile refresh of this step (TAC): this lane's plate sample at the sim step where the sensor's period elapses
  V3 tac = v3(0, 0, 0);
  bool tac_new = false;
  if (MODE == MODE_STEP) {
    // 1. JointPositionActionPrevPrev.process_actions (reference mdp/actions.py:30-44); action index = type*4 + leg
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      G.prev[k] = G.raw[k];  // prev_prev <- prev happens with the late loads below
      float x = env < L.n ? a.actions[env * 12 + k * 4 + leg] : 0.f;  // padded tail envs read no action
      x = x < -c.action_clip ? -c.action_clip : (x > c.action_clip ? c.action_clip : x);
      G.raw[k] = x * c.action_scale;
    }
    // 2. decimation loop
    const float h = c.sim_dt / (float)c.phys_substeps;
    // ContactSensor cadence [DEP]: the taxel forces refresh at the first sim step after a reset and then whenever
    // update_period (0.025 s = 5 sim steps) has elapsed: sim-step indices 0, 5, 10, ... counted from the reset
    const int tac_every = TAC ? (int)(c.tactile_update_period / c.sim_dt + 0.5f) : 1;
    int tac_phase = TAC ? (int)((hot.ep0 * (long long)c.decimation) % (long long)(tac_every > 0 ? tac_every : 1)) : 0;
    for (int d = 0; d < c.decimation; ++d) {
      float qd0[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        qd0[k] = G.qd[k];
        G.tau[k] = dc_motor(c, qdef[k] + G.raw[k], G.q[k], G.qd[k]);
      }
      Report rep;
      if (HELPERS) {
        HelperParts hp;
        hp.in = s_mb_in; hp.crba = s_mb_crba; hp.obj = s_mb_obj; hp.lane = lane;
#ifdef LT_STAMPS
        hp.wait = bar_wait_;
#else
        hp.wait = nullptr;
#endif
        for (int s = 0; s < c.phys_substeps; ++s) physics_substep<HAS_OBJ, TAC, HelperParts>(c, h, leg, sgn, B, G, O, X, rep, hp);
      } else {
        for (int s = 0; s < c.phys_substeps; ++s) physics_substep<HAS_OBJ, TAC>(c, h, leg, sgn, B, G, O, X, rep);
      }
      if (TAC) {
        if (tac_phase == 0) { tac = rep.plate; tac_new = true; }
        tac_phase = tac_phase + 1 >= tac_every ? 0 : tac_phase + 1;
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) G.qdd[k] = (G.qd[k] - qd0[k]) / c.sim_dt;  // Articulation.data.joint_acc [DEP]
      // K3 sensors: |F| history (newest first) + timers, at the sensor period = sim dt (locomotion_base_env_cfg.py:358-359)
#pragma unroll
      for (int ty = 0; ty < 4; ++ty) {
        G.fh[2][ty] = G.fh[1][ty];
        G.fh[1][ty] = G.fh[0][ty];
        G.fh[0][ty] = norm(rep.body[ty]);
      }
      timers_update(G.cur_air, G.cur_con, G.last_air, G.last_con, G.fh[0][3] > c.contact_force_threshold, c.sim_dt);
      X.trunk_fh[2] = X.trunk_fh[1]; X.trunk_fh[1] = X.trunk_fh[0];
      X.trunk_fh[0] = norm(qsum(rep.trunk_part));
      if (HAS_OBJ) timers_update(O.cur_air, O.cur_con, O.last_air, O.last_con, norm(qsum(rep.obj_part)) > c.contact_force_threshold, c.sim_dt);
    }
    if (HELPERS && HAS_OBJ) {  // the object as its helper wave left it (written before the last barrier B)
      O.p = v3(s_mb_ofin[0][lane], s_mb_ofin[1][lane], s_mb_ofin[2][lane]);
      O.q.w = s_mb_ofin[3][lane]; O.q.x = s_mb_ofin[4][lane]; O.q.y = s_mb_ofin[5][lane]; O.q.z = s_mb_ofin[6][lane];
      O.u = v3(s_mb_ofin[7][lane], s_mb_ofin[8][lane], s_mb_ofin[9][lane]);
      O.w = v3(s_mb_ofin[10][lane], s_mb_ofin[11][lane], s_mb_ofin[12][lane]);
    }
    foot_kinematics(sgn, B, G);
  }
  if (!HELPERS && MODE == MODE_STEP) s_P[lane & 31] = p_own;  // (LT_CMD_PARAMS_LEN = 32 floats; both half-waves hold the same values)
  LT_STAMP(2);
  // ---- late loads: state the physics never touches (compiler barrier: keep these below the decimation loop) ----
  asm volatile("" ::: "memory");
  int req_bits = 0;
  {
    float t;
    t = *F(LT_F_CMD, 0); X.cmd = v3(qbcast<0>(t), qbcast<1>(t), qbcast<2>(t)); X.cmd_time_left = qbcast<3>(t);
    t = *F(LT_F_CMD_BUF, 0); X.cmd_buf = v3(qbcast<0>(t), qbcast<1>(t), qbcast<2>(t)); X.cmd_standing = qbcast<3>(t);
    t = *F(LT_F_EVENT_TIMERS, 0); X.push_robot_left = qbcast<0>(t); X.push_obj_left = qbcast<1>(t); X.m_exy = qbcast<2>(t); X.m_eyaw = qbcast<3>(t);
    X.ep_len = ((const long long*)(arena + L.off_ep_len))[env];
    if (MODE == MODE_STEP) req_bits = ((const int*)(arena + L.off_term_bits))[env];  // last step's word: carries the caller's termination request (LT_T_USER)
    if (!(HELPERS && MODE == MODE_STEP)) {  // (helper form: wave 1 holds the gait class state and hands it back with the gait term)
      t = *F(LT_F_GAIT_CMD, 0); X.gait_cmd = v3(qbcast<0>(t), qbcast<1>(t), qbcast<2>(t)); X.gait_step = qbcast<3>(t);
      G.g_last_air = *F(LT_F_GAIT_LAST_AIR, 0); G.g_last_con = *F(LT_F_GAIT_LAST_CONTACT, 0);
      G.g_valid = *F(LT_F_GAIT_VALID_LAST_AIR, 0);
      G.g_flags = *(const int*)F(LT_F_GAIT_FLAGS, 0);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (MODE == MODE_STEP) {
        G.prev2[k] = *F(LT_F_ACT_PREV_RAW, k);  // old prev becomes prev_prev (mdp/actions.py:32-33)
      } else {
        G.prev[k] = *F(LT_F_ACT_PREV_RAW, k); G.prev2[k] = *F(LT_F_ACT_PREV_PREV_RAW, k);
        G.qdd[k] = *F(LT_F_JOINT_ACC, k); G.tau[k] = *F(LT_F_APPLIED_TORQUE, k);
      }
    }
    if (MODE != MODE_STEP) {
      G.foot_p = v3(*F(LT_F_FOOT_POS_W, 0), *F(LT_F_FOOT_POS_W, 1), *F(LT_F_FOOT_POS_W, 2));
      G.foot_v = v3(*F(LT_F_FOOT_VEL_W, 0), *F(LT_F_FOOT_VEL_W, 1), *F(LT_F_FOOT_VEL_W, 2));
    } else {
      X.ep_len += 1;  // 3.
    }
  }

  // =================================================================================================
  // stages 4-5: terminations and rewards
  // =================================================================================================
  constexpr bool OFFLOAD = HELPERS && MODE == MODE_STEP;  // gait / object terms evaluated by helper waves 1 / 2 (between barriers D, E)
  int bits = 0;
  bool terminated = false, time_out = false, reset = false;
  CurIn cur_in;
  cur_in.valid = env < L.n; cur_in.reset = false; cur_in.ep_len = cur_in.sum_lin = cur_in.sum_ang = 0.f; cur_in.cmd_nonzero = false;
  float sums[LT_REWARD_SLOTS / 4];  // this lane's episode sums: terms leg, leg+4, ...  (quad array q holds terms 4q..4q+3)
#pragma unroll
  for (int q = 0; q < LT_REWARD_SLOTS / 4; ++q) sums[q] = *F(LT_F_EPISODE_SUMS, q);
  if (MODE != MODE_RESET_ALL) {
    const M3 R0 = quat_to_mat(B.q.w, B.q.x, B.q.y, B.q.z);
    const V3 vb = tmul(R0, B.u), wb = tmul(R0, B.w);
    const V3 gb = -row(R0, 2);  // R0^T (0,0,-1)
    const V3 cmd = X.cmd;
    const float cn = norm(cmd);
    const float lin_err = fsqrt((cmd.x - vb.x) * (cmd.x - vb.x) + (cmd.y - vb.y) * (cmd.y - vb.y));
    const float ang_err = fabsf(cmd.z - wb.z);
    if (OFFLOAD) {  // hand-over to waves 1 (gait) and 2 (object terms); they work while this wave runs its terminations and other terms
      const float v[MB_FIN] = {B.p.x, B.p.y, B.p.z, B.q.w, B.q.x, B.q.y, B.q.z, B.u.x, B.u.y, B.u.z, B.w.x, B.w.y, B.w.z, lin_err, ang_err, cn,
                               G.cur_air, G.cur_con, G.last_air, O.last_con, O.cur_air, cmd.x, cmd.y, cmd.z};
#pragma unroll
      for (int i = 0; i < MB_FIN; ++i) s_mb_fin[i][lane] = v[i];
      wg_barrier_lds();  // D
    }
    // ---- 4. terminations (stock terms [DEP], cfg locomotion_base_env_cfg.py:296-313; mdp/terminations.py:10-23)
    {
      const V3 gq = qapply_inv(B.q, v3(0.f, 0.f, -1.f));
      if (c.term_enabled[LT_T_TIME_OUT] && X.ep_len >= (long long)c.max_episode_length) bits |= 1 << LT_T_TIME_OUT;
      // acos(x) > L  <=>  x < cos(L): the inverse function (~50 instructions of once-through code) only inside a band around the
      // threshold, where the two forms could round differently - the oracle's decision bit for bit, the common path two compares
      if (c.term_enabled[LT_T_BASE_ORIENTATION]) {
        const float x = -gq.z, cl = fcos(c.term_orientation_limit);
        const bool over = fabsf(x - cl) > 1e-4f ? x < cl : acosf(x) > c.term_orientation_limit;
        if (over) bits |= 1 << LT_T_BASE_ORIENTATION;
      }
      if (c.term_enabled[LT_T_BASE_HEIGHT] && B.p.z < c.term_min_height) bits |= 1 << LT_T_BASE_HEIGHT;
      if (c.term_enabled[LT_T_BASE_CONTACT]) {
        const float mx = fmaxf(X.trunk_fh[0], fmaxf(X.trunk_fh[1], X.trunk_fh[2]));
        if (mx > c.term_contact_threshold) bits |= 1 << LT_T_BASE_CONTACT;
      }
      if (c.term_enabled[LT_T_HIP_CONTACT]) {
        const float mx = fmaxf(G.fh[0][0], fmaxf(G.fh[1][0], G.fh[2][0]));
        if (qor(mx > c.term_contact_threshold ? 1 : 0)) bits |= 1 << LT_T_HIP_CONTACT;
      }
      if (HAS_OBJ) {
        if (c.term_enabled[LT_T_OBJECT_BELOW_ROBOT] && O.p.z < B.p.z) bits |= 1 << LT_T_OBJECT_BELOW_ROBOT;
        if (c.term_enabled[LT_T_OBJECT_BAD_ROLL]) {
          const V3 go = qapply_inv(O.q, v3(0.f, 0.f, -1.f));
          const float ay = fabsf(go.y), sl = fsin(c.term_object_roll_limit);  // |asin(y)| > L  <=>  |y| > sin(L), as above
          if (fabsf(ay - sl) > 1e-4f ? ay > sl : fabsf(asinf(go.y)) > c.term_object_roll_limit) bits |= 1 << LT_T_OBJECT_BAD_ROLL;
        }
      }
      // a termination the caller requested on the state the previous step left (include/lt_env.h, LT_T_USER)
      if ((req_bits >> LT_TERM_REQUEST_BIT) & 1) bits |= 1 << LT_T_USER;
      if ((req_bits >> LT_TIMEOUT_REQUEST_BIT) & 1) bits |= 1 << LT_T_USER_TIME_OUT;
      constexpr int k_time_out_bits = (1 << LT_T_TIME_OUT) | (1 << LT_T_USER_TIME_OUT);
      time_out = (bits & k_time_out_bits) != 0;
      terminated = (bits & ~k_time_out_bits) != 0;
    }
    const bool alive_in = MODE == MODE_TERMS ? ((const unsigned char*)(arena + L.off_terminated))[env] != 0 : terminated;
    // ---- 5. rewards (reference locotouch/mdp/rewards.py; weights/dt by the RewardManager [DEP])
    float terms[LT_REWARD_SLOTS];
#pragma unroll
    for (int i = 0; i < LT_REWARD_SLOTS; ++i) terms[i] = 0.f;
    const float* w = c.reward_weight;
    // Every term is evaluated and then kept or zeroed by its weight (a select, not a branch: a disabled term stays exactly 0, as
    // the guarded form and the oracle have it).  25 guarded terms were 25 scheduling regions; as straight-line code their
    // independent dependency chains interleave - a lone wave issues a DEPENDENT VALU operation only every 7 cycles.
    auto on = [&](int i, float v) { return w[i] != 0.f ? v : 0.f; };
    terms[LT_R_ALIVE] = on(LT_R_ALIVE, alive_in ? 0.f : 1.f);
    terms[LT_R_TRACK_LIN_VEL_XY] = on(LT_R_TRACK_LIN_VEL_XY, __expf(-(lin_err / c.track_sigma)));                 // :15-20
    terms[LT_R_TRACK_ANG_VEL_Z] = on(LT_R_TRACK_ANG_VEL_Z, __expf(-(ang_err / c.track_sigma)));                   // :22-27
    const float foot_pv = fsqrt(G.foot_v.x * G.foot_v.x + G.foot_v.y * G.foot_v.y);
    {                                                                                                         // :31-42
      const float mx = fmaxf(G.fh[0][3], fmaxf(G.fh[1][3], G.fh[2][3]));
      terms[LT_R_FOOT_SLIP] = on(LT_R_FOOT_SLIP, qsum(mx > c.foot_slip_threshold ? foot_pv : 0.f));
    }
    terms[LT_R_FOOT_DRAGGING] = on(LT_R_FOOT_DRAGGING, qsum((G.foot_p.z <= c.foot_drag_height && foot_pv > c.foot_drag_vel) ? 1.f : 0.f));  // :44-56
    // gait (rewards.py:60-392): the quad's four feet, the class state (G.g_*, X.gait_*) updated in place.  Helper form: wave 1 has
    // evaluated it beside the terms above (s_mb_rew), from the state wave 0 handed over after the physics.
    if (!OFFLOAD) {
      GaitIO gio{G.g_last_air, G.g_last_con, G.g_valid, G.g_flags, X.gait_cmd, X.gait_step};
      terms[LT_R_GAIT] = gait_term<HAS_OBJ>(c, leg, G.cur_air, G.cur_con, G.last_air, gio, B.p, B.q, O.p, cmd, lin_err, ang_err, P[26] != 0.f, step_dt);
      G.g_last_air = gio.last_air; G.g_last_con = gio.last_con; G.g_valid = gio.valid; G.g_flags = gio.flags;
      X.gait_cmd = gio.cmd; X.gait_step = gio.step;
    }
    { const float d = B.p.z - c.base_height_target; terms[LT_R_TRACK_BASE_HEIGHT] = on(LT_R_TRACK_BASE_HEIGHT, d * d); }  // :398-402
    terms[LT_R_BASE_Z_VELOCITY] = on(LT_R_BASE_Z_VELOCITY, vb.z * vb.z);                                        // :404-408
    terms[LT_R_BASE_ROLL_PITCH_ANGLE] = on(LT_R_BASE_ROLL_PITCH_ANGLE, gb.x * gb.x + gb.y * gb.y);              // :416-420
    terms[LT_R_BASE_ROLL_PITCH_VELOCITY] = on(LT_R_BASE_ROLL_PITCH_VELOCITY, fabsf(wb.x) + fabsf(wb.y));        // :410-414
    {
      float s_lim = 0.f, s_pos = 0.f, s_acc = 0.f, s_vel = 0.f, s_tau = 0.f, s_act = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float mid = (k_joint_lo[k] + k_joint_hi[k]) / 2.f, rng = k_joint_hi[k] - k_joint_lo[k];
        const float lo = mid - 0.5f * rng * LT_SOFT_LIMIT_FACTOR, hi = mid + 0.5f * rng * LT_SOFT_LIMIT_FACTOR;
        float x = G.q[k] - lo; x = x > 0.f ? 0.f : x;
        float y = G.q[k] - hi; y = y < 0.f ? 0.f : y;
        s_lim += -x + y;                                                                                      // :423-427
        const float dq = G.q[k] - qdef[k];
        s_pos += dq * dq; s_acc += G.qdd[k] * G.qdd[k]; s_vel += G.qd[k] * G.qd[k]; s_tau += G.tau[k] * G.tau[k];
        const float da = G.raw[k] - G.prev[k];
        s_act += da * da;
      }
      s_lim = qsum(s_lim); s_pos = qsum(s_pos); s_acc = qsum(s_acc); s_vel = qsum(s_vel); s_tau = qsum(s_tau); s_act = qsum(s_act);
      terms[LT_R_JOINT_POSITION_LIMIT] = on(LT_R_JOINT_POSITION_LIMIT, s_lim);
      {                                                                                                       // :429-440
        const float bv = fsqrt(vb.x * vb.x + vb.y * vb.y), r = fsqrt(s_pos);
        terms[LT_R_JOINT_POSITION] = on(LT_R_JOINT_POSITION, (cn > 0.f || bv > c.joint_pos_vel_threshold) ? r : c.joint_pos_stand_scale * r);
      }
      terms[LT_R_JOINT_ACCELERATION] = on(LT_R_JOINT_ACCELERATION, fsqrt(s_acc));                               // :446-448
      terms[LT_R_JOINT_VELOCITY] = on(LT_R_JOINT_VELOCITY, fsqrt(s_vel));                                       // :442-444
      terms[LT_R_JOINT_TORQUE] = on(LT_R_JOINT_TORQUE, fsqrt(s_tau));                                           // :450-452
      terms[LT_R_ACTION_RATE] = on(LT_R_ACTION_RATE, s_act);                                                    // :454-456
    }
    {                                                                                                         // :459-466
      const float m1 = fmaxf(G.fh[0][1], fmaxf(G.fh[1][1], G.fh[2][1])), m2 = fmaxf(G.fh[0][2], fmaxf(G.fh[1][2], G.fh[2][2]));
      terms[LT_R_THIGH_CALF_COLLISION] = on(LT_R_THIGH_CALF_COLLISION, qsum((m1 > c.thigh_calf_threshold ? 1.f : 0.f) + (m2 > c.thigh_calf_threshold ? 1.f : 0.f)));
    }
    if (HAS_OBJ && !OFFLOAD) {  // helper form: wave 2 (the object's owner) has evaluated them beside the terms above (s_mb_rew)
      float ot[8];
      object_terms(c, B, O, cn, ot);
#pragma unroll
      for (int i = 0; i < 8; ++i) terms[LT_R_OBJECT_XY_POSITION + i] = ot[i];
    }
    if (OFFLOAD) {
      wg_barrier_lds();  // E: wave 1's gait term + class state and wave 2's object terms are in LDS
      terms[LT_R_GAIT] = s_mb_rew[0][lane];
      G.g_last_air = s_mb_rew[1][lane]; G.g_last_con = s_mb_rew[2][lane]; G.g_valid = s_mb_rew[3][lane]; G.g_flags = __float_as_int(s_mb_rew[4][lane]);
      X.gait_cmd = v3(s_mb_rew[5][lane], s_mb_rew[6][lane], s_mb_rew[7][lane]); X.gait_step = s_mb_rew[8][lane];
      if (HAS_OBJ) {
#pragma unroll
        for (int i = 0; i < 8; ++i) terms[LT_R_OBJECT_XY_POSITION + i] = s_mb_rew[9 + i][lane];
      }
    }
    float rew = 0.f;
#pragma unroll
    for (int i = 0; i < LT_NUM_REWARD_TERMS; ++i) {
      const float v = terms[i] * w[i] * step_dt;
      rew += v;
      if ((i & 3) == leg) sums[i >> 2] += v;
    }
    if (leg == 0) {
      ((float*)(arena + L.off_reward))[env] = rew;
      ((int*)(arena + L.off_term_bits))[env] = bits;
      if (MODE == MODE_STEP && a.rec_rewards && env < L.n) {  // ppo.py:162-165 + rollout_storage.py:79-107
        a.rec_rewards[env] = rew + (time_out ? a.rec_gamma * a.rec_values[env] : 0.f);
        a.rec_dones[env] = (terminated || time_out) ? 1 : 0;
      }
    }
    if (c.debug_terms) {
#pragma unroll
      for (int q = 0; q < LT_REWARD_SLOTS / 4; ++q)
        *F(LT_F_REWARD_TERMS, q) = sel4(leg, terms[4 * q], terms[4 * q + 1], terms[4 * q + 2], terms[4 * q + 3]);
    }
  }

  LT_STAMP(3);
  // =================================================================================================
  // stage 6: reset (curriculum record, log snapshot, reset events, manager resets), 7 command, 8 pushes
  // =================================================================================================
  if (MODE == MODE_STEP) {
    reset = terminated || time_out;
    if (leg == 0) {
      ((unsigned char*)(arena + L.off_terminated))[env] = terminated ? 1 : 0;
      ((unsigned char*)(arena + L.off_time_out))[env] = time_out ? 1 : 0;
      ((long long*)(arena + L.off_dones))[env] = reset ? 1 : 0;
    }
    // this step's curriculum record (reset, ep_len, sum_lin, sum_ang): the sums live in lanes 1 and 2 of array 0
    cur_in.reset = reset;
    cur_in.ep_len = (float)X.ep_len;
    cur_in.sum_lin = qbcast<LT_R_TRACK_LIN_VEL_XY & 3>(sums[LT_R_TRACK_LIN_VEL_XY >> 2]);
    cur_in.sum_ang = qbcast<LT_R_TRACK_ANG_VEL_Z & 3>(sums[LT_R_TRACK_ANG_VEL_Z >> 2]);
    if (reset) {
#pragma unroll
      for (int q = 0; q < LT_REWARD_SLOTS / 4; ++q) *F(LT_F_LAST_EPISODE_SUMS, q) = sums[q];
      const float fin = *F(LT_F_LAST_EPISODE_INFO, 0);  // lane 0 holds episodes_finished
      *F(LT_F_LAST_EPISODE_INFO, 0) = sel4(leg, fin + 1.f, (float)X.ep_len, (float)bits, 0.f);
      // CommandTerm.reset [DEP]: the metrics this env contributes to its step's reset batch are the ones the last compute() left
      *F(LT_F_LAST_CMD_METRICS, 0) = sel4(leg, X.m_exy, X.m_eyaw, X.m_airvar, (float)(step & 0xFFFFFFull));
    }
  }
  // the step's event draws: from wave 3's bank (helper form: computed beside the physics) or inline Philox - the same uniforms
  auto draw = [&](uint32_t stream, int slot, int owner) -> U4 {
    if (HELPERS && MODE == MODE_STEP) {
      const int src = (lane & ~3) | owner;
      return U4{s_bank[slot][0][src], s_bank[slot][1][src], s_bank[slot][2][src], s_bank[slot][3][src]};
    }
    return rng4(c.seed, ekey, step, stream);
  };
  if ((MODE == MODE_STEP && reset) || MODE == MODE_RESET_ALL) {
    // E4 reset_root_state_uniform [DEP] (params locomotion_base_env_cfg.py:249-267 / object_transport_teacher...:144-160)
    U4 u = draw(RS_RESET_ROOT, 0, 0);
    B.p = v3(lerp2(c.reset_root_pos[0], u.a), lerp2(c.reset_root_pos[1], u.b), LT_ROOT_INIT_HEIGHT + lerp2(c.reset_root_pos[2], u.c));
    u = draw(RS_RESET_ROOT + 1, 0, 1);
    B.q = q_from_euler(lerp2(c.reset_root_rpy[0], u.a), lerp2(c.reset_root_rpy[1], u.b), lerp2(c.reset_root_rpy[2], u.c));
    u = draw(RS_RESET_ROOT + 2, 0, 2);
    U4 w4 = draw(RS_RESET_ROOT + 3, 0, 3);
    B.u = v3(lerp2(c.reset_root_vel[0], u.a), lerp2(c.reset_root_vel[1], u.b), lerp2(c.reset_root_vel[2], u.c));
    B.w = v3(lerp2(c.reset_root_vel[3], w4.a), lerp2(c.reset_root_vel[4], w4.b), lerp2(c.reset_root_vel[5], w4.c));
    // E5 reset_joints_by_offset [DEP] :269-276
    u = draw(RS_RESET_JOINT + leg, 1, leg);
    w4 = draw(RS_RESET_JOINT + 4 + leg, 2, leg);
    const float up[3] = {u.a, u.b, u.c}, uv[3] = {w4.a, w4.b, w4.c};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float mid = (k_joint_lo[k] + k_joint_hi[k]) / 2.f, rng = k_joint_hi[k] - k_joint_lo[k];
      const float lo = mid - 0.5f * rng * LT_SOFT_LIMIT_FACTOR, hi = mid + 0.5f * rng * LT_SOFT_LIMIT_FACTOR;
      G.q[k] = clampf(qdef[k] + lerp2(c.reset_joint_pos, up[k]), lo, hi);
      G.qd[k] = clampf(lerp2(c.reset_joint_vel, uv[k]), -c.velocity_limit, c.velocity_limit);
      G.qdd[k] = 0.f; G.tau[k] = 0.f;
      G.raw[k] = 0.f; G.prev[k] = 0.f; G.prev2[k] = 0.f;                                                       // actions.py:46-52
    }
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int ty = 0; ty < 4; ++ty) G.fh[s][ty] = 0.f;
    G.cur_air = G.cur_con = G.last_air = G.last_con = 0.f;
    G.g_last_air = G.g_last_con = G.g_valid = 0.f; G.g_flags = 0;                                               // rewards.py:107-114
    X.trunk_fh[0] = X.trunk_fh[1] = X.trunk_fh[2] = 0.f;
    X.gait_cmd = v3(0, 0, 0); X.gait_step = 0.f;
    if (HAS_OBJ) {
      u = draw(RS_RESET_MAT, 3, 0);                                                                            // E3 (events.py:160-196), E2
      X.trunk_mu = lerp2(c.trunk_friction, u.a); X.trunk_mu = X.trunk_mu > 1.f ? 1.f : X.trunk_mu;
      X.trunk_rest = lerp2(c.trunk_restitution, u.b);
      if (c.obj_material_buckets > 0) {  // E2: the object's material comes from a pool of obj_material_buckets entries
        const int b = min((int)(u.c * (float)c.obj_material_buckets), c.obj_material_buckets - 1);
        const U4 ub = (HELPERS && MODE == MODE_STEP) ? draw(0, 6, 0) : rng4(c.seed, (uint32_t)b, ~0ull, RS_BUCKET_OBJ);
        u.c = ub.a; u.d = ub.b;
      }
      O.mu = lerp2(c.obj_friction, u.c); O.mu = O.mu > 1.f ? 1.f : O.mu;
      X.obj_rest = lerp2(c.obj_restitution, u.d);
      u = draw(RS_RESET_OBJ, 3, 1);                                                                            // E6 (events.py:85-109)
      w4 = draw(RS_RESET_OBJ + 1, 3, 2);
      {
        // class variant: offset in world axes (events.py:98-99); function variant: rotated by the robot quat (:43-44)
        const V3 d = v3(lerp2(c.obj_reset_pos[0], u.a), lerp2(c.obj_reset_pos[1], u.b), lerp2(c.obj_reset_pos[2], u.c) + O.len / 2.f);
        O.p = B.p + (c.obj_reset_robot_frame ? qapply(B.q, d) : d);
      }
      O.q = qmul(B.q, q_from_euler(lerp2(c.obj_reset_rpy[0], w4.a), lerp2(c.obj_reset_rpy[1], w4.b), lerp2(c.obj_reset_rpy[2], w4.c)));
      O.u = B.u; O.w = B.w;
      O.mass = 1.0f + lerp2(c.obj_mass_add, w4.d);                                                             // E1
      O.cur_air = O.cur_con = O.last_air = O.last_con = 0.f;
    }
#pragma unroll
    for (int q = 0; q < LT_REWARD_SLOTS / 4; ++q) sums[q] = 0.f;
    command_resample(c, P, draw(RS_CMD_RESET, 4, 0), draw(RS_CMD_RESET + 1, 4, 1), X);
    u = draw(RS_RESET_EVENT, 3, 3);
    X.push_robot_left = lerp2(c.push_robot_interval, u.a);
    X.push_obj_left = lerp2(c.push_obj_interval, u.b);
    X.ep_len = 0;
    if (TAC) { tac = v3(0, 0, 0); tac_new = true; }  // ContactSensor.reset [DEP]: the reset envs' net forces are zeroed
    foot_kinematics(sgn, B, G);
    if (MODE == MODE_RESET_ALL && c.cmd_multi_sampling && 0 < (int)P[15]) X.cmd = v3(0, 0, 0);                 // commands.py:559
  }
  if (MODE == MODE_RESET_ALL) X.m_exy = X.m_eyaw = X.m_airvar = 0.f;  // CommandTerm.reset zeroes them; env.reset() runs no compute()
  if (MODE == MODE_STEP) {
    // 7. CommandTerm.compute [DEP]: _update_metrics first (commands.py:392-396, on the state the resets left and the command as it
    //    stands before this call's resample), then the timers, + MultiSampling._update_command (commands.py:561-576)
    {
      const V3 vb = qapply_inv(B.q, B.u), wb = qapply_inv(B.q, B.w);
      const float ex = X.cmd.x - vb.x, ey = X.cmd.y - vb.y;
      X.m_exy = sqrtf(ex * ex + ey * ey);
      X.m_eyaw = fabsf(X.cmd.z - wb.z);
      const float mean = 0.25f * qsum(G.last_air), d = G.last_air - mean;
      X.m_airvar = qsum(d * d) * (1.f / 3.f);  // torch.var: unbiased
    }
    X.cmd_time_left -= step_dt;
    if (X.cmd_time_left <= 0.f) command_resample(c, P, draw(RS_CMD_TIMER, 4, 2), draw(RS_CMD_TIMER + 1, 4, 3), X);
    if (c.cmd_multi_sampling) {
      const long long zs = (long long)(int)P[15];
      if (X.ep_len < zs) X.cmd = 0.0f * X.cmd_buf;
      if (X.ep_len == zs) X.cmd = X.cmd_buf;
    }
    if (X.cmd_standing != 0.f) X.cmd = v3(0, 0, 0);
    // 8. interval events: push_by_setting_velocity [DEP] (cfg locomotion_base_env_cfg.py:279-292, teacher :189-209)
    X.push_robot_left -= step_dt;
    if (X.push_robot_left < 1e-6f) {
      const U4 u = draw(RS_PUSH_ROBOT, 5, 0), w4 = draw(RS_PUSH_ROBOT + 1, 5, 1);
      X.push_robot_left = lerp2(c.push_robot_interval, u.d);
      B.u += v3(lerp2(c.push_robot_vel[0], u.a), lerp2(c.push_robot_vel[1], u.b), lerp2(c.push_robot_vel[2], u.c));
      B.w += v3(lerp2(c.push_robot_vel[3], w4.a), lerp2(c.push_robot_vel[4], w4.b), lerp2(c.push_robot_vel[5], w4.c));
    }
    if (HAS_OBJ) {
      X.push_obj_left -= step_dt;
      if (X.push_obj_left < 1e-6f) {
        const U4 u = draw(RS_PUSH_OBJ, 5, 2), w4 = draw(RS_PUSH_OBJ + 1, 5, 3);
        X.push_obj_left = lerp2(c.push_obj_interval, u.d);
        O.u += v3(lerp2(c.push_obj_vel[0], u.a), lerp2(c.push_obj_vel[1], u.b), lerp2(c.push_obj_vel[2], u.c));
        O.w += v3(lerp2(c.push_obj_vel[3], w4.a), lerp2(c.push_obj_vel[4], w4.b), lerp2(c.push_obj_vel[5], w4.c));
      }
    }
    // curriculum / population gate (lt_post.h).  Single-wave form: publish this tile's partials here; with helper waves the
    // record goes to wave 3 through LDS.  The global decision is lt_gate_decide_kernel's.
    cur_in.cmd_nonzero = X.cmd.x != 0.f || X.cmd.y != 0.f || X.cmd.z != 0.f;
    if (HELPERS) {
      s_cur[lane * 5 + 0] = cur_in.reset ? 1.f : 0.f; s_cur[lane * 5 + 1] = cur_in.ep_len;
      s_cur[lane * 5 + 2] = cur_in.sum_lin; s_cur[lane * 5 + 3] = cur_in.sum_ang; s_cur[lane * 5 + 4] = cur_in.cmd_nonzero ? 1.f : 0.f;
    } else {
      curriculum_publish(L, arena, P, gid, leg, cur_in, (int)(step & 1));
    }
  }

  LT_STAMP(4);
  // =================================================================================================
  // stage 9: observation frame (policy: noisy, critic: clean) -> LDS, then the 6-deep history rows
  //   term order: reference locomotion_base_env_cfg.py:74-109, object_state object_transport_teacher...:37-43
  // =================================================================================================
  {
    const int el = lane >> 2;
    const M3 R0 = quat_to_mat(B.q.w, B.q.x, B.q.y, B.q.z);
    const V3 wb = tmul(R0, B.w), gb = -row(R0, 2);
    const bool noisy = c.enable_corruption != 0;
    U4 uj, uvv;
    if (HELPERS) {
      wg_barrier_lds();  // C: wave 3's uniforms are in LDS
      uj.a = s_mb_rng[0][lane]; uj.b = s_mb_rng[1][lane]; uj.c = s_mb_rng[2][lane]; uj.d = 0.f;
      uvv.a = s_mb_rng[4][lane]; uvv.b = s_mb_rng[5][lane]; uvv.c = s_mb_rng[6][lane]; uvv.d = 0.f;
    } else {
      uj = rng4(c.seed, ekey, step, RS_NOISE_JPOS + leg);
      uvv = rng4(c.seed, ekey, step, RS_NOISE_JVEL + leg);
    }
    const float nj[3] = {uj.a, uj.b, uj.c}, nv[3] = {uvv.a, uvv.b, uvv.c};
    float* fp = s_frame[0][el];
    float* fc = s_frame[1][el];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float x = G.q[k] - qdef[k];
      fc[9 + k * 4 + leg] = x;
      fp[9 + k * 4 + leg] = x + (noisy ? nj[k] * (2.f * c.obs_noise_joint_pos) - c.obs_noise_joint_pos : 0.f);
      const float v = G.qd[k];
      fc[21 + k * 4 + leg] = v * c.obs_scale_joint_vel;
      fp[21 + k * 4 + leg] = (v + (noisy ? nv[k] * (2.f * c.obs_noise_joint_vel) - c.obs_noise_joint_vel : 0.f)) * c.obs_scale_joint_vel;
      fc[33 + k * 4 + leg] = G.raw[k];
      fp[33 + k * 4 + leg] = G.raw[k];
    }
    if (leg == 0) {
      U4 ua, ug;
      if (HELPERS) {  // lanes 2 and 3 of the quad hold the two base-noise blocks
        ua.a = s_mb_rng[12][lane + 2]; ua.b = s_mb_rng[13][lane + 2]; ua.c = s_mb_rng[14][lane + 2]; ua.d = 0.f;
        ug.a = s_mb_rng[12][lane + 3]; ug.b = s_mb_rng[13][lane + 3]; ug.c = s_mb_rng[14][lane + 3]; ug.d = 0.f;
      } else {
        ua = rng4(c.seed, ekey, step, RS_NOISE_BASE);
        ug = rng4(c.seed, ekey, step, RS_NOISE_BASE + 1);
      }
      const float na[3] = {ua.a, ua.b, ua.c}, ng[3] = {ug.a, ug.b, ug.c};
      const float cm[3] = {X.cmd.x, X.cmd.y, X.cmd.z}, wv[3] = {wb.x, wb.y, wb.z}, gv[3] = {gb.x, gb.y, gb.z};
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        fp[i] = cm[i]; fc[i] = cm[i];
        fc[3 + i] = wv[i] * c.obs_scale_ang_vel;
        fp[3 + i] = (wv[i] + (noisy ? na[i] * (2.f * c.obs_noise_ang_vel) - c.obs_noise_ang_vel : 0.f)) * c.obs_scale_ang_vel;
        fc[6 + i] = gv[i];
        fp[6 + i] = gv[i] + (noisy ? ng[i] * (2.f * c.obs_noise_gravity) - c.obs_noise_gravity : 0.f);
      }
      s_fill[el] = (reset || MODE != MODE_STEP) ? 1 : 0;
    }
    if (HAS_OBJ && leg == 1) {
      float u16[16];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        U4 t;
        if (HELPERS) {  // block b was drawn by lane b of the quad (this is lane 1)
          const int l = lane - 1 + b;
          t.a = s_mb_rng[8][l]; t.b = s_mb_rng[9][l]; t.c = s_mb_rng[10][l]; t.d = s_mb_rng[11][l];
        } else {
          t = rng4(c.seed, ekey, step, RS_NOISE_OBJ + b);
        }
        u16[4 * b] = t.a; u16[4 * b + 1] = t.b; u16[4 * b + 2] = t.c; u16[4 * b + 3] = t.d;
      }
      float o[13];
      object_state_obs(c, B, O, noisy, u16, o);
#pragma unroll
      for (int i = 0; i < 13; ++i) fp[45 + i] = o[i];
      object_state_obs(c, B, O, false, u16, o);
#pragma unroll
      for (int i = 0; i < 13; ++i) fc[45 + i] = o[i];
    }
  }
  wg_barrier_lds();  // B1: frame, reset flags and curriculum record are in LDS (helper form: for waves 1-3).  LDS traffic only - a
                     // full __syncthreads waited here for every global store issued so far
  LT_STAMP(5);
  // ---- store state ---- (a lambda with ONE live call site per kernel form: at the kernel's end, or - one-wave form with bf16 rows -
  // between the request for the critic group's old rows and the wait for them, where the ~80 stores and their address arithmetic run
  // under a DMA latency that nothing covered: 32768 envs 109.9 -> 102.8 us.  With f32 rows the same move cost 1.5 %: twice the row
  // stores are in flight in front of the wait.)
  auto store_state = [&]() __attribute__((always_inline)) {
    ST_STATE(F(LT_F_ROOT_POS, 0), sel4(leg, B.p.x, B.p.y, B.p.z, 0.f));
    ST_STATE(F(LT_F_ROOT_QUAT, 0), sel4(leg, B.q.w, B.q.x, B.q.y, B.q.z));
    ST_STATE(F(LT_F_ROOT_LIN_VEL_W, 0), sel4(leg, B.u.x, B.u.y, B.u.z, 0.f));
    ST_STATE(F(LT_F_ROOT_ANG_VEL_W, 0), sel4(leg, B.w.x, B.w.y, B.w.z, 0.f));
    if (HAS_OBJ || MODE == MODE_RESET_ALL) {
      ST_STATE(F(LT_F_OBJ_POS, 0), sel4(leg, O.p.x, O.p.y, O.p.z, 0.f));
      ST_STATE(F(LT_F_OBJ_QUAT, 0), sel4(leg, O.q.w, O.q.x, O.q.y, O.q.z));
      ST_STATE(F(LT_F_OBJ_LIN_VEL_W, 0), sel4(leg, O.u.x, O.u.y, O.u.z, 0.f));
      ST_STATE(F(LT_F_OBJ_ANG_VEL_W, 0), sel4(leg, O.w.x, O.w.y, O.w.z, 0.f));
      ST_STATE(F(LT_F_OBJ_TIMERS, 0), sel4(leg, O.cur_air, O.cur_con, O.last_air, O.last_con));
      ST_STATE(F(LT_F_OBJ_PARAMS, 0), sel4(leg, O.rad, O.len, O.mass, O.mu));
    }
    ST_STATE(F(LT_F_ENV_PARAMS, 0), sel4(leg, X.trunk_mass_add, X.trunk_mu, X.trunk_rest, X.obj_rest));
    ST_STATE(F(LT_F_TRUNK_FORCE_HIST, 0), sel4(leg, X.trunk_fh[0], X.trunk_fh[1], X.trunk_fh[2], X.m_airvar));
    ST_STATE(F(LT_F_CMD, 0), sel4(leg, X.cmd.x, X.cmd.y, X.cmd.z, X.cmd_time_left));
    ST_STATE(F(LT_F_CMD_BUF, 0), sel4(leg, X.cmd_buf.x, X.cmd_buf.y, X.cmd_buf.z, X.cmd_standing));
    ST_STATE(F(LT_F_EVENT_TIMERS, 0), sel4(leg, X.push_robot_left, X.push_obj_left, X.m_exy, X.m_eyaw));
    ST_STATE(F(LT_F_GAIT_CMD, 0), sel4(leg, X.gait_cmd.x, X.gait_cmd.y, X.gait_cmd.z, X.gait_step));
    if (leg == 0) ((long long*)(arena + L.off_ep_len))[env] = X.ep_len;
    if (TAC && tac_new) { ST_STATE(F(LT_F_PLATE_SAMPLES, 0), tac.x); ST_STATE(F(LT_F_PLATE_SAMPLES, 1), tac.y); ST_STATE(F(LT_F_PLATE_SAMPLES, 2), tac.z); }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      ST_STATE(F(LT_F_JOINT_POS, k), G.q[k]); ST_STATE(F(LT_F_JOINT_VEL, k), G.qd[k]);
      ST_STATE(F(LT_F_JOINT_ACC, k), G.qdd[k]); ST_STATE(F(LT_F_APPLIED_TORQUE, k), G.tau[k]);
      ST_STATE(F(LT_F_ACT_RAW, k), G.raw[k]); ST_STATE(F(LT_F_ACT_PREV_RAW, k), G.prev[k]); ST_STATE(F(LT_F_ACT_PREV_PREV_RAW, k), G.prev2[k]);
    }
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int ty = 0; ty < 4; ++ty) ST_STATE(F(LT_F_FORCE_HIST, s * 4 + ty), G.fh[s][ty]);
    ST_STATE(F(LT_F_FOOT_CUR_AIR, 0), G.cur_air); ST_STATE(F(LT_F_FOOT_CUR_CONTACT, 0), G.cur_con);
    ST_STATE(F(LT_F_FOOT_LAST_AIR, 0), G.last_air); ST_STATE(F(LT_F_FOOT_LAST_CONTACT, 0), G.last_con);
    ST_STATE(F(LT_F_FOOT_FRICTION, 0), G.mu);
    ST_STATE(F(LT_F_FOOT_POS_W, 0), G.foot_p.x); ST_STATE(F(LT_F_FOOT_POS_W, 1), G.foot_p.y); ST_STATE(F(LT_F_FOOT_POS_W, 2), G.foot_p.z);
    ST_STATE(F(LT_F_FOOT_VEL_W, 0), G.foot_v.x); ST_STATE(F(LT_F_FOOT_VEL_W, 1), G.foot_v.y); ST_STATE(F(LT_F_FOOT_VEL_W, 2), G.foot_v.z);
    ST_STATE(F(LT_F_GAIT_LAST_AIR, 0), G.g_last_air); ST_STATE(F(LT_F_GAIT_LAST_CONTACT, 0), G.g_last_con);
    ST_STATE(F(LT_F_GAIT_VALID_LAST_AIR, 0), G.g_valid);
    *(int*)F(LT_F_GAIT_FLAGS, 0) = G.g_flags;
#pragma unroll
    for (int q = 0; q < LT_REWARD_SLOTS / 4; ++q) ST_STATE(F(LT_F_EPISODE_SUMS, q), sums[q]);
  };
  if (!HELPERS)
  {
    // History rows (one-wave form).  The 16 env rows of this wave are one contiguous chunk of 16*OBS floats per group.  Row(t) is
    // built from row(t-1): every term block shifts left by one frame and takes the newest frame from LDS.  The old rows come in by
    // LDS-DMA (22 operations per group: the policy group's at kernel start - they land beside the physics - the critic group's
    // after the policy rows are done, into the same 22-KB area); lane l owns columns l, l + 64, ... of every row (conflict-free LDS
    // reads, table lookups once per lane), a row costs NCH coalesced stores.  In-place safety: the DMA of a group has landed
    // (vmcnt(0)) before any of its rows is stored.
    constexpr int NCH = (OBS + 63) / 64;
    int src[NCH], frm[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int col = i * 64 + lane;
      const int cc = col < OBS ? col : OBS - 1;
      src[i] = s_tab[0][cc];         // >= 0: old column (one slot newer); < 0: newest frame element -src-1
      frm[i] = s_tab[0][352 + cc];   // newest-frame element of this column's term (rows that were just reset)
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      if (g == 1) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's reads of the policy rows are done
        dma_old_rows(a.obs_prev[1], s_old);
        if (RB) store_state();
      }
      // vmcnt(0) as the BUILTIN (0x0F70 = vmcnt 0, expcnt / lgkmcnt untouched): the waitcnt pass then knows the DMA has landed; behind
      // an opaque asm wait it re-waits vmcnt(0) - i.e. for every store of the loop below - before each LDS read of the staged rows
      __builtin_amdgcn_s_waitcnt(0x0F70);
      float* const rows = (g ? a.obs_next[1] : a.obs_next[0]) + (long long)blockIdx.x * 16 * OBS * EB / 4;
      if constexpr (RB) {  // bf16 rows: one 32-bit word per column pair, old halves carried bit for bit, newest-frame halves rounded here
        unsigned* const rows32 = (unsigned*)rows;
        const unsigned short* const old16 = (const unsigned short*)s_old;
        // the newest frame of this group -> bf16 bit patterns, once (in place)
        for (int idx = lane; idx < 16 * 64; idx += 64) s_frame[g][idx >> 6][idx & 63] = __uint_as_float(bf16_rne(s_frame[g][idx >> 6][idx & 63]));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        int psrc[NPC][2], pfrm[NPC][2];
#pragma unroll
        for (int i = 0; i < NPC; ++i) {
          const int pr = i * 64 + lane, c0 = 2 * (pr < OBS / 2 ? pr : OBS / 2 - 1);
          psrc[i][0] = s_tab[0][c0]; psrc[i][1] = s_tab[0][c0 + 1];
          pfrm[i][0] = s_tab[0][352 + c0]; pfrm[i][1] = s_tab[0][352 + c0 + 1];
        }
#pragma unroll 4
        for (int r = 0; r < 16; ++r) {
          const bool fill = s_fill[r] != 0;
#pragma unroll
          for (int i = 0; i < NPC; ++i) {
            const int pr = i * 64 + lane;
            unsigned w[2];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
              const int fi = (fill ? pfrm[i][hf] : (-psrc[i][hf] - 1)) & 63;
              const int si = psrc[i][hf] >= 0 ? psrc[i][hf] : 0;
              const unsigned nv = __float_as_uint(s_frame[g][r][fi]), ov = old16[r * OBS + si];
              w[hf] = (fill || psrc[i][hf] < 0) ? nv : ov;
            }
            if (pr < OBS / 2) st_out(&rows32[r * (OBS / 2) + pr], w[0] | (w[1] << 16));
          }
        }
        continue;
      }
#pragma unroll 4
      for (int r = 0; r < 16; ++r) {
        const bool fill = s_fill[r] != 0;  // first push after a reset fills all 6 slots
        float fv[NCH], ov[NCH];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          const int fi = (fill ? frm[i] : (-src[i] - 1)) & 63;
          const int si = src[i] >= 0 ? src[i] : 0;
          fv[i] = s_frame[g][r][fi]; ov[i] = s_old[r * OBS + si];
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          const int col = i * 64 + lane;
          if (col < OBS) st_out(&rows[r * OBS + col], (fill || src[i] < 0) ? fv[i] : ov[i]);
        }
      }
    }
  }

  LT_STAMP(6);
  if (HELPERS || !RB) store_state();  // (one-wave form with bf16 rows: stored beside the critic rows' DMA, above)
#ifdef LT_STAMPS
  LT_STAMP(7);
  if (lane == 0)
    for (int q = 0; q < 7; ++q) *F(LT_F_LAST_EPISODE_SUMS, q) = (float)(long long)(stamps_[q + 1] - stamps_[q]);
  if (lane == 0) { *F(LT_F_REWARD_TERMS, 0) = (float)(long long)bar_wait_[0]; *F(LT_F_REWARD_TERMS, 1) = (float)(long long)bar_wait_[1]; }
  if (lane == 1) { for (int q = 0; q < 7; ++q) *F(LT_F_REWARD_TERMS, q) = (float)(long long)bar_wait_[4 + q]; }  // phase sums of the substeps (lane 1)
  if (lane == 0) { *F(LT_F_REWARD_TERMS, 4) = (float)(long long)(pro_a_ - stamps_[0]); *F(LT_F_REWARD_TERMS, 5) = (float)(long long)(pro_b_ - pro_a_); *F(LT_F_REWARD_TERMS, 6) = (float)(long long)(pro_c_ - pro_b_); }
  if (lane == 0) ((float*)(arena + L.quad_off[LT_F_REWARD_TERMS]) + 3 * L.npad * 4 + (long long)blockIdx.x * 64)[0] = (float)(long long)(stamps_[7] - stamps_[0]);
#endif
}

// =====================================================================================================
// the curriculum pass on caller-supplied records (parity-test hook for the reference's curriculum sequence): the step
// kernel's tail without a physics step.  Same grid and lane mapping as the step kernel.
// =====================================================================================================
__global__ __launch_bounds__(64) void lt_curriculum_kernel(const KArgs a, const float* __restrict__ records) {
  const lt_layout& L = a.d->layout;
  const int leg = threadIdx.x & 3;
  const long long gid = (long long)blockIdx.x * 64 + threadIdx.x, env = gid >> 2;
  CurIn in;
  in.valid = env < L.n;
  const float* r = records + (in.valid ? env : 0) * 4;
  in.reset = in.valid && r[0] != 0.f;
  in.ep_len = r[1]; in.sum_lin = r[2]; in.sum_ang = r[3];
  const float cm = ((const float*)(a.arena + L.quad_off[LT_F_CMD]))[gid];
  in.cmd_nonzero = qor((leg < 3 && cm != 0.f) ? 1 : 0) != 0;
  curriculum_publish(L, a.arena, (const float*)(a.arena + L.off_cmd_params), gid, leg, in, 0);
}
// the global half of the pass (lt_post.h): one wave behind the step kernel / the curriculum hook
// Four waves reduce the tiles' slots (lane-strided, all loads of a trip in flight together; lt_post.h slot_sums), meet in LDS in a
// fixed order, wave 0 decides.
__global__ __launch_bounds__(256) void lt_gate_decide_kernel(const KArgs a, int bump_counter) {
  __shared__ float s_part[4][LT_PARTIAL_FLOATS][64];
  const lt_layout& L = a.d->layout;
  const long long* const cnt = (const long long*)(a.arena + L.off_counters);
  const int set = bump_counter > 0 ? (int)((cnt[0] + bump_counter - 1) & 1) : 0;  // (the records hook publishes into set 0)
  const float* const slots = (const float*)(a.arena + L.off_partials) + (long long)set * (L.npad / 16) * LT_PARTIAL_FLOATS;
  float r[LT_PARTIAL_FLOATS];
  slot_sums<8>(slots, (unsigned)(L.npad / 16), threadIdx.x, 256u, r);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < LT_PARTIAL_FLOATS; ++i) s_part[wave][i][lane] = r[i];
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int i = 0; i < LT_PARTIAL_FLOATS; ++i) r[i] = ((s_part[0][i][lane] + s_part[1][i][lane]) + s_part[2][i][lane]) + s_part[3][i][lane];
  curriculum_decide(a.d->cfg, L, a.arena, bump_counter, 0, set, r);
}
// multi-rank curriculum gate on cross-rank sums (lt_post.h curriculum_apply_global); one wave
__global__ __launch_bounds__(64) void lt_gate_apply_kernel(const KArgs a, const float* __restrict__ ring_sums, int nsteps, float inv_n_total) {
  curriculum_apply_global(a.d->cfg, a.d->layout, a.arena, ring_sums, nsteps, inv_n_total);
}
// population gate of rewards.py:190 from the commands currently in the arena (lt_env_eval_terms; one block)
__global__ __launch_bounds__(1024) void lt_gate_kernel(const KArgs a) {
  const lt_layout& L = a.d->layout;
  const float4* cmd = (const float4*)(a.arena + L.quad_off[LT_F_CMD]);
  int nz = 0;
  for (long long e = threadIdx.x; e < L.n; e += blockDim.x) {
    const float4 v = cmd[e];
    nz |= (v.x != 0.f || v.y != 0.f || v.z != 0.f) ? 1 : 0;
  }
  nz = __syncthreads_or(nz);
  if (threadIdx.x == 0) ((float*)(a.arena + L.off_cmd_params))[26] = nz ? 1.f : 0.f;
}

// command/curriculum block initialisation (reference mdp/curriculums.py:187-193, commands.py:427-469)
__global__ void lt_init_params_kernel(const KArgs a) {
  const lt_cfg& c = a.d->cfg;
  float* P = (float*)(a.arena + a.d->layout.off_cmd_params);
  const int i = threadIdx.x;
  if (i >= LT_CMD_PARAMS_LEN) return;
  float v = 0.f;
  if (i < 6) v = c.cmd_range_init[i / 2][i & 1];
  else if (i < 12) v = c.cmd_range_init[(i - 6) / 2][i & 1];
  else if (i < 15) v = 1.f;
  else if (i == 15) v = (float)c.cmd_zero_steps;
  else if (i == 16) v = c.cmd_rel_standing;
  else if (i >= 21 && i < 24) v = c.cur_enabled ? (c.cmd_range_max[i - 21] - c.cmd_range_init[i - 21][1]) / (float)c.cur_bins[i - 21] : 0.f;
  else if (i == 24 || i == 25) v = c.cur_enabled ? 1.f : 0.f;
  else if (i == 26) v = 1.f;
  P[i] = v;
  if (i == 0) ((long long*)(a.arena + a.d->layout.off_counters))[0] = 1;
}
__global__ void lt_set_ranges_kernel(const KArgs a, float r0, float r1, float r2, float r3, float r4, float r5, int zero_steps, float rel_standing) {
  float* P = (float*)(a.arena + a.d->layout.off_cmd_params);
  if (threadIdx.x != 0) return;
  const float r[6] = {r0, r1, r2, r3, r4, r5};
  for (int d = 0; d < 3; ++d) set_range(P, d, r[2 * d], r[2 * d + 1]);
  P[15] = (float)zero_steps;
  P[16] = rel_standing;
}

KArgs make_args(const lt_env* env, const float* actions) {
  KArgs k;
  k.d = (const lt_dev_args*)((const char*)env->arena + env->layout.off_dev_args);
  k.arena = (char*)env->arena;
  k.actions = actions;
  k.ep_len = (const long long*)((const char*)env->arena + env->layout.off_ep_len);
  k.npad = env->layout.npad;
  float* const rp = (float*)((char*)env->arena + env->layout.off_obs_policy);
  float* const rc = (float*)((char*)env->arena + env->layout.off_obs_critic);
  k.obs_prev[0] = rp; k.obs_prev[1] = rc;
  k.obs_next[0] = rp; k.obs_next[1] = rc;
  k.rec_values = nullptr; k.rec_gamma = 0.f; k.rec_rewards = nullptr; k.rec_dones = nullptr;
  k.step_offset = 0; k.decide_first = 0;
  return k;
}

struct RecordArgs { const float* values; float gamma; float* rewards; unsigned char* dones; };

template <int MODE>
int launch_step(const lt_env* env, const float* actions, hipStream_t s, const float* const* prev = nullptr, float* const* next = nullptr,
                const RecordArgs* rec = nullptr, bool with_gate = true) {
  KArgs k = make_args(env, actions);
  if (rec) { k.rec_values = rec->values; k.rec_gamma = rec->gamma; k.rec_rewards = rec->rewards; k.rec_dones = rec->dones; }
  for (int g = 0; g < 2; ++g) {
    if (prev && prev[g]) k.obs_prev[g] = prev[g];
    if (next && next[g]) k.obs_next[g] = next[g];
  }
  const dim3 grid((unsigned)(env->layout.npad / 16));
  // the 4-wave (helper) form only where the grid leaves SIMDs idle: up to two 16-env tiles per CU
  // The 4-wave helper form needs a whole CU per 16-env tile (one ~310-register wave on each SIMD): it pays while every tile
  // gets its own CU, i.e. up to one workgroup per CU (4096 envs on an MI355X).  Beyond that a second round of tiles would wait
  // for the first (measured at 8192 envs: 85 us against 69 us for the one-wave form).  LT_STEP_HELPERS_MAX_WG overrides.
  static const unsigned helpers_max = [] {
    if (const char* e = getenv("LT_STEP_HELPERS_MAX_WG")) return (unsigned)atoi(e);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return (unsigned)cus;
  }();
  const bool helpers = MODE == MODE_STEP && grid.x <= helpers_max;
  // population pass of the step (lt_post.h): its own one-wave launch behind the step kernel (mode 0), the caller's (mode 1), or
  // chained into the next step launch (mode 2; helper form only - larger grids fall back to mode 0 behaviour)
  const int gate_mode = (MODE == MODE_STEP && with_gate) ? ((env->defer_gate == 2 && !helpers) ? 0 : env->defer_gate) : -1;  // (mode 3 is stored as 2)
  if (MODE == MODE_STEP) {
    k.step_offset = env->pending_steps;
    k.decide_first = (helpers && env->gate_pending) ? (env->test_chain_skew ? 3 : 1) : 0;
    if (k.decide_first == 3) env->test_chain_skew = 0;  // once
    if (env->gate_pending && !k.decide_first) {  // an outstanding pass this launch cannot absorb: run it now
      hipLaunchKernelGGL(lt_gate_decide_kernel, dim3(1), dim3(256), 0, s, k, env->pending_steps);
      env->pending_steps = 0; env->gate_pending = 0; k.step_offset = 0;
    }
  }
  // bf16 rows (lt_env_set_row_format): only where the caller hands over both row pointers of both groups (rollout-storage slots)
  const bool rb = MODE == MODE_STEP && env->rows_bf16 && prev && next && prev[0] && prev[1] && next[0] && next[1];
  // (no row pointer at all - lt_env_step, lt_env_step_profiled - means the arena's own f32 rows, whatever the format of caller rows)
  const bool any_rows = (prev && (prev[0] || prev[1])) || (next && (next[0] || next[1]));
  if (MODE == MODE_STEP && env->rows_bf16 && any_rows && !rb) return (int)hipErrorInvalidValue;
  if (rb && env->cfg.task == LT_TASK_LOCOMOTION) {
    if (helpers) hipLaunchKernelGGL((lt_step_kernel<LT_TASK_LOCOMOTION, MODE_STEP, true, true>), grid, dim3(256), 0, s, k);
    else hipLaunchKernelGGL((lt_step_kernel<LT_TASK_LOCOMOTION, MODE_STEP, false, true>), grid, dim3(64), 0, s, k);
  } else if (rb && !env->cfg.tactile_enabled) {
    if (helpers) hipLaunchKernelGGL((lt_step_kernel<LT_TASK_TRANSPORT_TEACHER, MODE_STEP, true, true>), grid, dim3(256), 0, s, k);
    else hipLaunchKernelGGL((lt_step_kernel<LT_TASK_TRANSPORT_TEACHER, MODE_STEP, false, true>), grid, dim3(64), 0, s, k);
  } else if (rb) {
    return (int)hipErrorInvalidValue;  // (tactile tasks keep f32 rows)
  } else if (env->cfg.task == LT_TASK_LOCOMOTION) {
    if (helpers) hipLaunchKernelGGL((lt_step_kernel<LT_TASK_LOCOMOTION, MODE, MODE == MODE_STEP>), grid, dim3(256), 0, s, k);
    else hipLaunchKernelGGL((lt_step_kernel<LT_TASK_LOCOMOTION, MODE, false>), grid, dim3(64), 0, s, k);
  } else if (env->cfg.tactile_enabled) {
    if (helpers) hipLaunchKernelGGL((lt_step_kernel<K_TASK_TACTILE, MODE, MODE == MODE_STEP>), grid, dim3(256), 0, s, k);
    else hipLaunchKernelGGL((lt_step_kernel<K_TASK_TACTILE, MODE, false>), grid, dim3(64), 0, s, k);
  } else {
    if (helpers) hipLaunchKernelGGL((lt_step_kernel<LT_TASK_TRANSPORT_TEACHER, MODE, MODE == MODE_STEP>), grid, dim3(256), 0, s, k);
    else hipLaunchKernelGGL((lt_step_kernel<LT_TASK_TRANSPORT_TEACHER, MODE, false>), grid, dim3(64), 0, s, k);
  }
  // the population pass of the step (curriculum decision, population gate, step counter): one wave behind the step kernel,
  // unless the caller places it itself (lt_env_defer_gate: beside the next policy launch in the rollout graph)
  if (MODE == MODE_STEP) {
    const int outstanding = env->pending_steps + 1;  // this step's bump joins the ones before it
    if (gate_mode == 0) {
      hipLaunchKernelGGL(lt_gate_decide_kernel, dim3(1), dim3(256), 0, s, k, outstanding);
      env->pending_steps = 0; env->gate_pending = 0;
    } else {  // modes 1, 2 (and the profiled launch, whose caller launches the pass behind its stop event)
      env->pending_steps = outstanding; env->gate_pending = 1;
    }
  }
  return (int)hipGetLastError();
}

}  // namespace

int lt_check_layout(const lt_layout* L) {
  for (int f = 0; f < LT_NUM_QUAD_FIELDS; ++f)
    if (L->quad_off[f] != (int64_t)make_cumq().v[f] * L->npad * 16 || field_quads_c(f) != lt_field_quads(f)) return 0;
  return 1;
}

int lt_launch_reset_all(const lt_env* env, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(env->arena, 0, (size_t)env->layout.off_obj_sizes, s);  // LT_F_OBJ_SIZES and the (cfg, layout) block survive
  if (e != hipSuccess) return (int)e;
  e = hipMemcpyAsync((char*)env->arena + env->layout.off_dev_args, &env->dev_args, sizeof(lt_dev_args), hipMemcpyHostToDevice, s);
  if (e != hipSuccess) return (int)e;
  const KArgs k = make_args(env, nullptr);
  hipLaunchKernelGGL(lt_init_params_kernel, dim3(1), dim3(64), 0, s, k);
  e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  return launch_step<MODE_RESET_ALL>(env, nullptr, s);
}

int lt_launch_step(const lt_env* env, const float* actions, void* stream) {
  return launch_step<MODE_STEP>(env, actions, (hipStream_t)stream);
}

int lt_launch_step_rows(const lt_env* env, const float* actions, const float* const prev[2], float* const next[2], const float* values,
                        float gamma, float* st_rewards, unsigned char* st_dones, void* stream) {
  const RecordArgs rec = {values, gamma, st_rewards, st_dones};
  return launch_step<MODE_STEP>(env, actions, (hipStream_t)stream, prev, next, st_rewards ? &rec : nullptr);
}

int lt_launch_step_profiled(lt_env* env, const float* actions, const float* const* prev, float* const* next, void* stream, float* ms) {
  hipStream_t s = (hipStream_t)stream;
  hipError_t e;
  if (!env->ev_start) {
    hipEvent_t a, b;
    if ((e = hipEventCreate(&a)) != hipSuccess) return (int)e;
    if ((e = hipEventCreate(&b)) != hipSuccess) return (int)e;
    env->ev_start = a; env->ev_stop = b;
  }
  if ((e = hipEventRecord((hipEvent_t)env->ev_start, s)) != hipSuccess) return (int)e;
  int rc = launch_step<MODE_STEP>(env, actions, s, prev, next, nullptr, false);  // the events bracket lt_step_kernel alone
  if (rc != 0) return rc;
  if ((e = hipEventRecord((hipEvent_t)env->ev_stop, s)) != hipSuccess) return (int)e;
  if (!env->defer_gate && (rc = lt_launch_gate_decide(env, -1, s)) != 0) return rc;
  if ((e = hipEventSynchronize((hipEvent_t)env->ev_stop)) != hipSuccess) return (int)e;
  return (int)hipEventElapsedTime(ms, (hipEvent_t)env->ev_start, (hipEvent_t)env->ev_stop);
}

void lt_release_events(lt_env* env) {
  if (env->ev_start) { (void)hipEventDestroy((hipEvent_t)env->ev_start); env->ev_start = nullptr; }
  if (env->ev_stop) { (void)hipEventDestroy((hipEvent_t)env->ev_stop); env->ev_stop = nullptr; }
}

int lt_launch_eval_terms(const lt_env* env, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  // refresh the population gate from the commands currently in the arena, then evaluate the terms
  const KArgs k = make_args(env, nullptr);
  hipLaunchKernelGGL(lt_gate_kernel, dim3(1), dim3(1024), 0, s, k);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  return launch_step<MODE_TERMS>(env, nullptr, s);
}

int lt_launch_curriculum(const lt_env* env, const float* records, void* stream) {
  const KArgs k = make_args(env, nullptr);
  hipLaunchKernelGGL(lt_curriculum_kernel, dim3((unsigned)(env->layout.npad / 16)), dim3(64), 0, (hipStream_t)stream, k, records);
  hipLaunchKernelGGL(lt_gate_decide_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, k, 0);
  return (int)hipGetLastError();
}

// the outstanding population pass (if any) with every outstanding step-counter bump; `bump_counter` < 0: those, else exactly that many
int lt_launch_gate_decide(const lt_env* env, int bump_counter, void* stream) {
  const KArgs k = make_args(env, nullptr);
  const int bump = bump_counter < 0 ? env->pending_steps : bump_counter;
  hipLaunchKernelGGL(lt_gate_decide_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, k, bump);
  env->pending_steps = 0; env->gate_pending = 0;
  return (int)hipGetLastError();
}

int lt_launch_curriculum_apply_global(const lt_env* env, const float* ring_sums, int nsteps, long long n_total, void* stream) {
  const KArgs k = make_args(env, nullptr);
  hipLaunchKernelGGL(lt_gate_apply_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, k, ring_sums, nsteps, 1.f / (float)n_total);
  return (int)hipGetLastError();
}

int lt_launch_set_command_ranges(const lt_env* env, const float ranges[6], int zero_steps, float rel_standing, void* stream) {
  const KArgs k = make_args(env, nullptr);
  hipLaunchKernelGGL(lt_set_ranges_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, k, ranges[0], ranges[1], ranges[2], ranges[3],
                     ranges[4], ranges[5], zero_steps, rel_standing);
  return (int)hipGetLastError();
}

// the device-side error word (counters[1]: chained hand-offs that timed out): read, cleared, returned in *count.  Host sync.
int lt_launch_check(const lt_env* env, void* stream, long long* count) {
  hipStream_t s = (hipStream_t)stream;
  long long* const word = (long long*)((char*)env->arena + env->layout.off_counters) + 1;
  hipError_t e = hipMemcpyAsync(count, word, sizeof(long long), hipMemcpyDeviceToHost, s);
  if (e != hipSuccess) return (int)e;
  if ((e = hipStreamSynchronize(s)) != hipSuccess) return (int)e;
  if (*count != 0) e = hipMemsetAsync(word, 0, sizeof(long long), s);
  return (int)e;
}

const char* lt_hip_error_string(int err) { return hipGetErrorString((hipError_t)err); }
