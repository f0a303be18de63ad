/*
 * lt_env.h - C ABI of the MI355X-native LocoTouch environment step.
 *
 * The reference has no FFI of its own: its hot path is `ManagerBasedRLEnv.step()` (IsaacLab, closed
 * PhysX underneath) called through the `VecEnv` protocol of loco_rl (reference
 * loco_rl/loco_rl/env/vec_env.py:12-101, used at loco_rl/loco_rl/runners/on_policy_runner.py:32,121-132,158).
 * This header is the boundary a maintainer binds instead (ctypes stub: INTEGRATION.md):
 *
 *   reference interface                                  entry point here
 *   ---------------------------------------------------  ---------------------------------------
 *   gym.make(task, cfg=env_cfg)  (scripts/train.py:98)    lt_cfg_default + lt_env_create + lt_env_bind
 *   env.reset() / RslRlVecEnvWrapper.__init__ [DEP]       lt_env_reset_all
 *   env.step(actions) (on_policy_runner.py:158)           lt_env_step
 *   env.episode_length_buf = ... (on_policy_runner:121)   lt_env_get_view(LT_F_EP_LEN) (int64, writable)
 *   env.get_observations() (on_policy_runner.py:32,127)   lt_env_get_view(LT_F_OBS_POLICY / _CRITIC)
 *   env.scene[...].data.* / sensors[...].data.* (B3)      lt_env_get_view(field)
 *   command_term.set_ranges (mdp/commands.py:471)         device-side in lt_env_step (tail of the step kernel),
 *                                                         host override: lt_env_set_command_ranges
 *   term-level evaluation for parity tests                lt_env_eval_terms
 *
 * Rules: extern "C", plain pointers and sizes, no torch types; every call returns 0 on success or a
 * negative LT_E* code (never throws); caller owns the device arena; every launch is stream-ordered on
 * the `stream` argument (a hipStream_t passed as void*) and performs NO host synchronisation, so a
 * whole rollout step is hipGraph-capturable.
 */
#ifndef LT_ENV_H
#define LT_ENV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LT_ABI_VERSION 17

/* error codes */
#define LT_OK 0
#define LT_EINVAL (-22)
#define LT_ENOMEM (-12)
#define LT_EFAULT (-14) /* arena not bound / too small */
#define LT_EHIP (-5)    /* a HIP runtime call failed (lt_last_error() has the text) */

/* tasks (gym ids: reference locotouch/config/locotouch/__init__.py:14,99) */
#define LT_TASK_LOCOMOTION 0        /* Isaac-Locomotion-LocoTouch-v1 */
#define LT_TASK_TRANSPORT_TEACHER 1 /* Isaac-RandCylinderTransportTeacher-LocoTouch-v1 (and, with cfg.tactile_enabled, the
                                     * student tasks: same scene + the 17 x 13 taxel sensor on the carrying plate) */
#define LT_TACTILE_ROWS 17          /* taxel grid: rows along x (front -> back), columns along y (left -> right);        */
#define LT_TACTILE_COLS 13          /* reference utils/urdf_processor/go1/generate_locotouch_urdf.py:4-8,59-72          */
#define LT_TACTILE_DIM 442          /* BinaryTactileSignals: two identical channels of the 17 x 13 contact map (mdp/observations.py:307-308) */
#define LT_TACTILE_WIDE_DIM 884 /* 4 channels x 17 x 13: the 4-channel formats and the arena's per-group capacity */
#define LT_TACTILE_BINARY 0     /* BinaryTactileSignals: (contact, contact) */
#define LT_TACTILE_NORMALIZED 1 /* NormalizedTactileSignals: (contact, min-max normalised force) */
#define LT_TACTILE_DISCRETE 2   /* DiscreteTactileSignals: (contact, discretised signal) */
#define LT_TACTILE_CONTINUOUS 3 /* CotinuousTactileSignals [sic]: (contact, clamp(force / maximal_force)) */
#define LT_TACTILE_PROCESSED 4  /* ProcessedTactileSignals: (contact, normalised force, min-max, discretised) after dropout / addition / noise */
#define LT_TACTILE_ORIGINAL 5   /* TactileSignals: the same four channels of the raw sensor reading */
#define LT_GATE_RING 32             /* passes kept in LT_F_GATE_RING (>= steps per rollout between two global gate evaluations) */

/* reward terms, in manager order (reference config/base/locomotion_base_env_cfg.py:139-218 then
 * config/locotouch/object_transport_teacher_env_cfg.py:88-105).  Zero-weight terms are not evaluated. */
enum lt_reward_term {
  LT_R_ALIVE = 0, LT_R_TRACK_LIN_VEL_XY, LT_R_TRACK_ANG_VEL_Z, LT_R_FOOT_SLIP, LT_R_FOOT_DRAGGING, LT_R_GAIT,
  LT_R_TRACK_BASE_HEIGHT, LT_R_BASE_Z_VELOCITY, LT_R_BASE_ROLL_PITCH_ANGLE, LT_R_BASE_ROLL_PITCH_VELOCITY,
  LT_R_JOINT_POSITION_LIMIT, LT_R_JOINT_POSITION, LT_R_JOINT_ACCELERATION, LT_R_JOINT_VELOCITY, LT_R_JOINT_TORQUE,
  LT_R_ACTION_RATE, LT_R_THIGH_CALF_COLLISION,
  LT_R_OBJECT_XY_POSITION, LT_R_OBJECT_XY_VELOCITY, LT_R_OBJECT_Z_CONTACT, LT_R_OBJECT_Z_VELOCITY,
  LT_R_OBJECT_ROLL_PITCH_ANGLE, LT_R_OBJECT_ROLL_PITCH_VELOCITY, LT_R_OBJECT_YAW_ALIGNMENT, LT_R_OBJECT_DANGEROUS_STATE,
  LT_NUM_REWARD_TERMS, /* 25 */
  LT_REWARD_SLOTS = 28 /* padded to a multiple of 4 (7 quad arrays) */
};

/* termination terms (bit index in LT_F_TERM_BITS) */
enum lt_term_bit {
  LT_T_TIME_OUT = 0, LT_T_BASE_ORIENTATION, LT_T_BASE_HEIGHT, LT_T_BASE_CONTACT, LT_T_HIP_CONTACT,
  LT_T_OBJECT_BELOW_ROBOT, LT_T_OBJECT_BAD_ROLL, LT_NUM_TERM_BITS
};
/* Termination terms outside the fused set (a cfg's own TerminationTermCfg, time_out = False - the TerminationManager ORs every
 * term into `terminated` [DEP]; reference cfg: locomotion_base_env_cfg.py:296-313): the caller evaluates the term on the state a step
 * left (lt_env_get_view) and sets LT_TERM_REQUEST_BIT in LT_F_TERM_BITS of the envs it fires for; the NEXT lt_env_step* reads the
 * bit before it rewrites the word, reports LT_T_USER among that step's bits and terminates the env (dones, the `alive` term and the
 * reset like any other termination) - one env step later than a fused term would.  LT_T_USER has no `term_enabled` entry.
 * A user term with `time_out = True` (TerminationTermCfg.time_out: the TerminationManager ORs it into `time_outs` instead [DEP])
 * sets LT_TIMEOUT_REQUEST_BIT: the next step reports LT_T_USER_TIME_OUT and ends the env by TIME-OUT (LT_F_TIME_OUT, no `alive`
 * penalty, value bootstrap in lt_env_step_rollout as for the stock time_out term, loco_rl/loco_rl/algorithms/ppo.py:162-165). */
#define LT_T_USER 7
#define LT_T_USER_TIME_OUT 8
#define LT_TERM_REQUEST_BIT 30
#define LT_TIMEOUT_REQUEST_BIT 29

/*
 * Environment configuration.  Every member after `seed` is a 4-byte int32_t or float (arrays of them), so
 * a binding can mirror the struct mechanically (locotouch_amd/_abi.py parses this very header).
 * lt_cfg_default() fills the resolved values of SURVEY.md Appendix A for a task.
 */
typedef struct lt_cfg {
  uint64_t seed;
  int32_t num_envs;
  int32_t task;
  /* timing: reference config/base/locomotion_base_env_cfg.py:343-349 */
  float sim_dt;
  int32_t decimation;
  int32_t phys_substeps; /* integrator substeps inside one sim step (engine detail; 1 = 200 Hz) */
  float episode_length_s;
  /* action term: reference mdp/actions.py:30-44, cfg locomotion_base_env_cfg.py:126-135 */
  float action_clip;
  float action_scale;
  /* DC-motor PD actuator: reference assets/go1.py:41-49 */
  float kp;
  float kd;
  float effort_limit;
  float saturation_effort;
  float velocity_limit;
  /* command term: reference mdp/commands.py:427-576, cfg locomotion_vel_cur_base_env_cfg.py:14-50 */
  float cmd_range_init[3][2]; /* lin_vel_x, lin_vel_y, ang_vel_z: (lo, hi) */
  float cmd_range_max[3];     /* curriculum maximum (symmetric) */
  float cmd_resample_time[2];
  float cmd_new_probs;        /* 0.15 -> bins (0.15, 0.70, 0.15) */
  float cmd_rel_standing;
  float cmd_rel_standing_final;
  int32_t cmd_zero_steps;
  int32_t cmd_zero_steps_final;
  int32_t cmd_multi_sampling; /* 0: UniformVelocityCommandGaitLogging, 1: ...MultiSampling */
  /* curriculum: reference mdp/curriculums.py:184-275 */
  int32_t cur_enabled;
  int32_t cur_bins[3];
  float cur_len_threshold;        /* 0.98 * max_episode_length_s, compared with steps (quirk Q4) */
  float cur_reward_threshold[2];  /* lin, ang: exp(-err/sigma) * weight * max_episode_length_s */
  int32_t cur_repeat_times[2];
  int32_t cur_max_distance_bins;
  /* rewards: weight per lt_reward_term (0 => term skipped), plus parameters */
  float reward_weight[28];
  float track_sigma;
  float foot_slip_threshold;
  float foot_drag_height;
  float foot_drag_vel;
  float base_height_target;
  float joint_pos_stand_scale;
  float joint_pos_vel_threshold;
  float thigh_calf_threshold;
  float danger_x_max;
  float danger_y_max;
  float danger_z_min;
  float danger_vel_xy_max;
  int32_t gait_with_object;       /* AdaptiveSymmetricGaitRewardwithObject */
  float gait_judge_time;
  float gait_air_bound;
  float gait_contact_bound;
  float gait_async_tolerance;
  float gait_stance_scale;
  float gait_soft_min_frequency;
  float gait_tolerance_proportion;
  float gait_rwd_upper;
  float gait_rwd_lower;
  float gait_vel_sigma;
  float gait_task_ratio;
  /* terminations: reference locomotion_base_env_cfg.py:296-313, mdp/terminations.py */
  int32_t term_enabled[8];        /* per lt_term_bit */
  float term_orientation_limit;
  float term_min_height;
  float term_contact_threshold;
  float term_object_roll_limit;
  /* observations: reference locomotion_base_env_cfg.py:69-122, object_transport_teacher_env_cfg.py:13-51 */
  int32_t obs_history;            /* 6 */
  float obs_noise_ang_vel;
  float obs_noise_gravity;
  float obs_noise_joint_pos;
  float obs_noise_joint_vel;
  float obs_scale_ang_vel;
  float obs_scale_joint_vel;
  float obj_noise[13];            /* half-widths: pos [0..2], lin vel [3..5], euler rpy [6..8], ang vel [9..11], [12] unused */
  float obj_scale[13];
  float obj_contact_time_threshold;
  /* reset events */
  float reset_root_pos[3][2];     /* x, y, z ranges */
  float reset_root_rpy[3][2];
  float reset_root_vel[6][2];
  float reset_joint_pos[2];
  float reset_joint_vel[2];
  float trunk_mass_add[2];        /* startup */
  float foot_friction[2];         /* startup, static & dynamic from the same range, consistent (min) */
  float foot_restitution[2];
  float trunk_friction[2];        /* reset */
  float trunk_restitution[2];
  float obj_friction[2];
  float obj_restitution[2];
  float obj_mass_add[2];
  float obj_reset_pos[3][2];
  float obj_reset_rpy[3][2];
  float obj_radius[2];            /* per-env cylinder radius range (sampled once, seeded) */
  float obj_length[2];
  /* interval events */
  float push_robot_interval[2];
  float push_robot_vel[6][2];
  float push_obj_interval[2];
  float push_obj_vel[6][2];
  /* contact-sensor thresholds */
  float contact_force_threshold;  /* 1.0 N [DEP default] */
  /* contact model (engine parameters; PhysX has no counterpart - see DESIGN.md "Physics") */
  float ground_kn;
  float ground_cn;
  float ground_ct;
  float ground_mu;                /* ground material (multiply combine) */
  float plate_kn;
  float plate_cn;
  float plate_ct;
  float contact_ramp;             /* damping ramp-in depth */
  float gravity;
  /* joint limits (URDF lower / upper, include/lt_go1_model.h): a unilateral spring-damper on the joint coordinate, integrated
   * implicitly INSIDE the dynamics solve like the ground contacts: with d the excursion beyond a limit (negative inside) the limit
   * torque is k d - (k h + c) qd_new along the inward direction, active when the joint would end the step beyond the limit at its
   * present velocity - it enters the joint's diagonal as h (k h + c) and its right-hand side, acts
   * between parent and child link (momentum-consistent) and never pulls.  PhysX solves limits as hard constraints
   * (reference assets/go1.py:18-29); engine parameters, no counterpart in the reference's cfg. */
  float joint_limit_kp;           /* N m / rad */
  float joint_limit_kd;           /* N m s / rad */
  int32_t enable_corruption;      /* policy-group noise on/off */
  int32_t debug_terms;            /* 1: write unweighted reward terms to LT_F_REWARD_TERMS every step */
  int32_t max_episode_length;     /* ceil(episode_length_s / step_dt) = 1000 (kept integral: bit-exact time_out) */
  int32_t obj_reset_robot_frame;  /* 0: ResetObjectStateUniform (class; offset in world axes, mdp/events.py:85-109);
                                   * 1: reset_object_state_uniform (function; offset rotated by the robot quat, :13-53) */
  int32_t obj_size_explicit;      /* 1: per-env (radius, length) are taken from LT_F_OBJ_SIZES (written by the host before
                                   * lt_env_reset_all) instead of the seeded draw from obj_radius / obj_length */
  /* tactile sensor + BinaryTactileSignals (student tasks): reference mdp/observations.py:95-308, cfg
   * config/locotouch/object_transport_student_env_cfg.py:13-43 (term params), :195-201 (sensor, 40 Hz) */
  int32_t tactile_enabled;
  float tactile_update_period;    /* 0.025 s: the taxel forces refresh every 5th sim step since the env's reset [DEP ContactSensor] */
  float tactile_threshold;        /* contact_threshold 0.05 N */
  float tactile_threshold_noise;  /* +- half-width of the per-(env, taxel) threshold offset drawn once: 0.05 * 0.2 */
  float tactile_dropout_prob;     /* contact_dropout_prob 0.005 */
  float tactile_addition_prob;    /* contact_addition_prob 0.005 */
  int32_t tactile_format;         /* LT_TACTILE_*: which TactileSignals class feeds the `tactile` group (observations.py:281-429);
                                   * only LT_TACTILE_BINARY is evaluated by the reference (object_transport_student_env_cfg.py:45) */
  float tactile_force_noise;      /* add_force_noise: contact forces x (1 + U(-p, p)), p = 0.1; 0 = off */
  float tactile_maximal_force;    /* 3.0 N: normalised force = clamp(force / maximal_force, 0, 1) */
  int32_t tactile_total_levels;   /* 5: discretisation levels of the min-max normalised signal */
  float tactile_level_noise;      /* add_level_noise: + U(-w, w) levels before re-scaling, w = 1; 0 = off */
  int32_t foot_material_buckets;  /* randomize_rigid_body_material num_buckets [DEP]: the materials are a POOL of this many (static, dynamic,
                                   * restitution) triples drawn once; every (env, shape) picks a pool entry.  4000 for the feet at startup
                                   * (locomotion_base_env_cfg.py:233-244), 8000 for the object at every reset
                                   * (object_transport_teacher_env_cfg.py:132-143).  Entry b of a pool is the Philox draw keyed by b (no
                                   * stored table; the same pool on every rank).  0: a fresh draw per (env, shape) */
  int32_t obj_material_buckets;
  int32_t tactile_aux_groups;     /* bit 0: group `original_tactile` (TactileSignals, 4 channels), bit 1: `processed_tactile`
                                   * (ProcessedTactileSignals, 4 channels) - the two extra groups of the student -Play- env
                                   * (object_transport_student_env_cfg.py:166-171); each term draws its own thresholds and noise */
  /* multi-rank runs (one process per GPU) */
  int32_t cur_gate_external;      /* 1: the step kernel keeps the per-env curriculum trackers and publishes its population sums
                                   * into LT_F_GATE_RING, but leaves the success test / widening (mdp/curriculums.py:224-238,
                                   * 245-259) to lt_env_curriculum_apply_global on sums all-reduced over the ranks */
  int32_t env_index_offset;       /* global index of this shard's env 0: the RNG streams are keyed (seed, offset + env, step,
                                   * stream), so R ranks with offsets r * N draw exactly what one R * N-env population draws */
  int32_t cmd_binary_maximal;     /* `binary_maximal_command` of the command terms (mdp/commands.py:95-104,189-197,452-461,518-521; off in every
                                   * registered config): a resample draws one of the 8 sign combinations (+-1, +-1, +-1) uniformly and
                                   * multiplies it by the current upper range bounds; the standing flag is not redrawn */
  int32_t reserved[1];
} lt_cfg;

/* Fields of the state arena (zero-copy views for the manager-term data contract). */
enum lt_field {
  LT_F_ROOT_POS = 0, LT_F_ROOT_QUAT, LT_F_ROOT_LIN_VEL_W, LT_F_ROOT_ANG_VEL_W,
  LT_F_JOINT_POS, LT_F_JOINT_VEL, LT_F_JOINT_ACC, LT_F_APPLIED_TORQUE,
  LT_F_ACT_RAW, LT_F_ACT_PREV_RAW, LT_F_ACT_PREV_PREV_RAW,
  LT_F_FORCE_HIST,      /* |F| history: [slot 3][type 4: hip,thigh,calf,foot] quad arrays (per leg) */
  LT_F_TRUNK_FORCE_HIST,/* (h0,h1,h2, foot_air_time_variance of the command term's metrics - see LT_F_EVENT_TIMERS) */
  LT_F_FOOT_CUR_AIR, LT_F_FOOT_CUR_CONTACT, LT_F_FOOT_LAST_AIR, LT_F_FOOT_LAST_CONTACT,
  LT_F_FOOT_POS_W,      /* 3 quad arrays: x,y,z per leg (derived, written each step) */
  LT_F_FOOT_VEL_W,      /* 3 quad arrays */
  LT_F_FOOT_FRICTION,   /* per-leg mu (startup randomisation) */
  LT_F_OBJ_POS, LT_F_OBJ_QUAT, LT_F_OBJ_LIN_VEL_W, LT_F_OBJ_ANG_VEL_W,
  LT_F_OBJ_TIMERS,      /* (cur_air, cur_contact, last_air, last_contact) */
  LT_F_OBJ_PARAMS,      /* (radius, length, mass, mu) */
  LT_F_ENV_PARAMS,      /* (trunk_mass_add, trunk_mu, trunk_restitution, obj_restitution) */
  LT_F_GAIT_LAST_AIR, LT_F_GAIT_LAST_CONTACT, LT_F_GAIT_VALID_LAST_AIR, LT_F_GAIT_FLAGS,
  LT_F_GAIT_CMD,        /* (last_cmd xyz, step_from_changing_cmd) */
  LT_F_CMD,             /* (vx, vy, wz, time_left) */
  LT_F_CMD_BUF,         /* (vx, vy, wz, is_standing) */
  LT_F_EVENT_TIMERS,    /* (push_robot_left, push_obj_left, error_vel_xy, error_vel_yaw): lanes 2, 3 and LT_F_TRUNK_FORCE_HIST lane 3 hold the
                         * command term's per-env metrics as CommandTerm.compute left them at the end of the last step
                         * (locotouch/mdp/commands.py:392-396: assigned, not accumulated; zero after lt_env_reset_all) */
  LT_F_EPISODE_SUMS,    /* 7 quad arrays: per reward term, weighted*dt accumulated */
  LT_F_LAST_EPISODE_SUMS,/* 7 quad arrays: snapshot at the last reset (logging) */
  LT_F_LAST_EPISODE_INFO,/* (episodes_finished, last_ep_len, last_term_bits, _) */
  LT_F_CURRICULUM,      /* 3 quad arrays: this step's record (reset, ep_len, sum_lin, sum_ang); trackers
                         * (reset_lin, len_lin, sum_lin, reset_ang); (len_ang, sum_ang, _, _).  The trackers lag the
                         * reference's by one pass: LT_F_CMD_PARAMS[27..30] holds the operations still to be applied */
  LT_F_REWARD_TERMS,    /* 7 quad arrays: unweighted term values of the last step (diagnostics / parity) */
  LT_F_LAST_CMD_METRICS,/* (error_vel_xy, error_vel_yaw, foot_air_time_variance, step id & 0xFFFFFF): the metrics of the env at its last
                         * reset, i.e. the values CommandTerm.reset averaged over that step's reset batch into `extras["log"]`
                         * (IsaacLab CommandTerm.reset [DEP]; the reference's own metrics: commands.py:392-417), and the
                         * common_step_counter of that step, so that the host can form the per-batch means */
  LT_F_PLATE_SAMPLES,   /* 3 quad arrays (lane = plate sample): contact point x, y in the trunk frame and normal force of the
                         * cylinder-on-plate contact at the last tactile refresh (zero after a reset); tactile tasks only.
                         * MUST stay the last quad field: it takes no space without cfg.tactile_enabled */
  LT_NUM_QUAD_FIELDS,
  /* plain (non-quad) arrays */
  LT_F_EP_LEN = 64,     /* int64 [N] */
  LT_F_OBS_POLICY,      /* float [N][obs_dim] */
  LT_F_OBS_CRITIC,      /* float [N][obs_dim] */
  LT_F_REWARD,          /* float [N] */
  LT_F_DONES,           /* int64 [N] = terminated | time_out */
  LT_F_TERMINATED,      /* uint8 [N] */
  LT_F_TIME_OUT,        /* uint8 [N] */
  LT_F_TERM_BITS,       /* int32 [N] which termination terms fired this step */
  LT_F_CMD_PARAMS,      /* float [LT_CMD_PARAMS_LEN], device-resident command/curriculum block */
  LT_F_COUNTERS,        /* int64 [4]: (common_step_counter, error word: chained hand-offs that timed out (lt_env_check), scratch: step id whose chained population pass is final (lt_env_defer_gate mode 2),
                         *             number of curriculum passes so far = next LT_F_GATE_RING slot, mod LT_GATE_RING) */
  LT_F_GATE_RING,       /* float [LT_GATE_RING][8]: the population sums of the last curriculum passes, one row per pass:
                         * (envs with a non-zero command, envs reset this step, lin trackers not all reset, sum ep_len lin,
                         *  sum reward lin, ang trackers not all reset, sum ep_len ang, sum reward ang) */
  LT_F_OBS_TACTILE,     /* float [N][442 or 884 by cfg.tactile_format]: observation group `tactile` (tactile tasks; no history) */
  LT_F_OBS_OBJECT_STATE,/* float [N][78] (row stride obs_dim): observation group `object_state` = the object-state block of the
                         * policy rows (same term, same parameters; the reference draws its noise separately) */
  LT_F_OBJ_SIZES,       /* float [N][2]: explicit per-env cylinder (radius, length), read by lt_env_reset_all when
                         * cfg.obj_size_explicit is set; lt_env_reset_all does not clear it */
  LT_F_OBS_TACTILE_ORIGINAL,  /* float [N][884]: group `original_tactile` (cfg.tactile_aux_groups bit 0) */
  LT_F_OBS_TACTILE_PROCESSED, /* float [N][884]: group `processed_tactile` (cfg.tactile_aux_groups bit 1) */
  LT_F_END
};

/* Device-resident command/curriculum block (LT_F_CMD_PARAMS), float[32]:
 *  [0..5]  current ranges (x_lo,x_hi,y_lo,y_hi,z_lo,z_hi)   [6..11] previous ranges
 *  [12..14] equal flags (1/0)  [15] zero_steps  [16] rel_standing
 *  [17] lin_forward_bins [18] ang_forward_bins [19] success_lin [20] success_ang
 *  [21..23] expansion per dim  [24] lin gate open (1/0)  [25] ang gate open
 *  [26] any env has a non-zero command (population gate of rewards.py:190)
 *  [27..30] tracker operations decided by the last step's curriculum pass and applied to the per-env trackers by the next
 *           one (the pass runs at the tail of the step kernel, where no wave may touch another wave's envs):
 *           [27] merge the last record into the lin trackers  [28] ... into the ang trackers
 *           [29] clear the lin trackers (lin gate passed)      [30] clear the ang trackers */
#define LT_CMD_PARAMS_LEN 32

typedef struct lt_view {
  void* ptr;         /* device pointer of element (env 0, component 0) */
  int32_t dtype;     /* 0 f32, 1 i64, 2 u8, 3 i32 */
  int32_t ndim;      /* 1..3 */
  int64_t shape[3];  /* quad fields: [N][Q][4] with component c = q*4 + lane */
  int64_t stride[3]; /* in elements */
} lt_view;

typedef struct lt_env lt_env; /* opaque */

int lt_abi_version(void);
size_t lt_cfg_sizeof(void);
const char* lt_last_error(void);

/* Fill `cfg` with the resolved task configuration.  `task` is an LT_TASK_* id. */
int lt_cfg_default(int task, lt_cfg* cfg);
/* Resolved configuration of a registered gym id (reference locotouch/config/locotouch/__init__.py:14-117), e.g.
 * "Isaac-CylinderTransportTeacher-LocoTouch-Play-v1": the task kind's defaults plus that registration's overrides and its
 * default num_envs.  LT_EINVAL for an unknown id.  lt_cfg_preset_id(i), i < lt_cfg_num_presets(), enumerates the ids. */
int lt_cfg_preset(const char* gym_id, lt_cfg* cfg);
int lt_cfg_num_presets(void);
const char* lt_cfg_preset_id(int i);
/* Observation width of a task (policy and critic groups are equally wide): 270 / 348. */
int lt_cfg_obs_dim(const lt_cfg* cfg);
/* Width of the `tactile` observation group: 442 (two-channel formats), 884 (LT_TACTILE_PROCESSED / _ORIGINAL), 0 without tactile. */
int lt_cfg_tactile_dim(const lt_cfg* cfg);

int lt_env_create(const lt_cfg* cfg, lt_env** out);
int lt_env_destroy(lt_env* env);
/* Size of the caller-owned device arena. */
int lt_env_state_bytes(const lt_cfg* cfg, size_t* bytes);
int lt_env_bind(lt_env* env, void* device_arena, size_t bytes);
/* Startup events + reset of every env + first observation (RslRlVecEnvWrapper.__init__ calls env.reset()). */
int lt_env_reset_all(lt_env* env, void* stream);
/* One ManagerBasedRLEnv.step(): actions float[N][12] (device).  Outputs live in the arena views
 * (OBS_POLICY, OBS_CRITIC, REWARD, DONES, TERMINATED, TIME_OUT). */
int lt_env_step(lt_env* env, const float* actions, void* stream);
/* Rollout form of lt_env_step for drivers that keep the observation rows in their own rollout storage
 * (loco_rl/loco_rl/storage/rollout_storage.py:79-107 copies obs / critic obs into slot t every step): the step kernel reads the previous
 * rows from (prev_policy, prev_critic) and writes the new ones to (next_policy, next_critic), each float[npad][obs_dim], 16-byte aligned,
 * e.g. storage slots t and t+1.  NULL = the arena's rows.  One launch, like lt_env_step (the curriculum pass and the step-counter
 * increment run at the tail of the step kernel). */
int lt_env_step_rows(lt_env* env, const float* actions, const float* prev_policy, const float* prev_critic, float* next_policy,
                     float* next_critic, void* stream);
/* lt_env_step_rows plus the rollout-storage writes of the transition (ppo.py:162-165, rollout_storage.py:79-107):
 * st_rewards[n] = reward + gamma * values * time_out, st_dones[n] = dones != 0.  `values` = V(obs_t), float[n]. */
int lt_env_step_rollout(lt_env* env, const float* actions, const float* prev_policy, const float* prev_critic, float* next_policy,
                        float* next_critic, const float* values, float gamma, float* st_rewards, uint8_t* st_dones, void* stream);
/* Element format of the observation rows behind the prev / next pointers of lt_env_step_rows and lt_env_step_rollout (BASELINE
 * config 5: bf16 observation rows, history and rollout-storage observations; the arena's own rows and all state stay f32).
 * LT_ROWS_BF16: rows are uint16 [n][obs_dim] (bf16 bit patterns); the newest frame is rounded to nearest-even as it enters a row,
 * older frames are carried bit for bit, so a frame is rounded exactly once.  All four row pointers must then be given (LT_EHIP /
 * invalid value for some but not all); entry points that take no row pointers (lt_env_step, lt_env_step_profiled) keep stepping
 * the arena's f32 rows.  Not available for the tactile tasks.  The reference keeps f32 everywhere
 * (loco_rl/loco_rl/storage/rollout_storage.py:36-44). */
enum lt_row_format { LT_ROWS_F32 = 0, LT_ROWS_BF16 = 1 };
int lt_env_set_row_format(lt_env* env, int format);
/* The population pass of a step - curriculum decision, population gate of rewards.py:190, common_step_counter += 1
 * (curriculums.py:184-275; csrc/lt_post.h) - needs sums over all envs; it is one wave of work.  lt_env_defer_gate(env, mode):
 *   0 (default) every step entry point launches it behind the step kernel: after a step the command block and the counters are final;
 *   1 the caller launches it: lt_env_gate_update(env, stream) once after every step and before the next one, on any stream ordered
 *     after the step's (nothing but the next STEP reads what it writes);
 *   2 chained: the pass of step t runs INSIDE the launch of step t + 1 (one workgroup's idle wave, beside the physics - a step reads
 *     the command block only after its physics), so a chain of steps is one launch per step.  A chain ends with lt_env_gate_update,
 *     before anything else reads the command block or the counters (between the steps of a chain both lag by the outstanding
 *     passes).  Grids beyond one workgroup per CU behave as mode 0.
 * Results are identical in all modes.  Mode 0 can only be selected while no pass is outstanding. */
int lt_env_defer_gate(lt_env* env, int mode);
int lt_env_gate_update(lt_env* env, void* stream);
/* Device-side error word (LT_F_COUNTERS[1]).  In a chained launch every workgroup polls - at most LT_CHAIN_POLL_MAX times, ~50 ms -
 * for the announcement that the command block of its step is final; a launch that never sees it (host bookkeeping and the device's
 * step counter disagree) counts itself there and goes on instead of hanging the GPU.  lt_env_check copies the word to the host
 * (WAITS for `stream`), clears it, and returns LT_EHIP with a message if it was set, LT_OK otherwise.  The reference has no
 * counterpart (its managers run on the host: on_policy_runner.py:154-214 is a Python loop).
 * Test hook: lt_env_defer_gate(env, 3) selects mode 2 and makes the NEXT chained launch announce a wrong step id, once. */
int lt_env_check(lt_env* env, void* stream);
/* Profiling variant of lt_env_step: HIP events bracket the step kernel on `stream`; the call WAITS for them (host
 * sync - never use it inside a captured region) and returns the step kernel's duration in milliseconds. */
int lt_env_step_profiled(lt_env* env, const float* actions, void* stream, float* step_kernel_ms);
/* ... of lt_env_step_rows (rows in the format of lt_env_set_row_format): what bench.py times for the bf16-row configuration. */
int lt_env_step_rows_profiled(lt_env* env, const float* actions, const float* prev_policy, const float* prev_critic, float* next_policy,
                              float* next_critic, void* stream, float* step_kernel_ms);
/* PPO minibatch loss with its gradients in one launch (reference loco_rl/loco_rl/algorithms/ppo.py:251-311; csrc/lt_ppo.hip).
 * mu, actions, old_mu, old_sigma: [M][A]; std: [A] (the policy's state-independent std); value, old_logp, adv, returns,
 * old_values: [M].  Outputs: dmu [M][A] and dvalue [M] = d loss / d mu, d loss / d value for
 * loss = mean(surrogate) + value_loss_coef * mean(value loss) - entropy_coef * entropy  (the entropy term depends on std only and
 * is the caller's); acc [24]: [0] sum surrogate, [1] sum value loss, [2] sum KL, [20] max |dmu|, [21] max |dvalue| (what
 * lt_mlp_backward_pair scales the gradients by), [4 + a] sum over rows of
 * d surrogate-row / d sigma_a scaled by 1 / M.  1 <= A <= 16.  Device pointers, f32.
 * out (optional, 24 floats): the finished scalars - [0] loss = mean surrogate + value_loss_coef * mean value loss - entropy_coef * entropy,
 * [1] mean surrogate, [2] mean value loss, [3] entropy (sum_a 0.5 + 0.5 log 2 pi + log sigma_a), [4] mean KL, [8 + a] d loss / d sigma_a
 * (incl. the entropy term) - written by a one-wave launch behind the main kernel.
 * idx (optional, int64 [M]): the batch tensors (actions ... old_sigma) are then the WHOLE rollout storage and minibatch row i is
 * their row idx[i] (rollout_storage.py:189-215 gathers them; here the gather is the kernel's load). */
int lt_ppo_loss(const float* mu, const float* std, const float* value, const float* actions, const float* old_logp, const float* adv,
                const float* returns, const float* old_values, const float* old_mu, const float* old_sigma, const int64_t* idx, int64_t M, int A,
                float clip, float value_loss_coef, float entropy_coef, int use_clipped_value_loss, float* dmu, float* dvalue, float* acc, float* out,
                void* stream);
/* GAE(lambda) of a rollout in one launch (RolloutStorage.compute_returns, loco_rl/loco_rl/storage/rollout_storage.py:170-186, before the
 * advantage normalisation): rewards, values, returns, advantages [T][N] f32, dones [T][N] uint8, last_values [N] = V(obs after the last
 * step).  returns = A + V, advantages = returns - V, as the reference forms them. */
int lt_gae(const float* rewards, const uint8_t* dones, const float* values, const float* last_values, float gamma, float lam, int T, int64_t N,
           float* returns, float* advantages, void* stream);
/* ELU backward fused with the bias gradient of the layer that fed it (the `Linear -> ELU` blocks of the actor / critic MLPs,
 * loco_rl/loco_rl/modules/actor_critic.py:45-66, in the backward pass of ppo.py:316): dz[M][N] = da * elu'(z) recovered from the
 * activation OUTPUT a (1 where a > 0, a + alpha elsewhere), db[N] = column sums of dz.  dz may alias da.  N a multiple of 4,
 * <= 1024; ws: lt_elu_backward_bias_ws_floats(M, N) floats of scratch. */
int lt_elu_backward_bias(const float* da, const float* a, int64_t M, int N, float alpha, float* dz, float* db, float* ws, void* stream);
/* The same, and per-block maxima of |dz| into amax_blocks[lt_elu_backward_bias_nblk(M)] (what lt_wgrad scales the gradient by). */
int lt_elu_backward_bias2(const float* da, const float* a, int64_t M, int N, float alpha, float* dz, float* db, float* ws, float* amax_blocks,
                          void* stream);
/* Weight gradient of a Linear layer over a minibatch, f32-equivalent on the f16 matrix cores (csrc/lt_wgrad.hip; the backward of
 * loco_rl/loco_rl/algorithms/ppo.py:316 for one layer): slabs[s][n][k] = sum over the s-th slice of the M rows of dz[m][n] x[m][k],
 * s < lt_wgrad_splits(M, N, K); the caller adds the slabs in order (lt_partial_sums: nblk = splits, stride = count = N * K).
 * dz [M][N], x [M][K] row-major f32, N and K multiples of 4; both operands are split into f16 (hi, lo) pairs in registers (three MFMAs per tile); dz is scaled
 * by a power of two taken from max |dz| = max over amax_blocks[nblk_amax] (NULL: no scaling - |dz| must then sit in f16's normal
 * range), |x| <= 65504.  slabs: lt_wgrad_ws_floats(M, N, K) floats.  db_slabs (optional, splits x N floats): the slices' column
 * sums of dz, i.e. the partials of the bias gradient (lt_partial_sums: nblk = splits, stride = count = N).  x_split: x is in the
 * SPLIT FORMAT (lt_mlp_forward_pair, lt_split_rows) - its halves go to the matrix cores as they are.  dz_split: so is dz, multiplied by
 * *dz_scale (a device float: lt_mlp_backward_pair's scales_out; |scaled dz| <= LT_MLP_INPUT_CLAMP) - amax_blocks is then unused.
 * Deterministic. */
int lt_wgrad(const float* dz, int dz_split, const float* dz_scale, const float* x, int x_split, int64_t M, int N, int K,
             const float* amax_blocks, int nblk_amax, float* slabs, float* db_slabs, void* stream);
/* out[i] = clamp(x[i], +-LT_MLP_INPUT_CLAMP) in the SPLIT FORMAT (lt_mlp_forward_pair), i < count (a multiple of 4): the observation rows
 * of a PPO update, converted once per update for the first layer's weight gradient (lt_wgrad, x_split = 1). */
int lt_split_rows(const float* x, void* out, int64_t count, void* stream);
int lt_wgrad_splits(int64_t M, int N, int K);
int64_t lt_wgrad_ws_floats(int64_t M, int N, int K);
int64_t lt_elu_backward_bias_ws_floats(int64_t M, int N);
/* Weight and bias gradient of a narrow head layer (the action-mean and value heads of actor_critic.py:45-66 in the backward pass):
 * dw[n][k] = sum_m dy[m][n] x[m][k], db[n] = sum_m dy[m][n] (db optional).  1 <= n <= 16, k a multiple of 4, <= 1024;
 * ws: lt_head_wgrad_ws_floats(M, n, k) floats of scratch.  x_split: x is in the SPLIT FORMAT (lt_mlp_forward_pair).  Deterministic
 * (no float atomics). */
int lt_head_wgrad(const float* dy, const float* x, int x_split, int64_t M, int n, int k, float* dw, float* db, float* ws, void* stream);
int64_t lt_head_wgrad_ws_floats(int64_t M, int n, int k);
/* clip_grad_norm_(max_norm) + Adam.step() on FLAT f32 buffers of n elements in two launches (ppo.py:318-319; torch's arithmetic
 * order, `step` = the 1-based count of this update).  grads are left scaled by the clip coefficient, as clip_grad_norm_ leaves
 * them; max_norm <= 0 disables the clip; grad_norm (optional, device) receives the pre-clip norm.
 * ws: lt_adam_clip_step_ws_floats(n) floats of scratch. */
int lt_adam_clip_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float max_norm, float lr, float beta1,
                      float beta2, float eps, float weight_decay, int64_t step, float* ws, float* grad_norm, void* stream);
/* The same with the learning rate read from device memory (*lr_dev, as lt_ppo_lr_rule leaves it): a PPO update then needs no host
 * read between its minibatch steps. */
int lt_adam_clip_step_dev(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float max_norm, const float* lr_dev,
                          float beta1, float beta2, float eps, float weight_decay, int64_t step, float* ws, float* grad_norm, void* stream);
/* The adaptive learning-rate rule of PPO.update on the device (loco_rl/loco_rl/algorithms/ppo.py:273-281): with kl = *kl_mean (the
 * minibatch's mean KL - lt_ppo_loss's out[4]; all-reduced over the ranks by the caller first),
 *   kl > 2 desired_kl -> *lr = max(lr_min, *lr / factor);   0 < kl < desired_kl / 2 -> *lr = min(lr_max, *lr * factor)
 * (the reference: 1e-5, 1e-2, 1.5).  kl_mean == NULL or desired_kl <= 0: fixed schedule, *lr untouched.  `stats` (optional, float[3]):
 * += (value loss, surrogate loss, entropy) taken from `scalars` = lt_ppo_loss's `out` - the statistics the runner logs, read once
 * per update instead of once per step (ppo.py:361-363).  `dstd_out` (optional, float[num_actions]): receives d loss / d sigma
 * (scalars[8 ..]) - the std parameter's slot of a flat gradient bucket.  One launch, one wave. */
int lt_ppo_lr_rule(const float* kl_mean, float desired_kl, float lr_min, float lr_max, float factor, float* lr, float* stats,
                   const float* scalars, float* dstd_out, int num_actions, void* stream);
/* njobs (<= 24) ordered partial sums in ONE launch: out0[j][e] = sum over b < nblk[j] of ws[j][b * stride[j] + e] for e < split[j], and
 * out1[j][e - split[j]] for split[j] <= e < count[j] (out1[j] may be NULL).  All arrays are HOST arrays of device pointers / ints.
 * lt_elu_backward_bias with db == NULL and lt_head_wgrad with dw == NULL leave their per-block partials in `ws` for this call
 * (lt_elu_backward_bias_nblk(M) blocks of N floats; lt_head_wgrad_nblk(M) blocks of n * k + 16 floats, weight part first). */
int lt_partial_sums(int njobs, const float* const* ws, const int* nblk, const int64_t* stride, const int* count, const int* split,
                    float* const* out0, float* const* out1, void* stream);
int lt_elu_backward_bias_nblk(int64_t M);
int lt_head_wgrad_nblk(int64_t M);
int64_t lt_adam_clip_step_ws_floats(int64_t n);
/* GRU recurrence of the student's tactile encoder over a padded batch of whole trajectories (reference
 * loco_rl/loco_rl/models/memory_module.py:10-14 -> nn.GRU, single layer; locotouch/distill/student.py:119-123 trains it on
 * (L, B, .) batches).  One launch per time step whose grid covers the chip (csrc/lt_gru.hip); the time loop runs here.
 *   forward : ig [L][B][3H] = X W_ih^T (no bias), h0 [B][H]  ->  out [L][B][H], ws [L][B][4H] (r, z, n, W_hn h + b_hn)
 *   backward: dout [L][B][H], dhn [B][H] | NULL  ->  dig, dhg [L][B][3H] (gate gradients, input / hidden side), dh0 [B][H];
 *             scratch [3][B][H].  The caller forms dW_hh = dhg^T H_prev, dW_ih = dig^T X, dX = dig W_ih and the bias sums.
 * Gate order and formulas are PyTorch's (r, z, n).  H must be a multiple of 64.  Device pointers, f32. */
int lt_gru_forward(const float* ig, const float* h0, const float* w_hh, const float* b_ih, const float* b_hh, int L, int B, int H,
                   float* out, float* ws, void* stream);
int lt_gru_backward(const float* dout, const float* dhn, const float* out, const float* ws, const float* h0, const float* w_hh, int L, int B,
                    int H, float* dig, float* dhg, float* scratch, float* dh0, void* stream);
/* Tactile observation pass (tactile tasks): taxel forces from LT_F_PLATE_SAMPLES -> thresholds -> dropout / addition ->
 * LT_F_OBS_TACTILE.  lt_env_step runs it after the step kernel; drivers that launch lt_env_step_rows / _rollout call it
 * themselves.  No-op (LT_OK) when cfg.tactile_enabled is 0. */
int lt_env_tactile_update(lt_env* env, void* stream);
/* Reward / termination / observation terms on the CURRENT arena contents, without physics, reset or
 * command update (parity-test hook: lets a test write a golden state into the views and read the terms).
 * `terminated_in` (uint8[N], device, may be NULL) feeds the `alive` term. */
int lt_env_eval_terms(lt_env* env, void* stream);
/* The curriculum / population-gate pass that lt_env_step runs at the tail of its step kernel, fed with `records`
 * (device, float[N][4]: reset flag, episode length, episode sum of track_lin_vel_xy, of track_ang_vel_z - what the step
 * kernel derives per env) instead of a physics step; does not advance the step counter.  Parity-test hook for the
 * reference's curriculum sequence (mdp/curriculums.py:184-275). */
int lt_env_curriculum_update(lt_env* env, const float* records, void* stream);
/* Multi-rank curriculum gate (cfg.cur_gate_external): `ring_sums` (device, float[LT_GATE_RING][8]) = LT_F_GATE_RING summed
 * over all ranks; replays the reference's decision sequence on the last `nsteps` passes with `n_total` envs and, at the first
 * pass that succeeds (per lin / ang), widens the command ranges and schedules the tracker clear for the next step.  Every
 * rank calls it with identical sums, so every rank widens on the same step.  Lags the reference by < nsteps steps. */
int lt_env_curriculum_apply_global(lt_env* env, const float* ring_sums, int nsteps, int64_t n_total, void* stream);
int lt_env_get_view(lt_env* env, int field, lt_view* view);
/* Host-side override of the command block (what `set_ranges` does in the reference; resume workflows). */
int lt_env_set_command_ranges(lt_env* env, const float ranges[6], int zero_steps, float rel_standing, void* stream);
/* ---- rollout-side fused kernels (trainer boundary; reference loco_rl/loco_rl/algorithms/ppo.py:129-170 and
 * loco_rl/loco_rl/storage/rollout_storage.py:79-107).  All pointers are device pointers; stream-ordered, no host sync. ----
 * lt_rollout_act: a = mu + sigma * N(0,1) (Philox keyed by (seed, env, *step_counter)), log_prob = sum_k log N(a_k; mu_k, sigma_k),
 * and the storage-slot writes: actions / mu / sigma [n][12], log_prob [n]; when `obs` is non-NULL also the observation rows
 * [n][obs_dim] (policy and critic) into (st_obs, st_critic_obs), and when `value` is non-NULL values [n] into st_values (drivers
 * that use lt_env_step_rows already have the rows in the slot and hand the values to lt_rollout_record instead).
 * `actions_out` [n][12] is the buffer handed to lt_env_step.  `step_counter`: int64 device scalar (the env's LT_F_COUNTERS view, or
 * a caller-owned copy of it that lt_rollout_record advances). */
int lt_rollout_act(int64_t n, int obs_dim, uint64_t seed, const int64_t* step_counter, const float* mu, const float* std12,
                   const float* value, const float* obs, const float* critic_obs, float* st_obs, float* st_critic_obs,
                   float* st_actions, float* st_mu, float* st_sigma, float* st_values, float* st_logp, float* actions_out, void* stream);
/* lt_rollout_record: st_rewards = reward + gamma * values * time_out (time-limit bootstrap, ppo.py:162-165); st_dones = dones != 0;
 * optional: st_values = values (NULL: skip), *bump_counter += 1 (NULL: skip). */
int lt_rollout_record(int64_t n, float gamma, const float* reward, const int64_t* dones, const uint8_t* time_out, const float* values,
                      float* st_rewards, uint8_t* st_dones, float* st_values, int64_t* bump_counter, void* stream);
/* ---- fused fp32 MLP inference (actor / critic evaluation inside the rollout loop; reference
 * loco_rl/loco_rl/modules/actor_critic.py:113-131: self.actor(obs), self.critic(critic_obs) - Linear layers with one activation between).
 * One launch per network (or both) on the f16 MFMA with error-compensated operand splitting - three MFMAs per f32-equivalent MAC, csrc/lt_mlp.hip;
 * weights are re-packed from the torch.nn.Linear layout once per policy update.
 * DOMAIN (applies to lt_mlp_forward, lt_mlp_forward_pair, lt_rollout_policy, lt_rollout_policy_value): every value that enters a
 * layer - the input rows x / obs and every hidden activation - is SATURATED to [-LT_MLP_INPUT_CLAMP, +LT_MLP_INPUT_CLAMP] before
 * the operand split (64 x must stay a finite f16 number).  Inside that range the results are f32-equivalent (error below an f32
 * GEMM's own rounding); outside it the kernel computes MLP(clamp(x)) per layer, which the reference's fp32 ActorCritic does not.
 * Weights and the outputs of the last layer are not bounded.  The trainer checks the bound on the device once per PPO update
 * (locotouch_amd/rl/mlp.py: PackedPair.domain_max) and warns when it is reached. ---- */
#define LT_MLP_INPUT_CLAMP 1000
#define LT_MLP_MAX_LAYERS 6
#define LT_MLP_MAX_WIDTH 1008
enum lt_activation { LT_ACT_NONE = 0, LT_ACT_ELU = 1, LT_ACT_RELU = 2, LT_ACT_TANH = 3 };
typedef struct lt_mlp_desc {
  int32_t num_layers;                   /* Linear layers */
  int32_t dims[LT_MLP_MAX_LAYERS + 1];  /* dims[0] = input width (<= LT_MLP_MAX_WIDTH), dims[l+1] = outputs of layer l (<= 512) */
  int32_t activation;                   /* lt_activation between layers (none after the last) */
  int32_t input_format;                 /* lt_row_format of the input rows (x / obs): LT_ROWS_F32, or LT_ROWS_BF16 = uint16 [m][dims[0]]
                                         * (bf16 bit patterns, widened exactly to f32 as they enter LDS; dims[0] % 4 == 0) */
} lt_mlp_desc;
/* Floats of the packed parameter buffer of a network. */
int lt_mlp_packed_floats(const lt_mlp_desc* desc, size_t* floats);
/* weights[l]: float[dims[l+1]][dims[l]] (row-major, torch.nn.Linear.weight), biases[l]: float[dims[l+1]]; device pointers in a HOST array. */
int lt_mlp_pack(const lt_mlp_desc* desc, const float* const* weights, const float* const* biases, float* packed, void* stream);
/* y[m][dims[L]] = MLP(x[m][dims[0]]). */
int lt_mlp_forward(const lt_mlp_desc* desc, const float* packed, const float* x, int64_t m, float* y, void* stream);
/* Training forward of the actor and the critic in one launch (the forward half of loco_rl/loco_rl/algorithms/ppo.py:251-262 on a
 * minibatch): y0 / y1 [m][out] and the activations behind every hidden layer, acts0[l] / acts1[l] [m][dims[l + 1]] (l < L - 1), which
 * the backward pass needs.  Same kernel and arithmetic as lt_mlp_forward. */
int lt_mlp_forward_pair(const lt_mlp_desc* d0, const float* packed0, const float* x0, const lt_mlp_desc* d1, const float* packed1, const float* x1,
                        int64_t m, float* y0, float* y1, float* const* acts0, float* const* acts1, int acts_split, void* stream);
/* acts_split != 0: the activations are written in the SPLIT FORMAT - one dword per element, low half = f16 hi, high half = f16 lo,
 * value = hi + lo / 64 (the pair the kernel forms for its own matrix-core operands; 2^-22 relative, |value| <= LT_MLP_INPUT_CLAMP) -
 * which lt_mlp_backward_pair, lt_wgrad and lt_head_wgrad read without converting (acts_split / x_split = 1). */
/* The backward DATA path of that pair in one launch (loss.backward() of loco_rl/loco_rl/algorithms/ppo.py:316, the part autograd runs
 * as three `dz @ W` GEMMs and three ELU-backward kernels per network): for every hidden layer l, from the last to the first,
 *     dz_l = (dz_{l+1} W_{l+1}) * ELU'(a_l),     dz_{L-1} = dy,
 * through the forward kernel with the TRANSPOSED weights (packed by lt_mlp_pack_backward into lt_mlp_backward_packed_floats floats;
 * `fwd` is the FORWARD descriptor: ELU, >= 2 layers, hidden widths multiples of 8, <= 64 outputs).  The gradients stay in LDS between
 * layers; each dz_l is written once, for lt_wgrad, with the per-workgroup max |dz_l| it scales by.
 * dy* [m][out]; acts*[l]: the forward activations (lt_mlp_forward_pair); OUT dz*[l] [m][dims[l + 1]]; OUT amax*[l]:
 * lt_mlp_backward_blocks(fwd0, fwd1, m) floats.  Every workgroup scales its own rows by a power of two (max |dy| -> [1, 2)) and
 * unscales what it writes; sat_count (optional, device float): += 1 per workgroup and layer whose scaled gradients reached
 * LT_MLP_INPUT_CLAMP (growth by > 500x through the chain) - those rows are saturated, not exact.
 * in_amax0 / in_amax1 (optional device scalars: max |dy0|, max |dy1|, e.g. acc[20], acc[21] of lt_ppo_loss): ONE scale per network for
 * the whole launch instead of one per workgroup; required for dz_split = 1: the dz* are then written in the SPLIT FORMAT, still
 * multiplied by that scale (scales_out[0 / 1], device floats) - the form lt_wgrad(dz_split = 1) reads without converting. */
int lt_mlp_backward_packed_floats(const lt_mlp_desc* fwd, size_t* floats);
int lt_mlp_pack_backward(const lt_mlp_desc* fwd, const float* const* weights, float* packed, void* stream);
int64_t lt_mlp_backward_blocks(const lt_mlp_desc* fwd0, const lt_mlp_desc* fwd1, int64_t m);
/* lt_mlp_pack of two networks and, where bpacked* is given, their lt_mlp_pack_backward - ONE launch (a training step re-packs all of
 * them: the optimizer has moved the weights). */
int lt_mlp_pack_training(const lt_mlp_desc* d0, const float* const* weights0, const float* const* biases0, float* packed0, float* bpacked0,
                         const lt_mlp_desc* d1, const float* const* weights1, const float* const* biases1, float* packed1, float* bpacked1,
                         void* stream);
int lt_mlp_backward_pair(const lt_mlp_desc* fwd0, const float* bpacked0, const float* dy0, const float* const* acts0, float* const* dz0,
                         float* const* amax0, const lt_mlp_desc* fwd1, const float* bpacked1, const float* dy1, const float* const* acts1,
                         float* const* dz1, float* const* amax1, int64_t m, int acts_split, const float* in_amax0, const float* in_amax1,
                         int dz_split, float* scales_out, float* sat_count, void* stream);
/* Actor forward + the sampling / log-prob / storage-slot writes of lt_rollout_act in one launch (the policy head must have 12 outputs).
 * Philox key step = *step_counter + step_offset. */
int lt_rollout_policy(const lt_mlp_desc* actor, const float* packed, const float* obs, int64_t n, uint64_t seed, const int64_t* step_counter,
                      int64_t step_offset, const float* std12, float* st_actions, float* st_mu, float* st_sigma, float* st_logp,
                      float* actions_out, void* stream);
/* lt_rollout_policy plus the critic forward (values[n] = critic(critic_obs)) in the SAME launch: the two networks' workgroups share
 * the chip, which hides each other's pipeline bubbles and leaves the env step kernel alone on the GPU afterwards. */
int lt_rollout_policy_value(const lt_mlp_desc* actor, const float* actor_packed, const float* obs, const lt_mlp_desc* critic,
                            const float* critic_packed, const float* critic_obs, float* values, int64_t n, uint64_t seed,
                            const int64_t* step_counter, int64_t step_offset, const float* std12, float* st_actions, float* st_mu,
                            float* st_sigma, float* st_logp, float* actions_out, void* stream);
/* Device kernel names and static resource usage, for profiling scripts. */
const char* lt_env_kernel_name(int which);

#ifdef __cplusplus
}
#endif
#endif /* LT_ENV_H */
