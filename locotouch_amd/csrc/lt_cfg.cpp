// Resolved task configurations behind lt_cfg_default() - host code, no device calls.
//
// Every number below is the value the reference's @configclass chain resolves to for the task
// (SURVEY.md Appendix A), cited per block as reference file:line.  The contact-model block at the end is
// the only part with no reference counterpart (PhysX is closed source; DESIGN.md "Physics").
#include <cmath>
#include <cstring>

#include "../../include/lt_env.h"

namespace {
constexpr float kPi = 3.14159265358979323846f;

void set2(float r[2], float lo, float hi) { r[0] = lo; r[1] = hi; }

void base_locomotion(lt_cfg* c) {
  // timing: config/base/locomotion_base_env_cfg.py:343-349
  c->sim_dt = 0.005f;
  c->decimation = 4;
  c->phys_substeps = 1;
  c->episode_length_s = 20.0f;
  c->max_episode_length = 1000;  // ceil(20.0 / 0.02)
  // action: locomotion_base_env_cfg.py:126-135
  c->action_clip = 100.0f;
  c->action_scale = 0.25f;
  // actuator: assets/go1.py:41-49
  c->kp = 25.0f;
  c->kd = 0.5f;
  c->effort_limit = 23.5f;
  c->saturation_effort = 23.5f;
  c->velocity_limit = 30.0f;
  // command: locomotion_base_env_cfg.py:54-66
  set2(c->cmd_range_init[0], -1.0f, 1.0f);
  set2(c->cmd_range_init[1], -0.6f, 0.6f);
  set2(c->cmd_range_init[2], -kPi / 2, kPi / 2);
  c->cmd_range_max[0] = 1.0f;
  c->cmd_range_max[1] = 0.6f;
  c->cmd_range_max[2] = kPi / 2;
  set2(c->cmd_resample_time, 8.0f, 8.0f);
  c->cmd_new_probs = 0.15f;
  c->cmd_rel_standing = 0.1f;
  c->cmd_rel_standing_final = 0.1f;
  c->cmd_zero_steps = 0;
  c->cmd_zero_steps_final = 0;
  c->cmd_multi_sampling = 0;
  c->cur_enabled = 0;
  // rewards: locomotion_base_env_cfg.py:139-218
  float* w = c->reward_weight;
  w[LT_R_ALIVE] = 10.0f;
  w[LT_R_TRACK_LIN_VEL_XY] = 1.0f;
  w[LT_R_TRACK_ANG_VEL_Z] = 0.5f;
  w[LT_R_FOOT_SLIP] = -1.0f;
  w[LT_R_FOOT_DRAGGING] = -0.1f;
  w[LT_R_GAIT] = 0.5f;
  w[LT_R_TRACK_BASE_HEIGHT] = -0.5f;
  w[LT_R_BASE_Z_VELOCITY] = -1.0f;
  w[LT_R_BASE_ROLL_PITCH_ANGLE] = -1.0f;
  w[LT_R_BASE_ROLL_PITCH_VELOCITY] = -0.2f;
  w[LT_R_JOINT_POSITION_LIMIT] = -10.0f;
  w[LT_R_JOINT_POSITION] = -0.5f;
  w[LT_R_JOINT_ACCELERATION] = -5.0e-6f;
  w[LT_R_JOINT_VELOCITY] = -5.0e-3f;
  w[LT_R_JOINT_TORQUE] = -2.5e-4f;
  w[LT_R_ACTION_RATE] = -0.75f;
  w[LT_R_THIGH_CALF_COLLISION] = -5.0f;
  c->track_sigma = 0.25f;
  c->foot_slip_threshold = 0.5f;
  c->foot_drag_height = 0.03f;
  c->foot_drag_vel = 0.1f;
  c->base_height_target = 0.42f;  // quirk Q6 (a Go2W value), kept for parity
  c->joint_pos_stand_scale = 5.0f;
  c->joint_pos_vel_threshold = 0.3f;
  c->thigh_calf_threshold = 0.1f;
  c->danger_x_max = 0.125f;
  c->danger_y_max = 0.097f;
  c->danger_z_min = 0.095f;
  c->danger_vel_xy_max = 2.5f;
  // gait params: locomotion_base_env_cfg.py:166-188
  c->gait_with_object = 0;
  c->gait_judge_time = 1.0e-6f;
  c->gait_air_bound = 0.5f;
  c->gait_contact_bound = 0.5f;
  c->gait_async_tolerance = 0.05f;
  c->gait_stance_scale = 1.0f;
  c->gait_soft_min_frequency = 2.0f;
  c->gait_tolerance_proportion = 0.2f;
  c->gait_rwd_upper = 1.0f;
  c->gait_rwd_lower = -5.0f;
  c->gait_vel_sigma = 0.25f;
  c->gait_task_ratio = 1.0f;
  // terminations: locomotion_base_env_cfg.py:296-313
  c->term_enabled[LT_T_TIME_OUT] = 1;
  c->term_enabled[LT_T_BASE_ORIENTATION] = 1;
  c->term_enabled[LT_T_BASE_HEIGHT] = 1;
  c->term_enabled[LT_T_BASE_CONTACT] = 1;
  c->term_enabled[LT_T_HIP_CONTACT] = 1;
  c->term_orientation_limit = kPi / 2;
  c->term_min_height = 0.15f;
  c->term_contact_threshold = 1.0f;
  c->term_object_roll_limit = kPi / 3;
  // observations: locomotion_base_env_cfg.py:69-122
  c->obs_history = 6;
  c->obs_noise_ang_vel = 0.2f;
  c->obs_noise_gravity = 0.05f;
  c->obs_noise_joint_pos = 0.01f;
  c->obs_noise_joint_vel = 1.5f;
  c->obs_scale_ang_vel = 0.25f;
  c->obs_scale_joint_vel = 0.05f;
  c->enable_corruption = 1;
  // reset events: locomotion_base_env_cfg.py:245-276 (degenerate constant ranges are the reference's, quirk Q5)
  set2(c->reset_root_pos[0], -0.3f, 0.3f);
  set2(c->reset_root_pos[1], -0.3f, 0.3f);
  set2(c->reset_root_pos[2], -0.05f, 0.05f);
  set2(c->reset_root_rpy[0], -kPi / 6, -kPi / 6);
  set2(c->reset_root_rpy[1], -kPi / 6, -kPi / 6);
  set2(c->reset_root_rpy[2], -kPi / 2, -kPi / 2);
  set2(c->reset_root_vel[0], -0.01f, 0.01f);
  set2(c->reset_root_vel[1], -0.01f, 0.01f);
  set2(c->reset_root_vel[2], 0.0f, 0.0f);
  set2(c->reset_root_vel[3], -kPi / 4, -kPi / 4);
  set2(c->reset_root_vel[4], -kPi / 4, -kPi / 4);
  set2(c->reset_root_vel[5], -kPi / 2, -kPi / 2);
  set2(c->reset_joint_pos, -0.03f, 0.03f);
  set2(c->reset_joint_vel, -0.1f, 0.1f);
  // startup events: locomotion_base_env_cfg.py:224-244
  set2(c->trunk_mass_add, -1.0f, 2.0f);
  set2(c->foot_friction, 0.4f, 2.0f);
  set2(c->foot_restitution, 0.0f, 0.5f);
  c->foot_material_buckets = 4000;                                   // :242
  c->obj_material_buckets = 0;
  // interval events: locomotion_base_env_cfg.py:279-292
  set2(c->push_robot_interval, 4.0f, 8.0f);
  set2(c->push_robot_vel[0], -1.0f, 1.0f);
  set2(c->push_robot_vel[1], -0.6f, 0.6f);
  set2(c->push_robot_vel[2], -0.2f, 0.2f);
  set2(c->push_robot_vel[3], -kPi / 4, -kPi / 4);
  set2(c->push_robot_vel[4], -kPi / 4, kPi / 4);
  set2(c->push_robot_vel[5], -kPi / 2, kPi / 2);
  c->contact_force_threshold = 1.0f;  // ContactSensorCfg.force_threshold default [DEP]
  // object defaults (unused by the locomotion task, kept well-formed)
  set2(c->obj_radius, 0.05f, 0.05f);
  set2(c->obj_length, 0.3f, 0.3f);
  set2(c->push_obj_interval, 1.0e9f, 1.0e9f);
  for (int i = 0; i < 13; ++i) c->obj_scale[i] = 1.0f;
  c->obj_contact_time_threshold = 1.0e-8f;
  // contact model (engine; no reference counterpart)
  c->ground_kn = 2.0e4f;
  c->ground_cn = 2.0e2f;
  c->ground_ct = 1.5e3f;
  c->ground_mu = 1.0f;  // terrain material 1.0/1.0, multiply combine: locomotion_base_env_cfg.py:19-30
  c->plate_kn = 2.0e4f;
  c->plate_cn = 2.0e2f;
  c->plate_ct = 1.0e3f;
  c->contact_ramp = 1.0e-3f;
  c->gravity = 9.81f;
  c->joint_limit_kp = 4.0e4f;  // 23.5 N m of motor torque rests 0.6 mrad beyond a limit; a leg (0.08 kg m^2 about its hip) arriving at 20 rad/s overshoots ~10 mrad
  c->joint_limit_kd = 120.f;   // ~critical for that leg; h (k h + c) = 1.6 kg m^2 against joint inertias of 0.003 .. 0.08: no rebound
}

void transport_teacher(lt_cfg* c) {
  // Isaac-RandCylinderTransportTeacher-LocoTouch-v1 = base -> vel-cur -> object transport -> cylinder -> rand cylinder
  // commands / curriculum: config/base/locomotion_vel_cur_base_env_cfg.py:14-50,
  //   config/locotouch/object_transport_teacher_env_cfg.py:75-81, rand_cylinder_transport_teacher_env_cfg.py:59-61
  set2(c->cmd_range_init[0], -0.2f, 0.2f);
  set2(c->cmd_range_init[1], -0.1f, 0.1f);
  set2(c->cmd_range_init[2], -kPi / 10, kPi / 10);
  c->cmd_range_max[0] = 0.5f;
  c->cmd_range_max[1] = 0.25f;
  c->cmd_range_max[2] = kPi / 4;
  c->cmd_multi_sampling = 1;
  c->cmd_rel_standing = 0.1f;
  c->cmd_rel_standing_final = 0.05f;
  c->cmd_zero_steps = 0;
  c->cmd_zero_steps_final = 50;
  c->cur_enabled = 1;
  c->cur_bins[0] = c->cur_bins[1] = c->cur_bins[2] = 20;
  c->cur_len_threshold = 0.98f * 20.0f;                                // mdp/curriculums.py:194 (quirk Q4)
  c->cur_reward_threshold[0] = (float)(std::exp(-0.08 / 0.25) * 1.0 * 20.0);  // :199
  c->cur_reward_threshold[1] = (float)(std::exp(-0.1 / 0.25) * 0.5 * 20.0);   // :200
  c->cur_repeat_times[0] = c->cur_repeat_times[1] = 1;
  c->cur_max_distance_bins = 4;
  // rewards: object_transport_teacher_env_cfg.py:88-105, cylinder_transport_teacher_env_cfg.py:41-45
  float* w = c->reward_weight;
  c->gait_with_object = 1;
  w[LT_R_OBJECT_XY_POSITION] = -50.0f;
  w[LT_R_OBJECT_XY_VELOCITY] = 0.0f;
  w[LT_R_OBJECT_Z_CONTACT] = 0.0f;
  w[LT_R_OBJECT_Z_VELOCITY] = -0.5f;
  w[LT_R_OBJECT_ROLL_PITCH_ANGLE] = -0.05f;     // func = object_relative_roll_angle_ngt
  w[LT_R_OBJECT_ROLL_PITCH_VELOCITY] = -0.05f;  // func = object_relative_roll_velocity_ngt
  w[LT_R_OBJECT_YAW_ALIGNMENT] = -0.1f;
  w[LT_R_OBJECT_DANGEROUS_STATE] = -50.0f;
  // terminations: object_transport_teacher_env_cfg.py:108-114, cylinder_...:47-52
  c->term_enabled[LT_T_BASE_CONTACT] = 0;
  c->term_enabled[LT_T_OBJECT_BELOW_ROBOT] = 1;
  c->term_enabled[LT_T_OBJECT_BAD_ROLL] = 1;
  // object-state observation: object_transport_teacher_env_cfg.py:13-30
  const float noise[13] = {0.01f, 0.01f, 0.005f, 0.2f, 0.2f, 0.2f, 0.05f, 0.05f, 0.05f, 0.2f, 0.2f, 0.2f, 0.0f};
  const float scale[13] = {1, 1, 1, 0.5f, 0.5f, 0.5f, 1, 1, 1, 1, 0.25f, 0.25f, 0.25f};
  std::memcpy(c->obj_noise, noise, sizeof(noise));
  std::memcpy(c->obj_scale, scale, sizeof(scale));
  // events: object_transport_teacher_env_cfg.py:117-209, rand_cylinder_...:21-56
  set2(c->foot_friction, 0.6f, 1.5f);
  set2(c->foot_restitution, 0.0f, 0.3f);
  set2(c->trunk_friction, 0.3f, 1.0f);
  set2(c->trunk_restitution, 0.0f, 0.2f);
  set2(c->obj_friction, 0.3f, 1.0f);
  set2(c->obj_restitution, 0.0f, 0.2f);
  c->obj_material_buckets = 8000;                                    // object_transport_teacher_env_cfg.py:141
  set2(c->obj_mass_add, -0.5f, 1.5f);
  set2(c->reset_root_pos[2], 0.0f, 0.0f);
  for (int i = 0; i < 3; ++i) set2(c->reset_root_rpy[i], 0.0f, 0.0f);
  for (int i = 2; i < 6; ++i) set2(c->reset_root_vel[i], 0.0f, 0.0f);
  set2(c->obj_reset_pos[0], -0.05f, 0.05f);
  set2(c->obj_reset_pos[1], -0.04f, 0.04f);
  set2(c->obj_reset_pos[2], 0.095f, 0.10f);
  set2(c->obj_reset_rpy[0], 0.0f, 0.0f);
  set2(c->obj_reset_rpy[1], -kPi, kPi);
  set2(c->obj_reset_rpy[2], -kPi / 6, kPi / 6);
  set2(c->obj_radius, 0.03f, 0.07f);
  set2(c->obj_length, 0.1f, 0.4f);
  set2(c->push_robot_interval, 6.0f, 10.0f);
  set2(c->push_robot_vel[0], -0.4f, 0.4f);
  set2(c->push_robot_vel[1], -0.3f, 0.3f);
  set2(c->push_robot_vel[2], -0.1f, 0.1f);
  for (int i = 3; i < 6; ++i) set2(c->push_robot_vel[i], 0.0f, 0.0f);
  set2(c->push_obj_interval, 6.0f, 8.0f);
  set2(c->push_obj_vel[0], -0.3f, 0.3f);
  set2(c->push_obj_vel[1], -0.3f, 0.3f);
  set2(c->push_obj_vel[2], -0.2f, 0.2f);
  set2(c->push_obj_vel[3], -kPi / 20, kPi / 20);
  set2(c->push_obj_vel[4], -kPi / 20, kPi / 20);
  set2(c->push_obj_vel[5], -kPi / 5, kPi / 5);
  // tactile sensor of the student tasks, OFF here: term parameters of object_transport_student_env_cfg.py:16-37 (shared by every
  // TactileSignals class), sensor cadence :195-201
  c->tactile_enabled = 0;
  c->tactile_update_period = 0.025f;
  c->tactile_threshold = 0.05f;
  c->tactile_threshold_noise = 0.05f * 0.2f;
  c->tactile_dropout_prob = 0.005f;
  c->tactile_addition_prob = 0.005f;
  c->tactile_format = LT_TACTILE_BINARY;                             // :49-53 NoisyBinaryTactileCfg
  c->tactile_force_noise = 0.1f;                                     // force_n_prop_min / max -+0.1
  c->tactile_maximal_force = 3.0f;
  c->tactile_total_levels = 5;
  c->tactile_level_noise = 1.0f;                                     // level_n_min / max -+1
  c->tactile_aux_groups = 0;
}
// Isaac-LocomotionVelCur-LocoTouch-v1: locomotion + the MultiSampling command term and the velocity curriculum
// (config/base/locomotion_vel_cur_base_env_cfg.py:14-50; maxima = the locomotion ranges)
void locomotion_vel_cur(lt_cfg* c) {
  set2(c->cmd_range_init[0], -0.2f, 0.2f);
  set2(c->cmd_range_init[1], -0.1f, 0.1f);
  set2(c->cmd_range_init[2], -kPi / 10, kPi / 10);
  c->cmd_multi_sampling = 1;
  c->cur_enabled = 1;
  c->cur_bins[0] = c->cur_bins[1] = c->cur_bins[2] = 20;
  c->cur_len_threshold = 0.98f * 20.0f;
  c->cur_reward_threshold[0] = (float)(std::exp(-0.056 / 0.25) * 1.0 * 20.0);  // error_threshold_lin :38
  c->cur_reward_threshold[1] = (float)(std::exp(-0.089 / 0.25) * 0.5 * 20.0);  // error_threshold_ang :39
  c->cur_repeat_times[0] = c->cur_repeat_times[1] = 1;
  c->cur_max_distance_bins = 4;
}

// Isaac-CylinderTransportTeacher-LocoTouch-v1: the transport teacher with ONE fixed cylinder
// (config/locotouch/cylinder_transport_teacher_env_cfg.py:16-52; it keeps object_transport_teacher_env_cfg.py's material
// ranges and the function variant of the object reset, which rand_cylinder_transport_teacher_env_cfg.py:52-56 replace)
void cylinder_teacher(lt_cfg* c) {
  set2(c->obj_radius, 0.05f, 0.05f);
  set2(c->obj_length, 0.3f, 0.3f);
  set2(c->trunk_friction, 0.1f, 0.8f);
  set2(c->obj_friction, 0.1f, 0.8f);
  c->obj_reset_robot_frame = 1;
  c->cur_reward_threshold[0] = (float)(std::exp(-0.07 / 0.25) * 1.0 * 20.0);  // cylinder_...:37-38
  c->cur_reward_threshold[1] = (float)(std::exp(-0.08 / 0.25) * 0.5 * 20.0);
}

// Isaac-RandCylinderTransportStudent_SingleBinaryTac_CNNRNN_Mon-LocoTouch-v1: the rand-cylinder teacher env in its final
// ("play") setup plus the tactile sensor (config/locotouch/object_transport_student_env_cfg.py:161-201)
void student_binary_tac(lt_cfg* c) {
  c->episode_length_s = 10.0f;                                       // :172
  c->max_episode_length = 500;                                       // ceil(10.0 / 0.02)
  for (int i = 0; i < 3; ++i) set2(c->cmd_range_init[i], -c->cmd_range_max[i], c->cmd_range_max[i]);  // :179-182
  c->cmd_zero_steps = c->cmd_zero_steps_final;                       // :183
  c->cmd_rel_standing = c->cmd_rel_standing_final;                   // :184
  // the curriculum term stays registered; its thresholds scale with max_episode_length_s (mdp/curriculums.py:194-200)
  c->cur_len_threshold = 0.98f * 10.0f;
  c->cur_reward_threshold[0] = (float)(std::exp(-0.08 / 0.25) * 1.0 * 10.0);
  c->cur_reward_threshold[1] = (float)(std::exp(-0.1 / 0.25) * 0.5 * 10.0);
  c->tactile_enabled = 1;                                            // :195-201; the term parameters are transport_teacher()'s
}

// ... -Play-v1 (:164-171): 20 envs and the two 4-channel groups a ROS publisher of the reference visualises
void student_binary_tac_play(lt_cfg* c) {
  student_binary_tac(c);
  c->tactile_aux_groups = 3;
}

struct Preset { const char* id; int task; void (*extra)(lt_cfg*); int num_envs; };
// gym ids of the reference registry (locotouch/config/locotouch/__init__.py:14-117); -Play- variants differ in num_envs only
// for the fused terms (locomotion_base_env_cfg.py:364-374 `smaller_scene_for_playing`: 50 envs; cylinder play cfgs: 20)
const Preset kPresets[] = {
    {"Isaac-Locomotion-LocoTouch-v1", LT_TASK_LOCOMOTION, nullptr, 4096},
    {"Isaac-Locomotion-LocoTouch-Play-v1", LT_TASK_LOCOMOTION, nullptr, 50},
    {"Isaac-LocomotionVelCur-LocoTouch-v1", LT_TASK_LOCOMOTION, locomotion_vel_cur, 4096},
    {"Isaac-LocomotionVelCur-LocoTouch-Play-v1", LT_TASK_LOCOMOTION, locomotion_vel_cur, 50},
    {"Isaac-CylinderTransportTeacher-LocoTouch-v1", LT_TASK_TRANSPORT_TEACHER, cylinder_teacher, 4096},
    {"Isaac-CylinderTransportTeacher-LocoTouch-Play-v1", LT_TASK_TRANSPORT_TEACHER, cylinder_teacher, 50},
    {"Isaac-RandCylinderTransportTeacher-LocoTouch-v1", LT_TASK_TRANSPORT_TEACHER, nullptr, 4096},
    {"Isaac-RandCylinderTransportTeacher-LocoTouch-Play-v1", LT_TASK_TRANSPORT_TEACHER, nullptr, 50},
    {"Isaac-RandCylinderTransportStudent_SingleBinaryTac_CNNRNN_Mon-LocoTouch-v1", LT_TASK_TRANSPORT_TEACHER, student_binary_tac, 405},
    {"Isaac-RandCylinderTransportStudent_SingleBinaryTac_CNNRNN_Mon-LocoTouch-Play-v1", LT_TASK_TRANSPORT_TEACHER, student_binary_tac_play, 20},
};
}  // namespace

extern "C" {

int lt_abi_version(void) { return LT_ABI_VERSION; }
size_t lt_cfg_sizeof(void) { return sizeof(lt_cfg); }

int lt_cfg_default(int task, lt_cfg* cfg) {
  if (!cfg || (task != LT_TASK_LOCOMOTION && task != LT_TASK_TRANSPORT_TEACHER)) return LT_EINVAL;
  std::memset(cfg, 0, sizeof(*cfg));
  cfg->seed = 42;  // loco_rl/utils/config_utils.py:80 default seed
  cfg->num_envs = 4096;
  cfg->task = task;
  base_locomotion(cfg);
  if (task == LT_TASK_TRANSPORT_TEACHER) transport_teacher(cfg);
  return LT_OK;
}

int lt_cfg_preset(const char* gym_id, lt_cfg* cfg) {
  if (!gym_id || !cfg) return LT_EINVAL;
  for (const Preset& p : kPresets) {
    if (std::strcmp(p.id, gym_id) != 0) continue;
    const int rc = lt_cfg_default(p.task, cfg);
    if (rc != LT_OK) return rc;
    if (p.extra) p.extra(cfg);
    cfg->num_envs = p.num_envs;
    return LT_OK;
  }
  return LT_EINVAL;
}

int lt_cfg_num_presets(void) { return (int)(sizeof(kPresets) / sizeof(kPresets[0])); }
const char* lt_cfg_preset_id(int i) { return (i >= 0 && i < lt_cfg_num_presets()) ? kPresets[i].id : nullptr; }

int lt_cfg_obs_dim(const lt_cfg* cfg) {
  if (!cfg) return LT_EINVAL;
  return (cfg->task == LT_TASK_LOCOMOTION ? 45 : 58) * cfg->obs_history;
}

int lt_cfg_tactile_dim(const lt_cfg* cfg) {
  if (!cfg) return LT_EINVAL;
  if (!cfg->tactile_enabled) return 0;
  return (cfg->tactile_format == LT_TACTILE_PROCESSED || cfg->tactile_format == LT_TACTILE_ORIGINAL) ? LT_TACTILE_WIDE_DIM : LT_TACTILE_DIM;
}

}  // extern "C"
