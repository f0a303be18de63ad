"""Runner / PPO configurations of the registered tasks, as plain dicts (`agent_cfg.to_dict()` in the reference).

Values: reference locotouch/config/locotouch/agents/rsl_rl_ppo_cfg.py:6-79 (LocomotionPPORunnerCfg and the
RandCylinderTransportTeacherPPORunnerCfg that derives from it).
"""
from __future__ import annotations

import copy

_BASE = {
    "seed": 42,
    "num_steps_per_env": 24,
    "max_iterations": 30000,
    "save_interval": 50,
    "empirical_normalization": False,
    "logger": "tensorboard",
    "policy": {"class_name": "ActorCritic", "init_noise_std": 1.0, "actor_hidden_dims": [512, 256, 128],
               "critic_hidden_dims": [512, 256, 128], "activation": "elu"},
    "algorithm": {"class_name": "PPO", "value_loss_coef": 1.0, "use_clipped_value_loss": True, "clip_param": 0.2,
                  "entropy_coef": 0.01, "num_learning_epochs": 5, "num_mini_batches": 4, "learning_rate": 1.0e-3,
                  "schedule": "adaptive", "gamma": 0.99, "lam": 0.95, "desired_kl": 0.01, "max_grad_norm": 1.0},
}

# experiment names: agents/rsl_rl_ppo_cfg.py:31,41,57,64 (-Play- registrations share their task's agent cfg,
# locotouch/config/locotouch/__init__.py:19-114)
_EXPERIMENTS = {
    "Isaac-Locomotion-LocoTouch": "locotouch_locomotion",
    "Isaac-LocomotionVelCur-LocoTouch": "locotouch_vel_cur",
    "Isaac-CylinderTransportTeacher-LocoTouch": "locotouch_cylinder_transport_teacher",
    "Isaac-RandCylinderTransportTeacher-LocoTouch": "locotouch_rand_cylinder_transport_teacher",
}
TRAIN_CFGS = {f"{k}{suffix}": dict(_BASE, experiment_name=v) for k, v in _EXPERIMENTS.items() for suffix in ("-v1", "-Play-v1")}
# the student registrations carry the rand-cylinder TEACHER's runner cfg (config/locotouch/__init__.py:133,143): it names the
# experiment the teacher checkpoint is loaded from
for _suffix in ("-v1", "-Play-v1"):
    TRAIN_CFGS[f"Isaac-RandCylinderTransportStudent_SingleBinaryTac_CNNRNN_Mon-LocoTouch{_suffix}"] = dict(
        _BASE, experiment_name="locotouch_rand_cylinder_transport_teacher")


def train_cfg(task: str) -> dict:
    return copy.deepcopy(TRAIN_CFGS[task])
