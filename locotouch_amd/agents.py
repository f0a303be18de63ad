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

TRAIN_CFGS = {
    "Isaac-Locomotion-LocoTouch-v1": dict(_BASE, experiment_name="locotouch_locomotion"),
    "Isaac-RandCylinderTransportTeacher-LocoTouch-v1": dict(_BASE, experiment_name="locotouch_rand_cylinder_transport_teacher"),
}


def train_cfg(task: str) -> dict:
    return copy.deepcopy(TRAIN_CFGS[task])
