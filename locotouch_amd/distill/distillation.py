"""DAgger distillation loop (reference locotouch/distill/distillation.py:20-212) on the MI355X-native student env.

    iteration 0:  collect `bc_data_steps` with the TEACHER acting        -> behaviour cloning
    iteration i:  collect `dagger_data_steps` with the STUDENT acting, teacher actions as labels (the buffer keeps growing)
    last:         evaluate the student on `evaluation_trajs_num` episodes

The env is the student registration (`cfg.tactile_enabled`): observation groups policy (proprioception 270 | object state 78),
tactile (442) and object_state (78).  The teacher is a trained PPO runner of the same task family (its checkpoint loads
with `weights_only=True`).
"""
from __future__ import annotations

import datetime
import os

import numpy as np
import torch

from .config import DistillationCfg
from .replay_buffer import ReplayBuffer
from .student import Student
from .tactile_recorder import TactileRecorder


class Distillation:
    def __init__(self, env, cfg: DistillationCfg, teacher_policy=None, teacher_encoder=None, teacher_backbone_weights=None,
                 training: bool = True, log_dir: str | None = None, checkpoint: str | None = None, verbose: bool = True):
        self.env, self.cfg, self.training = env, cfg, training
        obs, extras = env.get_observations()
        groups = extras["observations"]
        if "tactile" not in groups or "object_state" not in groups:
            raise ValueError("distillation needs a student task (observation groups `tactile` and `object_state`)")
        self.proprioception_dim = groups["policy"].shape[-1] - groups["object_state"].shape[-1]   # distillation.py:57
        self.tactile_signal_dim = groups["tactile"].shape[-1]
        if verbose:
            print(f"[INFO] Tactile signal dim: {self.tactile_signal_dim}, Proprioception dim: {self.proprioception_dim}")
        self.tactile_recorder = TactileRecorder(env.device, env.num_envs, self.tactile_signal_dim, cfg.min_delay, cfg.max_delay)
        cfg.device = str(env.device)
        self.logger = None
        if training:
            if teacher_policy is None:
                raise ValueError("training needs the teacher's inference policy")
            root = os.path.abspath(os.path.join(cfg.log_root_path, cfg.experiment_name))
            cfg.log_dir = log_dir or os.path.join(root, datetime.datetime.now().strftime("%Y-%m-%d_%H-%M-%S"))
            os.makedirs(cfg.log_dir, exist_ok=True)
            if verbose:
                print(f"[INFO] Logging student distillation in: {cfg.log_dir}")
            if cfg.logger == "tensorboard":
                from ..rl.tb_writer import SummaryWriter

                self.logger = SummaryWriter(log_dir=cfg.log_dir)
            elif cfg.logger == "wandb":
                try:
                    import wandb  # noqa: F401
                except ImportError:  # no wandb in this image: keep the run alive, log to TensorBoard event files instead
                    print("[WARN] wandb is not installed: logging scalars to TensorBoard event files instead")
                    from ..rl.tb_writer import SummaryWriter

                    self.logger = SummaryWriter(log_dir=cfg.log_dir)
                else:
                    self.logger = wandb.init(project=cfg.wandb_project, config=cfg.to_dict(), name=os.path.basename(cfg.log_dir))
            self.teacher_policy_inference = teacher_policy
            mono = cfg.distillation_type == "Monolithic"
            self.student = Student(cfg, self.proprioception_dim, self.tactile_signal_dim, env.num_actions,
                                   teacher_policy_inference=teacher_policy,
                                   teacher_encoder_inference=None if mono else teacher_encoder,
                                   teacher_backbone_weights=None if mono else teacher_backbone_weights, logger=self.logger, verbose=verbose)
            self.replay_buffer = ReplayBuffer(env, self.tactile_recorder, self.proprioception_dim)
        else:
            self.student = Student(cfg, self.proprioception_dim, self.tactile_signal_dim, env.num_actions, verbose=verbose)
            if checkpoint is not None:
                self.student.load_checkpoint(checkpoint)
                if verbose:
                    print(f"[INFO] Loading student policy checkpoint from: {checkpoint}")
        self.history: list[dict] = []

    def run(self):
        return self.train() if self.training else self.play()

    def train(self):
        c = self.cfg
        for it in range(c.num_iterations):
            print("-" * 100)
            rewards, lengths = self.replay_buffer.collect_data(
                teacher_policy=self.teacher_policy_inference, student_policy=self.student if it else None,
                num_steps=c.dagger_data_steps if it else c.bc_data_steps)
            rec = self.log_trajectory_rewards_and_lengths(rewards, lengths, it)
            self.student.train_on_data(self.replay_buffer, it)
            rec.update({"iter": it, **{f"train/{k}": v for k, v in self.student.last_stats.items()}})
            self.history.append(rec)
            if it == c.num_iterations - 1:
                self.replay_buffer.clear_buffer()
                rewards, lengths = self.replay_buffer.evaluate(self.student, c.evaluation_trajs_num)
                self.history.append({"iter": "eval", **self.log_trajectory_rewards_and_lengths(rewards, lengths, it + 1)})
                print("Log dir: ", self.student.log_dir)
        if self.logger is not None:
            (self.logger.close if hasattr(self.logger, "close") else self.logger.finish)()
        return self.history

    def log_trajectory_rewards_and_lengths(self, rewards: list, lengths: list, step: int = 0) -> dict:
        rewards, lengths = np.array(rewards, dtype=np.float64), np.array(lengths, dtype=np.float64)
        if rewards.size == 0:
            return {"collect/trj_num": 0}
        rec = {"collect/trj_num": int(rewards.size), "collect/trj_less_half_len_num": int((lengths < 0.5 * lengths.max()).sum()),
               "collect/step_reward_mean": float(np.mean(rewards / lengths)), "collect/trj_rwd_mean": float(rewards.mean()),
               "collect/trj_rwd_std": float(rewards.std()), "collect/trj_len_mean": float(lengths.mean()),
               "collect/trj_len_std": float(lengths.std())}
        if self.logger is not None:
            if hasattr(self.logger, "add_scalar"):
                for k, v in rec.items():
                    self.logger.add_scalar(k, v, step)
            else:
                self.logger.log(rec, commit=False)
        print(f"Collected {rec['collect/trj_num']} trajectories:")
        print(f"Trajectories less than half length: {rec['collect/trj_less_half_len_num']}")
        print(f"Mean step reward: {rec['collect/step_reward_mean']}")
        print(f"Mean reward: {rec['collect/trj_rwd_mean']}, Std reward: {rec['collect/trj_rwd_std']}")
        print(f"Mean length: {rec['collect/trj_len_mean']}, Std length: {rec['collect/trj_len_std']}")
        return rec

    def play(self, num_steps: int | None = None):
        """distillation.py:172-212: the student drives the env on DELAYED tactile rows; returns the actions of the last step."""
        self.student.eval()
        obs, extras = self.env.get_observations()
        groups = dict(extras["observations"])
        action, t = None, 0
        with torch.inference_mode():
            while num_steps is None or t < num_steps:
                self.tactile_recorder.record_new_tactile_signals(groups["tactile"])
                groups["tactile"] = self.tactile_recorder.get_tactile_signals().clone()
                action = self.student.extract_input_and_forward(groups)
                obs, _, dones, extras = self.env.step(action)
                groups = dict(extras["observations"])
                done_mask = dones != 0
                self.tactile_recorder.reset(done_mask)
                self.student.reset(done_mask)
                t += 1
        return action
