"""On-policy training loop: rollout (policy act -> env step -> storage) then PPO update, with checkpoints and logging.

Interface of the reference's `OnPolicyRunner` (loco_rl/loco_rl/runners/on_policy_runner.py:21-489):
`OnPolicyRunner(env, train_cfg: dict, log_dir, device)`, `.learn(n, init_at_random_ep_len)`, `.save/.load` with the
checkpoint dict keys `model_state_dict, optimizer_state_dict, iter, infos` (:369-385), `.get_inference_policy`.
Perf scalars keep the reference's names (`Perf/total_fps`, `Perf/collection time`, `Perf/learning_time`, :275,291-293).
The per-step host syncs of the reference's bookkeeping (:197-199) are replaced by device-side accumulators that are
read once per iteration.
"""
from __future__ import annotations

import json
import os
import time
from collections import deque

import torch

from .dist import Dist
from .modules import ActorCritic
from .normalizer import EmpiricalNormalization
from .ppo import PPO


class OnPolicyRunner:
    def __init__(self, env, train_cfg: dict, log_dir: str | None = None, device="cpu", dist: Dist | None = None):
        self.cfg = dict(train_cfg)
        self.alg_cfg = dict(train_cfg["algorithm"])
        self.policy_cfg = dict(train_cfg["policy"])
        self.device, self.env, self.dist = device, env, dist or Dist()
        obs, extras = env.get_observations()
        num_obs = obs.shape[1]
        num_critic_obs = extras["observations"]["critic"].shape[1] if "critic" in extras["observations"] else num_obs
        from . import modules as _m

        cls_name = self.policy_cfg.pop("class_name", "ActorCritic")  # on_policy_runner.py:42 (`eval(class_name)`)
        if cls_name not in ("ActorCritic", "ActorCriticRecurrent", "ActorCriticEncoder"):
            raise NotImplementedError(f"policy class {cls_name!r} is not implemented (ActorCritic, ActorCriticRecurrent, ActorCriticEncoder are)")
        self.alg_cfg.pop("class_name", None)
        ac = getattr(_m, cls_name)(num_obs, num_critic_obs, env.num_actions, **self.policy_cfg).to(device)
        self.alg = PPO(ac, device=device, dist=self.dist, **self.alg_cfg)
        self.num_steps_per_env = int(self.cfg["num_steps_per_env"])
        self.save_interval = int(self.cfg.get("save_interval", 50))
        self.alg.init_storage(env.num_envs, self.num_steps_per_env, [num_obs], [num_critic_obs], [env.num_actions])
        # observation normalisers (on_policy_runner.py:85-95); identity unless `empirical_normalization` is set
        self.empirical_normalization = bool(self.cfg.get("empirical_normalization", False))
        if self.empirical_normalization:
            self.obs_normalizer = EmpiricalNormalization(shape=[num_obs], until=1.0e8).to(device)
            self.critic_obs_normalizer = EmpiricalNormalization(shape=[num_critic_obs], until=1.0e8).to(device)
        else:
            self.obs_normalizer = self.critic_obs_normalizer = torch.nn.Identity().to(device)
        self.log_dir = log_dir if self.dist.is_main else None
        self.writer = None
        self.logger_type = str(self.cfg.get("logger", "tensorboard")).lower()
        self.tot_timesteps, self.tot_time, self.current_learning_iteration = 0, 0.0, 0
        self.history: list[dict] = []
        self.git_status_repos: list[str] = []

    def learn(self, num_learning_iterations: int, init_at_random_ep_len: bool = False) -> None:
        env, alg = self.env, self.alg
        if self.log_dir is not None and self.writer is None:  # on_policy_runner.py:98-118
            if self.logger_type == "tensorboard":
                from .tb_writer import SummaryWriter

                self.writer = SummaryWriter(self.log_dir, flush_secs=10)
            elif self.logger_type in ("wandb", "neptune"):
                raise NotImplementedError(f"logger {self.logger_type!r} needs a network service; use --logger tensorboard (quirk Q9: the "
                                          "reference's agent cfgs default to wandb)")
            else:
                raise ValueError("Logger type not found. Please choose 'neptune', 'wandb' or 'tensorboard'.")
        if init_at_random_ep_len:  # on_policy_runner.py:121-124
            env.episode_length_buf = torch.randint_like(env.episode_length_buf, high=int(env.max_episode_length))
        obs, extras = env.get_observations()
        critic_obs = extras["observations"].get("critic", obs)
        obs, critic_obs = obs.to(self.device), critic_obs.to(self.device)
        self.train_mode()
        n = env.num_envs
        fused = self._make_fused()
        rew_acc = torch.zeros(n, device=self.device)
        len_acc = torch.zeros(n, device=self.device)
        rewbuffer, lenbuffer = deque(maxlen=100), deque(maxlen=100)
        start = self.current_learning_iteration
        for it in range(start, start + num_learning_iterations):
            t0 = time.time()
            fin_rew, fin_len = [], []
            if fused is not None:
                # GPU path: GEMMs + fused kernels, episode statistics from the env's device-side accumulators
                fused.rollout(self.num_steps_per_env)
                with torch.inference_mode():
                    alg.compute_returns(env.obs_critic)
            else:
              with torch.inference_mode():
                if it == start:
                    obs, critic_obs = self.obs_normalizer(obs), self.critic_obs_normalizer(critic_obs)
                for _ in range(self.num_steps_per_env):
                    actions = alg.act(obs, critic_obs)
                    obs, rewards, dones, infos = env.step(actions.to(env.device))
                    obs, rewards, dones = obs.to(self.device), rewards.to(self.device), dones.to(self.device)
                    obs = self.obs_normalizer(obs)                                                    # on_policy_runner.py:163-169
                    critic_obs = self.critic_obs_normalizer(infos["observations"]["critic"].to(self.device)) if "critic" in infos["observations"] else obs
                    alg.process_env_step(rewards, dones, infos)
                    rew_acc += rewards
                    len_acc += 1
                    d = dones > 0
                    fin_rew.append(torch.where(d, rew_acc, torch.full_like(rew_acc, float("nan"))))
                    fin_len.append(torch.where(d, len_acc, torch.full_like(len_acc, float("nan"))))
                    rew_acc = torch.where(d, torch.zeros_like(rew_acc), rew_acc)
                    len_acc = torch.where(d, torch.zeros_like(len_acc), len_acc)
                alg.compute_returns(critic_obs)
            if hasattr(env, "curriculum_sync"):  # multi-rank: one all-reduce of the curriculum sums per rollout
                env.curriculum_sync(self.dist, self.num_steps_per_env)
            t1 = time.time()
            value_loss, surrogate_loss, entropy, _, _ = alg.update()
            t2 = time.time()
            # one host read per iteration for the episode statistics
            if fin_rew:
                fr, fl = torch.stack(fin_rew).flatten(), torch.stack(fin_len).flatten()
                keep = ~torch.isnan(fr)
                rewbuffer.extend(fr[keep].tolist())
                lenbuffer.extend(fl[keep].tolist())
            self.current_learning_iteration = it
            collect, learn = t1 - t0, t2 - t1
            steps = self.num_steps_per_env * n * self.dist.world_size
            self.tot_timesteps += steps
            self.tot_time += collect + learn
            rec = {"iter": it, "Perf/total_fps": steps / (collect + learn), "Perf/collection time": collect,
                   "Perf/learning_time": learn, "Loss/value_function": value_loss, "Loss/surrogate": surrogate_loss,
                   "Loss/entropy": entropy, "Loss/learning_rate": alg.learning_rate,
                   "Policy/mean_noise_std": float(alg.actor_critic.action_std.detach().mean()),
                   "Train/mean_reward": (sum(rewbuffer) / len(rewbuffer)) if rewbuffer else None,
                   "Train/mean_episode_length": (sum(lenbuffer) / len(lenbuffer)) if lenbuffer else None}
            if hasattr(env, "episode_log"):
                rec.update(env.episode_log())
                if fused is not None:  # the env's own per-episode accumulators stand in for rewbuffer / lenbuffer
                    rec["Train/mean_reward"] = rec.get("Episode/reward")
                    rec["Train/mean_episode_length"] = rec.get("Episode/length")
            self.history.append(rec)
            if self.log_dir is not None:
                os.makedirs(self.log_dir, exist_ok=True)
                with open(os.path.join(self.log_dir, "progress.jsonl"), "a") as f:
                    f.write(json.dumps(rec) + "\n")
                if self.writer is not None:  # the reference's scalar tags (on_policy_runner.py:268-309), same x axis
                    for k, v in rec.items():
                        if k != "iter" and isinstance(v, (int, float)):
                            self.writer.add_scalar(k, v, it)
                    if rec.get("Train/mean_reward") is not None:
                        self.writer.add_scalar("Train/mean_reward/time", rec["Train/mean_reward"], int(self.tot_time))
                        self.writer.add_scalar("Train/mean_episode_length/time", rec["Train/mean_episode_length"], int(self.tot_time))
                if it % self.save_interval == 0:
                    self.save(os.path.join(self.log_dir, f"model_{it}.pt"))
        # the final model, under the number of the LAST iteration run - the reference's naming (on_policy_runner.py:221,243-245):
        # `current_learning_iteration` is that iteration, and a resumed run starts again from it
        if self.log_dir is not None:
            self.save(os.path.join(self.log_dir, f"model_{self.current_learning_iteration}.pt"))
            if self.writer is not None:
                self.writer.flush()

    def _make_fused(self):
        """FusedRollout when the env is the HIP env on a GPU and the policy is the plain feed-forward ActorCritic."""
        try:
            from ..env import LocoTouchVecEnv
            from .fused import FusedRollout
        except Exception:
            return None
        target = self.env
        if not isinstance(target, LocoTouchVecEnv):  # the scripts' wrapper (compat/runtime.py) hands the HIP env over when it adds nothing to a step
            target = getattr(self.env, "fused_target", lambda: None)()
        if not isinstance(target, LocoTouchVecEnv) or self.cfg.get("fused_rollout", True) is False:
            return None
        if getattr(self.alg.actor_critic, "noise_std_type", "scalar") != "scalar" or type(self.alg.actor_critic) is not ActorCritic:
            return None  # the fused rollout packs the plain feed-forward actor / critic MLPs
        if self.empirical_normalization:  # the fused rollout feeds raw observation rows to the MLP kernel
            return None
        return FusedRollout(target, self.alg)

    # ---- checkpoints (reference on_policy_runner.py:369-422) -------------------------------------------
    def save(self, path: str, infos=None) -> None:
        if not self.dist.is_main:
            return
        opt_sd = self.alg.optimizer.state_dict()
        for g in opt_sd["param_groups"]:  # the graph-replayed update keeps the learning rate in a device scalar: save a float
            if torch.is_tensor(g.get("lr")):
                g["lr"] = float(g["lr"])
        saved = {"model_state_dict": self.alg.actor_critic.state_dict(),
                 "optimizer_state_dict": opt_sd,
                 "iter": self.current_learning_iteration, "infos": infos}
        if self.empirical_normalization:  # on_policy_runner.py:376-379
            saved["obs_norm_state_dict"] = self.obs_normalizer.state_dict()
            saved["critic_obs_norm_state_dict"] = self.critic_obs_normalizer.state_dict()
        torch.save(saved, path)

    def load(self, path: str, load_optimizer: bool = True, pretrained: bool = False):
        loaded = torch.load(path, map_location=self.device, weights_only=True)
        sd = loaded["model_state_dict"]
        if pretrained:  # actor only + reset exploration noise (reference :404-412)
            actor_sd = {k[len("actor."):]: v for k, v in sd.items() if k.startswith("actor.")}
            self.alg.actor_critic.actor.load_state_dict(actor_sd)
            self.alg.actor_critic.reset_init_std()
        else:
            self.alg.actor_critic.load_state_dict(sd)
            if self.empirical_normalization:  # on_policy_runner.py:413-415
                self.obs_normalizer.load_state_dict(loaded["obs_norm_state_dict"])
                self.critic_obs_normalizer.load_state_dict(loaded["critic_obs_norm_state_dict"])
            if load_optimizer:
                self.alg.optimizer.load_state_dict(loaded["optimizer_state_dict"])
            self.current_learning_iteration = loaded["iter"]
        if self.dist.world_size > 1:
            self.alg.broadcast_parameters()
        return loaded.get("infos")

    def train_mode(self) -> None:  # reference on_policy_runner.py:466-475
        self.alg.train_mode()
        if self.empirical_normalization:
            self.obs_normalizer.train()
            self.critic_obs_normalizer.train()

    def eval_mode(self) -> None:  # reference on_policy_runner.py:477-486
        self.alg.test_mode()
        if self.empirical_normalization:
            self.obs_normalizer.eval()
            self.critic_obs_normalizer.eval()

    def add_git_repo_to_log(self, repo_file_path: str) -> None:
        """Reference on_policy_runner.py:488-489: remembers code locations whose git state goes into the log directory.  There is
        no GitPython in this image, so the paths are recorded (`<log_dir>/git/repos.txt`) and no diff is taken."""
        self.git_status_repos.append(repo_file_path)
        if self.log_dir is not None:
            os.makedirs(os.path.join(self.log_dir, "git"), exist_ok=True)
            with open(os.path.join(self.log_dir, "git", "repos.txt"), "a") as f:
                f.write(str(repo_file_path) + "\n")

    def get_inference_encoder(self, device=None):  # reference on_policy_runner.py:435-448: None unless the policy has an encoder
        self.eval_mode()
        ac = self.alg.actor_critic
        if device is not None:
            ac.to(device)
        if callable(getattr(ac, "act_encoder_inference", None)):
            if self.empirical_normalization:
                return lambda x: ac.act_encoder_inference(self.obs_normalizer(x))
            return ac.act_encoder_inference
        return None

    def get_backbone_weights(self):  # :463-464 (RMA distillation: the teacher's actor becomes the student's frozen backbone)
        return self.alg.actor_critic.actor.state_dict()

    def get_inference_policy(self, device=None):  # reference on_policy_runner.py:424-436
        self.eval_mode()
        if device is not None:
            self.alg.actor_critic.to(device)
        policy = self.alg.actor_critic.act_inference
        if self.empirical_normalization:
            if device is not None:
                self.obs_normalizer.to(device)
            policy = lambda x: self.alg.actor_critic.act_inference(self.obs_normalizer(x))  # noqa: E731
        return policy
