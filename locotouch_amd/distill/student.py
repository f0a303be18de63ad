"""Student policy of the distillation (reference locotouch/distill/student.py:12-200): tactile image -> CNN head -> GRU
encoder -> embedding, concatenated with the proprioception -> MLP backbone -> action; trained by behaviour cloning on padded
whole trajectories against the teacher's actions (Monolithic) or its encoder's embedding (RMA).

Same constructor arguments, sub-module names (`pre_encoder`, `student_encoder`, `student_backbone`: checkpoints interchange),
forward / loss arithmetic, epoch schedule and checkpoint names.  The training loop keeps its statistics on the device and
reads them once per epoch (the reference calls `.item()` three times per batch).
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from ..rl.models import generate_model


class Student(nn.Module):
    def __init__(self, cfg, proprioception_dim: int, tactile_signal_dim, action_dim: int, teacher_policy_inference=None,
                 teacher_encoder_inference=None, teacher_backbone_weights=None, logger=None, verbose: bool = True):
        super().__init__()
        self.cfg = cfg
        self.logger = logger
        self.device = cfg.device
        self.log_dir = cfg.log_dir
        self.proprioception_dim = proprioception_dim
        self.tactile_signal_dim = tactile_signal_dim
        self.tactile_signal_img_shape = tuple(cfg.pre_encoder.img_shape)
        self.tactile_embedding_dim = cfg.tactile_encoder.embedding_dim
        self.action_dim = action_dim
        say = print if verbose else (lambda *a, **k: None)
        say("-------------- Construct Student Network --------------")
        self.use_pre_encoder = True if "CNN" in cfg.pre_encoder.model_type else (cfg.pre_encoder.hidden_dims is not None)
        if self.use_pre_encoder:
            self.pre_encoder = generate_model(tactile_signal_dim, cfg.pre_encoder.embedding_dim, cfg.pre_encoder).to(self.device)
            say(f"Pre Encoder: {self.pre_encoder}")
        self.student_encoder = generate_model(tactile_signal_dim if not self.use_pre_encoder else cfg.pre_encoder.embedding_dim,
                                              self.tactile_embedding_dim, cfg.tactile_encoder).to(self.device)
        say(f"Student Encoder: {self.student_encoder}")
        self.student_backbone = generate_model(proprioception_dim + self.tactile_embedding_dim, action_dim, cfg.student_policy).to(self.device)
        say(f"Student Backbone: {self.student_backbone}")
        self.MonolithicDistillation = cfg.distillation_type == "Monolithic"
        self.RMA_distillation = not self.MonolithicDistillation
        self.teacher_policy_inference = teacher_policy_inference
        self.teacher_encoder_inference = teacher_encoder_inference
        self.teacher_backbone_weights = teacher_backbone_weights
        if teacher_backbone_weights is not None:  # RMA: the teacher's actor is the student's (frozen) backbone
            self.student_backbone.model.load_state_dict(teacher_backbone_weights)
            for p in self.student_backbone.parameters():
                p.requires_grad = False
        self.max_iterations = cfg.num_iterations
        self.initial_epoches, self.incremental_epoches, self.final_epoches = cfg.initial_epoches, cfg.incremental_epoches, cfg.final_epoches
        self.batch_steps = cfg.batch_steps
        self._criterion = nn.MSELoss(reduction="none")
        self._distill_lr = cfg.distill_lr
        self._optimizer = torch.optim.AdamW(self.parameters(), lr=self._distill_lr)
        self.clip_actions, self.clip_range = cfg.clip_actions, cfg.clip_range
        self.action_scale_within_env = cfg.action_scale_within_env
        self.last_stats: dict = {}

    # ---- forward ---------------------------------------------------------------------------------------------------
    def encoder_forward(self, tactile_signal, hidden_states=None):
        if self.use_pre_encoder:
            shape = tactile_signal.shape                      # N x C x H x W, L x B x C x H x W, or flattened N x D / L x B x D
            if tactile_signal.dim() <= 3:
                tactile_signal = tactile_signal.reshape(*shape[:-1], *self.tactile_signal_img_shape)
                shape = tactile_signal.shape
            images = tactile_signal.reshape(-1, *shape[-3:])
            tactile_signal = self.pre_encoder(images).reshape(*shape[:-3], -1)
        return self.student_encoder(tactile_signal, hidden_states)

    def backbone_forward(self, proprioception, tactile_embedding):
        return self.student_backbone(torch.cat((proprioception, tactile_embedding), dim=-1))

    def forward(self, proprioception, tactile_signal, hidden_states=None):
        return self.backbone_forward(proprioception, self.encoder_forward(tactile_signal, hidden_states))

    def extract_input_and_forward(self, obs):
        return self.forward(obs["policy"][:, :self.proprioception_dim], obs["tactile"])

    def reset(self, dones=None):
        if self.use_pre_encoder:
            self.pre_encoder.reset(dones)
        self.student_encoder.reset(dones)
        self.student_backbone.reset(dones)

    def get_hidden_states(self):
        if hasattr(self.student_encoder, "get_hidden_states"):
            return self.student_encoder.get_hidden_states()

    # ---- training --------------------------------------------------------------------------------------------------
    def batch_loss(self, batch):
        """(loss, action_mse or None, action_mae) of one padded batch (student.py:119-152), all device scalars."""
        prop, enc_obs, tac, masks = batch["proprioceptions"], batch["teacher_encoder_obses"], batch["tactile_signals"], batch["masks"]
        denom = masks.sum()
        teacher_obs = torch.cat((prop, enc_obs), dim=-1)
        action_mse = None
        if self.MonolithicDistillation:
            student_actions = self.forward(prop, tac)
            with torch.no_grad():
                teacher_actions = self.teacher_policy_inference(teacher_obs)
            loss = self._criterion(student_actions, teacher_actions).mean(dim=-1)
        else:
            emb = self.encoder_forward(tac)
            with torch.no_grad():
                target = self.teacher_encoder_inference(enc_obs) if self.teacher_encoder_inference is not None else enc_obs
            loss = self._criterion(emb, target).mean(dim=-1)
            student_actions = self.backbone_forward(prop, emb)
            with torch.no_grad():
                teacher_actions = self.teacher_policy_inference(teacher_obs)
                action_mse = (((student_actions - teacher_actions) ** 2).mean(dim=-1) * masks).sum() / denom
        loss = (loss * masks).sum() / denom
        with torch.no_grad():
            sa, ta = student_actions, teacher_actions
            if self.clip_actions:
                sa, ta = sa.clamp(-self.clip_range, self.clip_range), ta.clamp(-self.clip_range, self.clip_range)
            action_mae = ((sa - ta).abs().mean(dim=-1) * masks).sum() / denom * self.action_scale_within_env
        return loss, action_mse, action_mae

    def _eager_step(self, batch):
        self._optimizer.zero_grad(set_to_none=True)
        loss, mse, mae = self.batch_loss(batch)
        loss.backward()
        self._optimizer.step()
        return loss.detach(), mse, mae

    def training_step(self, batch):
        """One optimizer step on a padded batch; (loss, action_mse | None, action_mae) as device scalars.
        (A hipGraph replay of this step was tried: +6 %, and capture of the ~3 k-launch step segfaulted in `capture_end` on
        this stack once the GRU's HIP time loop was inside - removed.)"""
        return self._eager_step(batch)

    def num_epoches(self, num_iter: int) -> int:
        n = self.initial_epoches + self.incremental_epoches * num_iter
        return n + (self.final_epoches if num_iter == self.max_iterations - 1 else 0)

    def train_on_data(self, replay_buffer, num_iter: int, progress: bool = False):
        self.train()
        batch_trajs = int(self.batch_steps / (replay_buffer.num_steps / replay_buffer.num_trajs)) + 1
        stats = {}
        for epoch in range(self.num_epoches(num_iter)):
            losses, mses, maes = [], [], []
            for batch in replay_buffer.to_recurrent_generator(batch_size=batch_trajs):
                loss, mse, mae = self.training_step(batch)
                losses.append(loss), maes.append(mae)
                if mse is not None:
                    mses.append(mse)
            stats = {"loss": float(torch.stack(losses).mean()), "action_mae": float(torch.stack(maes).mean())}
            stats["action_mse"] = float(torch.stack(mses).mean()) if mses else stats["loss"]
            if self.logger is not None:
                step = getattr(self, "_log_step", 0)
                self._log_step = step + 1
                scalars = {"train/Action MSE": stats["action_mse"], "train/Action MAE": stats["action_mae"]}
                if not self.MonolithicDistillation:
                    scalars["train/Encoder MSE"] = stats["loss"]
                if hasattr(self.logger, "add_scalar"):
                    for k, v in scalars.items():
                        self.logger.add_scalar(k, v, step)
                else:
                    self.logger.log(scalars)
            if progress:
                print(f"[Distillation iteration {num_iter}] epoch {epoch}: avg loss {stats['loss']:.4f}", flush=True)
        self.last_stats = stats
        if stats:
            print(f"[Distillation iteration {num_iter}] Action MSE: {stats['action_mse']}")
            print(f"[Distillation iteration {num_iter}] Action MAE: {stats['action_mae']}")
            if not self.MonolithicDistillation:
                print(f"[Distillation iteration {num_iter}] Encoder MSE: {stats['loss']}")
        self.save_model(num_iter)

    def save_model(self, iteration):
        torch.save(self.state_dict(), os.path.join(self.log_dir, f"model_{iteration}.pt"))

    def load_checkpoint(self, model_path):
        self.load_state_dict(torch.load(model_path, map_location=self.device, weights_only=True))
