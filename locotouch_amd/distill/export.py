"""Deployment export of the policy the reference actually deploys: the tactile student (CNN head -> GRU -> MLP embedding,
concatenated with the proprioception -> MLP backbone -> 12 joint-position actions; locotouch/distill/student.py:88-106,175-183),
as ONE TorchScript file, together with the observation-layout contract the robot-side runtime has to honour
(locotouch/scripts/play.py:140-144: `term_dims = [3, 3, 3, 12, 12, 12, 13]`, history 6, term-major; README.md:49-53 points at the
external Go1-Policy-Deployment repository that consumes such files).

    from locotouch_amd.distill.export import export_student_as_jit, OBS_LAYOUT
    path = export_student_as_jit(student, "exported", "student_policy.pt")
    policy = torch.jit.load(path)
    policy.reset()
    action = policy(proprioception[B, 270], delayed_tactile[B, 442])      # hidden state carried inside the module
    action, h = policy.step(proprioception, delayed_tactile, h)            # or explicit hidden-state in / out
    policy.reset_idx(dones)                                                # per-env reset at episode ends

The exported module holds plain `nn.Conv2d / nn.Linear / nn.GRU` copies of the trained parameters (the training-time nodes of
rl/linear.py, rl/models.py and the HIP GRU time loop are not scriptable, and the robot has no HIP device anyway).
ONNX is not offered: the image ships no `onnx` package (compat/runtime.py: export_policy_as_onnx says so).
"""
from __future__ import annotations

import copy
import os

import torch
import torch.nn as nn

# ---- the observation-layout contract of the registered student task (Isaac-RandCylinderTransportStudent_SingleBinaryTac_CNNRNN_Mon) ----
# policy group, 348 columns, TERM-major, each term's 6 frames oldest -> newest (ObservationManager history):
OBS_TERMS = [("velocity_commands", 3), ("base_ang_vel", 3), ("projected_gravity", 3), ("joint_pos", 12), ("joint_vel", 12),
             ("last_action", 12), ("object_state", 13)]
OBS_HISTORY = 6
OBS_LAYOUT = {
    "term_dims": [d for _, d in OBS_TERMS],          # play.py:140
    "history_length": OBS_HISTORY,                   # play.py:141
    "policy_dim": sum(d for _, d in OBS_TERMS) * OBS_HISTORY,                 # 348
    "proprioception_dim": sum(d for _, d in OBS_TERMS[:-1]) * OBS_HISTORY,    # 270 = policy minus object_state (distillation.py:57)
    "object_state_dim": 13 * OBS_HISTORY,            # 78: the teacher's privileged block, NOT an input of the student
    "object_state_terms": [3, 3, 4, 3],              # position, linear velocity, orientation (wxyz), angular velocity (play.py:150)
    "tactile_shape": (2, 17, 13),                    # two identical binary channels of the 17 x 13 taxel grid (observations.py:307-308)
    "tactile_dim": 442,
    "tactile_delay_steps": 1,                        # the policy sees the PREVIOUS env step's tactile frame (tactile_recorder.py:22)
    "joint_order": "a_FR, b_FL, c_RR, d_RL per joint type: 4 hips, 4 thighs, 4 calves",
    "scales": {"base_ang_vel": 0.25, "joint_vel": 0.05, "object_state": [1, 1, 1, .5, .5, .5, 1, 1, 1, 1, .25, .25, .25]},
    "action": "joint position targets = default_joint_pos + 0.25 * clamp(action, -100, 100) (mdp/actions.py:39-44)",
}


def term_slices() -> dict:
    """name -> slice of the policy row (term-major blocks of dim * history columns)."""
    out, off = {}, 0
    for name, d in OBS_TERMS:
        out[name] = slice(off, off + d * OBS_HISTORY)
        off += d * OBS_HISTORY
    return out


def newest_frame(row: torch.Tensor, name: str) -> torch.Tensor:
    """The most recent frame of term `name` in policy rows (N, 348)."""
    s = term_slices()[name]
    d = dict(OBS_TERMS)[name]
    return row[:, s.stop - d:s.stop]


def _plain(m: nn.Module) -> nn.Module:
    """Plain-PyTorch copy of a trained sub-network (same parameters): scriptable and free of this package's classes."""
    if isinstance(m, nn.Conv2d):
        c = nn.Conv2d(m.in_channels, m.out_channels, m.kernel_size, m.stride, m.padding, m.dilation, m.groups, bias=m.bias is not None)
        c.load_state_dict(m.state_dict())
        return c
    if isinstance(m, nn.Linear):
        lin = nn.Linear(m.in_features, m.out_features, bias=m.bias is not None)
        lin.load_state_dict(m.state_dict())
        return lin
    if isinstance(m, nn.Sequential):
        return nn.Sequential(*[_plain(x) for x in m])
    if isinstance(m, (nn.ReLU, nn.ELU, nn.SELU, nn.LeakyReLU, nn.Tanh, nn.Sigmoid, nn.MaxPool2d, nn.Identity, nn.BatchNorm2d)):
        return copy.deepcopy(m)
    raise NotImplementedError(f"export: no plain counterpart for {type(m).__name__}")


class StudentDeploy(nn.Module):
    """(proprioception [B, P], delayed tactile [B, 442]) -> actions [B, 12] with the GRU state inside the module (or passed
    explicitly through `step`)."""

    def __init__(self, student):
        super().__init__()
        from ..rl.models import MLP, RNN, CNN2dHead

        enc, back = student.student_encoder, student.student_backbone
        if not isinstance(enc, RNN) or not isinstance(enc.memory.rnn, nn.GRU) or enc.memory.rnn.num_layers != 1 or not isinstance(back, MLP):
            raise NotImplementedError("export_student_as_jit covers the registered student: [CNN head ->] one-layer GRU + MLP encoder, MLP backbone")
        self.use_pre_encoder = bool(student.use_pre_encoder)
        if self.use_pre_encoder:
            pre = student.pre_encoder
            if isinstance(pre, CNN2dHead):
                self.conv = _plain(pre.conv.conv)
                self.conv_head = _plain(pre.head.model) if isinstance(pre.head, MLP) else nn.Identity()
                self.is_cnn = True
            elif isinstance(pre, MLP):
                self.conv, self.conv_head, self.is_cnn = nn.Identity(), _plain(pre.model), False
            else:
                raise NotImplementedError(f"export: pre-encoder {type(pre).__name__}")
        else:
            self.conv, self.conv_head, self.is_cnn = nn.Identity(), nn.Identity(), False
        g = enc.memory.rnn
        self.gru = nn.GRU(input_size=g.input_size, hidden_size=g.hidden_size, num_layers=1)
        self.gru.load_state_dict(g.state_dict())
        self.encoder_mlp = _plain(enc.mlp.model)
        self.backbone = _plain(back.model)
        self.hidden_size: int = int(g.hidden_size)
        self.proprioception_dim: int = int(student.proprioception_dim)
        c, h, w = (int(v) for v in student.tactile_signal_img_shape)
        self.img_c, self.img_h, self.img_w = c, h, w
        self.register_buffer("hidden_state", torch.zeros(1, 1, self.hidden_size))

    def _embed(self, tactile: torch.Tensor, hidden: torch.Tensor):
        x = tactile
        if self.use_pre_encoder:
            if self.is_cnn:
                x = self.conv(x.reshape(-1, self.img_c, self.img_h, self.img_w)).flatten(1)
            x = self.conv_head(x)
        out, hidden = self.gru(x.unsqueeze(0), hidden)
        return self.encoder_mlp(out.squeeze(0)), hidden

    @torch.jit.export
    def step(self, proprioception: torch.Tensor, tactile: torch.Tensor, hidden: torch.Tensor):
        """One control step with the hidden state passed in and out: hidden is (1, B, hidden_size), zeros at episode start."""
        emb, hidden = self._embed(tactile, hidden)
        return self.backbone(torch.cat((proprioception, emb), dim=-1)), hidden

    def forward(self, proprioception: torch.Tensor, tactile: torch.Tensor) -> torch.Tensor:
        b = proprioception.shape[0]
        if self.hidden_state.shape[1] != b:  # first call of a run (or a new batch size): a fresh zero state
            self.hidden_state = torch.zeros(1, b, self.hidden_size, dtype=proprioception.dtype, device=proprioception.device)
        action, h = self.step(proprioception, tactile, self.hidden_state)
        self.hidden_state = h
        return action

    @torch.jit.export
    def forward_obs(self, policy_obs: torch.Tensor, tactile: torch.Tensor) -> torch.Tensor:
        """The same from whole policy rows (B, 348): the proprioception is their first `proprioception_dim` columns
        (locotouch/distill/student.py:175-178)."""
        return self.forward(policy_obs[:, :self.proprioception_dim], tactile)

    @torch.jit.export
    def reset(self) -> None:
        self.hidden_state = torch.zeros_like(self.hidden_state)

    @torch.jit.export
    def reset_idx(self, dones: torch.Tensor) -> None:
        """Zero the state of the envs whose `dones` entry is non-zero (shape (B,))."""
        keep = (dones.reshape(-1) == 0).to(self.hidden_state.dtype)
        self.hidden_state = self.hidden_state * keep.reshape(1, -1, 1)


def export_student_as_jit(student, path: str = ".", filename: str = "student_policy.pt") -> str:
    """TorchScript file `path/filename` of the student (CPU tensors; `torch.jit.load(...).to(device)` moves it)."""
    os.makedirs(path, exist_ok=True)
    mod = StudentDeploy(student).cpu().eval()
    out = os.path.join(path, filename)
    torch.jit.script(mod).save(out)
    return out


__all__ = ["export_student_as_jit", "StudentDeploy", "OBS_LAYOUT", "OBS_TERMS", "OBS_HISTORY", "term_slices", "newest_frame"]
