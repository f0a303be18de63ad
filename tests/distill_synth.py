"""Deterministic synthetic inputs shared by tools/gen_golden_distill.py (which feeds them to the reference's own
locotouch/distill + loco_rl.models classes) and tests/test_distill.py (which feeds them to locotouch_amd.distill).
No reference code here: a scripted env, a fixed linear "teacher", seeded tensors."""
from __future__ import annotations

import torch

PROPRIO, OBJ, TACTILE, ACTIONS = 270, 78, 442, 12


def grid(g, shape, lo=-1.0, hi=1.0):
    """Uniform values snapped to a 2^-8 grid (exact in fp32, small fixtures)."""
    return torch.round((torch.rand(*shape, generator=g) * (hi - lo) + lo) * 256.0) / 256.0


def teacher_policy():
    """A fixed linear map obs[348] -> action[12] (stands in for ActorCritic.act_inference)."""
    g = torch.Generator().manual_seed(101)
    W = grid(g, (PROPRIO + OBJ, ACTIONS), -0.05, 0.05)
    return lambda obs: obs @ W


def student_inputs(seed=7, n=5, L=7, B=3):
    g = torch.Generator().manual_seed(seed)
    steps = [dict(prop=grid(g, (n, PROPRIO)), tac=(torch.rand(n, TACTILE, generator=g) < 0.1).float()) for _ in range(4)]
    lengths = [7, 4, 6][:B]
    masks = torch.zeros(L, B, dtype=torch.bool)
    for b, ln in enumerate(lengths):
        masks[:ln, b] = True
    batch = dict(proprioceptions=grid(g, (L, B, PROPRIO)) * masks.unsqueeze(-1),
                 teacher_encoder_obses=grid(g, (L, B, OBJ)) * masks.unsqueeze(-1),
                 tactile_signals=(torch.rand(L, B, TACTILE, generator=g) < 0.1).float() * masks.unsqueeze(-1), masks=masks)
    return steps, batch


class ScriptedEnv:
    """num_envs envs whose rows are a pure function of (env, episode index, step in episode); episode k of env e lasts
    `3 + (5 * e + 7 * k) % 11` steps.  `form`: "tuple" = (obs, extras) / step -> (obs, rew, dones, extras) with
    extras["observations"]; "dict" = the group dict the reference's distillation code indexes (get_observations() -> dict,
    step -> (dict, rew, dones, extras))."""

    def __init__(self, num_envs=6, form="tuple"):
        self.num_envs, self.num_actions, self.device, self.form = num_envs, ACTIONS, torch.device("cpu"), form
        self.reset()

    def _length(self, e, k):
        return 3 + (5 * e + 7 * k) % 11

    def reset(self):
        self.ep = torch.zeros(self.num_envs, dtype=torch.long)
        self.t = torch.zeros(self.num_envs, dtype=torch.long)
        self.actions_seen = []
        return self.get_observations()

    def _groups(self):
        e = torch.arange(self.num_envs, dtype=torch.float32).unsqueeze(1)
        code = (e * 1000 + self.ep.unsqueeze(1).float() * 50 + self.t.unsqueeze(1).float()) / 4096.0
        pol = code + torch.arange(PROPRIO + OBJ, dtype=torch.float32).unsqueeze(0) / 1024.0
        tac = ((torch.arange(TACTILE).unsqueeze(0) + self.t.unsqueeze(1) + 3 * torch.arange(self.num_envs).unsqueeze(1)) % 7 == 0).float()
        return {"policy": pol, "critic": pol.clone(), "tactile": tac, "object_state": pol[:, PROPRIO:]}

    def get_observations(self):
        g = self._groups()
        return g if self.form == "dict" else (g["policy"], {"observations": g})

    def step(self, action):
        self.actions_seen.append(action.detach().clone())
        self.t += 1
        lens = torch.tensor([self._length(e, int(self.ep[e])) for e in range(self.num_envs)])
        dones = (self.t >= lens).long()
        reward = (torch.arange(self.num_envs, dtype=torch.float32) + 1.0) / 8.0 + self.t.float() / 64.0
        self.ep += dones
        self.t = torch.where(dones.bool(), torch.zeros_like(self.t), self.t)
        g = self._groups()
        extras = {"observations": g}
        return (g if self.form == "dict" else g["policy"]), reward, dones, extras
