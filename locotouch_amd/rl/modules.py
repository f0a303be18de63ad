"""Actor-critic network of the teacher policy.

Mirrors the interface of the reference's `ActorCritic` (loco_rl/loco_rl/modules/actor_critic.py:8-144): MLP actor
348->512->256->128->12 and critic ->1 with ELU, a state-independent Gaussian whose std is 12 free parameters
("scalar" noise type, init 1.0), and the same `state_dict` key names (`actor.{0,2,4,6}.*`, `critic.*`, `std`) so
checkpoints written by either side load into the other (SURVEY.md §5 checkpoint row).
"""
from __future__ import annotations

import torch
import torch.nn as nn
from torch.distributions import Normal

from .linear import Linear

_ACTIVATIONS = {"elu": nn.ELU, "relu": nn.ReLU, "selu": nn.SELU, "tanh": nn.Tanh, "sigmoid": nn.Sigmoid,
                "lrelu": nn.LeakyReLU, "identity": nn.Identity}


def build_mlp(in_dim: int, hidden: list[int], out_dim: int, activation: str) -> nn.Sequential:
    """Linear/activation stack with the activation after every hidden layer and none after the head."""
    act = _ACTIVATIONS[activation.lower()]
    sizes = [in_dim, *hidden]
    layers: list[nn.Module] = []
    for a, b in zip(sizes[:-1], sizes[1:]):
        layers += [Linear(a, b), act()]
    layers.append(Linear(sizes[-1], out_dim))
    return nn.Sequential(*layers)


class ActorCritic(nn.Module):
    is_recurrent = False

    def __init__(self, num_actor_obs: int, num_critic_obs: int, num_actions: int, actor_hidden_dims=(256, 256, 256),
                 critic_hidden_dims=(256, 256, 256), activation: str = "elu", init_noise_std: float = 1.0,
                 noise_std_type: str = "scalar", **unused):
        super().__init__()
        # construction order (actor, critic, std) fixes the RNG draw order of the initial weights
        self.actor = build_mlp(num_actor_obs, list(actor_hidden_dims), num_actions, activation)
        self.critic = build_mlp(num_critic_obs, list(critic_hidden_dims), 1, activation)
        self.init_noise_std = float(init_noise_std)
        self.noise_std_type = noise_std_type
        if noise_std_type == "scalar":
            self.std = nn.Parameter(self.init_noise_std * torch.ones(num_actions))
        elif noise_std_type == "log":
            self.log_std = nn.Parameter(torch.log(self.init_noise_std * torch.ones(num_actions)))
        else:
            raise ValueError(f"unknown noise_std_type {noise_std_type!r} (expected 'scalar' or 'log')")
        self.distribution: Normal | None = None
        Normal.set_default_validate_args(False)

    # -- distribution ------------------------------------------------------------------------------
    def _std_like(self, mean: torch.Tensor) -> torch.Tensor:
        s = self.std if self.noise_std_type == "scalar" else torch.exp(self.log_std)
        return s.expand_as(mean)

    def update_distribution(self, observations: torch.Tensor) -> None:
        mean = self.actor(observations)
        self.distribution = Normal(mean, self._std_like(mean))

    @property
    def action_mean(self) -> torch.Tensor:
        return self.distribution.mean

    @property
    def action_std(self) -> torch.Tensor:
        return self.distribution.stddev

    @property
    def entropy(self) -> torch.Tensor:
        return self.distribution.entropy().sum(dim=-1)

    # -- rollout / update API ------------------------------------------------------------------------
    def act(self, observations: torch.Tensor, **kwargs) -> torch.Tensor:
        self.update_distribution(observations)
        # == Normal.sample() draw for draw (same normal_(0,1) fill), but hipGraph-capturable: torch.normal(mean_t, std_t)
        # refuses stream capture on ROCm, randn_like does not.
        d = self.distribution
        with torch.no_grad():
            return d.mean + d.stddev * torch.randn_like(d.mean)

    def get_actions_log_prob(self, actions: torch.Tensor) -> torch.Tensor:
        return self.distribution.log_prob(actions).sum(dim=-1)

    def act_inference(self, observations: torch.Tensor) -> torch.Tensor:
        return self.actor(observations)

    def evaluate(self, critic_observations: torch.Tensor, **kwargs) -> torch.Tensor:
        return self.critic(critic_observations)

    def reset(self, dones=None) -> None:  # feed-forward policy: nothing to reset
        pass

    def reset_init_std(self) -> None:
        with torch.no_grad():
            if self.noise_std_type == "scalar":
                self.std.fill_(self.init_noise_std)
            else:
                self.log_std.fill_(float(torch.log(torch.tensor(self.init_noise_std))))

    def forward(self):
        raise NotImplementedError("use act / evaluate")
