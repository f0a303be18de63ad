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

from .linear import Linear, MLPSequential

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
    return MLPSequential(*layers)


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
        if self.distribution is None:  # fused rollout + fused update build no Normal object: the state-independent std itself
            return self.std if self.noise_std_type == "scalar" else torch.exp(self.log_std)
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


class PolicyMemory(nn.Module):
    """`Memory` of the reference's recurrent policy (loco_rl/loco_rl/modules/actor_critic_recurrent.py:66-94): batch mode (masks
    given) runs padded trajectories from their saved first hidden states and returns the outputs in (T, envs) layout; inference
    mode carries the module's own state.  Parameter names `rnn.*`."""

    def __init__(self, input_size, type="lstm", num_layers=1, hidden_size=256):
        super().__init__()
        self.rnn = (nn.GRU if type.lower() == "gru" else nn.LSTM)(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers)
        self.hidden_states = None

    def forward(self, input, masks=None, hidden_states=None):
        from .trajectories import unpad_trajectories

        if masks is not None:
            if hidden_states is None:
                raise ValueError("Hidden states not passed to memory module during policy update")
            out, _ = self.rnn(input, hidden_states)
            return unpad_trajectories(out, masks)
        out, self.hidden_states = self.rnn(input.unsqueeze(0), self.hidden_states)
        return out

    def reset(self, dones=None):
        if self.hidden_states is None:
            return
        keep = None if dones is None else (dones.reshape(-1) != 1)
        for h in (self.hidden_states if isinstance(self.hidden_states, tuple) else (self.hidden_states,)):
            if keep is None:
                h.zero_()
            else:
                h.mul_(keep.to(h.dtype)[None, :, None])


class ActorCriticRecurrent(ActorCritic):
    """Reference loco_rl/loco_rl/modules/actor_critic_recurrent.py:8-63: an RNN per network in front of the MLPs."""

    is_recurrent = True

    def __init__(self, num_actor_obs, num_critic_obs, num_actions, actor_hidden_dims=(256, 256, 256), critic_hidden_dims=(256, 256, 256),
                 activation="elu", rnn_type="lstm", rnn_hidden_size=256, rnn_num_layers=1, init_noise_std=1.0, **unused):
        super().__init__(num_actor_obs=rnn_hidden_size, num_critic_obs=rnn_hidden_size, num_actions=num_actions,
                         actor_hidden_dims=actor_hidden_dims, critic_hidden_dims=critic_hidden_dims, activation=activation,
                         init_noise_std=init_noise_std)
        self.memory_a = PolicyMemory(num_actor_obs, type=rnn_type, num_layers=rnn_num_layers, hidden_size=rnn_hidden_size)
        self.memory_c = PolicyMemory(num_critic_obs, type=rnn_type, num_layers=rnn_num_layers, hidden_size=rnn_hidden_size)

    def reset(self, dones=None):
        self.memory_a.reset(dones)
        self.memory_c.reset(dones)

    def act(self, observations, masks=None, hidden_states=None):
        return super().act(self.memory_a(observations, masks, hidden_states).squeeze(0))

    def update_distribution_recurrent(self, observations, masks, hidden_states):
        super().update_distribution(self.memory_a(observations, masks, hidden_states).squeeze(0))

    def act_inference(self, observations):
        return super().act_inference(self.memory_a(observations).squeeze(0))

    def evaluate(self, critic_observations, masks=None, hidden_states=None):
        return super().evaluate(self.memory_c(critic_observations, masks, hidden_states).squeeze(0))

    def get_hidden_states(self):
        return self.memory_a.hidden_states, self.memory_c.hidden_states


class ActorCriticEncoder(ActorCritic):
    """Reference loco_rl/loco_rl/modules/actor_critic_encoder.py:7-117: the tail of the observation row goes through an MLP
    encoder whose embedding is concatenated with the head of the row (the teacher form of RMA-style distillation:
    `act_encoder_inference` / `act_backbone_inference` are what `Distillation` asks the runner for)."""

    is_recurrent = False

    def __init__(self, actor_obs_dim, critic_obs_dim, num_actions, actor_flatten_obs_end_idx, actor_encoder_obs_start_idx,
                 actor_encoder_hidden_dims, actor_encoder_embedding_dim, actor_hidden_dims, critic_flatten_obs_end_idx=None,
                 critic_encoder_obs_start_idx=None, critic_encoder_hidden_dims=None, critic_encoder_embedding_dim=None,
                 critic_hidden_dims=(256, 256, 256), encoder_activation="elu", encoder_final_activation=None, activation="elu",
                 init_noise_std=1.0, **unused):
        from .models import MLP

        enc_dim = abs(actor_encoder_obs_start_idx) if actor_encoder_obs_start_idx < 0 else actor_obs_dim - actor_encoder_obs_start_idx
        flat_dim = actor_flatten_obs_end_idx if actor_flatten_obs_end_idx > 0 else actor_obs_dim - abs(actor_flatten_obs_end_idx)
        with_c = critic_encoder_hidden_dims is not None
        c_enc = c_flat = 0
        if with_c:
            assert None not in (critic_flatten_obs_end_idx, critic_encoder_obs_start_idx, critic_encoder_embedding_dim)
            c_enc = abs(critic_encoder_obs_start_idx) if critic_encoder_obs_start_idx < 0 else critic_obs_dim - critic_encoder_obs_start_idx
            c_flat = critic_flatten_obs_end_idx if critic_flatten_obs_end_idx > 0 else critic_obs_dim - abs(critic_flatten_obs_end_idx)
        super().__init__(num_actor_obs=flat_dim + actor_encoder_embedding_dim,
                         num_critic_obs=(c_enc + critic_encoder_embedding_dim) if with_c else critic_obs_dim,  # (sic, :49)
                         num_actions=num_actions, actor_hidden_dims=actor_hidden_dims, critic_hidden_dims=critic_hidden_dims,
                         activation=activation, init_noise_std=init_noise_std)
        self.actor_encoder_obs_dim, self.actor_flatten_obs_dim = enc_dim, flat_dim
        self.critic_with_encoder, self.critic_encoder_obs_dim, self.critic_flatten_obs_dim = with_c, c_enc, c_flat
        self.actor_encoder = MLP(enc_dim, actor_encoder_hidden_dims, actor_encoder_embedding_dim, activation=encoder_activation,
                                 final_layer_activation=encoder_final_activation)
        if with_c:
            self.critic_encoder = MLP(c_enc, critic_encoder_hidden_dims, critic_encoder_embedding_dim, activation=encoder_activation,
                                      final_layer_activation=encoder_final_activation)

    def _actor_input(self, obs):
        return torch.cat([obs[..., :self.actor_flatten_obs_dim], self.actor_encoder(obs[..., -self.actor_encoder_obs_dim:])], dim=-1)

    def update_distribution(self, observations):
        super().update_distribution(self._actor_input(observations))

    def act_inference(self, obs):
        return super().act_inference(self._actor_input(obs))

    def act_encoder_inference(self, encoder_obs):
        return self.actor_encoder(encoder_obs)

    def act_backbone_inference(self, flatten_obs, embedding):
        return super().act_inference(torch.cat([flatten_obs, embedding], dim=-1))

    def evaluate(self, obs, **kwargs):
        if self.critic_with_encoder:
            obs = torch.cat([obs[..., :self.critic_flatten_obs_dim], self.critic_encoder(obs[..., -self.critic_encoder_obs_dim:])], dim=-1)
        return super().evaluate(obs)

    def get_actor_critic_obs_from_obs_dict(self, obs_dict):
        actor_obs, enc = obs_dict["policy"], obs_dict["encoder"]
        critic_obs = obs_dict.get("critic", actor_obs)
        return torch.cat((actor_obs, enc), dim=1), (torch.cat((critic_obs, enc), dim=1) if self.critic_with_encoder else critic_obs)
