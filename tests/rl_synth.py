"""Seeded synthetic rollout for BASELINE.json configs[0] (64 envs x 24 steps, obs 348, 12 actions).

Shared by tools/gen_golden_rl.py (which feeds it to the reference's loco_rl) and
tests/test_rl_parity.py (which feeds it to locotouch_amd.rl); the data is regenerated from the seed
on both sides, so only the reference's *outputs* are stored in tests/golden/rl_ppo_cfg1.npz.
Hyper-parameters: reference locotouch/config/locotouch/agents/rsl_rl_ppo_cfg.py:11-30.
"""
import torch

N_ENVS, N_STEPS, N_OBS, N_ACT = 64, 24, 348, 12
POLICY_CFG = dict(init_noise_std=1.0, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128], activation="elu")
PPO_CFG = dict(value_loss_coef=1.0, use_clipped_value_loss=True, clip_param=0.2, entropy_coef=0.01, num_learning_epochs=5,
               num_mini_batches=4, learning_rate=1.0e-3, schedule="adaptive", gamma=0.99, lam=0.95, desired_kl=0.01,
               max_grad_norm=1.0)


def synth_rollout(seed: int = 123):
    g = torch.Generator().manual_seed(seed)
    obs = torch.randn(N_STEPS, N_ENVS, N_OBS, generator=g)
    critic_obs = obs + 0.05 * torch.randn(N_STEPS, N_ENVS, N_OBS, generator=g)
    rewards = torch.randn(N_STEPS, N_ENVS, generator=g)
    dones = (torch.rand(N_STEPS, N_ENVS, generator=g) < 0.05)
    time_outs = dones & (torch.rand(N_STEPS, N_ENVS, generator=g) < 0.4)
    last_critic_obs = torch.randn(N_ENVS, N_OBS, generator=g)
    return dict(obs=obs, critic_obs=critic_obs, rewards=rewards, dones=dones.long(), time_outs=time_outs,
                last_critic_obs=last_critic_obs)


def extra_inputs(seed: int = 77):
    """Seeded inputs for tools/gen_golden_rl_extra.py / tests/test_rl_extra.py (normaliser, trajectories, recurrent generator)."""
    g = torch.Generator().manual_seed(seed)
    T, N, D, A, H = 12, 8, 5, 3, 4
    dones = (torch.rand(T, N, 1, generator=g) < 0.2).to(torch.uint8)
    dones[:, 3] = 0                                   # an env without any done inside the window
    dones[5, 0] = 1; dones[6, 0] = 1                  # back-to-back dones: a length-1 trajectory
    x = dict(T=T, N=N, D=D, A=A, H=H, num_mini_batches=2,
             norm_batches=torch.randn(6, 16, 7, generator=g) * torch.tensor([1, 2, 0.5, 3, 1, 10, 0.1]) + torch.tensor([0, 1, -1, 5, 0, 2, 0.3]),
             norm_until=70,
             traj_tensor=torch.randn(T, N, D, generator=g), traj_dones=dones.clone(),
             obs=torch.randn(T, N, D, generator=g), cobs=torch.randn(T, N, D, generator=g), actions=torch.randn(T, N, A, generator=g),
             rewards=torch.randn(T, N, generator=g), dones=dones.squeeze(-1).clone(), values=torch.randn(T, N, 1, generator=g),
             logp=torch.randn(T, N, generator=g), mu=torch.randn(T, N, A, generator=g), sigma=torch.rand(T, N, A, generator=g) + 0.1,
             hid_a=torch.randn(T, 1, N, H, generator=g), hid_c=torch.randn(T, 1, N, H, generator=g),
             last_values=torch.randn(N, 1, generator=g))
    return x


def policy_case(kind: str, seed: int = 321):
    """Seeded inputs for tools/gen_golden_rl_policies.py / tests/test_rl_policies.py: a small rollout for the recurrent (GRU) and
    the encoder policy classes.  `args` / `kwargs` are the constructor arguments (same for the reference and this package)."""
    g = torch.Generator().manual_seed(seed + (0 if kind == "recurrent" else 1))
    N, T, A = 8, 10, 4
    D = 14
    dones = (torch.rand(T, N, generator=g) < 0.15).long()
    dones[:, 2] = 0
    case = dict(N=N, T=T, D=D, A=A, seed=seed, obs=torch.randn(T, N, D, generator=g), rewards=torch.randn(T, N, generator=g), dones=dones,
                last_cobs=torch.randn(N, D, generator=g))
    case["cobs"] = case["obs"] + 0.05 * torch.randn(T, N, D, generator=g)
    case["ppo"] = dict(value_loss_coef=1.0, use_clipped_value_loss=True, clip_param=0.2, entropy_coef=0.01, num_learning_epochs=2,
                       num_mini_batches=2, learning_rate=1.0e-3, schedule="adaptive", gamma=0.99, lam=0.95, desired_kl=0.01, max_grad_norm=1.0)
    if kind == "recurrent":
        case["args"] = (D, D, A)
        case["kwargs"] = dict(actor_hidden_dims=[32, 16], critic_hidden_dims=[32, 16], activation="elu", rnn_type="gru", rnn_hidden_size=24,
                              rnn_num_layers=1, init_noise_std=1.0)
    else:
        case["args"] = (D, D, A)
        case["kwargs"] = dict(actor_flatten_obs_end_idx=9, actor_encoder_obs_start_idx=-5, actor_encoder_hidden_dims=[16, 8],
                              actor_encoder_embedding_dim=6, actor_hidden_dims=[32, 16], critic_flatten_obs_end_idx=None,
                              critic_encoder_obs_start_idx=None, critic_encoder_hidden_dims=None, critic_encoder_embedding_dim=None,
                              critic_hidden_dims=[32, 16], encoder_activation="elu", encoder_final_activation=None, activation="elu",
                              init_noise_std=1.0)
    return case
