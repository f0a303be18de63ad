"""The import surface the reference's launch scripts run on (compat/runtime.py: ManagedEnv = what gym.make returns,
RslRlVecEnvWrapper, ObsGroups / ObsTensor) over the HIP env on the GPU - no reference file involved.  CPU-side,
tests/test_reference_scripts.py runs the reference's own train.py / play.py / distill.py on the same surface over the oracle env;
here the seam those scripts cross (train.py:98,116 `gym.make` + `RslRlVecEnvWrapper`; on_policy_runner.py:127,158 tuple form;
distillation.py:53-55, replay_buffer.py:35-37,51-53 mapping form) is exercised where the product runs."""
import glob
import os

import pytest

pytestmark = pytest.mark.gpu
TEACHER = "Isaac-RandCylinderTransportTeacher-LocoTouch-v1"
STUDENT = "Isaac-RandCylinderTransportStudent_SingleBinaryTac_CNNRNN_Mon-LocoTouch-v1"


def _wrapped(task, n, seed=2):
    from locotouch_amd.compat.runtime import ManagedEnv, RslRlVecEnvWrapper
    from locotouch_amd.env import LocoTouchVecEnv

    vec = LocoTouchVecEnv(task, num_envs=n, device="cuda:0", seed=seed)  # the registration's preset cfg
    env = ManagedEnv(task, None, vec)
    assert env.unwrapped is env and env.num_envs == n and env.device == vec.device
    return RslRlVecEnvWrapper(env), vec


def test_both_observation_call_forms_over_the_hip_env():
    import torch

    w, vec = _wrapped(TEACHER, 128)
    got = w.get_observations()
    obs, extras = got  # the runner / play.py / ReplayBuffer.evaluate form
    assert obs.shape == (128, 348) and set(extras["observations"]) >= {"policy", "critic"}
    assert torch.equal(got["policy"], obs) and dict(got.items()).keys() == extras["observations"].keys()  # the distillation form
    nxt, rew, dones, ex = w.step(torch.zeros(128, 12, device="cuda:0"))
    assert nxt.shape == (128, 348) and torch.equal(nxt["critic"], ex["observations"]["critic"])
    assert {k for k, _ in nxt.items()} == set(ex["observations"]) and nxt.to("cpu").shape == (128, 348)
    assert rew.shape == (128,) and dones.dtype == torch.long and "time_outs" in ex
    w.episode_length_buf = torch.randint_like(w.episode_length_buf, high=int(w.max_episode_length))  # on_policy_runner.py:121-124
    assert torch.equal(vec.episode_length_buf, w.episode_length_buf)


def test_runner_learns_through_the_wrapper(tmp_path):
    import torch

    from locotouch_amd.agents import train_cfg
    from locotouch_amd.rl import OnPolicyRunner

    w, vec = _wrapped(TEACHER, 256)
    agent = train_cfg(TEACHER)
    agent["save_interval"] = 1
    runner = OnPolicyRunner(w, agent, log_dir=str(tmp_path), device="cuda:0")
    # the wrapper adds nothing to a step here, so the trainer runs its fused rollout (hipGraph, two launches per step) on the HIP env
    # behind it; action clipping in the wrapper or a user term on the ManagedEnv sends it through `step` instead
    assert w.fused_target() is vec and runner._make_fused() is not None
    from locotouch_amd.compat.runtime import RslRlVecEnvWrapper
    assert RslRlVecEnvWrapper(w.env, clip_actions=1.0).fused_target() is None
    w.env.add_reward_term("zero", lambda e: e.scene["robot"].data.joint_vel[:, 0] * 0.0, 1.0)
    assert w.fused_target() is None
    w.env.extra = None
    runner.learn(num_learning_iterations=2, init_at_random_ep_len=True)
    assert sorted(os.path.basename(p) for p in glob.glob(str(tmp_path / "model_*.pt")))[-1] == "model_1.pt"
    assert all(torch.isfinite(p).all() for p in runner.alg.actor_critic.parameters())
    assert int(vec.counters[0]) >= 2 * 24


def test_dagger_collection_through_the_wrapper():
    import torch

    from locotouch_amd.agents import train_cfg
    from locotouch_amd.distill import ReplayBuffer, TactileRecorder
    from locotouch_amd.rl import OnPolicyRunner

    w, vec = _wrapped(STUDENT, 64, seed=3)
    teacher = OnPolicyRunner(w, train_cfg(STUDENT), log_dir=None, device="cuda:0").get_inference_policy(device="cuda:0")
    rb = ReplayBuffer(w, TactileRecorder(w.device, w.num_envs, 442, 1, 2), 270, check_every=8)  # indexes env_obs["policy"], next_obs.items()
    rewards, lengths = rb.collect_data(teacher, None, num_steps=400)
    assert rb.num_steps >= 400 and rb.num_trajs > 0 and all(torch.isfinite(torch.tensor(rewards[:rb.num_trajs])))
