"""Distillation on the real (HIP) student env: DAgger loop end to end on cuda:0 - teacher PPO actor -> device replay buffer ->
CNN/GRU student on MIOpen / hipBLASLt -> checkpoints -> play; TorchScript export of the teacher."""
import glob
import os

import pytest

pytestmark = pytest.mark.gpu
STUDENT = "Isaac-RandCylinderTransportStudent_SingleBinaryTac_CNNRNN_Mon-LocoTouch-v1"


def test_dagger_on_hip_student_env(tmp_path):
    import torch

    from locotouch_amd.agents import train_cfg
    from locotouch_amd.distill import Distillation, distillation_cfg
    from locotouch_amd.env import make
    from locotouch_amd.rl import OnPolicyRunner

    torch.manual_seed(0)
    env = make(STUDENT, num_envs=64, device="cuda:0", seed=3)
    assert env.tactile and env.max_episode_length == 500
    agent = train_cfg(STUDENT)
    runner = OnPolicyRunner(env, agent, log_dir=None, device="cuda:0")
    teacher = runner.get_inference_policy(device="cuda:0")
    cfg = distillation_cfg(STUDENT)
    cfg.logger, cfg.log_root_path = "tensorboard", str(tmp_path)
    cfg.num_iterations, cfg.bc_data_steps, cfg.dagger_data_steps = 2, 1500, 800
    cfg.initial_epoches, cfg.incremental_epoches, cfg.batch_steps, cfg.evaluation_trajs_num = 8, 2, 600, 16
    d = Distillation(env, cfg, teacher_policy=teacher, verbose=False)
    assert d.proprioception_dim == 270 and d.tactile_signal_dim == 442
    hist = d.train()
    assert [h["iter"] for h in hist] == [0, 1, "eval"]
    assert hist[0]["collect/trj_num"] > 0 and hist[2]["collect/trj_num"] >= 16
    rb = d.replay_buffer
    # BC: the student regresses the teacher's actions on the teacher's own state distribution
    assert hist[0]["train/loss"] < 0.5 and all(torch.isfinite(p).all() for p in d.student.parameters())
    assert sorted(os.path.basename(p) for p in glob.glob(os.path.join(cfg.log_dir, "model_*.pt"))) == ["model_0.pt", "model_1.pt"]
    assert next(d.student.parameters()).is_cuda and rb.num_trajs == 0  # cleared before the evaluation
    p = Distillation(env, distillation_cfg(STUDENT), training=False, checkpoint=os.path.join(cfg.log_dir, "model_1.pt"), verbose=False)
    a = p.play(num_steps=20)
    assert a.shape == (64, 12) and torch.isfinite(a).all()


def test_replay_buffer_rows_are_what_the_env_showed(tmp_path):
    """The kept trajectories hold exactly the policy rows the env exposed at each step, and the DELAYED tactile rows (delay 1:
    the previous step's map, the first step of an episode its own)."""
    import torch

    from locotouch_amd.distill import ReplayBuffer, TactileRecorder
    from locotouch_amd.env import make

    env = make(STUDENT, num_envs=32, device="cuda:0", seed=5)
    seen_pol, seen_tac, seen_done = [], [], []

    class Spy:
        num_envs, device, num_actions = env.num_envs, env.device, 12

        def get_observations(self):
            return env.get_observations()

        def reset(self):
            return env.reset()

        def step(self, a):
            obs, ex = env.get_observations()
            seen_pol.append(obs.clone()), seen_tac.append(ex["observations"]["tactile"].clone())
            out = env.step(a)
            seen_done.append(out[2].clone())
            return out

    g = torch.Generator(device="cuda").manual_seed(1)
    teacher = lambda obs: 1.5 * torch.randn(obs.shape[0], 12, device=obs.device, generator=g)  # noqa: E731  (falls quickly: short episodes)
    rb = ReplayBuffer(Spy(), TactileRecorder(env.device, env.num_envs, 442, 1, 2), 270, check_every=8)
    rewards, lengths = rb.collect_data(teacher, None, num_steps=600)
    assert rb.num_steps >= 600 and rb.num_trajs == len(rb._traj_len) > 3
    pol, tac, done = torch.stack(seen_pol), torch.stack(seen_tac), torch.stack(seen_done)
    (flat_pol, flat_tac), (first, length) = rb._materialise()
    n = env.num_envs
    for k in range(rb.num_trajs):
        f, ln = int(first[k]), int(length[k])
        t0, e = divmod(f, n)
        assert torch.equal(flat_pol[f:f + ln * n:n], pol[t0:t0 + ln, e])
        assert bool(done[t0 + ln - 1, e]) and not bool(done[t0:t0 + ln - 1, e].any())
        assert t0 == 0 or bool(done[t0 - 1, e])
        want = torch.cat([tac[t0:t0 + 1, e], tac[t0:t0 + ln - 1, e]])  # delay 1, first signal fills the register
        assert torch.equal(flat_tac[f:f + ln * n:n], want)
    assert sum(lengths[:rb.num_trajs]) == rb.num_steps


def test_export_policy_as_jit_matches_actor(tmp_path):
    import torch

    from locotouch_amd.agents import train_cfg
    from locotouch_amd.compat.runtime import export_policy_as_jit, export_policy_as_onnx
    from locotouch_amd.env import make
    from locotouch_amd.rl import OnPolicyRunner

    task = "Isaac-RandCylinderTransportTeacher-LocoTouch-Play-v1"
    env = make(task, device="cuda:0", seed=1)
    runner = OnPolicyRunner(env, train_cfg(task), log_dir=None, device="cuda:0")
    path = export_policy_as_jit(runner.alg.actor_critic, None, path=str(tmp_path / "exported"), filename="policy.pt")
    mod = torch.jit.load(path)
    obs, _ = env.get_observations()
    want = runner.get_inference_policy(device="cuda:0")(obs).cpu()
    torch.testing.assert_close(mod(obs.cpu()), want, rtol=1e-4, atol=1e-5)
    try:
        import onnx  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match="onnx"):
            export_policy_as_onnx(runner.alg.actor_critic, path=str(tmp_path / "exported"))
