"""GPU twin of tests/test_export_student.py: the TorchScript student, moved to the GPU, drives the HIP student env for an episode
(delayed tactile frames through the TactileRecorder, per-env resets at episode ends) and reproduces `Student`'s own inference
step for step; the observation-layout contract is checked on the HIP env's rows."""
import pytest

pytestmark = pytest.mark.gpu
STUDENT = "Isaac-RandCylinderTransportStudent_SingleBinaryTac_CNNRNN_Mon-LocoTouch-v1"


def test_scripted_student_drives_the_hip_env_like_the_student_module(tmp_path):
    import torch

    from locotouch_amd.distill import Student, TactileRecorder, distillation_cfg
    from locotouch_amd.distill.export import OBS_LAYOUT, export_student_as_jit, newest_frame
    from locotouch_amd.env import make

    n = 64
    env = make(STUDENT, num_envs=n, device="cuda:0", seed=3)
    env.episode_length_buf = torch.randint(440, 500, (n,), device="cuda:0")  # episodes end (time-out at 500) inside the 50 steps
    cfg = distillation_cfg(STUDENT)
    cfg.device, cfg.log_dir = "cuda:0", str(tmp_path)
    torch.manual_seed(5)
    st = Student(cfg, OBS_LAYOUT["proprioception_dim"], OBS_LAYOUT["tactile_dim"], 12, verbose=False).eval()
    pol = torch.jit.load(export_student_as_jit(st, str(tmp_path / "exported"))).to("cuda:0")
    pol.reset()
    st.reset()
    rec = TactileRecorder("cuda:0", n, OBS_LAYOUT["tactile_dim"], cfg.min_delay, cfg.max_delay)
    obs, extras = env.get_observations()
    resets = 0
    with torch.inference_mode():
        for t in range(50):
            rec.record_new_tactile_signals(extras["observations"]["tactile"])
            tac = rec.get_tactile_signals().clone()
            prop = obs[:, :OBS_LAYOUT["proprioception_dim"]].clone()
            want = st(prop, tac)
            got = pol(prop, tac)
            torch.testing.assert_close(got, want, rtol=2e-4, atol=2e-5, msg=lambda m, t=t: f"step {t}: {m}")
            torch.testing.assert_close(pol.hidden_state, st.get_hidden_states(), rtol=2e-4, atol=2e-5)
            obs, _, dones, extras = env.step(want)
            keep = dones == 0
            crit = extras["observations"]["critic"]
            torch.testing.assert_close(newest_frame(crit, "last_action")[keep], (0.25 * want.clamp(-100, 100))[keep], rtol=1e-5, atol=1e-6)
            assert torch.equal(extras["observations"]["object_state"], obs[:, 270:])
            if bool(dones.any()):
                resets += int(dones.sum())
                st.reset(dones)
                pol.reset_idx(dones)
                rec.reset(dones.nonzero(as_tuple=False).flatten())
    assert resets >= n // 2, "episodes must have ended inside the run (the per-env reset path is what is being tested)"
