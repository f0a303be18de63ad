"""SURVEY.md §8(f)4 - the policy that is deployed: the tactile student exported as TorchScript with its GRU state inside the module
(or passed explicitly), and the observation-layout contract the deployment runtime assumes (locotouch/scripts/play.py:140-144:
term_dims [3,3,3,12,12,12,13], history 6, term-major; README.md:49-53).  CPU; the GPU twin is tests/test_hip_export_student.py."""
import numpy as np
import torch

from locotouch_amd.distill import Student, distillation_cfg
from locotouch_amd.distill.export import OBS_LAYOUT, export_student_as_jit, newest_frame, term_slices
from tests import distill_synth as S

TASK = "Isaac-RandCylinderTransportStudent_SingleBinaryTac_CNNRNN_Mon-LocoTouch-v1"


def make_student(tmp):
    cfg = distillation_cfg(TASK)
    cfg.device, cfg.log_dir = "cpu", str(tmp)
    torch.manual_seed(77)
    return Student(cfg, S.PROPRIO, S.TACTILE, S.ACTIONS, teacher_policy_inference=S.teacher_policy(), verbose=False).eval()


def test_scripted_student_reproduces_the_students_inference_over_an_episode_with_resets(tmp_path):
    st = make_student(tmp_path)
    path = export_student_as_jit(st, str(tmp_path / "exported"))
    pol = torch.jit.load(path)
    assert pol.proprioception_dim == 270 and pol.hidden_size == 512
    b = 5
    g = torch.Generator().manual_seed(0)
    pol.reset()
    h = torch.zeros(1, b, 512)
    with torch.no_grad():
        for t in range(50):
            prop = torch.randn(b, S.PROPRIO, generator=g)
            tac = (torch.rand(b, S.TACTILE, generator=g) < 0.1).float()
            want = st(prop, tac)
            got = pol(prop, tac)
            got2, h = pol.step(prop, tac, h)
            torch.testing.assert_close(got, want, rtol=1e-5, atol=2e-6)
            torch.testing.assert_close(got2, want, rtol=1e-5, atol=2e-6)
            torch.testing.assert_close(pol.hidden_state, st.get_hidden_states(), rtol=1e-5, atol=2e-6)
            policy_rows = torch.cat((prop, torch.zeros(b, 78)), dim=1)
            if t == 20:  # an episode ends in envs 1 and 3: per-env reset on all three holders of the state
                dones = torch.tensor([0, 1, 0, 1, 0])
                st.reset(dones)
                pol.reset_idx(dones)
                h = h * (dones == 0).float().reshape(1, -1, 1)
                assert float(pol.hidden_state[0, 1].abs().sum()) == 0.0 and float(pol.hidden_state[0, 0].abs().sum()) > 0.0
            if t == 35:  # a full reset
                st.reset()
                pol.reset()
                h = torch.zeros(1, b, 512)
            _ = policy_rows
        # whole policy rows in: the proprioception is the first 270 columns
        prop, tac = torch.randn(b, 270, generator=g), torch.zeros(b, 442)
        a1 = pol.forward_obs(torch.cat((prop, torch.randn(b, 78, generator=g)), dim=1), tac)
        st(prop, tac)  # keeps the twin's state in step
        assert a1.shape == (b, 12)
    # a new batch size starts from a fresh zero state
    assert pol(torch.zeros(2, 270), torch.zeros(2, 442)).shape == (2, 12) and pol.hidden_state.shape == (1, 2, 512)


def test_observation_layout_contract_holds_on_the_student_env():
    """The contract the exported policy is deployed against, checked on the env itself (CPU: the oracle env): term-major blocks of
    6 frames, oldest -> newest; proprioception = first 270 columns; object_state = last 78 = the `object_state` group; tactile
    [2, 17, 13] binary with two identical channels."""
    from tests.oracle_vec_env import OracleVecEnv

    L = OBS_LAYOUT
    assert L["term_dims"] == [3, 3, 3, 12, 12, 12, 13] and L["history_length"] == 6 and L["policy_dim"] == 348
    assert L["proprioception_dim"] == 270 and L["tactile_dim"] == 442 and L["tactile_shape"] == (2, 17, 13)
    sl = term_slices()
    assert sl["velocity_commands"] == slice(0, 18) and sl["joint_pos"] == slice(54, 126) and sl["object_state"] == slice(270, 348)
    n = 24
    env = OracleVecEnv(TASK, num_envs=n, seed=2)
    g = torch.Generator().manual_seed(0)
    frames = []
    for t in range(9):
        act = 0.4 * torch.randn(n, 12, generator=g)
        obs, _, dones, extras = env.step(act)
        crit = extras["observations"]["critic"]  # the noise-free twin of the policy rows
        keep = dones == 0
        # newest frame of each proprioceptive term against the state the step left
        np.testing.assert_allclose(newest_frame(crit, "last_action")[keep], 0.25 * act.clamp(-100, 100)[keep], rtol=1e-6, atol=1e-7)
        q = env.field("LT_F_JOINT_POS").reshape(n, 12)
        dq = torch.tensor([-0.1, 0.1, -0.1, 0.1] + [0.9] * 4 + [-1.8] * 4)
        np.testing.assert_allclose(newest_frame(crit, "joint_pos")[keep], (q - dq)[keep], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(newest_frame(crit, "joint_vel")[keep], 0.05 * env.field("LT_F_JOINT_VEL").reshape(n, 12)[keep], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(newest_frame(crit, "velocity_commands")[keep], env.field("LT_F_CMD")[:, 0, :3][keep], atol=1e-6)
        # the object_state group is the last 78 columns of the policy rows
        assert torch.equal(extras["observations"]["object_state"], obs[:, 270:])
        tac = extras["observations"]["tactile"].reshape(n, 2, 17, 13)
        assert torch.equal(tac[:, 0], tac[:, 1]) and set(np.unique(tac.numpy())) <= {0.0, 1.0}
        frames.append((crit.clone(), keep.clone()))
    # history: frame k of step t is frame k+1 of step t-1 (oldest -> newest within a term), for envs that did not reset in between
    (prev, _), (cur, keep) = frames[-2], frames[-1]
    for name, d in zip(("velocity_commands", "base_ang_vel", "projected_gravity", "joint_pos", "joint_vel", "last_action", "object_state"), L["term_dims"]):
        s = sl[name]
        a, b = cur[:, s].reshape(n, 6, d), prev[:, s].reshape(n, 6, d)
        assert torch.equal(a[keep][:, :5], b[keep][:, 1:]), name
