"""Distillation path against tests/golden/distill.npz, which tools/gen_golden_distill.py recorded
from the REFERENCE's own locotouch/distill + loco_rl.models classes on the synthetic inputs of tests/distill_synth.py.
CPU only (the networks are torch modules; on the GPU box they run on MIOpen / hipBLASLt)."""
import os

import numpy as np
import pytest
import torch

from locotouch_amd.distill import ReplayBuffer, Student, TactileRecorder, distillation_cfg
from tests import distill_synth as S

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "distill.npz")
TASK = "Isaac-RandCylinderTransportStudent_SingleBinaryTac_CNNRNN_Mon-LocoTouch-v1"


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def make_student(tmp):
    cfg = distillation_cfg(TASK)
    cfg.device, cfg.log_dir = "cpu", str(tmp)
    torch.manual_seed(1234)
    return cfg, Student(cfg, S.PROPRIO, S.TACTILE, S.ACTIONS, teacher_policy_inference=S.teacher_policy(), verbose=False)


def test_distillation_cfg_matches_reference_values(gold):
    c = distillation_cfg(TASK)
    assert [c.initial_epoches, c.incremental_epoches, c.final_epoches, c.num_iterations, c.bc_data_steps, c.dagger_data_steps,
            c.batch_steps, c.evaluation_trajs_num] == gold["st_epoch_schedule"].tolist()
    np.testing.assert_allclose([c.distill_lr, c.clip_range, c.action_scale_within_env, c.min_delay, c.max_delay], gold["st_misc"])
    assert c.distillation_type == "Monolithic" and c.pre_encoder.model_type == "CNN2dHead" and c.tactile_encoder.model_type == "RNN"


def test_student_parameters_init_forward_loss_and_update_match_reference(gold, tmp_path):
    """Same parameter names / shapes (checkpoints interchange), same seeded initial weights (construction order), same
    inference-mode and batch-mode forward, same masked BC loss, gradients, and AdamW step."""
    torch.set_num_threads(1)
    cfg, st = make_student(tmp_path)
    sd = st.state_dict()
    assert list(sd.keys()) == gold["st_keys"].tolist()
    assert [str(tuple(v.shape)) for v in sd.values()] == gold["st_shapes"].tolist()
    assert sum(p.numel() for p in st.parameters()) == int(gold["st_num_params"]) == 1422420
    np.testing.assert_allclose([float(v.double().sum()) for v in sd.values()], gold["st_sums"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose([float(v.double().abs().sum()) for v in sd.values()], gold["st_abs_sums"], rtol=1e-9)
    steps, batch = S.student_inputs()
    st.eval()
    with torch.no_grad():
        for i, s_ in enumerate(steps):
            y = st(s_["prop"], s_["tac"])
            np.testing.assert_allclose(y.numpy(), gold["st_step_actions"][i], rtol=1e-5, atol=2e-6)
            assert abs(float(st.get_hidden_states().double().abs().sum()) - gold["st_step_hidden_abs_sum"][i]) < 1e-3
            if i == 1:
                st.reset()
    st.train()
    with torch.no_grad():
        np.testing.assert_allclose(st(batch["proprioceptions"], batch["tactile_signals"]).numpy(), gold["st_batch_actions"], rtol=1e-5, atol=2e-6)
    st._optimizer.zero_grad()
    loss, mse, mae = st.batch_loss(batch)
    loss.backward()
    assert mse is None and abs(float(loss) - float(gold["st_loss0"])) < 1e-6 and abs(float(mae) - float(gold["st_mae0"])) < 1e-6
    np.testing.assert_allclose([float(p.grad.double().norm()) for p in st.parameters()], gold["st_grad_norms"], rtol=2e-4, atol=1e-9)
    st._optimizer.step()
    with torch.no_grad():
        loss1, _, _ = st.batch_loss(batch)
    assert abs(float(loss1) - float(gold["st_loss1"])) < 1e-6 and float(loss1) < float(loss)
    # checkpoint round trip with the safe loader
    st.save_model(3)
    _, st2 = make_student(tmp_path)
    st2.load_checkpoint(os.path.join(str(tmp_path), "model_3.pt"))
    assert all(torch.equal(a, b) for a, b in zip(st.state_dict().values(), st2.state_dict().values()))


def test_memory_reset_zeroes_exactly_the_finished_envs(tmp_path):
    """Deliberate difference from the reference (quirk Q3: `state[..., dones, :] = 0` with a 0/1 LONG tensor zeroes envs 0 and 1
    whenever anything finished): a mask reset touches the finished envs only."""
    _, st = make_student(tmp_path)
    steps, _ = S.student_inputs()
    with torch.no_grad():
        st(steps[0]["prop"], steps[0]["tac"])
        h0 = st.get_hidden_states().clone()
        dones = torch.tensor([0, 0, 1, 0, 1])
        st.reset(dones)
    h1 = st.get_hidden_states()
    assert (h1[:, [2, 4]] == 0).all() and torch.equal(h1[:, [0, 1, 3]], h0[:, [0, 1, 3]]) and (h0[:, [0, 1]] != 0).any()


def test_tactile_recorder_matches_reference(gold):
    sig, n = torch.from_numpy(gold["rec_signals"]), gold["rec_signals"].shape[1]
    rec = TactileRecorder("cpu", n, sig.shape[-1], min_delay=1, max_delay=4)
    resets = dict(zip(gold["rec_reset_steps"].tolist(), gold["rec_reset_mask"]))
    delays = torch.from_numpy(gold["rec_delays"])
    for t in range(sig.shape[0]):
        rec.delay_steps = delays[t].clone()  # the delay draws are the reference's (its RNG consumption is not part of the contract)
        rec.record_new_tactile_signals(sig[t])
        np.testing.assert_array_equal(rec.get_tactile_signals().numpy(), gold["rec_out"][t])
        if t in resets:
            before = rec.delay_steps.clone()
            m = torch.from_numpy(resets[t]).bool()
            rec.reset(m.nonzero().flatten() if t % 2 else m)  # index form (reference call form) and mask form
            assert torch.equal(rec.delay_steps[~m], before[~m]) and ((rec.delay_steps[m] >= 1) & (rec.delay_steps[m] < 4)).all()
            assert (rec.tactile_buffer[m] == 0).all() and rec.first_signal_recorded[m].all() and not rec.first_signal_recorded[~m].any()


@pytest.mark.parametrize("check_every", [1, 4, 16])
def test_replay_buffer_keeps_the_reference_trajectories(gold, check_every):
    """The kept trajectories (order, lengths, first rows), returned rewards / lengths and a padded batch equal the reference's
    per-step bookkeeping, whatever the host-check period (the loop overshoots and discards)."""
    env = S.ScriptedEnv(6, form="tuple")
    rec = TactileRecorder("cpu", 6, S.TACTILE, min_delay=1, max_delay=2)
    rb = ReplayBuffer(env, rec, S.PROPRIO, check_every=check_every)
    teacher = S.teacher_policy()
    rewards, lengths = rb.collect_data(teacher_policy=teacher, student_policy=None, num_steps=60)
    np.testing.assert_allclose(rewards, gold["rb_rewards"], rtol=1e-6)
    assert lengths == gold["rb_lengths"].tolist()
    assert rb.num_trajs == int(gold["rb_num_trajs"]) and rb.num_steps == int(gold["rb_num_steps"])
    assert rb._traj_len == gold["rb_traj_lengths"].tolist()
    (policy, _), (first, _) = rb._materialise()
    np.testing.assert_allclose(policy[first, 0].numpy(), gold["rb_traj_first_prop0"], rtol=0, atol=0)
    assert int(gold["rb_env_steps"]) <= len(env.actions_seen) < int(gold["rb_env_steps"]) + check_every
    b = rb._prepare_padded_sequence(gold["rb_batch_idx"])
    np.testing.assert_array_equal(b["masks"].numpy(), gold["rb_batch_masks"])
    np.testing.assert_allclose(b["proprioceptions"].sum(dim=-1).numpy(), gold["rb_batch_prop_sum"], rtol=1e-6)
    np.testing.assert_allclose(b["teacher_encoder_obses"].sum(dim=-1).numpy(), gold["rb_batch_enc_sum"], rtol=1e-6)
    np.testing.assert_array_equal(b["tactile_signals"][..., :16].numpy(), gold["rb_batch_tac"])
    assert b["proprioceptions"].shape[-1] == S.PROPRIO and b["teacher_encoder_obses"].shape[-1] == S.OBJ
    if check_every == 1:  # a second collection appends; with the reference's stopping step the env is in the reference's state
        _, lengths2 = rb.collect_data(teacher_policy=teacher, student_policy=None, num_steps=30)
        assert lengths2 == gold["rb2_lengths"].tolist() and rb.num_trajs == int(gold["rb2_num_trajs"]) and rb.num_steps == int(gold["rb2_num_steps"])
        assert rb._traj_len == gold["rb2_traj_lengths"].tolist()
        seen = sum(int(b_["masks"][0].sum()) for b_ in rb.to_recurrent_generator(batch_size=4))  # every trajectory exactly once
        assert seen == rb.num_trajs
    # static shapes: same masked content, padded to (longest trajectory, batch size)
    torch.manual_seed(0)
    import numpy as _np
    _np.random.seed(3)
    tight = list(rb.to_recurrent_generator(batch_size=4, static_shapes=False))
    _np.random.seed(3)
    fixed = list(rb.to_recurrent_generator(batch_size=4, static_shapes=True))
    Lmax = max(rb._traj_len)
    for a, b_ in zip(tight, fixed):
        assert b_["masks"].shape == (Lmax, 4) and int(b_["masks"].sum()) == int(a["masks"].sum())
        l, w = a["masks"].shape
        assert torch.equal(b_["masks"][:l, :w], a["masks"]) and not b_["masks"][l:].any() and not b_["masks"][:, w:].any()
        assert torch.equal(b_["tactile_signals"][:l, :w], a["tactile_signals"]) and (b_["proprioceptions"][l:] == 0).all()
    rb.clear_buffer()
    assert rb.num_trajs == 0 and rb.num_steps == 0


def test_dagger_loop_end_to_end_on_scripted_env(tmp_path):
    """Distillation.train(): BC collection with the teacher, DAgger collections with the student, checkpoints per iteration,
    final evaluation; the loss falls."""
    from locotouch_amd.distill import Distillation

    cfg = distillation_cfg(TASK)
    cfg.logger, cfg.log_root_path = "tensorboard", str(tmp_path)
    cfg.num_iterations, cfg.bc_data_steps, cfg.dagger_data_steps = 2, 80, 40
    cfg.initial_epoches, cfg.incremental_epoches, cfg.batch_steps, cfg.evaluation_trajs_num = 6, 2, 60, 5
    torch.manual_seed(0)
    env = S.ScriptedEnv(6, form="tuple")
    d = Distillation(env, cfg, teacher_policy=S.teacher_policy(), verbose=False)
    assert d.proprioception_dim == 270 and d.tactile_signal_dim == 442
    hist = d.train()
    assert [h["iter"] for h in hist] == [0, 1, "eval"] and hist[2]["collect/trj_num"] >= 5
    assert sorted(f for f in os.listdir(cfg.log_dir) if f.endswith(".pt")) == ["model_0.pt", "model_1.pt"]
    from locotouch_amd.rl.tb_writer import read_events
    import glob

    tags = {t for _, t, _ in read_events(glob.glob(os.path.join(cfg.log_dir, "events.out.tfevents.*"))[0])}
    assert {"train/Action MSE", "train/Action MAE", "collect/trj_num", "collect/trj_len_mean"} <= tags
    # play mode from the saved checkpoint
    cfg2 = distillation_cfg(TASK)
    p = Distillation(S.ScriptedEnv(6, form="tuple"), cfg2, training=False, checkpoint=os.path.join(cfg.log_dir, "model_1.pt"), verbose=False)
    a = p.play(num_steps=5)
    assert a.shape == (6, 12) and torch.isfinite(a).all()
