#!/usr/bin/env python3
"""Golden vectors for the distillation path, produced by the REFERENCE's own classes on the
synthetic inputs of tests/distill_synth.py:

  D5  loco_rl.models (CNN2dHead, RNN, MLP via generate_model) + locotouch.distill.student.Student with the resolved
      DistillationRandCylinderCNNRNNMonCfg: parameter names / shapes / seeded-init checksums, inference-mode forward over
      consecutive steps (GRU state carried), batch-mode forward, the masked BC loss, per-parameter gradient norms, the loss
      after one AdamW step.
  D2  locotouch.distill.tactile_recorder.TactileRecorder: the delay line over a scripted sequence with resets.
  D3  locotouch.distill.replay_buffer.ReplayBuffer: which trajectories a collection keeps (order, lengths), the returned
      rewards / lengths, a padded batch.

Runs ONLY in the build container (imports /root/reference read-only; `loco_rl` here is the reference's own package, not this
repo's alias); writes data only: tests/golden/distill.npz.

    python tools/gen_golden_distill.py
"""
import os
import sys
import types

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from locotouch_amd.compat import runtime  # noqa: E402

runtime.install()  # isaaclab / gymnasium import surface, so that `import locotouch` (config classes) works
for name in [m for m in sys.modules if m == "loco_rl" or m.startswith("loco_rl.")]:
    del sys.modules[name]  # drop this repo's alias: the goldens must come from the reference's own loco_rl
sys.meta_path[:] = [f for f in sys.meta_path if "loco_rl" not in type(f).__name__.lower() and "LocoRl" not in type(f).__name__]
sys.modules.setdefault("git", types.ModuleType("git"))
sys.path.insert(0, "/root/reference/loco_rl")
sys.path.insert(0, "/root/reference")
import loco_rl  # noqa: E402

assert loco_rl.__file__.startswith("/root/reference/"), loco_rl.__file__
import locotouch  # noqa: E402,F401
from locotouch.config.locotouch.agents.distillation_cfg import DistillationRandCylinderCNNRNNMonCfg  # noqa: E402
from locotouch.distill.replay_buffer import ReplayBuffer  # noqa: E402
from locotouch.distill.student import Student  # noqa: E402
from locotouch.distill.tactile_recorder import TactileRecorder  # noqa: E402

from tests import distill_synth as S  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden", "distill.npz")


def gen_student(out):
    import contextlib
    import io
    import tempfile

    cfg = DistillationRandCylinderCNNRNNMonCfg()
    cfg.device = "cpu"
    cfg.log_dir = tempfile.mkdtemp()
    torch.manual_seed(1234)
    with contextlib.redirect_stdout(io.StringIO()):
        st = Student(cfg, S.PROPRIO, S.TACTILE, S.ACTIONS, teacher_policy_inference=S.teacher_policy())
    sd = st.state_dict()
    out["st_keys"] = np.array(list(sd.keys()))
    out["st_shapes"] = np.array([str(tuple(v.shape)) for v in sd.values()])
    out["st_sums"] = np.array([float(v.double().sum()) for v in sd.values()])
    out["st_abs_sums"] = np.array([float(v.double().abs().sum()) for v in sd.values()])
    out["st_num_params"] = np.array(sum(p.numel() for p in st.parameters()))
    steps, batch = S.student_inputs()
    st.eval()
    ys, hs = [], []
    with torch.no_grad():
        for i, s_ in enumerate(steps):
            ys.append(st(s_["prop"], s_["tac"]).clone())
            hs.append(st.get_hidden_states().clone())
            if i == 1:
                st.reset()  # all envs (dones=None): the path the reference's own code exercises correctly
    out["st_step_actions"], out["st_step_hidden_abs_sum"] = torch.stack(ys).numpy(), np.array([float(h.double().abs().sum()) for h in hs])
    st.train()
    with torch.no_grad():
        out["st_batch_actions"] = st(batch["proprioceptions"], batch["tactile_signals"]).numpy()
    # one training step exactly as student.py:119-147 does it (Monolithic)
    st._optimizer.zero_grad()
    sa = st.forward(batch["proprioceptions"], batch["tactile_signals"])
    ta = st.teacher_policy_inference(torch.cat((batch["proprioceptions"], batch["teacher_encoder_obses"]), dim=-1))
    loss = st._criterion(sa, ta).mean(dim=-1)
    loss = (loss * batch["masks"]).sum() / batch["masks"].sum()
    loss.backward()
    out["st_loss0"] = np.array(float(loss))
    out["st_grad_norms"] = np.array([float(p.grad.double().norm()) for p in st.parameters()])
    mae = (torch.abs(sa - ta).mean(dim=-1) * batch["masks"]).sum() / batch["masks"].sum() * st.action_scale_within_env
    out["st_mae0"] = np.array(float(mae))
    st._optimizer.step()
    with torch.no_grad():
        sa = st.forward(batch["proprioceptions"], batch["tactile_signals"])
        loss1 = ((st._criterion(sa, ta).mean(dim=-1)) * batch["masks"]).sum() / batch["masks"].sum()
    out["st_loss1"] = np.array(float(loss1))
    out["st_epoch_schedule"] = np.array([cfg.initial_epoches, cfg.incremental_epoches, cfg.final_epoches, cfg.num_iterations,
                                         cfg.bc_data_steps, cfg.dagger_data_steps, cfg.batch_steps, cfg.evaluation_trajs_num])
    out["st_misc"] = np.array([cfg.distill_lr, cfg.clip_range, cfg.action_scale_within_env, cfg.min_delay, cfg.max_delay])


def gen_recorder(out, n=6, T=12, d=5, seed=5):
    g = torch.Generator().manual_seed(seed)
    rec = TactileRecorder("cpu", n, d, min_delay=1, max_delay=4)
    sig = S.grid(g, (T, n, d))
    resets = {3: [1, 4], 7: [0], 8: [0, 5]}
    outs, delays = [], []
    for t in range(T):
        rec.record_new_tactile_signals(sig[t])
        outs.append(rec.get_tactile_signals().clone())
        delays.append(rec.delay_steps.clone())
        if t in resets:
            rec.reset(torch.tensor(resets[t]))
    out["rec_signals"], out["rec_out"], out["rec_delays"] = sig.numpy(), torch.stack(outs).numpy(), torch.stack(delays).numpy()
    out["rec_reset_steps"] = np.array(sorted(resets))
    out["rec_reset_mask"] = np.array([[1 if e in resets[t] else 0 for e in range(n)] for t in sorted(resets)])


def gen_replay(out, n=6, num_steps=60):
    env = S.ScriptedEnv(n, form="dict")
    rec = TactileRecorder("cpu", n, S.TACTILE, min_delay=1, max_delay=2)
    rb = ReplayBuffer(env, rec, S.PROPRIO)
    teacher = S.teacher_policy()
    import tqdm as _tqdm

    _tqdm.tqdm.__init__.__defaults__  # noqa: B018 (tqdm present)
    rewards, lengths = rb.collect_data(teacher_policy=teacher, student_policy=None, num_steps=num_steps)
    out["rb_rewards"], out["rb_lengths"] = np.array(rewards), np.array(lengths)
    out["rb_num_trajs"], out["rb_num_steps"] = np.array(rb.num_trajs), np.array(rb.num_steps)
    out["rb_traj_lengths"] = np.array([p.shape[0] for p in rb._proprioceptions])
    out["rb_traj_first_prop0"] = np.array([float(p[0, 0]) for p in rb._proprioceptions])  # identifies (env, episode, step)
    out["rb_env_steps"] = np.array(len(env.actions_seen))
    idx = [2, 0, rb.num_trajs - 1]
    b = rb._prepare_padded_sequence(np.array(idx))
    out["rb_batch_idx"] = np.array(idx)
    out["rb_batch_prop_sum"] = b["proprioceptions"].sum(dim=-1).numpy()
    out["rb_batch_enc_sum"] = b["teacher_encoder_obses"].sum(dim=-1).numpy()
    out["rb_batch_tac"] = b["tactile_signals"][..., :16].numpy()
    out["rb_batch_masks"] = b["masks"].numpy()
    # a second collection appends (the buffer keeps growing across DAgger iterations)
    rewards2, lengths2 = rb.collect_data(teacher_policy=teacher, student_policy=None, num_steps=30)
    out["rb2_lengths"], out["rb2_num_trajs"], out["rb2_num_steps"] = np.array(lengths2), np.array(rb.num_trajs), np.array(rb.num_steps)
    out["rb2_traj_lengths"] = np.array([p.shape[0] for p in rb._proprioceptions])


if __name__ == "__main__":
    torch.set_num_threads(1)
    out = {}
    gen_student(out)
    gen_recorder(out)
    gen_replay(out)
    np.savez_compressed(OUT, **out)
    print("distill.npz", {k: v.shape for k, v in out.items()}, os.path.getsize(OUT))
