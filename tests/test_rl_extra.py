"""Remaining `loco_rl` pieces (§8 a.7 P4, P10; the (f).2 log format) against what the reference's own code produced on the
same seeded inputs (tests/golden/rl_extra.npz, tools/gen_golden_rl_extra.py).  CPU torch."""
import glob
import os

import numpy as np
import torch

from locotouch_amd.rl import EmpiricalNormalization, RolloutStorage, split_and_pad_trajectories, unpad_trajectories
from tests.rl_synth import extra_inputs

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rl_extra.npz")


def test_empirical_normalization_matches_reference():
    g, x = np.load(GOLD), extra_inputs()
    nz = EmpiricalNormalization(shape=[x["norm_batches"].shape[-1]], until=x["norm_until"])
    ys = [nz(b).clone() for b in x["norm_batches"]]  # training mode: statistics update until `until` samples were seen
    nz.eval()
    ys.append(nz(x["norm_batches"][0]).clone())
    np.testing.assert_allclose(torch.stack(ys).numpy(), g["norm_y"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(nz.mean.numpy(), g["norm_mean"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(nz.std.numpy(), g["norm_std"], rtol=1e-6, atol=1e-7)
    assert int(nz.count) == int(g["norm_count"]) and int(nz.count) >= x["norm_until"]  # stopped learning past `until`
    np.testing.assert_allclose(nz.inverse(ys[-1]).numpy(), g["norm_inverse"], rtol=1e-5, atol=1e-5)
    assert sorted(nz.state_dict().keys()) == list(g["norm_keys"])  # obs_norm_state_dict interchange (on_policy_runner.py:376-379)


def test_trajectory_split_pad_and_unpad_match_reference():
    g, x = np.load(GOLD), extra_inputs()
    padded, masks = split_and_pad_trajectories(x["traj_tensor"], x["traj_dones"])
    np.testing.assert_array_equal(masks.numpy(), g["traj_masks"])
    np.testing.assert_array_equal(padded.numpy(), g["traj_padded"])
    back = unpad_trajectories(padded, masks)
    np.testing.assert_array_equal(back.numpy(), g["traj_unpadded"])
    assert torch.equal(back, x["traj_tensor"])
    assert int(masks.sum(0).min()) == 1  # the back-to-back dones give a length-1 trajectory


def test_recurrent_mini_batch_generator_matches_reference():
    g, x = np.load(GOLD), extra_inputs()
    T, N = x["T"], x["N"]
    st = RolloutStorage(N, T, x["D"], x["D"], x["A"], device="cpu")
    for t in range(T):
        st.add(x["obs"][t], x["cobs"][t], x["actions"][t], x["rewards"][t], x["dones"][t], x["values"][t], x["logp"][t], x["mu"][t], x["sigma"][t])
    st.compute_returns(x["last_values"], 0.99, 0.95)
    batches = list(st.recurrent_mini_batches(x["num_mini_batches"], num_epochs=1, hidden_states_a=[x["hid_a"]], hidden_states_c=[x["hid_c"]]))
    assert len(batches) == int(g["rec_num"])
    names = ("obs", "cobs", "act", "val", "adv", "ret", "lp", "mu", "sg")
    for i, b in enumerate(batches):
        for name, v in zip(names, b[:9]):
            np.testing.assert_allclose(v.numpy(), g[f"rec{i}_{name}"], rtol=1e-5, atol=1e-6, err_msg=f"batch {i} {name}")
        (ha, hc), mask = b[9], b[10]
        np.testing.assert_array_equal(ha.numpy(), g[f"rec{i}_ha"])
        np.testing.assert_array_equal(hc.numpy(), g[f"rec{i}_hc"])
        np.testing.assert_array_equal(mask.numpy(), g[f"rec{i}_mask"])


def test_event_file_writer_round_trip(tmp_path):
    """The scalar tags of the reference's runner, written in TensorBoard's event-file format and read back (CRCs checked)."""
    from locotouch_amd.rl.tb_writer import EventFileWriter, crc32c, read_events

    assert crc32c(b"123456789") == 0xE3069283  # CRC-32C check value
    w = EventFileWriter(str(tmp_path))
    tags = ["Loss/value_function", "Loss/surrogate", "Loss/entropy", "Loss/learning_rate", "Policy/mean_noise_std", "Perf/total_fps",
            "Perf/collection time", "Perf/learning_time", "Train/mean_reward", "Train/mean_episode_length", "Episode_Reward/alive"]
    for it in range(3):
        for k, tag in enumerate(tags):
            w.add_scalar(tag, 0.5 * it + k, it)
    w.close()
    files = glob.glob(os.path.join(str(tmp_path), "events.out.tfevents.*"))
    assert len(files) == 1
    ev = read_events(files[0])
    assert len(ev) == 3 * len(tags)
    assert ev[0] == (0, "Loss/value_function", 0.0) and ev[-1] == (2, "Episode_Reward/alive", 1.0 + len(tags) - 1)
    assert {t for _, t, _ in ev} == set(tags)
