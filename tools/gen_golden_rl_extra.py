#!/usr/bin/env python3
"""Golden vectors for the remaining `loco_rl` pieces of §8 a.7: P10 EmpiricalNormalization (modules/normalizer.py:14-76),
P4 recurrent minibatch generator + split_and_pad_trajectories / unpad_trajectories (storage/rollout_storage.py:246-318,
utils/utils.py:37-83).  Runs ONLY in the build container: imports the reference's own `loco_rl` (two in-memory stubs for
the absent `git` / `isaaclab` modules) on seeded inputs; writes data only: tests/golden/rl_extra.npz.

    python tools/gen_golden_rl_extra.py
"""
import os
import sys
import types

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference/loco_rl")
sys.modules.setdefault("git", types.ModuleType("git"))
il, ilu = types.ModuleType("isaaclab"), types.ModuleType("isaaclab.utils")
ilu.configclass = lambda c: c
il.utils = ilu
sys.modules.update({"isaaclab": il, "isaaclab.utils": ilu})

import numpy as np  # noqa: E402
import torch  # noqa: E402

from loco_rl.modules.normalizer import EmpiricalNormalization  # noqa: E402
from loco_rl.storage import RolloutStorage  # noqa: E402
from loco_rl.utils import split_and_pad_trajectories, unpad_trajectories  # noqa: E402

from tests.rl_synth import extra_inputs  # noqa: E402


def main():
    torch.set_num_threads(1)
    x = extra_inputs()
    out = {}
    # ---- normaliser: 6 batches in training mode (the 5th crosses `until`), then eval mode
    nz = EmpiricalNormalization(shape=[x["norm_batches"].shape[-1]], until=x["norm_until"])
    ys = []
    for b in x["norm_batches"]:
        ys.append(nz(b).clone())
    nz.eval()
    ys.append(nz(x["norm_batches"][0]).clone())
    out["norm_y"] = torch.stack(ys).numpy()
    out["norm_mean"], out["norm_std"], out["norm_count"] = nz.mean.numpy(), nz.std.numpy(), np.array(int(nz.count))
    out["norm_inverse"] = nz.inverse(ys[-1]).numpy()
    out["norm_keys"] = np.array(sorted(nz.state_dict().keys()))
    # ---- trajectories
    padded, masks = split_and_pad_trajectories(x["traj_tensor"], x["traj_dones"])
    out["traj_padded"], out["traj_masks"] = padded.numpy(), masks.numpy()
    out["traj_unpadded"] = unpad_trajectories(padded, masks).numpy()
    # ---- recurrent generator on a filled storage
    T, N, D, A, H = x["T"], x["N"], x["D"], x["A"], x["H"]
    st = RolloutStorage(N, T, [D], [D], [A], device="cpu")
    for t in range(T):
        tr = RolloutStorage.Transition()
        tr.observations, tr.critic_observations = x["obs"][t], x["cobs"][t]
        tr.actions, tr.rewards, tr.dones = x["actions"][t], x["rewards"][t], x["dones"][t]
        tr.values, tr.actions_log_prob = x["values"][t], x["logp"][t]
        tr.action_mean, tr.action_sigma = x["mu"][t], x["sigma"][t]
        tr.hidden_states = (x["hid_a"][t], x["hid_c"][t])
        st.add_transitions(tr)
    st.compute_returns(x["last_values"], 0.99, 0.95)
    batches = list(st.recurrent_mini_batch_generator(x["num_mini_batches"], num_epochs=1))
    out["rec_num"] = np.array(len(batches))
    for i, b in enumerate(batches):
        obs_b, cobs_b, act_b, val_b, adv_b, ret_b, lp_b, mu_b, sg_b, (ha, hc), mask_b, _ = b
        for name, v in (("obs", obs_b), ("cobs", cobs_b), ("act", act_b), ("val", val_b), ("adv", adv_b), ("ret", ret_b), ("lp", lp_b),
                        ("mu", mu_b), ("sg", sg_b), ("ha", ha), ("hc", hc), ("mask", mask_b)):
            out[f"rec{i}_{name}"] = v.numpy()
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "rl_extra.npz"), **out)
    print("rl_extra.npz", {k: v.shape for k, v in out.items() if not k.startswith("rec") or k.startswith("rec0")})


if __name__ == "__main__":
    main()
