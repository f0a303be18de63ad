#!/usr/bin/env python3
"""Golden vectors for the recurrent and the encoder policy classes of `loco_rl`: the reference's own
PPO + ActorCriticRecurrent (GRU) / ActorCriticEncoder on the seeded synthetic rollout of tests/rl_synth.py - parameter
names / shapes / seeded-init checksums, the rollout's per-step actions' log-probs and values (hidden state carried and reset),
the three losses of one update and the post-update parameter checksums.  Runs ONLY in the build container (imports the
reference's `loco_rl` with the two in-memory stubs of SURVEY.md Appendix E); writes data only: tests/golden/rl_policies.npz.

    python tools/gen_golden_rl_policies.py
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

import contextlib  # noqa: E402
import io  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

from loco_rl.algorithms import PPO  # noqa: E402
from loco_rl.modules import ActorCriticEncoder, ActorCriticRecurrent  # noqa: E402

from tests.rl_synth import policy_case  # noqa: E402


def run(kind, out):
    case = policy_case(kind)
    torch.manual_seed(case["seed"])
    with contextlib.redirect_stdout(io.StringIO()):
        ac = (ActorCriticRecurrent if kind == "recurrent" else ActorCriticEncoder)(*case["args"], **case["kwargs"])
    alg = PPO(ac, device="cpu", **case["ppo"])
    N, T, D, A = case["N"], case["T"], case["D"], case["A"]
    alg.init_storage(N, T, [D], [D], [A])
    sd = ac.state_dict()
    out[f"{kind}_keys"] = np.array(list(sd.keys()))
    out[f"{kind}_shapes"] = np.array([str(tuple(v.shape)) for v in sd.values()])
    out[f"{kind}_init_sums"] = np.array([float(v.double().sum()) for v in sd.values()])
    logp, vals = [], []
    torch.manual_seed(case["seed"] + 1)
    with torch.inference_mode():  # as the runner's rollout (on_policy_runner.py:154)
        for t in range(T):
            alg.act(case["obs"][t], case["cobs"][t])
            logp.append(alg.transition.actions_log_prob.clone())
            vals.append(alg.transition.values.clone())
            alg.process_env_step(case["rewards"][t], case["dones"][t], {})
        alg.compute_returns(case["last_cobs"])
    out[f"{kind}_actions"] = alg.storage.actions.numpy().copy()
    out[f"{kind}_logp"], out[f"{kind}_values"] = torch.stack(logp).numpy(), torch.stack(vals).numpy()
    out[f"{kind}_adv"] = alg.storage.advantages.numpy().copy()
    torch.manual_seed(case["seed"] + 2)
    losses = alg.update()
    out[f"{kind}_losses"] = np.array([float(losses[0]), float(losses[1]), float(losses[2])])
    out[f"{kind}_lr"] = np.array(alg.learning_rate)
    out[f"{kind}_post_sums"] = np.array([float(v.double().sum()) for v in ac.state_dict().values()])
    out[f"{kind}_post_abs"] = np.array([float(v.double().abs().sum()) for v in ac.state_dict().values()])


if __name__ == "__main__":
    torch.set_num_threads(1)
    out = {}
    for kind in ("recurrent", "encoder"):
        run(kind, out)
    path = os.path.join(REPO, "tests", "golden", "rl_policies.npz")
    np.savez_compressed(path, **out)
    print("rl_policies.npz", {k: v.shape for k, v in out.items()}, os.path.getsize(path))
