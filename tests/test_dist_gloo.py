"""N>1 path on CPU: world_size-2 gloo.  Two ranks with half of the envs each must reproduce the single-process update
(flat gradient all-reduce, all-reduced KL for the adaptive LR, all-reduced advantage moments) - SURVEY.md §8(e)."""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from locotouch_amd.rl import PPO, ActorCritic, Dist
from tests.rl_synth import N_ACT, N_OBS, POLICY_CFG, PPO_CFG

N_TOTAL, T = 32, 8
CFG = dict(PPO_CFG, num_learning_epochs=2, num_mini_batches=1)


def _fill(alg, seed, sl):
    """Deterministic synthetic rollout for env slice `sl` of the N_TOTAL-env population."""
    g = torch.Generator().manual_seed(seed)
    obs = torch.randn(T, N_TOTAL, N_OBS, generator=g)
    cobs = obs + 0.01 * torch.randn(T, N_TOTAL, N_OBS, generator=g)
    eps = torch.randn(T, N_TOTAL, N_ACT, generator=g)
    rew = torch.randn(T, N_TOTAL, generator=g)
    dones = (torch.rand(T, N_TOTAL, generator=g) < 0.1).long()
    last = torch.randn(N_TOTAL, N_OBS, generator=g)
    ac = alg.actor_critic
    with torch.no_grad():
        for t in range(T):
            o, c = obs[t, sl], cobs[t, sl]
            ac.update_distribution(o)
            act = ac.action_mean + ac.action_std * eps[t, sl]
            alg._t = dict(actions=act, values=ac.evaluate(c), log_prob=ac.get_actions_log_prob(act), mu=ac.action_mean.clone(),
                          sigma=ac.action_std.clone(), obs=o, critic_obs=c)
            alg.process_env_step(rew[t, sl], dones[t, sl], {})
        alg.compute_returns(last[sl])


def _single():
    torch.manual_seed(0)
    alg = PPO(ActorCritic(N_OBS, N_OBS, N_ACT, **POLICY_CFG), device="cpu", **CFG)
    alg.init_storage(N_TOTAL, T, [N_OBS], [N_OBS], [N_ACT])
    _fill(alg, 9, slice(0, N_TOTAL))
    adv = alg.storage.advantages.clone()
    out = alg.update()
    return alg, adv, out


def _worker(rank, world, port, outdir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist = Dist.from_env(backend="gloo")
    torch.manual_seed(rank * 1000 + 5)  # different initial weights per rank: the broadcast must fix that
    ac = ActorCritic(N_OBS, N_OBS, N_ACT, **POLICY_CFG)
    if rank == 0:
        torch.manual_seed(0)
        ac = ActorCritic(N_OBS, N_OBS, N_ACT, **POLICY_CFG)
    alg = PPO(ac, device="cpu", dist=dist, **CFG)
    n = N_TOTAL // world
    alg.init_storage(n, T, [N_OBS], [N_OBS], [N_ACT])
    _fill(alg, 9, slice(rank * n, (rank + 1) * n))
    adv = alg.storage.advantages.clone()
    out = alg.update()
    flat = torch.cat([p.detach().flatten() for p in alg.actor_critic.parameters()])
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), params=flat.numpy(), adv=adv.numpy(), losses=np.array(out[:3]), lr=alg.learning_rate)
    dist.barrier()
    dist.shutdown()


@pytest.mark.parametrize("world", [2, 8])
def test_multi_rank_update_equals_single_process(world):
    """2 ranks, and the node's 8 (4 envs each): index offsets, advantage moments, KL rule and the gradient mean at the real rank count."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ref, ref_adv, ref_out = _single()
    ref_flat = torch.cat([p.detach().flatten() for p in ref.actor_critic.parameters()]).numpy()
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, port, d), nprocs=world, join=True)
        rs = [dict(np.load(os.path.join(d, f"rank{r}.npz"))) for r in range(world)]
    n = N_TOTAL // world
    for r, rec in enumerate(rs):
        np.testing.assert_array_equal(rec["params"], rs[0]["params"])  # replicas stay bit-identical
        np.testing.assert_allclose(rec["adv"], ref_adv[:, r * n:(r + 1) * n].numpy(), rtol=1e-4, atol=1e-5)  # global advantage normalisation
        assert float(rec["lr"]) == ref.learning_rate  # same adaptive-LR decisions on every rank
    # Adam divides by sqrt(v): where a gradient is ~0 a differently ordered 8-way sum moves the step by a visible fraction of lr, so
    # the tight band may be left by a handful of the 687 513 parameters (counted), never the loose one (2 steps x lr 1e-3)
    tight = np.isclose(rs[0]["params"], ref_flat, rtol=2e-4, atol=2e-5)
    assert (~tight).sum() <= 1e-5 * ref_flat.size, int((~tight).sum())
    np.testing.assert_allclose(rs[0]["params"], ref_flat, rtol=2e-4, atol=4e-4)
