"""Child process of tests/test_hip_dist.py: one rank of a data-parallel PPO update on the GPU (fused path: packed MLP forward,
lt_ppo_loss, flat gradient bucket -> all-reduce -> lt_adam_clip_step).  WORLD_SIZE ranks share ONE card (gloo through the host,
LT_DIST_BACKEND=gloo: RCCL refuses several ranks per device); WORLD_SIZE=1 is the single process holding the whole population."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from locotouch_amd.rl import PPO, ActorCritic, Dist  # noqa: E402
from tests.rl_synth import N_ACT, N_OBS, POLICY_CFG, PPO_CFG  # noqa: E402

N_TOTAL, T = 512, 24


def main(out):
    dist = Dist.from_env()
    dev = torch.device("cuda:0")
    torch.manual_seed(1000 * dist.rank + 5)  # different initial weights per rank: the broadcast must fix that
    ac = ActorCritic(N_OBS, N_OBS, N_ACT, **POLICY_CFG)
    if dist.rank == 0:
        torch.manual_seed(0)
        ac = ActorCritic(N_OBS, N_OBS, N_ACT, **POLICY_CFG)
    alg = PPO(ac, device=dev, dist=dist, **dict(PPO_CFG, num_learning_epochs=2, num_mini_batches=int(os.environ.get("LT_TEST_MINIBATCHES", "2"))))
    assert alg._flat_adam is not None and alg.fused_loss and alg.packed_forward, "the fused update path must be the one under test"
    n = N_TOTAL // dist.world_size
    sl = slice(dist.rank * n, (dist.rank + 1) * n)
    alg.init_storage(n, T, [N_OBS], [N_OBS], [N_ACT])
    g = torch.Generator().manual_seed(9)
    obs = torch.randn(T, N_TOTAL, N_OBS, generator=g)
    cobs = obs + 0.01 * torch.randn(T, N_TOTAL, N_OBS, generator=g)
    eps = torch.randn(T, N_TOTAL, N_ACT, generator=g)
    rew = torch.randn(T, N_TOTAL, generator=g)
    dones = (torch.rand(T, N_TOTAL, generator=g) < 0.1).long()
    last = torch.randn(N_TOTAL, N_OBS, generator=g)
    a = alg.actor_critic
    with torch.no_grad():
        for t in range(T):
            o, c = obs[t, sl].to(dev), cobs[t, sl].to(dev)
            a.update_distribution(o)
            act = a.action_mean + a.action_std * eps[t, sl].to(dev)
            alg._t = dict(actions=act, values=a.evaluate(c), log_prob=a.get_actions_log_prob(act), mu=a.action_mean.clone(),
                          sigma=a.action_std.clone(), obs=o, critic_obs=c)
            alg.process_env_step(rew[t, sl].to(dev), dones[t, sl].to(dev), {})
        alg.compute_returns(last[sl].to(dev))
    # (with one minibatch per epoch the two shards' mean gradient IS the single process's full-batch gradient; with more, each
    #  rank permutes its own shard and the partitions differ - the replicas must agree with each other bit for bit in any case)
    torch.manual_seed(77)
    losses = alg.update()
    flat = torch.cat([p.detach().flatten() for p in alg.actor_critic.parameters()]).cpu().numpy()
    np.savez(os.path.join(out, f"rank{dist.rank}of{dist.world_size}_mb{os.environ.get('LT_TEST_MINIBATCHES', '2')}.npz"), params=flat, losses=np.array(losses[:3]), lr=alg.learning_rate)
    dist.barrier()
    dist.shutdown()


if __name__ == "__main__":
    main(sys.argv[1])
