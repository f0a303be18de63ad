#!/usr/bin/env python3
"""Train a teacher policy: the reference's `locotouch/scripts/train.py` flow on the MI355X-native env.

    python -m locotouch_amd.scripts.train --task Isaac-RandCylinderTransportTeacher-LocoTouch-v1 --num_envs 4096 --headless
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m locotouch_amd.scripts.train --task ... (one rank per GPU)

Flags follow the reference CLI (locotouch/scripts/cli_args.py:11-33, train.py:17-29); `--headless` / `--video` are accepted
and ignored (there is no renderer).  Logs: logs/rsl_rl/<experiment>/<timestamp>/{progress.jsonl, model_<it>.pt, params/{env,agent}.{yaml,pkl}}.
"""
from __future__ import annotations

import argparse
import datetime
import os

import torch


def dump_params(log_dir: str, env_cfg: dict, agent_cfg: dict) -> None:
    """params/env.yaml, agent.yaml, env.pkl, agent.pkl (reference locotouch/scripts/train.py:150-153, isaaclab.utils.io.dump_yaml /
    dump_pickle [DEP]).  The env config is the resolved lt_cfg the kernels run on, as a plain dict (the reference pickles its
    cfg object; a dict loads without this package)."""
    import pickle

    import yaml

    d = os.path.join(log_dir, "params")
    os.makedirs(d, exist_ok=True)
    for name, obj in (("env", env_cfg), ("agent", agent_cfg)):
        with open(os.path.join(d, name + ".yaml"), "w") as f:
            yaml.safe_dump(obj, f)
        with open(os.path.join(d, name + ".pkl"), "wb") as f:
            pickle.dump(obj, f)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--task", default="Isaac-RandCylinderTransportTeacher-LocoTouch-v1")
    ap.add_argument("--num_envs", type=int, default=4096)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--max_iterations", type=int, default=None)
    ap.add_argument("--headless", action="store_true")
    ap.add_argument("--video", action="store_true")
    ap.add_argument("--resume", action="store_true")
    ap.add_argument("--load_run", default=None)
    ap.add_argument("--checkpoint", default=None)
    ap.add_argument("--logger", default="tensorboard")
    ap.add_argument("--log_root", default="logs/rsl_rl")
    args, _unknown = ap.parse_known_args()  # hydra-style overrides of the reference CLI are tolerated and ignored

    from locotouch_amd.agents import train_cfg
    from locotouch_amd.env import make
    from locotouch_amd.rl import Dist, OnPolicyRunner

    dist = Dist.from_env()
    cfg = train_cfg(args.task)
    cfg["logger"] = args.logger
    if args.seed is not None:
        cfg["seed"] = args.seed
    if args.max_iterations is not None:
        cfg["max_iterations"] = args.max_iterations
    device = f"cuda:{dist.local_rank}"
    torch.cuda.set_device(device)
    torch.manual_seed(cfg["seed"])
    # one population over all ranks: same seed, RNG streams keyed by the GLOBAL env index (rank r owns envs [r*N, (r+1)*N)),
    # and - with more than one rank - the curriculum gate decided on cross-rank sums
    env = make(args.task, num_envs=args.num_envs, device=device, seed=cfg["seed"], env_index_offset=dist.rank * args.num_envs,
               cur_gate_external=1 if dist.world_size > 1 else 0)
    log_dir = os.path.join(args.log_root, cfg["experiment_name"], datetime.datetime.now().strftime("%Y-%m-%d_%H-%M-%S"))
    runner = OnPolicyRunner(env, cfg, log_dir=log_dir, device=device, dist=dist)
    if args.resume:  # train.py:131-137 of the reference: newest matching run / checkpoint under the experiment's log root
        from locotouch_amd.compat.runtime import get_checkpoint_path

        ckpt = args.checkpoint if (args.checkpoint and os.path.isfile(args.checkpoint)) else get_checkpoint_path(
            os.path.join(args.log_root, cfg["experiment_name"]), args.load_run or ".*", args.checkpoint or "model_.*.pt")
        print(f"[INFO]: Loading model checkpoint from: {ckpt}")
        runner.load(ckpt)
    if dist.is_main:  # train.py:150-153 of the reference: params/{env,agent}.{yaml,pkl}
        dump_params(log_dir, dict(env.cfg.to_dict(), gym_id=args.task), cfg)
    runner.learn(cfg["max_iterations"], init_at_random_ep_len=True)
    if dist.is_main and runner.history:
        last = runner.history[-1]
        print({k: last[k] for k in ("iter", "Perf/total_fps", "Loss/value_function", "Loss/surrogate", "Loss/learning_rate")})
    dist.shutdown()


if __name__ == "__main__":
    main()
