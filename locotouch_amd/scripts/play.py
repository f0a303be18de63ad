#!/usr/bin/env python3
"""Play a trained policy: the reference's `locotouch/scripts/play.py` flow on the MI355X-native env (no renderer: the loop
steps the policy, prints episode statistics, and optionally exports the actor as TorchScript for the robot-side runtime).

    python -m locotouch_amd.scripts.play --task Isaac-RandCylinderTransportTeacher-LocoTouch-Play-v1 --num_envs 50 --steps 1000 --export
"""
from __future__ import annotations

import argparse
import os

import torch


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--task", default="Isaac-RandCylinderTransportTeacher-LocoTouch-Play-v1")
    ap.add_argument("--num_envs", type=int, default=None)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--headless", action="store_true")
    ap.add_argument("--video", action="store_true")
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--load_run", default=None)
    ap.add_argument("--checkpoint", default=None)
    ap.add_argument("--steps", type=int, default=None, help="stop after this many env steps (default: run until interrupted)")
    ap.add_argument("--export", action="store_true", help="write <run>/exported/policy.pt (TorchScript: normaliser -> actor)")
    args, _unknown = ap.parse_known_args()

    from locotouch_amd.agents import train_cfg
    from locotouch_amd.compat.runtime import export_policy_as_jit, get_checkpoint_path
    from locotouch_amd.env import make
    from locotouch_amd.rl import OnPolicyRunner

    agent = train_cfg(args.task)
    root = os.path.abspath(os.path.join("logs", "rsl_rl", agent["experiment_name"]))
    print(f"[INFO] Loading experiment from directory: {root}")
    resume = args.checkpoint if (args.checkpoint and os.path.isfile(args.checkpoint)) else get_checkpoint_path(
        root, args.load_run or ".*", args.checkpoint or "model_.*.pt")
    torch.cuda.set_device(args.device)
    env = make(args.task, num_envs=args.num_envs, device=args.device, seed=args.seed if args.seed is not None else agent["seed"])
    runner = OnPolicyRunner(env, agent, log_dir=None, device=args.device)
    print(f"[INFO]: Loading model checkpoint from: {resume}")
    runner.load(resume)
    policy = runner.get_inference_policy(device=args.device)
    if args.export:
        out = export_policy_as_jit(runner.alg.actor_critic, runner.obs_normalizer if runner.empirical_normalization else None,
                                   path=os.path.join(os.path.dirname(resume), "exported"), filename="policy.pt")
        print(f"[INFO]: Exported policy to: {out}")
    obs, _ = env.get_observations()
    t, finished, ret_sum = 0, 0, torch.zeros(env.num_envs, device=env.device)
    with torch.inference_mode():
        while args.steps is None or t < args.steps:
            obs, rew, dones, _ = env.step(policy(obs))
            ret_sum += rew
            t += 1
            if t % 200 == 0:
                log = env.episode_log()
                print(f"[play] step {t}: " + ", ".join(f"{k}={v:.3f}" for k, v in sorted(log.items()) if k.startswith(("Episode/", "Metrics/"))))


if __name__ == "__main__":
    main()
