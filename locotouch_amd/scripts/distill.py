#!/usr/bin/env python3
"""Distil a trained teacher into the tactile student: the reference's `locotouch/scripts/distill.py` flow on the MI355X-native env.

    python -m locotouch_amd.scripts.distill --task Isaac-RandCylinderTransportStudent_SingleBinaryTac_CNNRNN_Mon-LocoTouch-v1 --training --headless
    python -m locotouch_amd.scripts.distill --task Isaac-RandCylinderTransportStudent_SingleBinaryTac_CNNRNN_Mon-LocoTouch-Play-v1 \\
        --log_dir_distill 2025-... --checkpoint_distill model_7.pt

Flags follow the reference CLI (locotouch/scripts/cli_args.py:11-33,92-124, distill.py:8-20).  The teacher checkpoint is looked
up as the reference does: logs/rsl_rl/<teacher experiment>/<--load_run>/<--checkpoint> (latest matching).
"""
from __future__ import annotations

import argparse
import os

import torch


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--task", default="Isaac-RandCylinderTransportStudent_SingleBinaryTac_CNNRNN_Mon-LocoTouch-v1")
    ap.add_argument("--num_envs", type=int, default=None)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--headless", action="store_true")
    ap.add_argument("--video", action="store_true")
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--training", action="store_true", default=False)
    ap.add_argument("--load_run", default=None)
    ap.add_argument("--checkpoint", default=None)
    ap.add_argument("--log_dir_distill", default=None)
    ap.add_argument("--checkpoint_distill", default=None)
    ap.add_argument("--distill_lr", type=float, default=None)
    ap.add_argument("--logger", default=None, choices=["wandb", "tensorboard"])
    ap.add_argument("--play_steps", type=int, default=None, help="play mode: stop after this many env steps (default: run until interrupted)")
    args, _unknown = ap.parse_known_args()

    from locotouch_amd.agents import train_cfg
    from locotouch_amd.compat.runtime import get_checkpoint_path
    from locotouch_amd.distill import Distillation, distillation_cfg
    from locotouch_amd.env import make
    from locotouch_amd.rl import OnPolicyRunner

    cfg = distillation_cfg(args.task)
    cfg.device = args.device
    if args.logger is not None:
        cfg.logger = args.logger
    if args.distill_lr is not None:
        cfg.distill_lr = args.distill_lr
    torch.cuda.set_device(args.device)
    agent = train_cfg(args.task)
    env = make(args.task, num_envs=args.num_envs, device=args.device, seed=args.seed if args.seed is not None else agent["seed"])
    distill_root = os.path.abspath(os.path.join(cfg.log_root_path, cfg.experiment_name))
    if args.training:
        teacher_root = os.path.abspath(os.path.join("logs", "rsl_rl", agent["experiment_name"]))
        resume = get_checkpoint_path(teacher_root, args.load_run or ".*", args.checkpoint or "model_.*.pt")
        print(f"[INFO] Loading teacher policy checkpoint from: {resume}")
        runner = OnPolicyRunner(env, agent, log_dir=None, device=args.device)
        runner.load(resume)
        mono = cfg.distillation_type == "Monolithic"
        d = Distillation(env, cfg, teacher_policy=runner.get_inference_policy(device=args.device),
                         teacher_encoder=None if mono else runner.get_inference_encoder(device=args.device),
                         teacher_backbone_weights=None if mono else runner.get_backbone_weights(), training=True)
        d.train()
    else:
        ckpt = get_checkpoint_path(distill_root, args.log_dir_distill or ".*", args.checkpoint_distill or "model_.*.pt")
        d = Distillation(env, cfg, training=False, checkpoint=ckpt)
        d.play(num_steps=args.play_steps)


if __name__ == "__main__":
    main()
