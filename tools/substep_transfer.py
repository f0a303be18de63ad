#!/usr/bin/env python3
"""Integrator-refinement transfer (VERDICT r03 #6c) and leg-leg interpenetration census (#6d) of a freshly trained teacher.

    python tools/substep_transfer.py [--iters 800] [--envs 4096] [--out gpurun_out/r04_substep_transfer.json]

1. trains the teacher (Isaac-RandCylinderTransportTeacher-LocoTouch-v1, the reference's agent cfg) at phys_substeps = 1;
2. evaluates its deterministic policy for one whole episode (1000 steps) on fresh envs at phys_substeps = 1, 2 and 4 (the integrator
   step 5 / 2.5 / 1.25 ms, everything else equal): a policy that lives on integrator artefacts loses its return when the step is
   refined - the claim tested is that it keeps >= 90 % of return and episode length;
3. during the phys_substeps = 1 evaluation, measures how often links of DIFFERENT legs interpenetrate (the engine has no
   self-collision; the reference enables it, assets/go1.py:26): thigh and calf as capsules between their joint origins
   (radii 0.02 / 0.012 m), foot spheres (0.02 m), by forward kinematics over the exported state."""
import argparse, json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch

TASK = "Isaac-RandCylinderTransportTeacher-LocoTouch-v1"


def seg_seg_dist(p1, q1, p2, q2):
    """Distance between segments [p1, q1] and [p2, q2] (batched, (..., 3)); clamped closest-point parameters."""
    d1, d2, r = q1 - p1, q2 - p2, p1 - p2
    a, e, f = (d1 * d1).sum(-1), (d2 * d2).sum(-1), (d2 * r).sum(-1)
    b, c = (d1 * d2).sum(-1), (d1 * r).sum(-1)
    den = (a * e - b * b).clamp(min=1e-12)
    s = ((b * f - c * e) / den).clamp(0, 1)
    t = ((b * s + f) / e.clamp(min=1e-12)).clamp(0, 1)
    s = ((b * t - c) / a.clamp(min=1e-12)).clamp(0, 1)
    return ((p1 + d1 * s.unsqueeze(-1)) - (p2 + d2 * t.unsqueeze(-1))).norm(dim=-1)


def evaluate(policy, substeps, n, seed, census=False):
    from locotouch_amd.compat.scene_views import link_kinematics
    from locotouch_amd.env import make
    env = make(TASK, num_envs=n, device="cuda:0", seed=seed, phys_substeps=substeps)
    obs, _ = env.get_observations()
    ret = torch.zeros(n, device="cuda:0"); length = torch.zeros(n, device="cuda:0"); alive = torch.ones(n, dtype=torch.bool, device="cuda:0")
    hits = torch.zeros((), device="cuda:0"); samples = 0; deepest = torch.full((), 1.0, device="cuda:0")
    pairs = [(0, 1), (2, 3), (0, 2), (1, 3), (0, 3), (1, 2)]  # FR-FL, RR-RL, FR-RR, FL-RL, diagonals
    with torch.inference_mode():
        for t in range(env.max_episode_length):
            obs, rew, dones, _ = env.step(policy(obs))
            ret += rew * alive; length += alive.float()
            alive &= dones == 0
            if census and t % 4 == 0:
                f = env.field
                pos, _, _, _ = link_kinematics(f("LT_F_ROOT_POS")[:, 0, :3], f("LT_F_ROOT_QUAT")[:, 0, :4], f("LT_F_ROOT_LIN_VEL_W")[:, 0, :3],
                                               f("LT_F_ROOT_ANG_VEL_W")[:, 0, :3], f("LT_F_JOINT_POS").reshape(n, 12), f("LT_F_JOINT_VEL").reshape(n, 12))
                thigh, calf, foot = pos[:, 5:9], pos[:, 9:13], pos[:, 13:17]  # [n, leg, 3]: thigh origin, calf origin (knee), foot centre
                worst = torch.full((n,), 1.0, device="cuda:0")
                for a, b in pairs:
                    for (pa, qa, ra) in ((thigh[:, a], calf[:, a], 0.02), (calf[:, a], foot[:, a], 0.016)):
                        for (pb, qb, rb) in ((thigh[:, b], calf[:, b], 0.02), (calf[:, b], foot[:, b], 0.016)):
                            worst = torch.minimum(worst, seg_seg_dist(pa, qa, pb, qb) - (ra + rb))
                live = dones == 0
                hits += ((worst < 0) & live).sum(); samples += int(live.sum()); deepest = torch.minimum(deepest, worst[live].min() if bool(live.any()) else deepest)
    out = {"phys_substeps": substeps, "mean_return": float(ret.mean()), "mean_episode_length": float(length.mean()),
           "finished_early_frac": float((length < env.max_episode_length).float().mean())}
    if census:
        out["interpenetration"] = {"env_step_samples": samples, "frac_with_leg_leg_overlap": float(hits) / max(samples, 1), "smallest_leg_leg_clearance_m": float(deepest)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=800)
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--eval-envs", type=int, default=2048)
    ap.add_argument("--out", default=os.path.join(REPO, "gpurun_out", "r04_substep_transfer.json"))
    args = ap.parse_args()
    from locotouch_amd.agents import train_cfg
    from locotouch_amd.env import make
    from locotouch_amd.rl import OnPolicyRunner
    cfg = train_cfg(TASK)
    torch.manual_seed(cfg["seed"])
    env = make(TASK, num_envs=args.envs, device="cuda:0", seed=cfg["seed"])
    import tempfile

    log_dir = tempfile.mkdtemp(prefix="lt_transfer_")  # (checkpoints every 50 iterations: kept out of gpurun_out/)
    runner = OnPolicyRunner(env, cfg, log_dir=log_dir, device="cuda:0")
    t0 = time.time()
    runner.learn(args.iters, init_at_random_ep_len=True)
    train_s = time.time() - t0
    last = runner.history[-1] if runner.history else {}
    policy = runner.get_inference_policy(device="cuda:0")
    res = {"task": TASK, "train": {"iterations": args.iters, "envs": args.envs, "seconds": train_s,
                                   "final": {k: last.get(k) for k in ("Train/mean_reward", "Train/mean_episode_length", "Perf/total_fps", "Loss/learning_rate")}},
           "eval": [evaluate(policy, s, args.eval_envs, 1000 + s, census=(s == 1)) for s in (1, 2, 4)]}
    base = res["eval"][0]
    for e in res["eval"]:
        e["return_vs_substeps_1"] = e["mean_return"] / base["mean_return"]
        e["length_vs_substeps_1"] = e["mean_episode_length"] / base["mean_episode_length"]
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(res, open(args.out, "w"), indent=1)
    print(json.dumps(res, indent=1))
    if os.path.exists(os.path.join(log_dir, "progress.jsonl")):
        import shutil
        shutil.copy(os.path.join(log_dir, "progress.jsonl"), os.path.join(os.path.dirname(args.out), "r04_train_teacher_4096_progress.jsonl"))


if __name__ == "__main__":
    main()
