#!/usr/bin/env python3
"""End-to-end evidence for BASELINE.json configs[3] on one MI355X: train a rand-cylinder teacher with PPO on the HIP env, distil
it into the tactile student (CNN head -> GRU -> MLP) with a shortened DAgger schedule on the 405-env student registration, and
evaluate both.  Writes JSON lines (progress + summary) to --out.

    python tools/run_distill_demo.py --teacher_iters 400 --out gpurun_out/r02d/distill_demo.jsonl
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--teacher_iters", type=int, default=400)
    ap.add_argument("--teacher_envs", type=int, default=4096)
    ap.add_argument("--num_envs", type=int, default=405)
    ap.add_argument("--iterations", type=int, default=3)
    ap.add_argument("--bc_steps", type=int, default=100000)
    ap.add_argument("--dagger_steps", type=int, default=50000)
    ap.add_argument("--epochs", type=int, default=60)
    ap.add_argument("--inc_epochs", type=int, default=20)
    ap.add_argument("--eval_trajs", type=int, default=400)
    ap.add_argument("--out", default="gpurun_out/distill_demo.jsonl")
    args = ap.parse_args()
    from locotouch_amd.agents import train_cfg
    from locotouch_amd.distill import Distillation, distillation_cfg
    from locotouch_amd.env import make
    from locotouch_amd.rl import OnPolicyRunner

    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    fout = open(args.out, "w")

    def emit(rec):
        fout.write(json.dumps(rec) + "\n")
        fout.flush()
        print(json.dumps(rec), flush=True)

    teacher_task = "Isaac-RandCylinderTransportTeacher-LocoTouch-v1"
    student_task = "Isaac-RandCylinderTransportStudent_SingleBinaryTac_CNNRNN_Mon-LocoTouch-v1"
    torch.manual_seed(0)
    # ---- 1. teacher
    env_t = make(teacher_task, num_envs=args.teacher_envs, device="cuda:0", seed=42)
    agent = train_cfg(teacher_task)
    agent["max_iterations"] = args.teacher_iters
    log_dir = "/tmp/lt_demo_teacher"
    runner = OnPolicyRunner(env_t, agent, log_dir=log_dir, device="cuda:0")
    t0 = time.perf_counter()
    runner.learn(args.teacher_iters, init_at_random_ep_len=True)
    for h in runner.history[::50] + runner.history[-1:]:
        emit({"phase": "teacher", "iter": h["iter"], "mean_reward": h.get("Train/mean_reward"),
              "mean_episode_length": h.get("Train/mean_episode_length"), "total_fps": h.get("Perf/total_fps")})
    emit({"phase": "teacher", "done": True, "elapsed_s": time.perf_counter() - t0})
    teacher_sd = {k: v.clone() for k, v in runner.alg.actor_critic.state_dict().items()}
    del env_t, runner
    torch.cuda.empty_cache()
    # ---- 2. distillation on the student env
    env = make(student_task, num_envs=args.num_envs, device="cuda:0", seed=7)
    runner = OnPolicyRunner(env, train_cfg(student_task), log_dir=None, device="cuda:0")
    runner.alg.actor_critic.load_state_dict(teacher_sd)
    teacher = runner.get_inference_policy(device="cuda:0")
    cfg = distillation_cfg(student_task)
    cfg.logger, cfg.log_root_path = "tensorboard", "/tmp/lt_demo_distill"
    cfg.num_iterations, cfg.bc_data_steps, cfg.dagger_data_steps = args.iterations, args.bc_steps, args.dagger_steps
    cfg.initial_epoches, cfg.incremental_epoches, cfg.evaluation_trajs_num = args.epochs, args.inc_epochs, args.eval_trajs
    d = Distillation(env, cfg, teacher_policy=teacher, verbose=False)
    rb, st = d.replay_buffer, d.student

    def sync_time(fn):
        torch.cuda.synchronize()
        t = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        return r, time.perf_counter() - t

    for it in range(cfg.num_iterations):
        rows0 = rb._rows_total
        (rew, lens), dt_c = sync_time(lambda: rb.collect_data(teacher, st if it else None, cfg.dagger_data_steps if it else cfg.bc_data_steps))
        rec = d.log_trajectory_rewards_and_lengths(rew, lens, it)
        _, dt_t = sync_time(lambda: st.train_on_data(rb, it))
        ep = st.num_epoches(it)
        emit({"phase": "distill", "iter": it, "actor": "student" if it else "teacher", "collect_s": dt_c,
              "collect_env_steps_per_s": (rb._rows_total - rows0) / dt_c, "buffer_steps": rb.num_steps, "buffer_trajs": rb.num_trajs,
              "epochs": ep, "train_s": dt_t, "train_kept_steps_per_s": rb.num_steps * ep / dt_t, **rec,
              **{f"train/{k}": v for k, v in st.last_stats.items()}})
    # ---- 3. evaluation: student (raw tactile, as the reference's evaluate) and the teacher on the same env
    rb.clear_buffer()
    env.reset()
    st.reset()
    st.eval()
    (rew_s, len_s), dt = sync_time(lambda: rb.evaluate(st, cfg.evaluation_trajs_num))

    class TeacherAsStudent:
        def __call__(self, prop, tac):
            return teacher(self.obs())

        def reset(self, dones=None):
            pass

    tw = TeacherAsStudent()
    tw.obs = lambda: env.obs_policy
    rb.clear_buffer()
    env.reset()
    (rew_t, len_t), _ = sync_time(lambda: rb.evaluate(tw, cfg.evaluation_trajs_num))
    emit({"phase": "eval", "student_trajs": len(rew_s), "student_len_mean": float(np.mean(len_s)), "student_rwd_mean": float(np.mean(rew_s)),
          "student_step_reward": float(np.sum(rew_s) / np.sum(len_s)), "teacher_trajs": len(rew_t), "teacher_len_mean": float(np.mean(len_t)),
          "teacher_rwd_mean": float(np.mean(rew_t)), "teacher_step_reward": float(np.sum(rew_t) / np.sum(len_t)),
          "max_episode_length": env.max_episode_length})


if __name__ == "__main__":
    main()
