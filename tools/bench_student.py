#!/usr/bin/env python3
"""BASELINE.json configs[3]: student distillation at the registration's 405 envs - throughput of the three phases of a
DAgger iteration on one MI355X (teacher-driven collection, student-driven collection, BC epochs), and the average duration
of the env kernels involved.  Prints one JSON line (kept under profiles/ as r02_student_405.json).

    python tools/bench_student.py [--num_envs 405] [--steps 2000] [--epochs 3]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num_envs", type=int, default=405)
    ap.add_argument("--steps", type=int, default=4000)
    ap.add_argument("--epochs", type=int, default=3)
    args = ap.parse_args()
    from locotouch_amd.agents import train_cfg
    from locotouch_amd.distill import Distillation, distillation_cfg
    from locotouch_amd.env import make
    from locotouch_amd.rl import OnPolicyRunner

    task = "Isaac-RandCylinderTransportStudent_SingleBinaryTac_CNNRNN_Mon-LocoTouch-v1"
    torch.manual_seed(0)
    env = make(task, num_envs=args.num_envs, device="cuda:0", seed=1)
    runner = OnPolicyRunner(env, train_cfg(task), log_dir=None, device="cuda:0")
    teacher = runner.get_inference_policy(device="cuda:0")
    cfg = distillation_cfg(task)
    cfg.logger, cfg.log_root_path = "none", "/tmp/lt_bench_student"
    d = Distillation(env, cfg, teacher_policy=teacher, verbose=False)
    rb, st = d.replay_buffer, d.student
    out = {"workload": "student_distillation", "num_envs": args.num_envs, "tactile_dim": 442}

    def timed(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        return r, time.perf_counter() - t0

    rb.collect_data(teacher, None, 200)  # warm-up (MIOpen find, allocator)
    rb.collect_data(teacher, st, 100)
    rb.clear_buffer()
    n0 = len(getattr(env, "_dummy", []))
    (_, lens), dt = timed(lambda: rb.collect_data(teacher, None, args.steps))
    env_steps = rb._rows_total  # rows kept = env steps x envs of this block
    out["collect_teacher_env_steps_per_s"] = env_steps / dt
    out["collect_teacher_kept_steps"] = rb.num_steps
    (_, _), dt = timed(lambda: rb.collect_data(teacher, st, args.steps // 2))
    out["collect_student_env_steps_per_s"] = (rb._rows_total - env_steps) / dt
    st.train()
    batch_trajs = int(cfg.batch_steps / (rb.num_steps / rb.num_trajs)) + 1
    def epochs():
        n = 0
        for _ in range(args.epochs):
            for batch in rb.to_recurrent_generator(batch_size=batch_trajs):
                st._optimizer.zero_grad(set_to_none=True)
                loss, _, _ = st.batch_loss(batch)
                loss.backward()
                st._optimizer.step()
                n += int(batch["masks"].numel())
        return n
    epochs()
    n, dt = timed(epochs)
    out["train_padded_steps_per_s"] = n / dt
    out["train_kept_steps_per_s"] = rb.num_steps * args.epochs / dt
    out["buffer_trajs"], out["buffer_steps"], out["batch_trajs"] = rb.num_trajs, rb.num_steps, batch_trajs
    # env kernels alone (HIP events on the launch stream)
    a = torch.zeros(args.num_envs, 12, device="cuda:0")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    for _ in range(20):
        env.step(a)
    torch.cuda.synchronize()
    K = 200
    ev[0].record()
    for _ in range(K):
        env.step_raw(a.data_ptr())  # lt_env_step = step kernel + tactile kernel
    ev[1].record()
    for _ in range(K):
        env.tactile_update()
    ev[2].record()
    torch.cuda.synchronize()
    out["env_step_us"] = ev[0].elapsed_time(ev[1]) / K * 1e3
    out["tactile_kernel_us"] = ev[1].elapsed_time(ev[2]) / K * 1e3
    print(json.dumps(out))


if __name__ == "__main__":
    main()
