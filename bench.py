#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of the synthetic rollout of Isaac-RandCylinderTransportTeacher-LocoTouch-v1
at 4096 envs per GPU (BASELINE.json metric), one process per GPU.

A "step" = one pass of the hot path over one batch: policy act (random-init 348->512->256->128->12 ActorCritic, fp32)
-> lt_env_step (HIP kernels through the C ABI) -> PPO.process_env_step (time-out bootstrap + rollout-storage write).
All inputs are HBM-resident; the 24-step rollout is replayed as one hipGraph when capture succeeds.

    python bench.py --gpus 1 --steps 2400 --warmup 1000
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     - the dominant kernel (lt_step_kernel) timed live with HIP events (lt_env_step_profiled) against the
                 HBM roofline: algorithmic bytes = 6720 B per env-step x envs per launch.
  cpu_baseline - the CPU oracle (a port: oracle/lt_oracle.c, OpenMP over envs) timed on this box's host cores on a
                 bounded sample of the same workload.  A reported baseline, not the optimisation target.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

ALGO_BYTES_PER_ENV_STEP = {"teacher": 6720, "locomotion": 5344}  # 1680 / 1336 words of 4 bytes
# bf16 observation rows (BASELINE config 5; state quad arrays stay f32): the 580 R + 696 W row words of the teacher table
# (locomotion: 450 + 540) move as 2 bytes each: 6720 - 2 * 1276 = 4168 B, 5344 - 2 * 990 = 3364 B per env-step
ALGO_BYTES_PER_ENV_STEP_BF16_ROWS = {"teacher": 4168, "locomotion": 3364}
MFMA_F32_PEAK_TFLOPS = 157.3  # dense f32-INPUT MFMA peak (MI355X_MICROARCH.md): what an ideal f32-MFMA kernel could reach
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense f16 MFMA peak: the pipe lt_mlp_kernel runs on, at 3 MFMAs per f32-equivalent MAC
L2_STREAM_PEAK_GBS = 17800.0  # rows shared by every workgroup, served from the XCDs' L2s: 16.8-18.8 TB/s chip-wide (MI355X_MICROARCH.md, "Indexed rows")
VALU_PEAK_GINST = 1024 * 2.4 / 4  # wave64 VALU instructions per ns the chip can issue: 1024 SIMDs x 2.4 GHz / 4 cycles
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); measured float4-copy ceiling is 6290 GB/s
TASKS = {"teacher": "Isaac-RandCylinderTransportTeacher-LocoTouch-v1", "locomotion": "Isaac-Locomotion-LocoTouch-v1"}
POLICY_CFG = dict(init_noise_std=1.0, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128], activation="elu")
PPO_CFG = dict(value_loss_coef=1.0, use_clipped_value_loss=True, clip_param=0.2, entropy_coef=0.01, num_learning_epochs=5,
               num_mini_batches=4, learning_rate=1.0e-3, schedule="adaptive", gamma=0.99, lam=0.95, desired_kl=0.01,
               max_grad_norm=1.0)
ROLLOUT = 24  # num_steps_per_env (agents/rsl_rl_ppo_cfg.py:7)


def cpu_baseline(task: str, num_envs: int, budget_s: float = 12.0) -> dict:
    """The oracle timed on the host cores (bounded sample).  Only this leg of bench.py touches oracle/."""
    import numpy as np

    from locotouch_amd import _abi
    from tests import oracle_lib

    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:  # honour the container's CPU share (cgroup v2 quota), not just the affinity mask
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(float(quota) / float(period))))
    except Exception:
        pass
    cores = min(cores, int(os.environ.get("LT_CPU_BASELINE_THREADS", "64")))
    cfg = _abi.default_cfg(_abi.CONSTS["LT_TASK_TRANSPORT_TEACHER" if task == "teacher" else "LT_TASK_LOCOMOTION"], num_envs=num_envs, seed=42)
    env = oracle_lib.OracleEnv(cfg)
    env.reset_all()
    rng = np.random.default_rng(1234)
    acts = (0.5 * rng.standard_normal((8, num_envs, 12))).astype(np.float32)
    for i in range(2):
        env.step(acts[i], nthreads=cores)
    t0, steps = time.perf_counter(), 0
    while True:
        env.step(acts[steps % 8], nthreads=cores)
        steps += 1
        el = time.perf_counter() - t0
        if el > budget_s or steps >= 20000:
            break
    return {"value": num_envs * steps / el, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{steps} steps x {num_envs} envs of the same task, 0.5*N(0,1) actions, OpenMP over envs, {el:.1f} s"}


def _spawn_ranks(args, script: str | None = None, argv: list | None = None) -> int:
    """`python bench.py --gpus N` typed plainly (no torchrun): this parent starts N rank processes of the same script -
    one per GPU, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* in their environment - BEFORE touching torch or the GPU itself
    (a process that has initialised HIP must never exec or fork into another GPU program on this pool), forwards
    rank 0's single JSON line, and exits with the worst child code.  (`script` / `argv`: the child to start instead of this
    file - tests/test_bench_spawn.py checks the rank plumbing with a stub child.)"""
    import socket
    import subprocess

    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + (sys.argv[1:] if argv is None else argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        p.wait()
        rc = rc or p.returncode
    sys.stdout.write(out)
    sys.stdout.flush()
    return rc


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2400)
    ap.add_argument("--warmup", type=int, default=1000)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--task", default="teacher", choices=list(TASKS))
    ap.add_argument("--obs-dtype", default="f32", choices=["f32", "bf16"],
                    help="element type of the observation rows, their history and the rollout-storage observations (bf16: BASELINE config 5; state stays f32)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager-torch", action="store_true", help="reference-shaped rollout (torch elementwise ops) instead of the fused kernels")
    ap.add_argument("--update-iters", type=int, default=3, help="PPO iterations timed for train_total_fps (0 = skip)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(_spawn_ranks(args))

    import torch

    from locotouch_amd.env import LocoTouchVecEnv
    from locotouch_amd.rl import PPO, ActorCritic, Dist, FusedRollout

    dist = Dist.from_env()
    if dist.world_size != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={dist.world_size}")
    dev = torch.device(f"cuda:{os.environ.get('LT_FORCE_DEVICE', dist.local_rank)}")  # LT_FORCE_DEVICE: multi-rank rehearsal on one GPU
    torch.cuda.set_device(dev)
    n = args.envs
    # one population over all ranks: same seed, RNG streams keyed by the global env index; with more than one rank the
    # curriculum gate is decided on cross-rank sums, one 1-KiB all-reduce per rollout
    env = LocoTouchVecEnv(TASKS[args.task], num_envs=n, device=dev, seed=42, env_index_offset=dist.rank * n,
                          cur_gate_external=1 if dist.world_size > 1 else 0)
    torch.manual_seed(1234)  # identical random-init policy on every rank
    ac = ActorCritic(env.num_obs, env.num_obs, 12, **POLICY_CFG)
    alg = PPO(ac, device=dev, dist=dist, **PPO_CFG)
    obs_dtype = torch.bfloat16 if args.obs_dtype == "bf16" else torch.float32
    alg.init_storage(n, ROLLOUT, [env.num_obs], [env.num_obs], [12], obs_dtype=obs_dtype)
    obs, extras = env.get_observations()
    critic_obs = extras["observations"]["critic"]

    fused = None if args.eager_torch else FusedRollout(env, alg)

    def rollout_steps(k: int) -> None:
        """k consecutive rollout steps starting at storage slot 0 (k <= ROLLOUT)."""
        if fused is not None:  # [actor+critic MLPs + sampling] -> [env step + storage record (+ curriculum tail)]
            fused.rollout(k)
            return
        alg.storage.clear()  # reference-shaped eager path (every elementwise op its own launch)
        with torch.inference_mode():
            for _ in range(k):
                actions = alg.act(obs, critic_obs)
                _, rew, dones, infos = env.step(actions)
                alg.process_env_step(rew, dones, infos)

    # One hipGraph per distinct chunk length the run needs (a full 24-step rollout and, when --steps / --warmup are not
    # multiples of 24, their remainders), all captured before the timed region.
    graphs: dict[int, "torch.cuda.CUDAGraph"] = {}

    def capture(k: int) -> None:
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            rollout_steps(k)  # warm every op / allocation before capture
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            rollout_steps(k)
        torch.cuda.synchronize(dev)
        graphs[k] = g

    if not args.no_graph:
        try:
            for k in sorted({ROLLOUT, args.steps % ROLLOUT, args.warmup % ROLLOUT} - {0}, reverse=True):
                capture(k)
        except Exception as exc:  # capture is an optimisation, not a requirement
            print(f"[bench] hipGraph capture failed ({type(exc).__name__}: {exc}); falling back to eager launches", file=sys.stderr)
            graphs.clear()
            torch.cuda.synchronize(dev)

    launched = {"graph": 0, "eager": 0}

    def run(k: int) -> None:
        full, rem = divmod(k, ROLLOUT)
        for chunk in [ROLLOUT] * full + ([rem] if rem else []):
            if chunk in graphs:
                graphs[chunk].replay()
                launched["graph"] += chunk
            else:
                rollout_steps(chunk)
                launched["eager"] += chunk

    run(args.warmup)
    launched = {"graph": 0, "eager": 0}
    dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    run(args.steps)
    torch.cuda.synchronize(dev)
    dist.barrier()
    elapsed = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    dist.all_reduce_max_(elapsed)
    elapsed = float(elapsed)
    value = n * args.gpus * args.steps / elapsed

    # ---- roofline of the dominant kernel: live HIP-event timing of lt_step_kernel on the launch stream ----
    with torch.inference_mode():
        act = alg.act(obs, critic_obs).clone()
    if args.obs_dtype == "bf16":  # the kernel variant the rollout runs: bf16 rows from one storage slot into the next
        st = alg.storage
        env.set_row_format(torch.bfloat16)
        prof = lambda: env.step_rows_profiled(act, st.observations[0], st.privileged_observations[0], st.observations[1], st.privileged_observations[1])  # noqa: E731
    else:
        prof = lambda: env.step_profiled(act)  # noqa: E731
    try:
        for _ in range(20):
            prof()
        ms = [prof() for _ in range(200)]
    finally:  # (lt_env_step* without row pointers refuses to run while the bf16 row format is selected, include/lt_env.h)
        if args.obs_dtype == "bf16":
            env.set_row_format(torch.float32)
    k_ms = sum(ms) / len(ms)
    algo_bytes = (ALGO_BYTES_PER_ENV_STEP_BF16_ROWS if args.obs_dtype == "bf16" else ALGO_BYTES_PER_ENV_STEP)[args.task] * n
    achieved = algo_bytes / (k_ms * 1e-3) / 1e9
    key = f"{args.task}_{n}" + ("_bf16rows" if args.obs_dtype == "bf16" else "")
    # The PMC numbers (HBM traffic, VALU instruction count) come from committed rocprofv3 --pmc summaries of this same command
    # (profiles/traffic.json, profiles/sq_counters.json; tools/profile_round.sh).  Each entry carries the stamp of the kernel
    # sources it was measured on: an entry whose stamp differs from this tree's is NOT reported (`traffic` null, `stale_profile`).
    from locotouch_amd.build import step_kernel_source_hash

    tree_hash = step_kernel_source_hash()

    def committed(fname: str) -> tuple[dict, bool]:
        """(entry, stale) of this configuration's record in profiles/<fname>."""
        try:
            entry = json.load(open(os.path.join(REPO, "profiles", fname))).get(key, {})
        except Exception:
            return {}, False
        if entry and entry.get("source_hash") != tree_hash:
            return {}, True
        return entry, False

    tentry, tstale = committed("traffic.json")
    traffic = tentry.get("hbm_bytes_per_launch")
    roofline = {"bound": "hbm", "kernel": "lt_step_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel_ms": k_ms, "algorithmic_bytes_per_launch": algo_bytes,
                "env_only_steps_per_s": n / (k_ms * 1e-3), "kernel_source_hash": tree_hash}
    # what actually bounds the step kernel: VALU issue.  Wave-instructions per launch from the SQ counters of the same command
    # (profiles/sq_counters.json, tools/pmc_sq.py) against what 1024 SIMDs can issue in the measured kernel time.
    sentry, sstale = committed("sq_counters.json")
    if tstale or sstale:
        roofline["stale_profile"] = True  # profiles/*.json were measured on other kernel sources: re-run tools/profile_round.sh
    if True:
        valu = sentry.get("valu_wave_insts_per_launch")
        if valu:
            rate = valu / (k_ms * 1e6)  # G wave-instructions / s
            roofline["valu"] = {"bound": "valu-issue", "kernel": "lt_step_kernel", "achieved": rate, "peak": VALU_PEAK_GINST,
                                "unit": "G wave-inst/s", "frac": rate / VALU_PEAK_GINST, "valu_wave_insts_per_launch": valu,
                                "note": "SQ_INSTS_VALU per launch / kernel time against 1024 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction"}
    # the other launch of a rollout step: actor + critic MLPs + sampling in one kernel, against the dense f32 MFMA peak
    if fused is not None and fused.actor_mlp is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            fused.policy_value_launch(0)
        e0.record()
        for _ in range(50):
            fused.policy_value_launch(0)
        e1.record()
        torch.cuda.synchronize(dev)
        mlp_ms = e0.elapsed_time(e1) / 50
        dims = [env.num_obs, 512, 256, 128]
        macs = sum(a * b for a, b in zip(dims[:-1], dims[1:]))
        flop = 2.0 * n * (2 * macs + dims[-1] * (12 + 1))
        # The kernel evaluates an f32-equivalent MAC as THREE f16 MFMAs (csrc/lt_mlp.hip), so its matrix-pipe peak is the f16
        # peak / 3; what bounds it in practice is the weight stream every workgroup pulls from its XCD's L2 (2 f16 planes =
        # 4 bytes per weight, once per workgroup of 16 x rt rows).  Both fractions are <= 1 by construction.
        tflops = flop / (mlp_ms * 1e-3) / 1e12
        t16 = 2 * ((n + 15) // 16)  # 16-row tiles of both networks; row tiles per workgroup as csrc/lt_mlp.hip::pick_row_tiles
        rt = 4 if t16 // 4 >= 256 else (2 if t16 // 2 >= 256 else 1)
        wg_per_net = (n + 16 * rt - 1) // (16 * rt)
        stream_bytes = 2 * wg_per_net * 4.0 * (macs + dims[-1] * 6.5)  # actor (12 outputs) + critic (1): (12 + 1) / 2 per net
        l2_rate = stream_bytes / (mlp_ms * 1e-3) / 1e9
        roofline["mlp"] = {"bound": "l2-stream", "kernel": "lt_mlp_kernel (actor + critic + sampling)", "flop_per_launch": flop,
                           "kernel_ms": mlp_ms, "achieved": tflops, "peak": MFMA_F16_PEAK_TFLOPS / 3.0, "unit": "TFLOP/s (f32-equivalent)",
                           "frac": tflops / (MFMA_F16_PEAK_TFLOPS / 3.0),
                           "rows_per_workgroup": 16 * rt,
                           "l2_stream": {"bytes_per_launch": stream_bytes, "achieved": l2_rate, "peak": L2_STREAM_PEAK_GBS, "unit": "GB/s",
                                         "frac": l2_rate / L2_STREAM_PEAK_GBS},
                           "ideal_f32_mfma_kernel_tflops": MFMA_F32_PEAK_TFLOPS}

    extra = {}
    if args.update_iters > 0:  # rollout + GAE + PPO update with the gradient all-reduce (Perf/total_fps of the reference)
        iters = args.update_iters

        def iteration() -> None:
            if ROLLOUT in graphs:
                graphs[ROLLOUT].replay()
            else:
                rollout_steps(ROLLOUT)
            env.curriculum_sync(dist, ROLLOUT)
            with torch.inference_mode():
                alg.compute_returns(critic_obs)
            alg.update()

        iteration()  # warm the update path (allocator, hipBLASLt heuristics)
        dist.barrier(); torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(iters):
            iteration()
        torch.cuda.synchronize(dev); dist.barrier()
        el = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        dist.all_reduce_max_(el)
        extra["train_total_fps"] = n * args.gpus * ROLLOUT * iters / float(el)
        extra["train_iteration_ms"] = 1e3 * float(el) / iters

    cpu = None
    if dist.rank == 0 and args.gpus == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.task, n)

    env.check()  # a chained step launch that lost its population-pass hand-off (LT_F_COUNTERS[1]) fails the run loudly
    if dist.rank == 0:
        task_name = "TransportTeacher" if args.task == "teacher" else "Locomotion"
        out = {"metric": f"env-steps/sec (whole node), {task_name} {n} envs/GPU", "value": value, "unit": "env-steps/s",
               "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32" if args.obs_dtype == "f32" else "f32 state and arithmetic / bf16 observation rows", "data": "synthetic",
               "config": {"workload": f"{TASKS[args.task]} rollout (policy act + env step + storage), {n} envs/GPU, "
                                      f"random-init ActorCritic [512,256,128], seed 42, env RNG keyed by global env index"
                                      + ("; observation rows, their 6-deep history and the rollout-storage observations in bf16, all state "
                                         "quad arrays and the arithmetic in f32" if args.obs_dtype == "bf16" else ""),
                          "obs_dtype": args.obs_dtype,
                          "envs_per_gpu": n, "rollout_len": ROLLOUT, "hipgraph_replayed": launched["graph"] > 0 and launched["eager"] == 0,
                          "steps_replayed_from_graphs": launched["graph"], "steps_launched_eagerly": launched["eager"],
                          "fused_rollout": fused is not None, "launches_per_step": fused.launches_per_step if fused is not None else None},
               "roofline": roofline, "cpu_baseline": cpu}
        out.update(extra)
        print(json.dumps(out))
    dist.shutdown()


if __name__ == "__main__":
    main()
