#!/usr/bin/env python3
"""Diagnostic: wall time of the first, second, ... replay of a freshly captured 20-step rollout graph (the driver's bench shape times
ONE replay of a graph that has never run), with and without hipGraphUpload in front of it."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from locotouch_amd.env import LocoTouchVecEnv
from locotouch_amd.rl import PPO, ActorCritic, FusedRollout
from tests.rl_synth import POLICY_CFG, PPO_CFG

dev = torch.device("cuda:0")
n, K = 4096, int(os.environ.get("K", "20"))
env = LocoTouchVecEnv("Isaac-RandCylinderTransportTeacher-LocoTouch-v1", num_envs=n, device=dev, seed=42)
alg = PPO(ActorCritic(env.num_obs, env.num_obs, 12, **POLICY_CFG), device=dev, **PPO_CFG)
alg.init_storage(n, 24, [env.num_obs], [env.num_obs], [12])
fused = FusedRollout(env, alg)


def capture(k, keep):
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        fused.rollout(k)
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize(dev)
    g = torch.cuda.CUDAGraph(keep_graph=True) if keep else torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fused.rollout(k)
    torch.cuda.synchronize(dev)
    return g


for mode in ("plain", "upload"):
    g = capture(K, mode == "upload")
    if mode == "upload":
        g.instantiate()
        hip = ctypes.CDLL("libamdhip64.so")
        rc = hip.hipGraphUpload(ctypes.c_void_p(g.raw_cuda_graph_exec()), ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        torch.cuda.synchronize(dev)
        print("hipGraphUpload rc", rc)
    ts = []
    for i in range(5):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        g.replay()
        torch.cuda.synchronize(dev)
        ts.append((time.perf_counter() - t0) * 1e6)
    print(mode, "replay wall us:", " ".join(f"{t:8.1f}" for t in ts), f"  -> per step {ts[0] / K:.2f} first, {min(ts) / K:.2f} best")

# how the host learns that the replay is done: torch.cuda.synchronize() alone, or an event-query spin in front of it
g = capture(K, False)
for mode in ("sync", "spin+sync"):
    ts = []
    for i in range(8):
        torch.cuda.synchronize(dev)
        ev = torch.cuda.Event()
        t0 = time.perf_counter()
        g.replay()
        if mode != "sync":
            ev.record()
            while not ev.query():
                pass
        torch.cuda.synchronize(dev)
        ts.append((time.perf_counter() - t0) * 1e6)
    print(mode, "replay wall us:", " ".join(f"{t:8.1f}" for t in ts[1:]), f"  -> best per step {min(ts) / K:.2f}")
