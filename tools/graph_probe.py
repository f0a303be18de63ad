"""Which part of the rollout step refuses hipGraph capture?  (diagnostic)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from locotouch_amd.env import LocoTouchVecEnv
from locotouch_amd.rl import PPO, ActorCritic
import bench

dev = torch.device("cuda:0")
env = LocoTouchVecEnv(bench.TASKS["teacher"], num_envs=4096, device=dev)
ac = ActorCritic(348, 348, 12, **bench.POLICY_CFG)
alg = PPO(ac, device=dev, **bench.PPO_CFG)
alg.init_storage(4096, 24, [348], [348], [12])
obs, extras = env.get_observations()
cobs = extras["observations"]["critic"]
act = torch.zeros(4096, 12, device=dev)

def try_capture(name, fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        torch.cuda.synchronize()
        g.replay(); torch.cuda.synchronize()
        print(name, "OK")
    except Exception as e:
        print(name, "FAILED:", str(e).split("\n")[0])
        torch.cuda.synchronize()

def f_env():
    env.step_raw(act.data_ptr())
def f_env_full():
    env.step(act)
def f_act():
    with torch.inference_mode():
        alg.act(obs, cobs)
def f_store():
    with torch.inference_mode():
        alg.storage.clear()
        a = alg.act(obs, cobs)
        alg.process_env_step(env.reward_buf, env.dones_buf, {"time_outs": env.time_out_buf.bool()})
try_capture("env.step_raw", f_env)
try_capture("env.step", f_env_full)
try_capture("policy act", f_act)
try_capture("act+store", f_store)

def f_mlp():
    with torch.inference_mode():
        alg.actor_critic.actor(obs)
def f_randn():
    with torch.inference_mode():
        torch.randn(4096, 12, device=dev)
def f_normal():
    with torch.inference_mode():
        m = torch.zeros(4096, 12, device=dev)
        torch.normal(m, torch.ones_like(m))
def f_dist():
    with torch.inference_mode():
        m = torch.zeros(4096, 12, device=dev)
        d = torch.distributions.Normal(m, alg.actor_critic.std.expand_as(m))
        a = d.sample()
        d.log_prob(a).sum(-1)
def f_nograd():
    with torch.no_grad():
        alg.act(obs, cobs)
try_capture("mlp", f_mlp)
try_capture("randn", f_randn)
try_capture("normal", f_normal)
try_capture("dist", f_dist)
try_capture("act under no_grad", f_nograd)
