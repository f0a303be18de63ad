from locotouch_amd.rl.ppo import PPO

__all__ = ["PPO"]
