"""PPO trainer (PyTorch-ROCm): the `loco_rl` half of the path, driving the HIP env through the VecEnv protocol."""
from .dist import Dist
from .fused import FusedRollout
from .modules import ActorCritic, ActorCriticEncoder, ActorCriticRecurrent
from .normalizer import EmpiricalNormalization
from .ppo import PPO
from .runner import OnPolicyRunner
from .storage import RolloutStorage
from .trajectories import split_and_pad_trajectories, unpad_trajectories

__all__ = ["Dist", "FusedRollout", "ActorCritic", "ActorCriticRecurrent", "ActorCriticEncoder", "EmpiricalNormalization", "PPO", "OnPolicyRunner", "RolloutStorage",
           "split_and_pad_trajectories", "unpad_trajectories"]
