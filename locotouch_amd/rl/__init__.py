"""PPO trainer (PyTorch-ROCm): the `loco_rl` half of the path, driving the HIP env through the VecEnv protocol."""
from .dist import Dist
from .fused import FusedRollout
from .modules import ActorCritic
from .ppo import PPO
from .runner import OnPolicyRunner
from .storage import RolloutStorage

__all__ = ["Dist", "FusedRollout", "ActorCritic", "PPO", "OnPolicyRunner", "RolloutStorage"]
