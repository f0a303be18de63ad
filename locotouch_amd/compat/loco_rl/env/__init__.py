"""VecEnv protocol marker (reference loco_rl/loco_rl/env/vec_env.py:12-101): attributes and methods a trainer-facing env has."""
from abc import ABC, abstractmethod


class VecEnv(ABC):
    num_envs: int
    num_actions: int
    max_episode_length: int
    device: object

    @abstractmethod
    def get_observations(self): ...

    @abstractmethod
    def step(self, actions): ...

    @abstractmethod
    def reset(self): ...


__all__ = ["VecEnv"]
