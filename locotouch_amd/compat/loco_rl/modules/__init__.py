from locotouch_amd.rl.modules import ActorCritic, ActorCriticEncoder, ActorCriticRecurrent
from locotouch_amd.rl.normalizer import EmpiricalNormalization

__all__ = ["ActorCritic", "ActorCriticRecurrent", "ActorCriticEncoder", "EmpiricalNormalization"]
