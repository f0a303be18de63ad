from locotouch_amd.rl.modules import ActorCritic

__all__ = ["ActorCritic"]
