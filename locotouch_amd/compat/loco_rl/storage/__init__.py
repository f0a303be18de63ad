from locotouch_amd.rl.storage import RolloutStorage

__all__ = ["RolloutStorage"]
