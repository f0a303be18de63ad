from locotouch_amd.rl.runner import OnPolicyRunner

__all__ = ["OnPolicyRunner"]
