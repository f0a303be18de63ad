"""`from loco_rl.models.model_generation import generate_model` (reference locotouch/distill/student.py:6)."""
from locotouch_amd.rl.models import generate_model

__all__ = ["generate_model"]
