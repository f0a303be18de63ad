"""`from loco_rl.models.model_cfg import ModelCfg`."""
from . import ModelCfg

__all__ = ["ModelCfg"]
