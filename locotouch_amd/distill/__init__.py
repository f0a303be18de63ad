"""Teacher -> student distillation (BASELINE.json configs[3]; reference locotouch/distill/)."""
from .config import DistillationCfg, ModelCfg, distillation_cfg
from .distillation import Distillation
from .replay_buffer import ReplayBuffer
from .student import Student
from .tactile_recorder import TactileRecorder

__all__ = ["Distillation", "DistillationCfg", "ModelCfg", "ReplayBuffer", "Student", "TactileRecorder", "distillation_cfg"]
