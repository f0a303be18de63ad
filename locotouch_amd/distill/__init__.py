"""Teacher -> student distillation (BASELINE.json configs[3]; reference locotouch/distill/)."""
import os as _os

# MIOpen's default "find" runs an exhaustive solver search (and sometimes a kernel compile: 151 s measured) for every new
# convolution shape; the student's 3 tiny convolutions gain nothing from it.  Read once, when MIOpen initialises.
_os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")

from .config import DistillationCfg, ModelCfg, distillation_cfg
from .distillation import Distillation
from .replay_buffer import ReplayBuffer
from .student import Student
from .tactile_recorder import TactileRecorder

__all__ = ["Distillation", "DistillationCfg", "ModelCfg", "ReplayBuffer", "Student", "TactileRecorder", "distillation_cfg"]
