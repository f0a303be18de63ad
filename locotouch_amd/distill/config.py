"""Distillation configuration records (reference locotouch/config/locotouch/agents/distillation_cfg.py:1-106 and
loco_rl/loco_rl/models/model_cfg.py:4-26), as plain dataclasses: same field names, same values."""
from __future__ import annotations

import copy
from dataclasses import asdict, dataclass, field


@dataclass
class ModelCfg:
    model_type: str = "MLP"
    hidden_dims: list | None = field(default_factory=lambda: [512, 256, 128])
    activation: str = "elu"
    final_layer_activation: str | None = None
    rnn_type: str = "gru"
    rnn_hidden_size: int = 256
    rnn_num_layers: int = 1
    img_shape: tuple = (2, 17, 13)
    cnn_channels: tuple = (24, 24, 24)
    cnn_kernel_size: tuple = (4, 3, 2)
    cnn_stride: tuple = (2, 1, 1)
    cnn_nonlinearity: str = "relu"
    cnn_padding: tuple | None = None
    cnn_use_maxpool: bool = True
    cnn_normlayer: str | None = None
    embedding_dim: int | None = None


@dataclass
class DistillationCfg:
    distillation_type: str = "Monolithic"  # or "RMA"
    pre_encoder: ModelCfg = field(default_factory=lambda: ModelCfg(model_type="MLP", hidden_dims=None, embedding_dim=None))
    tactile_encoder: ModelCfg = field(default_factory=lambda: ModelCfg(model_type="MLP", hidden_dims=[256, 128, 64], embedding_dim=64))
    student_policy: ModelCfg = field(default_factory=ModelCfg)
    device: str = "cuda:0"
    log_root_path: str = "logs/distillation"
    experiment_name: str = "object"
    log_dir: str = "specify_log_dir"
    log_dir_distill: str = "specify_log_dir_distill"
    checkpoint_distill: str = "specify_checkpoint_distill"
    logger: str = "wandb"
    wandb_project: str = "Transport_Distillation"
    num_iterations: int = 8
    bc_data_steps: int = 400000
    dagger_data_steps: int = 200000
    initial_epoches: int = 2000
    incremental_epoches: int = 500
    final_epoches: int = 0
    batch_steps: int = 20000
    distill_lr: float = 5.0e-4
    evaluation_trajs_num: int = 2000
    clip_actions: bool = False
    clip_range: float = 100.0
    action_scale_within_env: float = 0.25
    min_delay: int = 1
    max_delay: int = 2

    def to_dict(self) -> dict:
        return asdict(self)


def _rand_cylinder_cnn_rnn_mon() -> DistillationCfg:
    """DistillationRandCylinderCNNRNNMonCfg (distillation_cfg.py:96-104)."""
    c = DistillationCfg()
    c.pre_encoder.model_type = "CNN2dHead"
    c.pre_encoder.embedding_dim = 64
    c.tactile_encoder.model_type = "RNN"
    c.tactile_encoder.rnn_hidden_size = 512
    c.experiment_name = "rand_cylinder"
    return c


# registry kwarg `distillation_cfg_entry_point` of the student ids (locotouch/config/locotouch/__init__.py:127-146)
_CFGS = {
    "Isaac-RandCylinderTransportStudent_SingleBinaryTac_CNNRNN_Mon-LocoTouch-v1": _rand_cylinder_cnn_rnn_mon,
    "Isaac-RandCylinderTransportStudent_SingleBinaryTac_CNNRNN_Mon-LocoTouch-Play-v1": _rand_cylinder_cnn_rnn_mon,
}


def distillation_cfg(task: str) -> DistillationCfg:
    return copy.deepcopy(_CFGS[task]())
