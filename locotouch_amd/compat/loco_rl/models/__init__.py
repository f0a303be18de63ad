"""Model configuration record used by the reference's distillation configs (`from loco_rl.models import ModelCfg`,
locotouch/config/locotouch/agents/distillation_cfg.py:2).  Field names and defaults follow the reference record
(loco_rl/loco_rl/models/model_cfg.py:4-25) so that `ModelCfg(model_type="CNN2dHead", ...)` in unmodified configs resolves."""
from locotouch_amd.compat.configclass import configclass


@configclass
class ModelCfg:
    model_type: str = "MLP"          # "MLP" | "RNN" | "CNN2d" | "CNN2dHead"
    hidden_dims: list = [512, 256, 128]
    activation: str = "elu"
    final_layer_activation: object = None
    rnn_type: str = "gru"
    rnn_hidden_size: int = 256
    rnn_num_layers: int = 1
    img_shape: tuple = (2, 17, 13)   # two identical binary channels of the 17 x 13 taxel grid (observations.py:307-308)
    cnn_channels: tuple = (24, 24, 24)
    cnn_kernel_size: tuple = (4, 3, 2)
    cnn_stride: tuple = (2, 1, 1)
    cnn_nonlinearity: str = "relu"
    cnn_padding: object = None
    cnn_use_maxpool: bool = True
    cnn_normlayer: object = None


from locotouch_amd.rl.models import MLP, RNN, CNN2d, CNN2dHead, Memory, generate_model  # noqa: E402

__all__ = ["ModelCfg", "MLP", "RNN", "CNN2d", "CNN2dHead", "Memory", "generate_model"]
