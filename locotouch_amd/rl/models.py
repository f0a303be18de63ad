"""Network building blocks of the student / distillation path: MLP, GRU/LSTM memory + MLP head, tactile CNN (+ MLP head).

Behavioural twins of the reference's `loco_rl.models` (loco_rl/loco_rl/models/{mlp,rnn,memory_module,cnn_2d,model_generation}.py):
same constructor signatures, same parameter names (checkpoints interchange: `model.N.*`, `memory.rnn.*`, `mlp.model.N.*`,
`conv.N.*`, `conv.conv.N.*`, `head.model.N.*`) and the same module construction order (equal seeds give equal initial weights).
The reference student (locotouch/distill/student.py:40-60) is `CNN2dHead` (2x17x13 binary taxel image -> 64) -> `RNN`
(GRU 512 + MLP [256,128,64] -> 64) -> `MLP` (270 + 64 -> [512,256,128] -> 12); the GRU runs on MIOpen.

One deliberate difference: `Memory.reset(dones)` zeroes the hidden state of the envs whose `dones` is
non-zero.  The reference indexes with the long tensor itself (`state[..., dones, :] = 0`, memory_module.py:26-27), i.e. it
zeroes envs 0 and 1 whenever anything finished.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .gru import gru_sequence
from .linear import Linear

_ACTIVATIONS = {"elu": nn.ELU, "selu": nn.SELU, "relu": nn.ReLU, "crelu": nn.ReLU, "lrelu": nn.LeakyReLU, "tanh": nn.Tanh,
                "sigmoid": nn.Sigmoid}


def get_activation(name: str) -> nn.Module:
    if name not in _ACTIVATIONS:
        raise ValueError(f"Invalid activation function: {name}")
    return _ACTIVATIONS[name]()


class MLP(nn.Module):
    def __init__(self, input_dim, hidden_dims, output_dim, activation="elu", final_layer_activation=None):
        super().__init__()
        act = get_activation(activation)  # ONE module instance shared by all layers, as in the reference (stateless)
        dims = [input_dim] + list(hidden_dims or [])
        layers: list[nn.Module] = []
        for a, b in zip(dims[:-1], dims[1:]):
            layers += [Linear(a, b), act]
        layers.append(Linear(dims[-1], output_dim))
        if final_layer_activation is not None:
            layers.append(get_activation(final_layer_activation))
        self.model = nn.Sequential(*layers)

    def forward(self, x):
        return self.model(x)

    def reset(self, dones=None):
        pass


class Memory(nn.Module):
    """`input` (L, B, D): a padded batch of whole trajectories, hidden state given (or zero); `input` (B, D): one env step,
    the module's own hidden state carried across calls."""

    def __init__(self, memory_type, input_dim, hidden_size, num_layers):
        super().__init__()
        cls = nn.GRU if memory_type.lower() == "gru" else nn.LSTM
        self.rnn = cls(input_size=input_dim, hidden_size=hidden_size, num_layers=num_layers)
        self.hidden_states = None
        self.use_miopen_sequence = False  # True: whole-trajectory batches through nn.GRU's own (MIOpen) path

    def forward(self, input, hidden_states=None):
        if input.dim() == 3:
            if input.is_cuda and isinstance(self.rnn, nn.GRU) and self.rnn.num_layers == 1 and not self.use_miopen_sequence:
                out, _ = gru_sequence(self.rnn, input, hidden_states)  # rl/gru.py: ~15x MIOpen's RNN path at L ~ 500, B ~ 47
                return out
            out, _ = self.rnn(input, hidden_states)
            return out
        if isinstance(self.rnn, nn.GRU) and self.rnn.num_layers == 1:
            # one env step: the fused cell kernel on the same weights (3x faster than a length-1 MIOpen sequence at 405 envs)
            r = self.rnn
            h = self.hidden_states[0] if self.hidden_states is not None else input.new_zeros(input.shape[0], r.hidden_size)
            h = torch.gru_cell(input, h, r.weight_ih_l0, r.weight_hh_l0, r.bias_ih_l0, r.bias_hh_l0)
            self.hidden_states = h.unsqueeze(0)
            return h
        out, self.hidden_states = self.rnn(input.unsqueeze(0), self.hidden_states)
        return out.squeeze(0)

    def reset(self, dones=None):
        if self.hidden_states is None:
            return
        if dones is None:
            self.hidden_states = None
            return
        keep = (dones.reshape(-1) == 0)
        for state in (self.hidden_states if isinstance(self.hidden_states, tuple) else (self.hidden_states,)):
            state.mul_(keep.to(state.dtype)[None, :, None])  # masked multiply: no host sync (a bool-mask index_put has one)

    def get_hidden_states(self):
        return self.hidden_states


class RNN(nn.Module):
    def __init__(self, input_dim, hidden_dims, output_dim, activation="elu", rnn_memory_type="gru", rnn_hidden_size=256, rnn_num_layers=1):
        super().__init__()
        self.memory = Memory(rnn_memory_type, input_dim, rnn_hidden_size, rnn_num_layers)
        self.mlp = MLP(rnn_hidden_size, hidden_dims, output_dim, activation)

    def forward(self, x, hidden_states=None):
        return self.mlp(self.memory(x, hidden_states=hidden_states))

    def reset(self, dones=None):
        self.memory.reset(dones=dones)

    def get_hidden_states(self):
        return self.memory.get_hidden_states()


class Conv2dAsGemm(nn.Conv2d):
    """`nn.Conv2d` for tiny images and huge batches: on CUDA the layer is applied as ONE dense GEMM.

    The student's three convolutions work on 2 x 17 x 13 ... 24 x 5 x 3 maps but on N = L * B ~ 50 000 images per batch.  MIOpen
    needs a solver search per new N (0.1-0.3 s, once a 151 s kernel compile, tools/shape_probe.py) and without the search
    (MIOPEN_FIND_MODE=FAST) takes 42 ms for the stack forward + backward.  A convolution on a C x H x W image is a linear map
    of C*H*W inputs to C'*H'*W' outputs; its matrix is the layer's response to the identity basis,
    `conv2d(eye(C*H*W).view(-1, C, H, W), weight)` - a convolution on a FIXED batch of C*H*W (442 / 840 / 360) tiny images,
    differentiable w.r.t. the weights.  The batch then passes through `x.flatten(1) @ matrix`: 14x the FLOPs of the
    convolution but at GEMM efficiency, the same few milliseconds at every N, no search.  Same parameters and state_dict as
    nn.Conv2d; equal to it to fp32 rounding (tests/test_rl_conv.py)."""

    as_gemm = True

    def forward(self, x):
        if not (Conv2dAsGemm.as_gemm and x.is_cuda and x.dim() == 4 and x.shape[0] >= 1024):
            return super().forward(x)
        from .linear import split_k_matmul

        n, c, h, w = x.shape
        key = (c, h, w, x.device)
        basis = getattr(self, "_basis", {}).get(key)
        if basis is None:
            basis = torch.eye(c * h * w, device=x.device, dtype=x.dtype).view(c * h * w, c, h, w)
            self._basis = {key: basis}
        resp = self._conv_forward(basis, self.weight, None)  # (C H W, C', H', W'): row i = the layer's response to input element i
        ho, wo = resp.shape[2], resp.shape[3]
        y = split_k_matmul(x.flatten(1), resp.flatten(1))
        if self.bias is not None:
            y = y + self.bias.repeat_interleave(ho * wo)
        return y.view(n, self.out_channels, ho, wo)


def conv2d_output_shape(h, w, kernel_size=1, stride=1, padding=0, dilation=1):
    pair = lambda v: v if isinstance(v, tuple) else (v, v)  # noqa: E731
    (kh, kw), (sh, sw), (ph, pw) = pair(kernel_size), pair(stride), pair(padding)
    return (h + 2 * ph - dilation * (kh - 1) - 1) // sh + 1, (w + 2 * pw - dilation * (kw - 1) - 1) // sw + 1


class CNN2d(nn.Module):
    """Conv stack on [B, C, H, W].  `use_maxpool`: the convs run at stride 1 and a MaxPool2d(stride) follows every conv whose
    configured stride is > 1 (for the student's (2, 1, 1): one 2x2 pool after the first conv)."""

    def __init__(self, in_channels=2, channels=(2, 4, 8), kernel_sizes=(5, 4, 3), strides=(2, 1, 1), paddings=None, nonlinearity="relu",
                 use_maxpool=True, normlayer=None):
        super().__init__()
        n = len(channels)
        paddings = [0] * n if paddings is None else paddings
        assert n == len(kernel_sizes) == len(strides) == len(paddings)
        act = get_activation(nonlinearity)
        norm = getattr(nn, normlayer) if isinstance(normlayer, str) else normlayer
        conv_strides = [1] * n if use_maxpool else list(strides)
        pool_strides = list(strides) if use_maxpool else [1] * n
        ins = [in_channels] + list(channels)[:-1]
        convs = [Conv2dAsGemm(in_channels=i, out_channels=o, kernel_size=k, stride=s, padding=p)
                 for i, o, k, s, p in zip(ins, channels, kernel_sizes, conv_strides, paddings)]  # all convs first: the RNG order
        seq: list[nn.Module] = []
        for conv, o, ps in zip(convs, channels, pool_strides):
            seq += [conv] + ([norm(o)] if norm is not None else []) + [act]
            if ps > 1:
                seq.append(nn.MaxPool2d(ps))
        self.conv = nn.Sequential(*seq)

    def forward(self, x):
        return self.conv(x)

    def conv_out_size(self, h, w, c=None):
        for m in self.conv.children():
            if isinstance(m, (nn.Conv2d, nn.MaxPool2d)):
                h, w = conv2d_output_shape(h, w, m.kernel_size, m.stride, m.padding)
            if isinstance(m, nn.Conv2d):
                c = m.out_channels
        return h * w * c

    def reset(self, dones=None):
        pass


class CNN2dHead(nn.Module):
    def __init__(self, image_shape, channels=(2, 4, 8), kernel_sizes=(5, 4, 3), strides=(2, 1, 1), paddings=None, hidden_sizes=None,
                 output_size=None, nonlinearity="relu", use_maxpool=False, normlayer=None):
        super().__init__()
        c, h, w = image_shape
        self.conv = CNN2d(in_channels=c, channels=channels, kernel_sizes=kernel_sizes, strides=strides, paddings=paddings,
                          nonlinearity=nonlinearity, use_maxpool=use_maxpool, normlayer=normlayer)
        flat = self.conv.conv_out_size(h, w)
        if hidden_sizes or output_size:
            self.head = MLP(flat, hidden_sizes, output_size, activation=nonlinearity)
            self._output_size = output_size if output_size is not None else (hidden_sizes if isinstance(hidden_sizes, int) else hidden_sizes[-1])
        else:
            self.head = lambda x: x
            self._output_size = flat

    def forward(self, x):
        return self.head(self.conv(x).reshape(x.shape[0], -1))

    @property
    def output_size(self):
        return self._output_size

    def reset(self, dones=None):
        pass


def generate_model(input_dim, output_dim: int, cfg):
    """Reference loco_rl/loco_rl/models/model_generation.py:3-22: a network from a `ModelCfg` record."""
    t = cfg.model_type
    if t == "MLP":
        return MLP(input_dim, cfg.hidden_dims, output_dim, cfg.activation, cfg.final_layer_activation)
    if t == "RNN":
        return RNN(input_dim, cfg.hidden_dims, output_dim, cfg.activation, cfg.rnn_type, cfg.rnn_hidden_size, cfg.rnn_num_layers)
    if t == "CNN2d":
        return CNN2d(input_dim, cfg.cnn_channels, cfg.cnn_kernel_size, cfg.cnn_stride, cfg.cnn_padding, cfg.cnn_nonlinearity,
                     cfg.cnn_use_maxpool, cfg.cnn_normlayer)
    if t == "CNN2dHead":
        return CNN2dHead(cfg.img_shape, cfg.cnn_channels, cfg.cnn_kernel_size, cfg.cnn_stride, cfg.cnn_padding, cfg.hidden_dims, output_dim,
                         cfg.cnn_nonlinearity, cfg.cnn_use_maxpool, cfg.cnn_normlayer)
    raise NotImplementedError(f"Model type {t} not implemented")
