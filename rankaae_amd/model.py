"""Parameter containers for the networks the HIP engine trains.

These ``nn.Module``s exist for three reasons only:
  * they own the parameters/buffers (as views into the engine's flat arena) under the
    SAME sub-module names and ``state_dict`` keys as the reference's classes
    (``sc/clustering/model.py``), so ``final.pt`` keeps the reference's output format;
  * constructing them in the reference's order draws the same initial weights from the
    global torch generator (seed parity);
  * ``forward`` is a plain-PyTorch *inference* convenience for consumers of ``final.pt``
    (the reference's report tool calls ``model["Encoder"](spec)``).
The training and validation hot path never calls these ``forward`` methods -- it runs
the HIP kernels through ``rankaae_amd.engine`` and fails loudly without them.
"""
import math

import torch
from torch import nn


class GradientReversal(torch.autograd.Function):
    """Identity forward, ``-beta * grad`` backward (reference ``model.py:8-22``)."""

    @staticmethod
    def forward(ctx, x, beta):
        ctx.beta = beta
        return x

    @staticmethod
    def backward(ctx, grad):
        return (grad.clone() if ctx.beta is None else -grad * ctx.beta), None


def _final_activation(name):
    table = {"ReLu": nn.ReLU, "Softplus": lambda: nn.Softplus(beta=2)}
    if name not in table:
        raise ValueError(f"Unknow activation function \"{name}\", please use one available in Pytorch")
    return table[name]()


def _mlp(dims, dropout_rate, batchnorm):
    """[Linear, PReLU, (BatchNorm1d), Dropout] per hidden layer; Sequential indices match
    the reference so ``main.<i>.weight`` keys line up."""
    mods = []
    for d_in, d_out in zip(dims[:-1], dims[1:]):
        mods += [nn.Linear(d_in, d_out), nn.PReLU(num_parameters=d_out, init=0.01)]
        if batchnorm:
            mods.append(nn.BatchNorm1d(d_out, affine=False))
        mods.append(nn.Dropout(p=dropout_rate))
    return mods


class FCEncoder(nn.Module):
    """Dense encoder, reference ``model.py:330-378``."""

    def __init__(self, dropout_rate=0.2, nstyle=5, dim_in=256, n_layers=3, hidden_size=64):
        super().__init__()
        mods = _mlp([dim_in] + [hidden_size] * (n_layers - 1), dropout_rate, True)
        mods += [nn.Linear(hidden_size, nstyle), nn.BatchNorm1d(nstyle, affine=False)]
        self.main = nn.Sequential(*mods)

    def forward(self, spec):
        return self.main(spec)


class FCDecoder(nn.Module):
    """Dense decoder, reference ``model.py:518-570``."""

    def __init__(self, dropout_rate=0.2, nstyle=5, debug=False, dim_out=256, last_layer_activation="ReLu",
                 n_layers=3, hidden_size=64):
        super().__init__()
        act = _final_activation(last_layer_activation)
        mods = _mlp([nstyle] + [hidden_size] * (n_layers - 1), dropout_rate, True)
        mods += [nn.Linear(hidden_size, dim_out), act]
        self.main = nn.Sequential(*mods)
        self.nstyle = nstyle
        self.debug = debug

    def forward(self, z_gauss):
        return self.main(z_gauss)


class DiscriminatorFC(nn.Module):
    """Gaussian-prior discriminator, reference ``model.py:631-663``."""

    def __init__(self, hiden_size=64, dropout_rate=0.2, nstyle=5, noise=0.1, layers=3):
        super().__init__()
        mods = _mlp([nstyle] + [hiden_size] * (layers - 1), dropout_rate, False)
        mods.append(nn.Linear(hiden_size, 1))
        self.main = nn.Sequential(*mods)
        self.nstyle = nstyle
        self.noise = noise

    def forward(self, x, beta):
        if self.training:
            x = x + self.noise * torch.randn_like(x, requires_grad=False)
        return self.main(GradientReversal.apply(x, beta))


class _ConvBlockBase(nn.Module):
    """Shared body of the two residual block types: main(conv-conv) + shortcut + excitation."""

    def _excitation(self, ci, co, in_len, out_len, excitation, dropout_rate):
        self.dropout_1 = nn.Dropout(p=dropout_rate) if in_len > 10 else None
        self.fc1 = nn.Linear(in_len, excitation)
        self.relu_excit_1 = nn.PReLU(num_parameters=ci, init=0.01)
        self.fc2 = nn.Linear(excitation, out_len)
        self.relu_excit_2 = nn.PReLU(num_parameters=ci, init=0.01)
        if ci != co:
            self.bn_excit = nn.BatchNorm1d(ci, affine=False)
            self.relu_excit_3 = nn.PReLU(num_parameters=co, init=0.01)
            self.conv_excit = nn.Conv1d(ci, co, kernel_size=1, stride=1, groups=math.gcd(ci, co))
        else:
            self.bn_excit = self.relu_excit_3 = self.conv_excit = None

    def forward(self, x):
        r = x if self.bn1 is None else self.bn1(x)
        y = self.relu2(self.conv2(self.bn2(self.relu1(self.conv1(r)))))
        s = r if self.conv_short is None else self.relu_short(self.conv_short(r))
        e = r if self.dropout_1 is None else self.dropout_1(r)
        e = self.relu_excit_2(self.fc2(self.relu_excit_1(self.fc1(e))))
        if self.conv_excit is not None:
            e = self.relu_excit_3(self.conv_excit(self.bn_excit(e)))
        return y + s + e


class EncodingBlock(_ConvBlockBase):
    """Strided Conv1d residual block, reference ``model.py:24-100``."""

    def __init__(self, in_channels, out_channels, in_len, out_len, kernel_size=7, stride=2, excitation=4,
                 dropout_rate=0.2):
        super().__init__()
        ci, co, k = in_channels, out_channels, kernel_size
        self.bn1 = nn.BatchNorm1d(ci, affine=False) if ci > 1 else None
        self.relu1 = nn.PReLU(num_parameters=co, init=0.01)
        self.conv1 = nn.Conv1d(ci, co, kernel_size=k, padding=(k - 1) // 2, padding_mode="replicate",
                               stride=in_len // (out_len * stride))
        self.bn2 = nn.BatchNorm1d(co, affine=False)
        self.relu2 = nn.PReLU(num_parameters=co, init=0.01)
        self.conv2 = nn.Conv1d(co, co, kernel_size=k, padding=(k - 1) // 2, stride=stride)
        self._excitation(ci, co, in_len, out_len, excitation, dropout_rate)
        if stride > 1 or ci != co:
            q = in_len // out_len
            self.conv_short = nn.Conv1d(ci, co, kernel_size=q, stride=q, groups=math.gcd(ci, co))
            self.relu_short = nn.PReLU(num_parameters=co, init=0.01)
        else:
            self.conv_short = None


class DecodingBlock(_ConvBlockBase):
    """Transposed-conv (kernel == stride) residual block, reference ``model.py:103-174``."""

    def __init__(self, in_channels, out_channels, in_len, excitation=4, dropout_rate=0.2, out_len=None):
        super().__init__()
        ci, co = in_channels, out_channels
        out_len = in_len * 4 if out_len is None else out_len
        self.bn1 = nn.BatchNorm1d(ci, affine=False) if in_len > 1 else None
        self.relu1 = nn.PReLU(num_parameters=co, init=0.01)
        self.conv1 = nn.ConvTranspose1d(ci, co, kernel_size=2, stride=2)
        self.bn2 = nn.BatchNorm1d(co, affine=False)
        self.relu2 = nn.PReLU(num_parameters=co, init=0.01)
        q2 = out_len // (in_len * 2)
        self.conv2 = nn.ConvTranspose1d(co, co, kernel_size=q2, stride=q2)
        self._excitation(ci, co, in_len, out_len, excitation, dropout_rate)
        q = out_len // in_len
        self.conv_short = nn.ConvTranspose1d(ci, co, kernel_size=q, stride=q, groups=math.gcd(ci, co))
        self.relu_short = nn.PReLU(num_parameters=co, init=0.01)


class CompactEncoder(nn.Module):
    """1-D conv encoder, reference ``model.py:264-295``."""

    def __init__(self, dropout_rate=0.2, nstyle=5, dim_in=256, n_layers=3):
        super().__init__()
        d = dict(dropout_rate=dropout_rate)
        self.main = nn.Sequential(
            EncodingBlock(1, 4, dim_in, 64, kernel_size=11, stride=2, excitation=4, **d),
            EncodingBlock(4, 4, 64, 16, kernel_size=7, stride=2, excitation=2, **d),
            EncodingBlock(4, 4, 16, 8, kernel_size=5, stride=2, excitation=1, **d))
        self.lin3 = nn.Linear(32, nstyle)
        self.bn_style = nn.BatchNorm1d(nstyle, affine=False)

    def forward(self, spec):
        h = self.main(spec.unsqueeze(dim=1))
        return self.bn_style(self.lin3(h.reshape(spec.size(0), 32)))


class CompactDecoder(nn.Module):
    """1-D conv decoder, reference ``model.py:430-474``."""

    def __init__(self, dropout_rate=0.2, nstyle=5, debug=False, last_layer_activation="ReLu", dim_out=256,
                 n_layers=3):
        super().__init__()
        act = _final_activation(last_layer_activation)
        d = dict(dropout_rate=dropout_rate)
        self.main = nn.Sequential(
            DecodingBlock(nstyle, 8, 1, excitation=1, out_len=8, **d),
            DecodingBlock(8, 4, 8, excitation=2, out_len=64, **d),
            DecodingBlock(4, 4, 64, excitation=4, **d),
            EncodingBlock(4, 4, 256, dim_out, kernel_size=11, stride=1, excitation=2, **d),
            nn.BatchNorm1d(4, affine=False), nn.Conv1d(4, 1, kernel_size=1, stride=1), act)
        self.nstyle = nstyle
        self.debug = debug

    def forward(self, z_gauss):
        return self.main(z_gauss.unsqueeze(dim=2)).squeeze(dim=1)


AE_CLS_DICT = {
    "compact": {"encoder": CompactEncoder, "decoder": CompactDecoder},
    "FC": {"encoder": FCEncoder, "decoder": FCDecoder},
}
