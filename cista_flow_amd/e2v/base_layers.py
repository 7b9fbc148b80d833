"""Parameter holders of the CISTA-LSTC layer library (reference: e2v/base_layers.py).

These classes keep the reference's constructor signatures, attribute names and state_dict layout
(so checkpoints load unchanged) but hold NO arithmetic: the parent network (e2v_model.CistaLSTCNet)
hands their parameters to libcistaflow, where each layer is a fused HIP kernel launch.  Calling
`forward` on a holder raises -- there is deliberately no PyTorch fallback.
"""
import torch
import torch.nn as nn
from torch.nn import Parameter


def _no_eager(name):
    raise RuntimeError(
        "%s is a parameter holder in cista_flow_amd: its arithmetic runs inside the fused HIP graph of the "
        "parent network (CistaLSTCNet.forward); there is no eager PyTorch path" % name)


def softshrink(x, lambd):
    """base_layers.py:11-12 -- fused into the P-conv epilogue (EPI_ADD_AUX_SHRINK)."""
    _no_eager("softshrink")


def connect_cat(x1, x2):
    """base_layers.py:17-18 -- torch.cat never materialises: convs read channel segments."""
    _no_eager("connect_cat")


class ConvLayer(nn.Module):
    """base_layers.py:137-163 (norm is never enabled in CISTA-Flow)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, activation=None, norm=None, groups=1):
        super(ConvLayer, self).__init__()
        if norm is not None or groups != 1:
            raise NotImplementedError("CISTA-Flow never instantiates ConvLayer with norm/groups")
        self.conv2d = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, bias=True, padding_mode='reflect')
        self.activation = activation
        self.norm = norm

    def forward(self, x):
        _no_eager("ConvLayer")


class UpsampleConvLayer(nn.Module):
    """base_layers.py:168-212: bilinear x2 (align_corners=False) -> ReflectionPad2d -> conv."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, activation=None, norm=None):
        super(UpsampleConvLayer, self).__init__()
        if norm is not None:
            raise NotImplementedError("CISTA-Flow never instantiates UpsampleConvLayer with norm")
        self._kernel_size = kernel_size
        self.conv2d = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, bias=True)
        self.activation = activation
        self.norm = norm

    def forward(self, conv, out_dim=None):
        _no_eager("UpsampleConvLayer")


class ConvLSTM(nn.Module):
    """base_layers.py:75-132; gate chunk order: in, remember, out, cell."""

    def __init__(self, input_size, hidden_size, kernel_size):
        super(ConvLSTM, self).__init__()
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.zero_tensors = {}
        self.Gates = nn.Conv2d(input_size + hidden_size, 4 * hidden_size, kernel_size, padding=kernel_size // 2,
                               padding_mode='reflect')

    def forward(self, input_, prev_state=None):
        _no_eager("ConvLSTM")


class ConvLSTC(nn.Module):
    """base_layers.py:38-71."""

    def __init__(self, x_size, z_size, output_size, kernel_size):
        super(ConvLSTC, self).__init__()
        self.x_size = x_size
        self.z_size = z_size
        self.output_size = output_size
        pad = kernel_size // 2
        self.gates = nn.Conv2d(x_size + z_size, 2 * output_size, kernel_size, padding=pad, padding_mode='reflect')
        self.out_gates = nn.Conv2d(z_size + output_size, output_size, kernel_size, padding=pad, padding_mode='reflect')
        self.P0 = nn.Conv2d(x_size, output_size, kernel_size, padding=pad, padding_mode='reflect')

    def forward(self, x, z=None, prev_state=None):
        _no_eager("ConvLSTC")


class RecurrentConvLayer(nn.Module):
    """base_layers.py:216-227."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=0, activation=None, norm=None):
        super(RecurrentConvLayer, self).__init__()
        self.conv = ConvLayer(in_channels, out_channels, kernel_size, stride, padding, activation, norm)
        self.recurrent_block = ConvLSTM(input_size=out_channels, hidden_size=out_channels, kernel_size=3)

    def forward(self, x, prev_state):
        _no_eager("RecurrentConvLayer")


class IstaBlock(nn.Module):
    """base_layers.py:21-35: D (2c->c), P (c->2c), Lambda [1,2c,1,1]."""

    def __init__(self, base_channels=32, kernel_size=3, stride=1, padding=1, activation=None, norm=None, is_recurrent=False):
        super(IstaBlock, self).__init__()
        if is_recurrent:
            raise NotImplementedError("CISTA-Flow uses IstaBlock(is_recurrent=False)")
        self.D = ConvLayer(2 * base_channels, base_channels, kernel_size, stride, padding, activation, norm)
        self.P = ConvLayer(base_channels, 2 * base_channels, kernel_size, stride, padding, activation, norm)
        self.Lambda = Parameter(0.001 * torch.rand((1, 2 * base_channels, 1, 1)))
