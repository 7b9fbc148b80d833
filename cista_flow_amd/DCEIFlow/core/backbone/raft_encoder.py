"""Parameter holders of the RAFT-style encoder (reference: DCEIFlow/core/backbone/raft_encoder.py:6-59,125-203).

norm_fn 'instance' -> nn.InstanceNorm2d (no parameters, statistics computed per call by
csrc/pointwise.hip), 'batch' -> nn.BatchNorm2d (eval statistics folded into the packed conv weights).
The arithmetic lives in csrc/cf_api.hip::encoder_forward.
"""
import torch.nn as nn


def _norm(norm_fn, planes):
    if norm_fn == 'batch':
        return nn.BatchNorm2d(planes)
    if norm_fn == 'instance':
        return nn.InstanceNorm2d(planes)
    raise NotImplementedError("CISTA-Flow's DCEIFlow uses norm_fn 'instance' (fnet/enet) and 'batch' (cnet) only")


class ResidualBlock(nn.Module):
    def __init__(self, in_planes, planes, norm_fn='group', stride=1):
        super(ResidualBlock, self).__init__()
        self.conv1 = nn.Conv2d(in_planes, planes, kernel_size=3, padding=1, stride=stride)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, padding=1)
        self.relu = nn.ReLU(inplace=True)
        self.norm1 = _norm(norm_fn, planes)
        self.norm2 = _norm(norm_fn, planes)
        if not stride == 1:
            self.norm3 = _norm(norm_fn, planes)
        if stride == 1:
            self.downsample = None
        else:
            self.downsample = nn.Sequential(nn.Conv2d(in_planes, planes, kernel_size=1, stride=stride), self.norm3)

    def forward(self, x):
        raise RuntimeError("ResidualBlock is a parameter holder; the encoder runs as fused HIP kernels")


class BasicEncoder(nn.Module):
    def __init__(self, ds=8, input_dim=3, output_dim=128, norm_fn='batch', dropout=0.0):
        super(BasicEncoder, self).__init__()
        if ds != 8:
            raise NotImplementedError("only ds=8 (utils/configs.py:25 default) is built")
        self.norm_fn = norm_fn
        self.norm1 = _norm(norm_fn, 64)
        self.conv1 = nn.Conv2d(input_dim, 64, kernel_size=7, stride=2, padding=3)
        self.relu1 = nn.ReLU(inplace=True)
        self.in_planes = 64
        self.layer1 = self._make_layer(64, stride=1)
        self.layer2 = self._make_layer(96, stride=2)
        self.layer3 = self._make_layer(128, stride=2)
        self.conv2 = nn.Conv2d(128, output_dim, kernel_size=1)
        self.dropout = None
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, (nn.BatchNorm2d, nn.InstanceNorm2d)):
                if m.weight is not None:
                    nn.init.constant_(m.weight, 1)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def _make_layer(self, dim, stride=1):
        layer1 = ResidualBlock(self.in_planes, dim, self.norm_fn, stride=stride)
        layer2 = ResidualBlock(dim, dim, self.norm_fn, stride=1)
        self.in_planes = dim
        return nn.Sequential(layer1, layer2)

    def forward(self, x):
        raise RuntimeError("BasicEncoder is a parameter holder; DCEIFlow.forward runs it as fused HIP kernels")
