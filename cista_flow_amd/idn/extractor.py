"""Parameter holder of IDNet's LiteEncoder (reference: idn/extractor.py:5-125; norm-free residual blocks).
Arithmetic: csrc/cf_api.hip::idnet_forward (all B*5 bins encoded in one batch)."""
import torch.nn as nn


class ResidualBlock(nn.Module):
    def __init__(self, in_planes, planes, norm_fn='group', stride=1):
        super(ResidualBlock, self).__init__()
        if norm_fn != 'none':
            raise NotImplementedError("IDNet's LiteEncoder uses norm_fn='none'")
        self.conv1 = nn.Conv2d(in_planes, planes, kernel_size=3, padding=1, stride=stride)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, padding=1)
        self.relu = nn.ReLU(inplace=True)
        self.norm1 = nn.Sequential()
        self.norm2 = nn.Sequential()
        if not stride == 1:
            self.norm3 = nn.Sequential()
        if stride == 1:
            self.downsample = None
        else:
            self.downsample = nn.Sequential(nn.Conv2d(in_planes, planes, kernel_size=1, stride=stride), self.norm3)


class LiteEncoder(nn.Module):
    def __init__(self, output_dim=32, stride=2, dropout=0.0, n_first_channels=1):
        super(LiteEncoder, self).__init__()
        if stride != 2:
            raise NotImplementedError("only downsample=8 (stride=2) is built")
        self.conv1 = nn.Conv2d(n_first_channels, output_dim, kernel_size=7, stride=2, padding=3)
        self.relu1 = nn.ReLU(inplace=True)
        self.in_planes = output_dim
        self.layer1 = self._make_layer(output_dim, stride=2)
        self.layer2 = self._make_layer(output_dim * 2, stride=2)
        self.dropout = None
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')

    def _make_layer(self, dim, stride=1):
        layer1 = ResidualBlock(self.in_planes, dim, 'none', stride=stride)
        layer2 = ResidualBlock(dim, dim, 'none', stride=1)
        self.in_planes = dim
        return nn.Sequential(layer1, layer2)
