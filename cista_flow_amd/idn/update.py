"""Parameter holders of IDNet's update block (reference: idn/update.py:18-85)."""
import torch.nn as nn


class FlowHead(nn.Module):
    def __init__(self, input_dim=128, hidden_dim=256):
        super(FlowHead, self).__init__()
        self.conv1 = nn.Conv2d(input_dim, hidden_dim, 3, padding=1)
        self.conv2 = nn.Conv2d(hidden_dim, 2, 3, padding=1)
        self.relu = nn.ReLU(inplace=True)


class ConvGRU(nn.Module):
    def __init__(self, hidden_dim=128, input_dim=192 + 128):
        super(ConvGRU, self).__init__()
        self.convz = nn.Conv2d(hidden_dim + input_dim, hidden_dim, 3, padding=1)
        self.convr = nn.Conv2d(hidden_dim + input_dim, hidden_dim, 3, padding=1)
        self.convq = nn.Conv2d(hidden_dim + input_dim, hidden_dim, 3, padding=1)


class LiteUpdateBlock(nn.Module):
    def __init__(self, hidden_dim=32, input_dim=16, num_outputs=1, downsample=8):
        super(LiteUpdateBlock, self).__init__()
        self.upsample_mask_dim = downsample * downsample
        self.num_outputs = num_outputs
        assert self.num_outputs in [1, 2]
        self.gru = ConvGRU(hidden_dim=hidden_dim, input_dim=input_dim)
        self.flow_head = FlowHead(hidden_dim, hidden_dim=hidden_dim)
        self.mask = nn.Sequential(nn.Conv2d(hidden_dim, 256, 3, padding=1), nn.ReLU(inplace=True),
                                  nn.Conv2d(256, self.upsample_mask_dim * 9, 1, padding=0))
        if self.num_outputs == 2:
            self.flow_head2 = FlowHead(hidden_dim, hidden_dim=hidden_dim)
            self.mask2 = nn.Sequential(nn.Conv2d(hidden_dim, 256, 3, padding=1), nn.ReLU(inplace=True),
                                       nn.Conv2d(256, self.upsample_mask_dim * 9, 1, padding=0))
