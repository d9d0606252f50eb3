"""Squeeze-and-excitation gates used by ResNetBlockSE (reference pytorch3dunet/unet3d/se.py:18-114).
Parameter names (cSE.fc1/fc2, sSE.conv) follow the reference so checkpoints load unchanged."""
import torch
from torch import nn


class ChannelSELayer3D(nn.Module):
    """cSE: global average -> fc1 -> ReLU -> fc2 -> sigmoid -> per-channel scale (se.py:18-51)."""

    def __init__(self, num_channels, reduction_ratio=2):
        super().__init__()
        hidden = num_channels // reduction_ratio
        self.reduction_ratio = reduction_ratio
        self.avg_pool = nn.AdaptiveAvgPool3d(1)
        self.fc1 = nn.Linear(num_channels, hidden, bias=True)
        self.fc2 = nn.Linear(hidden, num_channels, bias=True)
        self.relu = nn.ReLU()
        self.sigmoid = nn.Sigmoid()

    def forward(self, x):
        n, c = x.shape[:2]
        gate = self.sigmoid(self.fc2(self.relu(self.fc1(self.avg_pool(x).view(n, c)))))
        return x * gate.view(n, c, 1, 1, 1)


class SpatialSELayer3D(nn.Module):
    """sSE: 1x1x1 conv C->1 -> sigmoid -> per-voxel scale (se.py:54-93)."""

    def __init__(self, num_channels):
        super().__init__()
        self.conv = nn.Conv3d(num_channels, 1, 1)
        self.sigmoid = nn.Sigmoid()

    def forward(self, x, weights=None):
        n, c, d, h, w = x.shape
        if weights:
            squeeze = torch.nn.functional.conv2d(x, weights.view(1, c, 1, 1))
        else:
            squeeze = self.conv(x)
        return x * self.sigmoid(squeeze).view(n, 1, d, h, w)


class ChannelSpatialSELayer3D(nn.Module):
    """scSE = elementwise max of the two gates (se.py:96-114)."""

    def __init__(self, num_channels, reduction_ratio=2):
        super().__init__()
        self.cSE = ChannelSELayer3D(num_channels, reduction_ratio)
        self.sSE = SpatialSELayer3D(num_channels)

    def forward(self, x):
        return torch.max(self.cSE(x), self.sSE(x))
