"""Building blocks of the U-Net variants — the module tree (and therefore the `state_dict` keys) of the
reference's pytorch3dunet/unet3d/buildingblocks.py, re-expressed for the MI355X path.

These modules are PARAMETER CONTAINERS first: on a gfx950 device the enclosing `AbstractUNet` does not call
them — it hands their parameters to the fused HIP executor (pytorch3dunet_amd/engine.py).  Their own
`forward`s are the ordinary torch.nn semantics and are what runs for CPU tensors (`device: cpu` in the YAML) and
for configurations the native executor does not cover yet.

Reference citations (file:line under /root/reference/pytorch3dunet/unet3d/):
  layer-order mini language ............ buildingblocks.py:10-96
  SingleConv / DoubleConv .............. :99-135 / :138-227
  ResNetBlock / ResNetBlockSE .......... :230-288 / :291-307
  Encoder / Decoder .................... :310-384 / :387-493
  create_encoders / create_decoders .... :496-544 / :547-574
  upsampling classes ................... :577-675
"""
from functools import partial

import torch
from torch import nn
from torch.nn import functional as F

from .se import ChannelSELayer3D, ChannelSpatialSELayer3D, SpatialSELayer3D

_NONLINEAR = "rle"


def _activation(char):
    if char == "r":
        return "ReLU", nn.ReLU(inplace=True)
    if char == "l":
        return "LeakyReLU", nn.LeakyReLU(inplace=True)
    return "ELU", nn.ELU(inplace=True)


def create_conv(in_channels, out_channels, kernel_size, order, num_groups, padding, dropout_prob, is3d):
    """List of (name, module) for one conv layer described by `order` (g c r l e b d D), buildingblocks.py:10-96.

    Semantics kept from the reference: GroupNorm/BatchNorm act on the conv INPUT channels when they precede 'c';
    GroupNorm falls back to a single group when channels < num_groups; the conv has a bias only when no norm
    layer is present; an unknown character raises ValueError."""
    assert "c" in order, "Conv layer MUST be present"
    assert order[0] not in _NONLINEAR, "Non-linearity cannot be the first operation in the layer"
    conv_at = order.index("c")
    has_norm = ("g" in order) or ("b" in order)
    conv_cls = nn.Conv3d if is3d else nn.Conv2d
    bn_cls = nn.BatchNorm3d if is3d else nn.BatchNorm2d

    layers = []
    for pos, char in enumerate(order):
        norm_channels = in_channels if pos < conv_at else out_channels
        if char in _NONLINEAR:
            layers.append(_activation(char))
        elif char == "c":
            layers.append(("conv", conv_cls(in_channels, out_channels, kernel_size, padding=padding, bias=not has_norm)))
        elif char == "g":
            groups = num_groups if norm_channels >= num_groups else 1
            assert norm_channels % groups == 0, (
                f"Expected number of channels in input to be divisible by num_groups. "
                f"num_channels={norm_channels}, num_groups={groups}"
            )
            layers.append(("groupnorm", nn.GroupNorm(num_groups=groups, num_channels=norm_channels)))
        elif char == "b":
            layers.append(("batchnorm", bn_cls(norm_channels)))
        elif char == "d":
            layers.append(("dropout", nn.Dropout(p=dropout_prob)))
        elif char == "D":
            layers.append(("dropout2d", nn.Dropout2d(p=dropout_prob)))
        else:
            raise ValueError(f"Unsupported layer type '{char}'. MUST be one of ['b', 'g', 'r', 'l', 'e', 'c', 'd', 'D']")
    return layers


class SingleConv(nn.Sequential):
    """One conv + optional norm / non-linearity / dropout in the given order (buildingblocks.py:99-135)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, order="gcr", num_groups=8, padding=1,
                 dropout_prob=0.1, is3d=True):
        super().__init__()
        self.order = order
        for name, module in create_conv(in_channels, out_channels, kernel_size, order, num_groups, padding,
                                        dropout_prob, is3d):
            self.add_module(name, module)


class DoubleConv(nn.Sequential):
    """Two SingleConvs (buildingblocks.py:138-227).  Encoder: in -> max(out//2, in) -> out (out//2 skipped when
    upscale == 1); decoder: in -> out -> out."""

    def __init__(self, in_channels, out_channels, encoder, kernel_size=3, order="gcr", num_groups=8, padding=1,
                 upscale=2, dropout_prob=0.1, is3d=True):
        super().__init__()
        if encoder:
            mid = out_channels if upscale == 1 else out_channels // 2
            mid = max(mid, in_channels)
        else:
            mid = out_channels
        if isinstance(dropout_prob, (list, tuple)):
            drop1, drop2 = dropout_prob[0], dropout_prob[1]
        else:
            drop1 = drop2 = dropout_prob
        self.add_module("SingleConv1", SingleConv(in_channels, mid, kernel_size, order, num_groups, padding=padding,
                                                  dropout_prob=drop1, is3d=is3d))
        self.add_module("SingleConv2", SingleConv(mid, out_channels, kernel_size, order, num_groups, padding=padding,
                                                  dropout_prob=drop2, is3d=is3d))


class ResNetBlock(nn.Module):
    """Residual block (buildingblocks.py:230-288): optional 1x1 conv to out_channels, two SingleConvs (the second
    without non-linearity), in-place residual add, then the non-linearity (LeakyReLU slope 0.1 / ELU / ReLU)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, order="cge", num_groups=8, is3d=True, **kwargs):
        super().__init__()
        if in_channels != out_channels:
            self.conv1 = (nn.Conv3d if is3d else nn.Conv2d)(in_channels, out_channels, 1)
        else:
            self.conv1 = nn.Identity()
        self.conv2 = SingleConv(out_channels, out_channels, kernel_size=kernel_size, order=order, num_groups=num_groups,
                                is3d=is3d)
        linear_order = "".join(ch for ch in order if ch not in _NONLINEAR)
        self.conv3 = SingleConv(out_channels, out_channels, kernel_size=kernel_size, order=linear_order,
                                num_groups=num_groups, is3d=is3d)
        if "l" in order:
            self.non_linearity = nn.LeakyReLU(negative_slope=0.1, inplace=True)
        elif "e" in order:
            self.non_linearity = nn.ELU(inplace=True)
        else:
            self.non_linearity = nn.ReLU(inplace=True)

    def forward(self, x):
        residual = self.conv1(x)
        out = self.conv3(self.conv2(residual))
        out += residual
        return self.non_linearity(out)


class ResNetBlockSE(ResNetBlock):
    """ResNetBlock followed by a squeeze-and-excitation gate (buildingblocks.py:291-307)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, order="cge", num_groups=8, se_module="scse", **kwargs):
        super().__init__(in_channels, out_channels, kernel_size=kernel_size, order=order, num_groups=num_groups, **kwargs)
        assert se_module in ["scse", "cse", "sse"]
        if se_module == "scse":
            self.se_module = ChannelSpatialSELayer3D(num_channels=out_channels, reduction_ratio=1)
        elif se_module == "cse":
            self.se_module = ChannelSELayer3D(num_channels=out_channels, reduction_ratio=1)
        else:
            self.se_module = SpatialSELayer3D(num_channels=out_channels)

    def forward(self, x):
        return self.se_module(super().forward(x))


class Encoder(nn.Module):
    """[pooling] -> basic_module (buildingblocks.py:310-384)."""

    def __init__(self, in_channels, out_channels, conv_kernel_size=3, apply_pooling=True, pool_kernel_size=2,
                 pool_type="max", basic_module=DoubleConv, conv_layer_order="gcr", num_groups=8, padding=1, upscale=2,
                 dropout_prob=0.1, is3d=True):
        super().__init__()
        assert pool_type in ["max", "avg"]
        self.pooling = None
        if apply_pooling:
            pools = {("max", True): nn.MaxPool3d, ("max", False): nn.MaxPool2d, ("avg", True): nn.AvgPool3d,
                     ("avg", False): nn.AvgPool2d}
            self.pooling = pools[(pool_type, bool(is3d))](kernel_size=pool_kernel_size)
        self.basic_module = basic_module(in_channels, out_channels, encoder=True, kernel_size=conv_kernel_size,
                                         order=conv_layer_order, num_groups=num_groups, padding=padding, upscale=upscale,
                                         dropout_prob=dropout_prob, is3d=is3d)

    def forward(self, x):
        if self.pooling is not None:
            x = self.pooling(x)
        return self.basic_module(x)


class AbstractUpsampling(nn.Module):
    """Upsample `x` to the spatial size of `encoder_features` (buildingblocks.py:577-595)."""

    def __init__(self, upsample):
        super().__init__()
        self.upsample = upsample

    def forward(self, encoder_features, x):
        return self.upsample(x, encoder_features.size()[2:])


class InterpolateUpsampling(AbstractUpsampling):
    """F.interpolate to the skip's size (buildingblocks.py:598-614)."""

    def __init__(self, mode="nearest"):
        super().__init__(partial(self._interpolate, mode=mode))
        self.mode = mode  # read by the native executor (plain attribute: no parameters, the state_dict is unchanged)

    @staticmethod
    def _interpolate(x, size, mode):
        return F.interpolate(x, size=size, mode=mode)


class TransposeConvUpsampling(AbstractUpsampling):
    """ConvTranspose(k=3, stride=scale, pad=1, no bias) -> 2n-1 outputs -> nearest resize to the skip's size
    (buildingblocks.py:617-664).  Parameter path: upsampling.upsample.conv_transposed.weight."""

    class Upsample(nn.Module):
        def __init__(self, conv_transposed, is3d):
            super().__init__()
            self.conv_transposed = conv_transposed
            self.is3d = is3d

        def forward(self, x, size):
            return F.interpolate(self.conv_transposed(x), size=size)

    def __init__(self, in_channels, out_channels, kernel_size=3, scale_factor=2, is3d=True):
        ct = nn.ConvTranspose3d if is3d is True else nn.ConvTranspose2d
        super().__init__(self.Upsample(ct(in_channels, out_channels, kernel_size=kernel_size, stride=scale_factor,
                                          padding=1, bias=False), is3d))


class NoUpsampling(AbstractUpsampling):
    """identity (buildingblocks.py:667-675)."""

    def __init__(self):
        super().__init__(self._no_upsampling)

    @staticmethod
    def _no_upsampling(x, size):
        return x


class Decoder(nn.Module):
    """upsample -> join with the skip (concat for DoubleConv, sum for residual blocks) -> basic_module
    (buildingblocks.py:387-493)."""

    def __init__(self, in_channels, out_channels, conv_kernel_size=3, scale_factor=2, basic_module=DoubleConv,
                 conv_layer_order="gcr", num_groups=8, padding=1, upsample="default", dropout_prob=0.1, is3d=True):
        super().__init__()
        concat, adapt_channels = True, False
        if upsample is not None and upsample != "none":
            if upsample == "default":
                if basic_module == DoubleConv:
                    upsample, concat, adapt_channels = "nearest", True, False
                elif basic_module in (ResNetBlock, ResNetBlockSE):
                    upsample, concat, adapt_channels = "deconv", False, True
            if upsample == "deconv":
                self.upsampling = TransposeConvUpsampling(in_channels=in_channels, out_channels=out_channels,
                                                          kernel_size=conv_kernel_size, scale_factor=scale_factor,
                                                          is3d=is3d)
            else:
                self.upsampling = InterpolateUpsampling(mode=upsample)
        else:
            self.upsampling = NoUpsampling()
        self.concat = concat
        self.joining = partial(self._joining, concat=concat)
        if adapt_channels:
            in_channels = out_channels
        self.basic_module = basic_module(in_channels, out_channels, encoder=False, kernel_size=conv_kernel_size,
                                         order=conv_layer_order, num_groups=num_groups, padding=padding,
                                         dropout_prob=dropout_prob, is3d=is3d)

    def forward(self, encoder_features, x):
        x = self.upsampling(encoder_features=encoder_features, x=x)
        return self.basic_module(self.joining(encoder_features, x))

    @staticmethod
    def _joining(encoder_features, x, concat):
        # skip channels FIRST (buildingblocks.py:491)
        return torch.cat((encoder_features, x), dim=1) if concat else encoder_features + x


def create_encoders(in_channels, f_maps, basic_module, conv_kernel_size, conv_padding, conv_upscale, dropout_prob,
                    layer_order, num_groups, pool_kernel_size, is3d):
    """len(f_maps) encoders; the first one has no pooling (buildingblocks.py:496-544)."""
    encoders = []
    prev = in_channels
    for level, width in enumerate(f_maps):
        extra = {} if level == 0 else {"pool_kernel_size": pool_kernel_size}
        encoders.append(Encoder(prev, width, apply_pooling=level > 0, basic_module=basic_module,
                                conv_layer_order=layer_order, conv_kernel_size=conv_kernel_size, num_groups=num_groups,
                                padding=conv_padding, upscale=conv_upscale, dropout_prob=dropout_prob, is3d=is3d, **extra))
        prev = width
    return nn.ModuleList(encoders)


def create_decoders(f_maps, basic_module, conv_kernel_size, conv_padding, layer_order, num_groups, upsample,
                    dropout_prob, is3d):
    """len(f_maps)-1 decoders; DoubleConv decoders take skip + upsampled channels (buildingblocks.py:547-574)."""
    widths = list(reversed(f_maps))
    decoders = []
    for deep, shallow in zip(widths[:-1], widths[1:]):
        concat_in = basic_module == DoubleConv and upsample != "deconv"
        decoders.append(Decoder(deep + shallow if concat_in else deep, shallow, basic_module=basic_module,
                                conv_layer_order=layer_order, conv_kernel_size=conv_kernel_size, num_groups=num_groups,
                                padding=conv_padding, upsample=upsample, dropout_prob=dropout_prob, is3d=is3d))
    return nn.ModuleList(decoders)
