"""TEST INFRASTRUCTURE ONLY (see oracle/README or DESIGN.md §6): a CPU restatement, on plain torch tensor algebra, of the
sub-pixel identities that csrc/u3d_subpix.hip builds on.  Nothing under pytorch-3dunet_amd/ imports this module.

The reference computes, for the upsampled half of a decoder's first convolution (buildingblocks.py:491 torch.cat,
:614 F.interpolate(mode='nearest'), :56 nn.Conv3d(k=3, padding=1)):

    y = conv3d(nearest2x(low), w)                                   full-res, 27 taps per output voxel

Per dimension, output voxel 2j + p reads low-res voxels j + p - 1 + e, e in {0, 1}, with the taps that hit the same low-res
voxel summed:  p=0: e=0 <- {t0}, e=1 <- {t1, t2};   p=1: e=0 <- {t0, t1}, e=1 <- {t2}.
"""
import itertools

import torch
import torch.nn.functional as F

# taps of parity p that read low-res offset (p - 1 + e)
TAPS = {(0, 0): (0,), (0, 1): (1, 2), (1, 0): (0, 1), (1, 1): (2,)}


def presum_weights(w):
    """w (Cout, C1, 3, 3, 3) -> dict[(pz,py,px)] = (Cout, C1, 2, 2, 2): the 2x2x2 kernel of every output parity class"""
    out = {}
    for p in itertools.product((0, 1), repeat=3):
        k = torch.zeros(w.shape[0], w.shape[1], 2, 2, 2, dtype=w.dtype)
        for e in itertools.product((0, 1), repeat=3):
            for tz in TAPS[(p[0], e[0])]:
                for ty in TAPS[(p[1], e[1])]:
                    for tx in TAPS[(p[2], e[2])]:
                        k[:, :, e[0], e[1], e[2]] += w[:, :, tz, ty, tx]
        out[p] = k
    return out


def _shifted(low, p):
    """zero-padded low-res tensor such that a VALID 2x2x2 correlation yields, at j, the sum over e of low[j + p - 1 + e]"""
    pads = []
    for d in (2, 1, 0):  # F.pad order: last dimension first
        pads += [1 - p[d], p[d]]
    return F.pad(low, pads)


def forward(low, w):
    """8 parity-class 2x2x2 convolutions over the low-res grid, interleaved into the full-res output"""
    N, C1, D1, H1, W1 = low.shape
    y = torch.zeros(N, w.shape[0], 2 * D1, 2 * H1, 2 * W1, dtype=low.dtype)
    for p, k in presum_weights(w).items():
        y[:, :, p[0]::2, p[1]::2, p[2]::2] = F.conv3d(_shifted(low, p), k)
    return y


def dgrad_low(dz, w):
    """gradient with respect to `low` (children sum of the nearest upsampling included): adjoint of forward()"""
    N, K, D, H, W = dz.shape
    C1 = w.shape[1]
    dlow = torch.zeros(N, C1, D // 2, H // 2, W // 2, dtype=dz.dtype)
    for p, k in presum_weights(w).items():
        g = F.conv_transpose3d(dz[:, :, p[0]::2, p[1]::2, p[2]::2], k)  # gradient of the padded tensor
        D1, H1, W1 = dlow.shape[2:]
        dlow += g[:, :, 1 - p[0]:1 - p[0] + D1, 1 - p[1]:1 - p[1] + H1, 1 - p[2]:1 - p[2] + W1]
    return dlow


def wgrad(low, dz):
    """dw (Cout, C1, 3,3,3) from the 64 (class, tap half) matrices, folded 8 per tap"""
    N, C1, D1, H1, W1 = low.shape
    K = dz.shape[1]
    dw = torch.zeros(K, C1, 3, 3, 3, dtype=low.dtype)
    for p in itertools.product((0, 1), repeat=3):
        dzp = dz[:, :, p[0]::2, p[1]::2, p[2]::2]
        lp = _shifted(low, p)
        for e in itertools.product((0, 1), repeat=3):
            a = lp[:, :, e[0]:e[0] + D1, e[1]:e[1] + H1, e[2]:e[2] + W1]
            m = torch.einsum("nkzyx,nczyx->kc", dzp, a)  # dWc[p][e]
            for tz in TAPS[(p[0], e[0])]:
                for ty in TAPS[(p[1], e[1])]:
                    for tx in TAPS[(p[2], e[2])]:
                        dw[:, :, tz, ty, tx] += m
    return dw
