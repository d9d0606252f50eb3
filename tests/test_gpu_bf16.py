"""-m gpu: the opt-in bf16-operand path (BASELINE config 4: "bf16 compute", fp32 master weights; csrc/u3d_bf16.hip).
Kernel level: u3d_conv3d_bf16 through the C-ABI against the SAME arithmetic restated on the CPU — operands rounded to bf16
(round-to-nearest-even, after the fp32 GroupNorm affine), products accumulated in fp32/fp64 — so the comparison is tight
(1e-3 of the output range covers the rare operand that rounds the other way after a 1-ulp difference in the affine)."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

import gpu_utils as U
from pytorch3dunet_amd import _native as nat
from pytorch3dunet_amd.engine import _p, _stream

pytestmark = pytest.mark.gpu


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


def pack_bf16(w, mode):
    Cout, Cin = w.shape[:2]
    n = nat.get_lib().u3d_packed_weight_bf16_elems(Cin, Cout, mode)
    assert n > 0
    out = torch.empty(n, dtype=torch.bfloat16, device=U.DEV)
    wd = w.contiguous().to(U.DEV)
    nat.call("u3d_pack_weights_bf16", 0, _stream(U.DEV), _p(wd), Cout, Cin, mode, _p(out))
    return out


def conv_bf16(x, w, mode=0, affine=None, relu=0, out_stats=None, gx=None, gstats=None, residual=None):
    """x: (N,C,D,H,W) cpu; returns (N,K,D,H,W) cpu"""
    N, C, D, H, W = x.shape
    K = w.shape[0] if mode == 0 else w.shape[1]
    wp = pack_bf16(w, mode)
    xd = U.ndhwc(x)
    y = torch.empty((N, D, H, W, K), dtype=torch.float32, device=U.DEV)
    nat.call("u3d_conv3d_bf16", 0, _stream(U.DEV), _p(xd), _p(affine), _p(wp), _p(y), N, D, H, W, C, K, relu, _p(out_stats),
             _p(gx), _p(gstats), _p(residual))
    torch.cuda.synchronize()
    return U.ncdhw(y)


@pytest.mark.parametrize("shape,C,K", [
    ((2, 9, 13, 17), 16, 32),     # ragged, one 32-channel n-tile, 4-plane tiles
    ((1, 16, 24, 24), 32, 64),    # 64 channels per block, 4-plane tiles
    ((1, 12, 20, 28), 48, 96),    # three chunks, 96 = 3 n-tiles of 32 (32 per block)
    ((1, 64, 64, 64), 16, 64),    # 512 blocks -> 8-plane tiles, 64 channels per block
    ((1, 64, 64, 64), 16, 32),    # 8-plane tiles, 32 channels per block
    ((1, 17, 70, 66), 32, 128),   # ragged 8-plane tiles (>= 512 blocks), two 64-channel blocks
])
def test_conv3d_bf16_forward_affine_relu_stats(shape, C, K):
    N, D, H, W = shape
    torch.manual_seed(1)
    x = torch.randn(N, C, D, H, W)
    w = torch.randn(K, C, 3, 3, 3) / (27 * C) ** 0.5
    a = 1.0 + 0.3 * torch.randn(N, C)
    b = 0.2 * torch.randn(N, C)
    g = (x.double() * a.double().view(N, C, 1, 1, 1) + b.double().view(N, C, 1, 1, 1)).float()
    ref = F.conv3d(bf16_round(g).double(), bf16_round(w).double(), None, padding=1)
    aff = torch.stack((a, b), dim=-1).contiguous().to(U.DEV)
    stats = torch.zeros((N, K, 2), dtype=torch.float64, device=U.DEV)
    y = conv_bf16(x, w, 0, affine=aff, relu=1, out_stats=stats)
    ref_r = ref.clamp_min(0)
    scale = ref.abs().max().item()
    assert (y.double() - ref_r).abs().max().item() < 1e-3 * scale
    st = stats.cpu()
    assert torch.allclose(st[..., 0], y.double().sum(dim=(2, 3, 4)), rtol=1e-6, atol=1e-6 * y.numel() / (N * K) * scale)
    assert torch.allclose(st[..., 1], (y.double() ** 2).sum(dim=(2, 3, 4)), rtol=1e-6, atol=1e-6)
    # and within the stated bf16 tolerance of the exact fp32 convolution (operand rounding 2^-9 per factor)
    exact = F.conv3d(g.double(), w.double(), None, padding=1).clamp_min(0)
    assert (y.double() - exact).abs().max().item() < 2e-2 * scale


@pytest.mark.parametrize("shape,C,K", [((2, 9, 13, 17), 16, 32), ((1, 64, 64, 64), 16, 64), ((1, 16, 24, 24), 64, 64)])
def test_conv3d_bf16_data_gradient_with_groupnorm_sums_and_residual(shape, C, K):
    """mode-1 image on dz = the data gradient of a (K -> C)-channel convolution... here: forward conv has `K` inputs and `C`
    outputs; dz has C channels and the gradient K channels"""
    N, D, H, W = shape
    cin_f, cout_f = K, C  # forward layer: cin_f -> cout_f; dz has cout_f channels, the gradient cin_f
    torch.manual_seed(2)
    w = torch.randn(cout_f, cin_f, 3, 3, 3) / (27 * cin_f) ** 0.5
    dz = torch.randn(N, cout_f, D, H, W)
    xin = torch.randn(N, cin_f, D, H, W)
    ref = F.conv_transpose3d(bf16_round(dz).double(), bf16_round(w).double(), None, padding=1)
    gst = torch.zeros((N, cin_f, 2), dtype=torch.float64, device=U.DEV)
    xin_d = U.ndhwc(xin)
    dg = conv_bf16(dz, w, 1, gx=xin_d, gstats=gst)
    scale = ref.abs().max().item()
    assert (dg.double() - ref).abs().max().item() < 1e-3 * scale
    st = gst.cpu()
    assert torch.allclose(st[..., 0], dg.double().sum(dim=(2, 3, 4)), rtol=1e-6, atol=1e-6 * dg.abs().sum().item() / (N * cin_f))
    assert torch.allclose(st[..., 1], (dg.double() * xin.double()).sum(dim=(2, 3, 4)), rtol=1e-6, atol=1e-6 * dg.abs().sum().item() / (N * cin_f))
    # residual epilogue (forward direction)
    res = torch.randn(N, cin_f, D, H, W)
    res_d = U.ndhwc(res)
    y = conv_bf16(dz, w, 1, relu=1, residual=res_d)
    assert (y.double() - (ref + res.double()).clamp_min(0)).abs().max().item() < 1e-3 * max(scale, 1.0)


def test_conv3d_bf16_rejects_unsupported_channel_counts():
    assert nat.get_lib().u3d_conv3d_bf16_supported(64, 64) == 1
    assert nat.get_lib().u3d_conv3d_bf16_supported(8, 32) == 0 and nat.get_lib().u3d_conv3d_bf16_supported(16, 48) == 0
    x = torch.zeros(1, 2, 2, 2, 8, device=U.DEV)
    with pytest.raises(nat.U3DError):
        nat.call("u3d_conv3d_bf16", 0, _stream(U.DEV), _p(x), None, _p(x), _p(x), 1, 2, 2, 2, 8, 32, 0, None, None, None, None)


def wgrad_bf16(x, dz, affine=None):
    N, C, D, H, W = x.shape
    K = dz.shape[1]
    lib = nat.get_lib()
    need = lib.u3d_wgrad_bf16_workspace_floats(N, D, H, W, C, K)
    assert need > 0
    ws = torch.empty(need, dtype=torch.float32, device=U.DEV)
    dw = torch.full((K, C, 3, 3, 3), float("nan"), dtype=torch.float32, device=U.DEV)
    xd, dzd = U.ndhwc(x), U.ndhwc(dz)  # keep the device tensors alive: _p() only takes the pointer
    nat.call("u3d_conv3d_wgrad_bf16", 0, _stream(U.DEV), _p(xd), _p(affine), _p(dzd), _p(dw), N, D, H, W, C, K, _p(ws), need)
    torch.cuda.synchronize()
    return dw.cpu()


@pytest.mark.parametrize("shape,C,K", [
    ((2, 5, 9, 19), 32, 64),      # ragged in every dimension, one (ci, co) pair, few tiles
    ((1, 16, 24, 40), 64, 128),   # 2 x 2 pairs
    ((1, 32, 64, 64), 32, 64),    # many tiles per split
    ((1, 5, 10, 10), 128, 256),   # bottom-of-the-U shape: one tile per split, many pairs
])
def test_conv3d_wgrad_bf16(shape, C, K):
    N, D, H, W = shape
    torch.manual_seed(3)
    x = torch.randn(N, C, D, H, W)
    dz = torch.randn(N, K, D, H, W)
    a = 1.0 + 0.3 * torch.randn(N, C)
    b = 0.2 * torch.randn(N, C)
    g = (x.double() * a.double().view(N, C, 1, 1, 1) + b.double().view(N, C, 1, 1, 1)).float()
    gr = bf16_round(g).double().requires_grad_(False)
    w0 = torch.zeros(K, C, 3, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv3d(gr, w0, None, padding=1).backward(bf16_round(dz).double())
    ref = w0.grad
    aff = torch.stack((a, b), dim=-1).contiguous().to(U.DEV)
    dw = wgrad_bf16(x, dz, aff)
    scale = ref.abs().max().item()
    assert torch.isfinite(dw).all()
    assert (dw.double() - ref).abs().max().item() < 1e-3 * scale
    dw2 = wgrad_bf16(x, dz, aff)
    assert torch.equal(dw, dw2)  # fixed-order split reduction: run-to-run identical
    # the stated bf16 tolerance against the exact fp32 operands
    w1 = torch.zeros(K, C, 3, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv3d(g.double(), w1, None, padding=1).backward(dz.double())
    assert (dw.double() - w1.grad).abs().max().item() < 2e-2 * scale
