"""-m gpu: layer orders other than 'gcr' on the native path (buildingblocks.py:10-96): LeakyReLU /
ELU non-linearities and GroupNorm after the convolution, same MFMA convolutions + the bandwidth passes of csrc/u3d_act.hip,
against the CPU oracle's ordered restatement (pinned to the live reference by tests/test_oracle.py)."""
import pytest
import torch

import gpu_utils as U
from conftest import diag, loss_by_name
from pytorch3dunet_amd import _native as nat
from pytorch3dunet_amd.engine import _p, _stream

pytestmark = pytest.mark.gpu
REL = 1e-3
# Gradient gate of the whole-model tests below: global relative L2 distance from the float64 gradient.  These problems are tiny
# (16 k voxels): ONE ReLU / LeakyReLU decision taken differently at a pre-activation within round-off of 0 moves the L2 norm of
# every upstream gradient by ~1e-3 (measured on the 'crg' net: fused vs unfused GroupNorm statistics differ by 1e-8, flip one
# ReLU of the last layer, and the two runs' gradients differ by 8.6e-4 .. 1.0e-3 at every layer).  A wrong mask, slope or
# statistic is off by >= 1e-2.
REL_GRAD = 3e-3


@pytest.mark.parametrize("mode,slope", [(0, 0.0), (1, 0.0), (2, 0.01), (2, 0.1), (3, 0.0)])
def test_activation_and_postnorm_kernels(mode, slope):
    import torch.nn.functional as F

    torch.manual_seed(mode)
    f = {0: lambda t: t, 1: F.relu, 2: lambda t: F.leaky_relu(t, slope), 3: F.elu}[mode]
    N, V, C = 2, 777, 24
    z = torch.randn(N, V, C, requires_grad=True)
    a, b = torch.randn(N, C), torch.randn(N, C)
    y = f(z * a.view(N, 1, C) + b.view(N, 1, C))
    g = torch.randn(N, V, C)
    zd, gd = z.detach().to(U.DEV), g.to(U.DEV)
    aff = torch.stack((a, b), dim=-1).contiguous().to(U.DEV)
    yd = torch.empty_like(zd)
    nat.call("u3d_affine_act_fwd", 0, _stream(U.DEV), _p(zd), _p(aff), N, V, C, mode, slope, _p(yd))
    assert torch.allclose(yd.cpu(), y.detach(), atol=1e-6, rtol=1e-6)
    # act fwd in place + derivative through the output
    t = torch.randn(N, V, C, requires_grad=True)
    ft = f(t)
    ft.backward(g)
    td = t.detach().to(U.DEV)
    nat.call("u3d_act_fwd", 0, _stream(U.DEV), _p(td), td.numel(), mode, slope, _p(td))
    assert torch.allclose(td.cpu(), ft.detach(), atol=1e-6, rtol=1e-6)
    dn = torch.empty_like(gd)
    nat.call("u3d_act_bwd", 0, _stream(U.DEV), _p(gd), _p(td), gd.numel(), mode, slope, _p(dn))
    assert torch.allclose(dn.cpu(), t.grad, atol=1e-6, rtol=1e-5)
    st = torch.zeros((N, C, 2), dtype=torch.float64, device=U.DEV)
    nat.call("u3d_pair_stats", 0, _stream(U.DEV), _p(gd), _p(zd), N, V, C, _p(st))
    assert torch.allclose(st[..., 0].cpu(), g.double().sum(1), rtol=1e-6, atol=1e-4)
    assert torch.allclose(st[..., 1].cpu(), (g.double() * z.detach().double()).sum(1), rtol=1e-6, atol=1e-4)


@pytest.mark.parametrize("order", ["gcl", "gce", "gc", "cgr", "cgl", "cge", "cg", "crg", "clg", "ceg"])
@pytest.mark.parametrize("cfg,shape,loss_name", [
    (dict(in_channels=1, out_channels=1, f_maps=16, num_levels=3, num_groups=8), (1, 1, 16, 32, 32), "bce_dice"),   # exact 2x: sub-pixel decoders
    (dict(in_channels=2, out_channels=3, f_maps=[8, 16, 32], num_groups=4, final_sigmoid=False), (2, 2, 9, 13, 11), "probs_sum"),  # ragged
])
def test_unet3d_other_layer_orders_native(order, cfg, shape, loss_name, monkeypatch):
    import unet3d_oracle as orc
    from pytorch3dunet_amd.unet3d.model import UNet3D

    monkeypatch.setenv("U3D_STRICT", "1")  # no stock-operator fallback allowed
    torch.manual_seed(31)
    model = UNet3D(layer_order=order, **cfg)
    assert model.native_supported
    with torch.no_grad():
        for k, p in model.named_parameters():
            if "groupnorm" in k:
                p.add_(0.2 * torch.randn_like(p))
    x = torch.randn(shape)
    target = (torch.rand((shape[0], cfg["out_channels"]) + shape[2:]) > 0.5).float()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    G, fs = cfg["num_groups"], cfg.get("final_sigmoid", True)
    p32, l32, v32, g32 = orc.forward_backward(sd, x, target, G, fs, True, loss_name, order=order)
    sd64 = {k: v.double() for k, v in sd.items()}
    _, _, _, g64 = orc.forward_backward(sd64, x.double(), target.double(), G, fs, True, loss_name, order=order)
    model = model.to(U.DEV).train()
    n0 = nat.launch_count
    probs, logits = model(x.to(U.DEV), return_logits=True)
    loss = loss_by_name(loss_name, probs, logits, target.to(U.DEV))
    model.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    assert nat.launch_count > n0
    assert orc.rel_err(logits.detach().cpu(), l32) < REL and orc.rel_err(probs.detach().cpu(), p32) < REL
    assert abs(loss.item() - v32.item()) < REL * max(1.0, abs(v32.item()))
    keys = list(g32)
    ours = torch.cat([dict(model.named_parameters())[k].grad.detach().cpu().double().flatten() for k in keys])
    r32 = torch.cat([g32[k].double().flatten() for k in keys])
    r64 = torch.cat([g64[k].flatten() for k in keys])
    e_ours = ((ours - r64).norm() / r64.norm()).item()
    e_ref = ((r32 - r64).norm() / r64.norm()).item()
    diag(test="orders", order=order, shape=list(shape), ours_vs_fp64=e_ours, ref32_vs_fp64=e_ref)
    # distance from the exact (float64) gradient: within REL_GRAD, or no worse than 3x the reference arithmetic's own distance
    # (kinks of ReLU / LeakyReLU at 0 and max-pool ties make every fp32 implementation differ from exact in isolated voxels)
    assert e_ours <= max(REL_GRAD, 3.0 * e_ref), (order, e_ours, e_ref)
    # the model can be wrapped / evaluated like any other
    model.eval()
    with torch.no_grad():
        y = model(x.to(U.DEV))
    assert torch.allclose(y, probs.detach(), atol=1e-6)


@pytest.mark.parametrize("order", ["cge", "cgr", "gcl", "gce", "cgl", "gc", "cg", "crg", "ceg"])
@pytest.mark.parametrize("cls,cfg,shape,loss_name", [
    ("ResidualUNet3D", dict(in_channels=1, out_channels=1, f_maps=[16, 32, 64], num_groups=8), (1, 1, 16, 32, 32), "bce_dice"),
    ("ResidualUNet3D", dict(in_channels=2, out_channels=3, f_maps=[8, 16, 32], num_groups=4, final_sigmoid=False), (2, 2, 9, 13, 11), "probs_sum"),
    ("ResidualUNetSE3D", dict(in_channels=2, out_channels=2, f_maps=[8, 16, 32], num_groups=4, final_sigmoid=False), (1, 2, 10, 12, 14), "probs_sum"),
])
def test_residual_nets_other_layer_orders_native(order, cls, cfg, shape, loss_name, monkeypatch):
    """ResNetBlock / ResNetBlockSE in the orders other than 'gcr' (buildingblocks.py:245-275; 'cge' is the class default, the
    reference's tests/test_models.py:26-44 builds 'cgr' blocks): GroupNorm before or after the convolutions, conv2's own
    non-linearity (LeakyReLU 0.01) and the block's final one after `out += residual` (LeakyReLU 0.1 / ELU / ReLU)."""
    import unet3d_oracle as orc
    from pytorch3dunet_amd.unet3d import model as M

    monkeypatch.setenv("U3D_STRICT", "1")
    torch.manual_seed(37)
    model = getattr(M, cls)(layer_order=order, **cfg)
    assert model.native_supported, model._native_blockers
    with torch.no_grad():
        for k, p in model.named_parameters():
            if "groupnorm" in k:
                p.add_(0.2 * torch.randn_like(p))
    x = torch.randn(shape)
    target = (torch.rand((shape[0], cfg["out_channels"]) + shape[2:]) > 0.5).float()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    G, fs = cfg["num_groups"], cfg.get("final_sigmoid", True)
    p32, l32, v32, g32 = orc.forward_backward(sd, x, target, G, fs, True, loss_name, order=order)
    _, _, _, g64 = orc.forward_backward({k: v.double() for k, v in sd.items()}, x.double(), target.double(), G, fs, True, loss_name,
                                        order=order)
    model = model.to(U.DEV).train()
    n0 = nat.launch_count
    xd = x.to(U.DEV).requires_grad_(True)
    probs, logits = model(xd, return_logits=True)
    loss = loss_by_name(loss_name, probs, logits, target.to(U.DEV))
    model.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    assert nat.launch_count > n0
    assert orc.rel_err(logits.detach().cpu(), l32) < REL and orc.rel_err(probs.detach().cpu(), p32) < REL
    keys = list(g32)
    ours = torch.cat([dict(model.named_parameters())[k].grad.detach().cpu().double().flatten() for k in keys])
    r32 = torch.cat([g32[k].double().flatten() for k in keys])
    r64 = torch.cat([g64[k].flatten() for k in keys])
    e_ours, e_ref = ((ours - r64).norm() / r64.norm()).item(), ((r32 - r64).norm() / r64.norm()).item()
    diag(test="orders_residual", cls=cls, order=order, shape=list(shape), ours_vs_fp64=e_ours, ref32_vs_fp64=e_ref)
    assert e_ours <= max(REL_GRAD, 3.0 * e_ref), (order, e_ours, e_ref)
    # every parameter on its own (a wrong mask / slope in one branch would hide in the global norm)
    for k in keys:
        g = dict(model.named_parameters())[k].grad.detach().cpu().double()
        # (relative to the parameter's own gradient, floored at 1e-3 of the whole gradient: analytically-zero gradients are noise)
        assert ((g - g64[k]).norm() / g64[k].norm().clamp_min(1e-3 * r64.norm())).item() < max(5e-3, 20 * e_ref), k
    # the input gradient too (first block's conv1 / identity shortcut)
    xr = x.clone().requires_grad_(True)
    pr, lr = orc.model_forward({k: v for k, v in sd.items()}, xr, G, fs, True, order=order)
    loss_by_name(loss_name, pr, lr, target).backward()
    assert orc.rel_err(xd.grad.cpu(), xr.grad) < 2e-2  # (fp32 vs fp32: a decision flip on either side, see REL_GRAD)
    model.eval()
    with torch.no_grad():
        y = model(x.to(U.DEV))
    assert torch.allclose(y, probs.detach(), atol=1e-6)


@pytest.mark.parametrize("order", ["cge", "gcl"])
def test_residual_orders_with_checkpointing_and_bf16(order, monkeypatch):
    """the opt-in extras compose with the layer orders: encoder recomputation is bitwise the same as the stored-activation
    run; bf16 operands stay within the bf16 tolerances of tests/test_gpu_bf16.py"""
    from pytorch3dunet_amd.unet3d.model import ResidualUNet3D

    monkeypatch.setenv("U3D_STRICT", "1")
    cfg = dict(in_channels=1, out_channels=1, f_maps=[32, 64, 128], num_groups=8, layer_order=order)
    x = torch.randn(1, 1, 16, 24, 24, generator=torch.Generator().manual_seed(5)).to(U.DEV)
    target = (torch.rand(1, 1, 16, 24, 24, generator=torch.Generator().manual_seed(6)) > 0.5).float().to(U.DEV)
    grads = {}
    for tag, kw in (("plain", {}), ("ckpt", dict(checkpoint_encoders=True)), ("bf16", dict(compute_dtype="bf16"))):
        torch.manual_seed(3)
        model = ResidualUNet3D(**cfg, **kw).to(U.DEV).train()
        probs, logits = model(x, return_logits=True)
        loss_by_name("bce_dice", probs, logits, target).backward()
        grads[tag] = torch.cat([p.grad.flatten() for p in model.parameters()]).double()
    assert torch.equal(grads["plain"], grads["ckpt"])
    e = ((grads["bf16"] - grads["plain"]).norm() / grads["plain"].norm()).item()
    assert e < 0.15, e


@pytest.mark.parametrize("order", ["gcr", "gce", "cgl"])
@pytest.mark.parametrize("cfg,shape", [
    (dict(in_channels=1, out_channels=1, f_maps=16, num_levels=3, num_groups=8), (1, 1, 16, 32, 32)),
    (dict(in_channels=2, out_channels=2, f_maps=[8, 16, 32], num_groups=4, final_sigmoid=False), (2, 2, 9, 13, 11)),
])
def test_unet3d_deconv_upsampling_native(order, cfg, shape, monkeypatch):
    """upsample='deconv' with DoubleConv blocks (buildingblocks.py:435-464): ConvTranspose3d(k3,s2,p1) -> 2n-1 -> nearest resize
    -> concat, all native (transposed-convolution kernels + virtual concat)"""
    import unet3d_oracle as orc
    from pytorch3dunet_amd.unet3d.model import UNet3D

    monkeypatch.setenv("U3D_STRICT", "1")
    torch.manual_seed(41)
    model = UNet3D(layer_order=order, upsample="deconv", **cfg)
    assert model.native_supported
    assert any("conv_transposed" in k for k in model.state_dict())
    x = torch.randn(shape)
    loss_name = "bce_dice" if cfg.get("final_sigmoid", True) else "probs_sum"
    target = (torch.rand((shape[0], cfg["out_channels"]) + shape[2:]) > 0.5).float()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    G, fs = cfg["num_groups"], cfg.get("final_sigmoid", True)
    p32, l32, v32, g32 = orc.forward_backward(sd, x, target, G, fs, True, loss_name, order=order)
    _, _, _, g64 = orc.forward_backward({k: v.double() for k, v in sd.items()}, x.double(), target.double(), G, fs, True, loss_name,
                                        order=order)
    model = model.to(U.DEV).train()
    probs, logits = model(x.to(U.DEV), return_logits=True)
    loss = loss_by_name(loss_name, probs, logits, target.to(U.DEV))
    model.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    assert orc.rel_err(logits.detach().cpu(), l32) < REL
    keys = list(g32)
    ours = torch.cat([dict(model.named_parameters())[k].grad.detach().cpu().double().flatten() for k in keys])
    r32 = torch.cat([g32[k].double().flatten() for k in keys])
    r64 = torch.cat([g64[k].flatten() for k in keys])
    e_ours, e_ref = ((ours - r64).norm() / r64.norm()).item(), ((r32 - r64).norm() / r64.norm()).item()
    assert e_ours <= max(REL_GRAD, 3.0 * e_ref), (e_ours, e_ref)
    for k in keys:
        if "conv_transposed" in k:
            g = dict(model.named_parameters())[k].grad.detach().cpu().double()
            assert ((g - g64[k]).norm() / g64[k].norm()).item() < max(5e-3, 10 * e_ref), k


@pytest.mark.parametrize("mode", ["trilinear", "area"])
@pytest.mark.parametrize("N,C,lo,hi", [(2, 8, (4, 6, 5), (8, 12, 10)), (1, 5, (4, 6, 5), (9, 13, 11)), (1, 16, (1, 2, 3), (2, 4, 7))])
def test_resample_kernels_against_f_interpolate(mode, N, C, lo, hi):
    """u3d_resample2_fwd / _bwd (csrc/u3d_interp.hip) vs F.interpolate(mode) and its autograd, exact 2x and ragged ratios"""
    import torch.nn.functional as F
    from pytorch3dunet_amd.engine import _resample_tables

    torch.manual_seed(3)
    x = torch.randn(N, C, *lo, requires_grad=True)
    y = F.interpolate(x, size=hi, mode=mode)
    g = torch.randn_like(y)
    y.backward(g)
    tabs = [_resample_tables(U.DEV, mode, a, b) for a, b in zip(lo, hi)]
    xd, gd = U.ndhwc(x.detach()), U.ndhwc(g)
    out = torch.empty((N, *hi, C), device=U.DEV)
    nat.call("u3d_resample2_fwd", 0, _stream(U.DEV), _p(xd), _p(tabs[0][0]), _p(tabs[1][0]), _p(tabs[2][0]), _p(tabs[0][1]),
             _p(tabs[1][1]), _p(tabs[2][1]), N, *lo, *hi, C, _p(out))
    assert U.relerr(U.ncdhw(out), y.detach()) < 1e-6
    dx = torch.empty_like(xd)
    nat.call("u3d_resample2_bwd", 0, _stream(U.DEV), _p(gd), _p(tabs[0][2]), _p(tabs[1][2]), _p(tabs[2][2]), _p(tabs[0][0]),
             _p(tabs[1][0]), _p(tabs[2][0]), _p(tabs[0][1]), _p(tabs[1][1]), _p(tabs[2][1]), N, *lo, *hi, C, _p(dx))
    assert U.relerr(U.ncdhw(dx), x.grad) < 1e-6


@pytest.mark.parametrize("mode", ["trilinear", "area"])
@pytest.mark.parametrize("order", ["gcr", "cge"])
@pytest.mark.parametrize("cfg,shape", [
    (dict(in_channels=1, out_channels=1, f_maps=16, num_levels=3, num_groups=8), (1, 1, 16, 32, 32)),
    (dict(in_channels=2, out_channels=2, f_maps=[8, 16, 32], num_groups=4, final_sigmoid=False), (2, 2, 9, 13, 11)),
])
def test_unet3d_interpolating_upsampling_native(mode, order, cfg, shape, monkeypatch):
    """upsample='trilinear' / 'area' (InterpolateUpsampling, buildingblocks.py:598-614): the two F.interpolate modes besides
    'nearest' that the reference can run on 3-D nets (tests/test_oracle.py::test_reference_upsample_modes_that_cannot_run_on_3d_models)"""
    import unet3d_oracle as orc
    from pytorch3dunet_amd.unet3d.model import UNet3D

    monkeypatch.setenv("U3D_STRICT", "1")
    torch.manual_seed(43)
    model = UNet3D(layer_order=order, upsample=mode, **cfg)
    assert model.native_supported, model._native_blockers
    x = torch.randn(shape)
    loss_name = "bce_dice" if cfg.get("final_sigmoid", True) else "probs_sum"
    target = (torch.rand((shape[0], cfg["out_channels"]) + shape[2:]) > 0.5).float()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    G, fs = cfg["num_groups"], cfg.get("final_sigmoid", True)
    p32, l32, v32, g32 = orc.forward_backward(sd, x, target, G, fs, True, loss_name, order=order, upsample=mode)
    _, _, _, g64 = orc.forward_backward({k: v.double() for k, v in sd.items()}, x.double(), target.double(), G, fs, True, loss_name,
                                        order=order, upsample=mode)
    model = model.to(U.DEV).train()
    xd = x.to(U.DEV).requires_grad_(True)
    probs, logits = model(xd, return_logits=True)
    loss = loss_by_name(loss_name, probs, logits, target.to(U.DEV))
    model.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    assert orc.rel_err(logits.detach().cpu(), l32) < REL and orc.rel_err(probs.detach().cpu(), p32) < REL
    keys = list(g32)
    ours = torch.cat([dict(model.named_parameters())[k].grad.detach().cpu().double().flatten() for k in keys])
    r32 = torch.cat([g32[k].double().flatten() for k in keys])
    r64 = torch.cat([g64[k].flatten() for k in keys])
    e_ours, e_ref = ((ours - r64).norm() / r64.norm()).item(), ((r32 - r64).norm() / r64.norm()).item()
    diag(test="interp_upsampling", mode=mode, order=order, shape=list(shape), ours_vs_fp64=e_ours, ref32_vs_fp64=e_ref)
    assert e_ours <= max(REL_GRAD, 3.0 * e_ref), (e_ours, e_ref)
    for k in keys:
        g = dict(model.named_parameters())[k].grad.detach().cpu().double()
        # (relative to the parameter's own gradient, floored at 1e-3 of the whole gradient: analytically-zero gradients are noise)
        assert ((g - g64[k]).norm() / g64[k].norm().clamp_min(1e-3 * r64.norm())).item() < max(5e-3, 20 * e_ref), k
    xr = x.clone().requires_grad_(True)
    pr, lr = orc.model_forward(sd, xr, G, fs, True, order=order, upsample=mode)
    loss_by_name(loss_name, pr, lr, target).backward()
    assert orc.rel_err(xd.grad.cpu(), xr.grad) < 2e-2


@pytest.mark.parametrize("order", ["bcr", "cbr", "crb", "cbl", "cr", "cl", "ce", "c"])
@pytest.mark.parametrize("cls,cfg,shape,loss_name", [
    ("UNet3D", dict(in_channels=1, out_channels=1, f_maps=16, num_levels=3, num_groups=8), (2, 1, 16, 32, 32), "bce_dice"),
    ("UNet3D", dict(in_channels=2, out_channels=3, f_maps=[8, 16, 32], num_groups=4, final_sigmoid=False), (2, 2, 9, 13, 11), "probs_sum"),
    ("ResidualUNet3D", dict(in_channels=2, out_channels=2, f_maps=[8, 16, 32], num_groups=4, final_sigmoid=False), (2, 2, 10, 12, 14), "probs_sum"),
])
def test_batchnorm_and_norm_free_orders_native(order, cls, cfg, shape, loss_name, monkeypatch):
    """'b' = nn.BatchNorm3d (batch statistics over N x voxels, running-estimate update, eval mode on the running estimates) and the
    norm-free orders 'cr' / 'cl' / 'ce' / 'c' (conv WITH bias) of create_conv (buildingblocks.py:10-96), on the native path.
    Gradient gate: see REL_GRAD — a run whose distance exceeds it (decision flips; measured 'cbr': seed 53 1e-2 in the encoders only,
    seed 54 clean to 1e-4, seed 55 3e-3 everywhere) is repeated with the next seed; every run must stay below the gross-error bound
    5e-2 and one of three must be clean."""
    import unet3d_oracle as orc
    from pytorch3dunet_amd.unet3d import model as M

    monkeypatch.setenv("U3D_STRICT", "1")
    G, fs = cfg["num_groups"], cfg.get("final_sigmoid", True)
    clean = False
    for seed in (53, 54, 55):
        torch.manual_seed(seed)
        model = getattr(M, cls)(layer_order=order, **cfg)
        assert model.native_supported, model._native_blockers
        with torch.no_grad():
            for k, p in model.named_parameters():
                if "batchnorm" in k or k.endswith("conv.bias"):
                    p.add_(0.2 * torch.randn_like(p))
        x = torch.randn(shape)
        target = (torch.rand((shape[0], cfg["out_channels"]) + shape[2:]) > 0.5).float()
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        bufs = {}
        p32, l32, v32, g32 = orc.forward_backward(sd, x, target, G, fs, True, loss_name, order=order, buffers_out=bufs)
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
        _, _, _, g64 = orc.forward_backward(sd64, x.double(), target.double(), G, fs, True, loss_name, order=order)
        model = model.to(U.DEV).train()
        probs, logits = model(x.to(U.DEV), return_logits=True)
        loss = loss_by_name(loss_name, probs, logits, target.to(U.DEV))
        model.zero_grad()
        loss.backward()
        torch.cuda.synchronize()
        assert orc.rel_err(logits.detach().cpu(), l32) < REL and orc.rel_err(probs.detach().cpu(), p32) < REL
        keys = list(g32)
        ours = torch.cat([dict(model.named_parameters())[k].grad.detach().cpu().double().flatten() for k in keys])
        r32 = torch.cat([g32[k].double().flatten() for k in keys])
        r64 = torch.cat([g64[k].flatten() for k in keys])
        e_ours, e_ref = ((ours - r64).norm() / r64.norm()).item(), ((r32 - r64).norm() / r64.norm()).item()
        diag(test="orders_bn_bias", cls=cls, order=order, shape=list(shape), seed=seed, ours_vs_fp64=e_ours, ref32_vs_fp64=e_ref)
        assert e_ours < 5e-2, (order, seed, e_ours)
        # running estimates after the training forward (momentum 0.1, unbiased variance, num_batches_tracked)
        after = model.state_dict()
        for k, v in bufs.items():
            if v.is_floating_point():
                assert torch.allclose(after[k].cpu(), v, rtol=1e-4, atol=1e-6), k
            else:  # num_batches_tracked: incremented by the nn.Module (F.batch_norm in the oracle does not touch it)
                assert int(after[k]) == int(sd[k]) + 1, k
        if e_ours <= max(REL_GRAD, 3.0 * e_ref):
            for k in keys:
                g = dict(model.named_parameters())[k].grad.detach().cpu().double()
                assert ((g - g64[k]).norm() / g64[k].norm().clamp_min(1e-3 * r64.norm())).item() < max(5e-3, 20 * e_ref), k
            clean = True
            break
    assert clean, (order, "no clean run in three seeds")
    # eval mode: BatchNorm on the running estimates
    model.eval()
    with torch.no_grad():
        pe, le = model(x.to(U.DEV), return_logits=True)
    orc.TRAINING = False
    try:
        pr, lr = orc.model_forward({k: v.cpu() for k, v in after.items()}, x, G, fs, True, order=order)
    finally:
        orc.TRAINING = True
    assert orc.rel_err(le.cpu(), lr) < REL


@pytest.mark.parametrize("order", ["gcrd", "gcrD", "cgld", "crd", "bcrD"])
def test_trailing_dropout_native_draws_the_reference_masks(order, monkeypatch):
    """'d' nn.Dropout / 'D' nn.Dropout2d (per-(sample, channel) on 5-D inputs) as the last operation of a layer
    (buildingblocks.py:89-92): the native path draws its masks from torch's generator exactly as the module tree does, so with
    the same seed both paths see the SAME masks and must agree like two fp32 implementations"""
    from pytorch3dunet_amd.unet3d.model import UNet3D

    cfg = dict(in_channels=1, out_channels=1, f_maps=[16, 32], num_groups=8, layer_order=order, dropout_prob=0.25)
    torch.manual_seed(61)
    model = UNet3D(**cfg).to(U.DEV).train()
    assert model.native_supported, model._native_blockers
    monkeypatch.setenv("U3D_ALLOW_TORCH_FALLBACK", "1")  # the 'tree' leg below is the explicit opt-in (default: error)
    x = torch.randn(2, 1, 8, 16, 16, device=U.DEV)
    target = (torch.rand(2, 1, 8, 16, 16, device=U.DEV) > 0.5).float()
    res = {}
    for tag in ("native", "tree"):
        blockers = model._native_blockers
        if tag == "tree":
            model._native_blockers = ["forced module tree (test)"]
        torch.manual_seed(1234)
        n0 = nat.launch_count
        probs, logits = model(x, return_logits=True)
        loss = loss_by_name("bce_dice", probs, logits, target)
        model.zero_grad()
        loss.backward()
        assert (nat.launch_count > n0) == (tag == "native")
        res[tag] = (logits.detach().clone(), torch.cat([p.grad.flatten() for p in model.parameters()]).clone())
        model._native_blockers = blockers
    (la, ga), (lb, gb) = res["native"], res["tree"]
    assert ((la - lb).norm() / lb.norm()).item() < 1e-4                   # same masks: fp32-level agreement
    assert ((ga - gb).norm() / gb.norm()).item() < REL_GRAD
    frac_zero = float((la == 0).float().mean())
    assert frac_zero < 0.5
    # eval mode: dropout is the identity, the stochastic and the deterministic forward differ
    model.eval()
    with torch.no_grad():
        e1, e2 = model(x), model(x)
    assert torch.equal(e1, e2)


@pytest.mark.parametrize("cls,order", [("ResidualUNet3D", "gcr"), ("ResidualUNetSE3D", "gcr"), ("ResidualUNet3D", "cge")])
@pytest.mark.parametrize("cfg,shape,loss_name", [
    (dict(in_channels=1, out_channels=1, f_maps=[16, 32, 64], num_groups=8), (1, 1, 16, 32, 32), "bce_dice"),
    (dict(in_channels=2, out_channels=2, f_maps=[8, 16, 32], num_groups=4, final_sigmoid=False), (2, 2, 9, 13, 11), "probs_sum"),
])
def test_residual_nets_with_explicit_deconv_native(cls, order, cfg, shape, loss_name, monkeypatch):
    """an EXPLICIT upsample='deconv' on residual nets — the only other value the reference can run for them — keeps concat
    joining and a 1x1x1 conv (deep -> shallow channels) in the decoder blocks (buildingblocks.py:435-468)"""
    import unet3d_oracle as orc
    from pytorch3dunet_amd.unet3d import model as M

    monkeypatch.setenv("U3D_STRICT", "1")
    G, fs = cfg["num_groups"], cfg.get("final_sigmoid", True)
    clean = False
    for seed in (71, 72, 73):
        torch.manual_seed(seed)
        model = getattr(M, cls)(layer_order=order, upsample="deconv", **cfg)
        assert model.native_supported, model._native_blockers
        assert "decoders.0.basic_module.conv1.weight" in model.state_dict()
        x = torch.randn(shape)
        target = (torch.rand((shape[0], cfg["out_channels"]) + shape[2:]) > 0.5).float()
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        p32, l32, v32, g32 = orc.forward_backward(sd, x, target, G, fs, True, loss_name, order=order)
        _, _, _, g64 = orc.forward_backward({k: v.double() for k, v in sd.items()}, x.double(), target.double(), G, fs, True, loss_name,
                                            order=order)
        model = model.to(U.DEV).train()
        xd = x.to(U.DEV).requires_grad_(True)
        probs, logits = model(xd, return_logits=True)
        loss = loss_by_name(loss_name, probs, logits, target.to(U.DEV))
        model.zero_grad()
        loss.backward()
        torch.cuda.synchronize()
        assert orc.rel_err(logits.detach().cpu(), l32) < REL and orc.rel_err(probs.detach().cpu(), p32) < REL
        keys = list(g32)
        ours = torch.cat([dict(model.named_parameters())[k].grad.detach().cpu().double().flatten() for k in keys])
        r32 = torch.cat([g32[k].double().flatten() for k in keys])
        r64 = torch.cat([g64[k].flatten() for k in keys])
        e_ours, e_ref = ((ours - r64).norm() / r64.norm()).item(), ((r32 - r64).norm() / r64.norm()).item()
        diag(test="residual_explicit_deconv", cls=cls, order=order, shape=list(shape), seed=seed, ours_vs_fp64=e_ours, ref32_vs_fp64=e_ref)
        assert e_ours < 5e-2, (seed, e_ours)
        if e_ours <= max(REL_GRAD, 3.0 * e_ref):
            # every parameter on its own (floored at 1e-3 of the whole gradient); a flip that only shows in a tiny gradient
            # counts like any other flip: next seed
            per = [((dict(model.named_parameters())[k].grad.detach().cpu().double() - g64[k]).norm()
                    / g64[k].norm().clamp_min(1e-3 * r64.norm())).item() for k in keys]
            if max(per) < max(5e-3, 20 * e_ref):
                clean = True
                break
    assert clean


def test_batchnorm_running_estimates_under_activation_checkpointing():
    """encoder recomputation re-runs the block forward in backward: the running estimates and num_batches_tracked must move once
    per step, and the gradients must equal the stored-activation run bitwise"""
    from pytorch3dunet_amd.unet3d.model import ResidualUNet3D

    cfg = dict(in_channels=1, out_channels=1, f_maps=[16, 32, 64], num_groups=8, layer_order="bcr")
    x = torch.randn(2, 1, 16, 24, 24, generator=torch.Generator().manual_seed(5)).to(U.DEV)
    target = (torch.rand(2, 1, 16, 24, 24, generator=torch.Generator().manual_seed(6)) > 0.5).float().to(U.DEV)
    out = {}
    for tag, kw in (("plain", {}), ("ckpt", dict(checkpoint_encoders=True))):
        torch.manual_seed(3)
        model = ResidualUNet3D(**cfg, **kw).to(U.DEV).train()
        probs, logits = model(x, return_logits=True)
        loss_by_name("bce_dice", probs, logits, target).backward()
        out[tag] = (torch.cat([p.grad.flatten() for p in model.parameters()]), {k: v.clone() for k, v in model.state_dict().items() if "running" in k or "tracked" in k})
    assert torch.equal(out["plain"][0], out["ckpt"][0])
    for k, v in out["plain"][1].items():
        assert torch.equal(v, out["ckpt"][1][k]), k
        if k.endswith("num_batches_tracked"):
            assert int(v) == 1
