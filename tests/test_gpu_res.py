"""-m gpu: the operators the residual variants add (csrc/u3d_res.hip + the residual conv epilogue), each through the
C-ABI against torch CPU operators — the ATen call sites of buildingblocks.py:248-255 (1x1x1 conv + bias), :277-288
(out += residual; ReLU), :653-662 (ConvTranspose3d k3 s2 p1), :650-651 + :493 (nearest resize + sum joining)."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = 2e-5


def _mods():
    import gpu_utils as U
    from pytorch3dunet_amd import _native as nat
    from pytorch3dunet_amd.engine import VSrc, _maps, _p, _stream

    return U, nat, VSrc, _maps, _p, _stream


@pytest.mark.parametrize("N,Cin,Cout,size", [(2, 1, 16, (4, 6, 5)), (1, 32, 64, (8, 8, 8)), (2, 3, 7, (3, 5, 4)),
                                             (1, 64, 128, (4, 8, 16)), (1, 130, 70, (2, 3, 5))])
def test_conv1x1_forward_backward(N, Cin, Cout, size):
    U, nat, VSrc, _maps, _p, _stream = _mods()
    torch.manual_seed(Cin + Cout)
    x = torch.randn(N, Cin, *size, requires_grad=True)
    w = (torch.randn(Cout, Cin, 1, 1, 1) / Cin ** 0.5).requires_grad_(True)
    b = torch.randn(Cout, requires_grad=True)
    y = F.conv3d(x, w, b)
    dy = torch.randn_like(y)
    y.backward(dy)
    V = size[0] * size[1] * size[2]
    xd, wd, bd = U.ndhwc(x.detach()), w.detach().reshape(Cout, Cin).contiguous().to(U.DEV), b.detach().to(U.DEV)
    yd = torch.empty((N, *size, Cout), device=U.DEV)
    st = torch.zeros((N, Cout, 2), dtype=torch.float64, device=U.DEV)
    nat.call("u3d_conv1x1_fwd", 0, _stream(U.DEV), _p(xd), _p(wd), _p(bd), _p(yd), N, V, Cin, Cout, _p(st))
    assert U.relerr(U.ncdhw(yd), y.detach()) < TOL
    yy = y.detach().double()
    s_ref = torch.stack([yy.sum(dim=(2, 3, 4)), (yy * yy).sum(dim=(2, 3, 4))], dim=-1)
    assert U.relerr(st.cpu(), s_ref) < 1e-5
    dyd = U.ndhwc(dy)
    dx = torch.empty_like(xd)
    acc = torch.zeros(Cout * Cin + Cout, dtype=torch.float64, device=U.DEV)
    nat.call("u3d_conv1x1_bwd", 0, _stream(U.DEV), _p(dyd), _p(xd), _p(wd), N, V, Cin, Cout, _p(dx), _p(acc))
    assert U.relerr(U.ncdhw(dx), x.grad) < TOL
    assert U.relerr(acc[: Cout * Cin].cpu().float().view(Cout, Cin), w.grad.view(Cout, Cin)) < 1e-4
    assert U.relerr(acc[Cout * Cin:].cpu().float(), b.grad) < 1e-4


@pytest.mark.parametrize("N,Cin,Cout,size", [(1, 8, 4, (3, 4, 5)), (2, 32, 16, (4, 4, 4)), (1, 6, 3, (1, 2, 3)),
                                             (1, 64, 32, (5, 10, 10)), (1, 16, 16, (2, 1, 7))])
@pytest.mark.parametrize("relu_mask", [0, 1])
def test_convtranspose3d_forward_backward(N, Cin, Cout, size, relu_mask):
    U, nat, VSrc, _maps, _p, _stream = _mods()
    torch.manual_seed(Cin * 5 + Cout)
    x = torch.randn(N, Cin, *size)
    if relu_mask:
        x = F.relu(x)
    x.requires_grad_(True)
    w = (torch.randn(Cin, Cout, 3, 3, 3) / (Cin * 3.4) ** 0.5).requires_grad_(True)
    t = F.conv_transpose3d(x, w, None, stride=2, padding=1)
    D1, H1, W1 = size
    assert tuple(t.shape[2:]) == (2 * D1 - 1, 2 * H1 - 1, 2 * W1 - 1)
    dt = torch.randn_like(t)
    t.backward(dt)
    xd, wd = U.ndhwc(x.detach()), w.detach().contiguous().to(U.DEV)
    td = torch.empty((N, 2 * D1 - 1, 2 * H1 - 1, 2 * W1 - 1, Cout), device=U.DEV)
    nat.call("u3d_convtr3d_fwd", 0, _stream(U.DEV), _p(xd), _p(wd), _p(td), N, D1, H1, W1, Cin, Cout, None)
    assert U.relerr(U.ncdhw(td), t.detach()) < TOL
    pk0, pk1 = torch.empty(27 * Cin * Cout, device=U.DEV), torch.empty(27 * Cin * Cout, device=U.DEV)
    nat.call("u3d_pack_convtr_weights", 0, _stream(U.DEV), _p(wd), Cin, Cout, 0, _p(pk0))
    nat.call("u3d_pack_convtr_weights", 0, _stream(U.DEV), _p(wd), Cin, Cout, 1, _p(pk1))
    td2 = torch.empty_like(td)
    nat.call("u3d_convtr3d_fwd", 0, _stream(U.DEV), _p(xd), _p(wd), _p(td2), N, D1, H1, W1, Cin, Cout, _p(pk0))
    assert U.relerr(U.ncdhw(td2), t.detach()) < TOL
    if Cin % 4 == 0 and Cout % 4 == 0:
        # the sub-pixel MFMA forward: all 8 output parity classes from one staged input halo tile
        pk2 = torch.empty(nat.get_lib().u3d_convtr3d_subpixel_packed_floats(Cin, Cout), device=U.DEV)
        nat.call("u3d_pack_convtr3d_subpixel", 0, _stream(U.DEV), _p(wd), Cin, Cout, _p(pk2))
        td3 = torch.full_like(td, float("nan"))
        nat.call("u3d_convtr3d_fwd_subpixel", 0, _stream(U.DEV), _p(xd), _p(pk2), _p(td3), N, D1, H1, W1, Cin, Cout)
        assert U.relerr(U.ncdhw(td3), t.detach()) < TOL
    dtd = U.ndhwc(dt)
    dx = torch.empty_like(xd)
    acc = torch.zeros(Cin * Cout * 27, dtype=torch.float64, device=U.DEV)
    nat.call("u3d_convtr3d_bwd", 0, _stream(U.DEV), _p(dtd), _p(xd), _p(wd), N, D1, H1, W1, Cin, Cout, relu_mask, _p(dx),
             _p(acc), None)
    ref_dx = x.grad * (x.detach() > 0) if relu_mask else x.grad
    assert U.relerr(U.ncdhw(dx), ref_dx) < TOL
    dx2 = torch.empty_like(xd)
    acc2 = torch.zeros_like(acc)
    nat.call("u3d_convtr3d_bwd", 0, _stream(U.DEV), _p(dtd), _p(xd), _p(wd), N, D1, H1, W1, Cin, Cout, relu_mask, _p(dx2),
             _p(acc2), _p(pk1))
    assert U.relerr(U.ncdhw(dx2), ref_dx) < TOL
    assert U.relerr(acc.cpu().float().view(Cin, Cout, 3, 3, 3), w.grad) < 1e-4


@pytest.mark.parametrize("N,C,lo,hi", [(2, 8, (3, 4, 5), (6, 8, 10)), (1, 6, (5, 10, 10), (10, 20, 20)), (1, 3, (2, 3, 4), (5, 7, 9)),
                                       (2, 64, (2, 2, 2), (4, 4, 4)), (1, 5, (1, 2, 2), (3, 5, 4))])
def test_nearest_add_and_sum(N, C, lo, hi):
    """t has the transposed conv's (2n-1) size; F.interpolate(t, size=skip) then skip + t (buildingblocks.py:650-651,:493)"""
    U, nat, VSrc, _maps, _p, _stream = _mods()
    torch.manual_seed(C)
    Dt, Ht, Wt = (2 * v - 1 for v in lo)
    t = torch.randn(N, C, Dt, Ht, Wt, requires_grad=True)
    skip = torch.randn(N, C, *hi, requires_grad=True)
    j = skip + F.interpolate(t, size=hi)
    dj = torch.randn_like(j)
    j.backward(dj)
    maps, los = zip(*(_maps(U.DEV, a, b) for a, b in zip((Dt, Ht, Wt), hi)))
    td, sd = U.ndhwc(t.detach()), U.ndhwc(skip.detach())
    out = torch.empty_like(sd)
    st = torch.zeros((N, C, 2), dtype=torch.float64, device=U.DEV)
    nat.call("u3d_nearest_add_fwd", 0, _stream(U.DEV), _p(sd), _p(td), _p(maps[0]), _p(maps[1]), _p(maps[2]), N, *hi, Dt, Ht, Wt,
             C, _p(out), _p(st))
    assert U.relerr(U.ncdhw(out), j.detach()) < 1e-6
    jj = j.detach().double()
    assert U.relerr(st.cpu(), torch.stack([jj.sum(dim=(2, 3, 4)), (jj * jj).sum(dim=(2, 3, 4))], dim=-1)) < 1e-5
    djd = U.ndhwc(dj)
    dtd = torch.empty_like(td)
    nat.call("u3d_nearest_sum_bwd", 0, _stream(U.DEV), _p(djd), _p(los[0]), _p(los[1]), _p(los[2]), N, *hi, Dt, Ht, Wt, C, _p(dtd))
    assert U.relerr(U.ncdhw(dtd), t.grad) < 1e-5


@pytest.mark.parametrize("N,Cs,Ct,lo,hi", [(2, 32, 64, (4, 6, 5), (8, 12, 10)),   # channel quads (16-byte path), exact 2x
                                          (1, 8, 4, (3, 4, 5), (5, 7, 9)),        # quads, a (2n-1)-sized tensor resized to the skip
                                          (1, 6, 5, (2, 3, 4), (4, 6, 8))])       # element path
def test_nearest_cat(N, Cs, Ct, lo, hi):
    """torch.cat((skip, F.interpolate(t, size=skip.shape[2:])), dim=1) (buildingblocks.py:491 after :650-651 / :614) written out:
    the explicit-`deconv` residual decoders and, in bf16 mode, the DoubleConv decoders' first convolution read it"""
    U, nat, VSrc, _maps, _p, _stream = _mods()
    torch.manual_seed(Cs + Ct)
    t = torch.randn(N, Ct, *lo)
    skip = torch.randn(N, Cs, *hi)
    want = torch.cat((skip, F.interpolate(t, size=hi)), dim=1)
    maps = [_maps(U.DEV, a, b)[0] for a, b in zip(lo, hi)]
    td, sd = U.ndhwc(t), U.ndhwc(skip)
    out = torch.full((N, *hi, Cs + Ct), float("nan"), device=U.DEV)
    nat.call("u3d_nearest_cat_fwd", 0, _stream(U.DEV), _p(sd), _p(td), _p(maps[0]), _p(maps[1]), _p(maps[2]), N, *hi, *lo, Cs, Ct, _p(out))
    torch.cuda.synchronize()
    assert torch.equal(U.ncdhw(out).cpu(), want)


@pytest.mark.parametrize("N,Cin,Cout,size", [(1, 16, 32, (8, 16, 16)), (2, 32, 32, (4, 8, 8)), (1, 8, 12, (5, 9, 7)),
                                             (1, 64, 96, (4, 8, 16)), (1, 6, 5, (3, 4, 5))])
def test_conv3d_residual_epilogue(N, Cin, Cout, size):
    """out = relu(conv(GN-affine(x)) + residual): aligned sizes take the persistent kernel, ragged ones the generic"""
    U, nat, VSrc, _maps, _p, _stream = _mods()
    torch.manual_seed(Cin + 7 * Cout)
    x = torch.randn(N, Cin, *size)
    w = torch.randn(Cout, Cin, 3, 3, 3) / (27 * Cin) ** 0.5
    ab = torch.randn(N, Cin, 2)
    res = torch.randn(N, Cout, *size)
    g = x * ab[:, :, 0].view(N, Cin, 1, 1, 1) + ab[:, :, 1].view(N, Cin, 1, 1, 1)
    ref = F.relu(F.conv3d(g, w, None, padding=1) + res)
    src = VSrc(U.ndhwc(x))
    aff = ab.contiguous().to(U.DEV)  # must outlive the call: the struct holds a raw pointer
    s = src.struct(aff)
    wp = U.pack(w, 0)
    resd = U.ndhwc(res)
    y = torch.empty((N, *size, Cout), device=U.DEV)
    st = torch.zeros((N, Cout, 2), dtype=torch.float64, device=U.DEV)
    nat.call("u3d_conv3d_residual", 0, _stream(U.DEV), ctypes.byref(s), _p(wp), _p(y), N, *size, Cout, 1, _p(st), _p(resd))
    assert U.relerr(U.ncdhw(y), ref) < TOL
    rr = ref.double()
    assert U.relerr(st.cpu(), torch.stack([rr.sum(dim=(2, 3, 4)), (rr * rr).sum(dim=(2, 3, 4))], dim=-1)) < 1e-5


def test_gn_bwd_apply_add():
    U, nat, VSrc, _maps, _p, _stream = _mods()
    torch.manual_seed(2)
    N, C, V = 2, 12, 77
    dg, x, add = (torch.randn(N, V, C, device=U.DEV) for _ in range(3))
    coef = torch.randn(N, 3, C, device=U.DEV)
    out = torch.empty_like(x)
    for relu in (0, 1):
        nat.call("u3d_gn_bwd_apply_add", 0, _stream(U.DEV), _p(dg), C, 0, _p(x), C, _p(coef), C, V, N, relu, _p(add), _p(out))
        ref = coef[:, 0:1] * dg + coef[:, 1:2] * x + coef[:, 2:3] + add
        if relu:
            ref = ref * (x > 0)
        assert U.relerr(out, ref) < 1e-6


@pytest.mark.parametrize("signed", [False, True])
@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("N,C,size", [(2, 16, (4, 6, 5)), (1, 96, (3, 4, 4)), (1, 8, (2, 3, 3)), (2, 256, (2, 2, 3)), (1, 1024, (1, 2, 2))])
def test_se_gates_forward_backward(mode, N, C, size, signed):
    """scSE / cSE / sSE (se.py:18-114) on a block output — post-ReLU (>= 0), or signed as after ELU / LeakyReLU blocks, where
    max(y*gc, y*a) picks the SMALLER gate for y < 0: forward and every gradient against the torch modules"""
    U, nat, VSrc, _maps, _p, _stream = _mods()
    from pytorch3dunet_amd.unet3d.se import ChannelSELayer3D, ChannelSpatialSELayer3D, SpatialSELayer3D

    torch.manual_seed(C + mode)
    mod = [ChannelSpatialSELayer3D(C, 1), ChannelSELayer3D(C, 1), SpatialSELayer3D(C)][mode]
    cse = mod.cSE if mode == 0 else (mod if mode == 1 else None)
    sse = mod.sSE if mode == 0 else (mod if mode == 2 else None)
    pre = torch.randn(N, C, *size, requires_grad=True)
    y = F.elu(pre) if signed else F.relu(pre)
    out = mod(y)
    dout = torch.randn_like(out)
    out.backward(dout)
    V = size[0] * size[1] * size[2]
    yd = U.ndhwc(y.detach())
    yy = y.detach().double()
    st = torch.stack([yy.sum(dim=(2, 3, 4)), (yy * yy).sum(dim=(2, 3, 4))], dim=-1).contiguous().to(U.DEV)
    dev_ = lambda t: t.detach().contiguous().to(U.DEV)  # noqa: E731
    gc = s = h = a = ws = bs = None
    if cse is not None:
        w1, b1, w2, b2 = dev_(cse.fc1.weight), dev_(cse.fc1.bias), dev_(cse.fc2.weight), dev_(cse.fc2.bias)
        s, h, gc = (torch.empty((N, C), device=U.DEV) for _ in range(3))
        nat.call("u3d_se_gate_fwd", 0, _stream(U.DEV), _p(st), float(V), _p(w1), _p(b1), _p(w2), _p(b2), N, C, C, _p(s), _p(h), _p(gc))
    if sse is not None:
        ws, bs = dev_(sse.conv.weight.view(C)), dev_(sse.conv.bias)
        a = torch.empty(N * V, device=U.DEV)
    od = torch.empty_like(yd)
    nat.call("u3d_se_apply_fwd", 0, _stream(U.DEV), _p(yd), _p(gc), _p(ws), _p(bs), N, V, C, mode, _p(od), _p(a))
    assert U.relerr(U.ncdhw(od), out.detach()) < TOL
    # backward
    dd = U.ndhwc(dout)
    acc_gc = torch.zeros(N * C, dtype=torch.float64, device=U.DEV) if cse is not None else None
    acc_ws = torch.zeros(C + 1, dtype=torch.float64, device=U.DEV) if sse is not None else None
    dls = torch.empty(N * V, device=U.DEV) if sse is not None else None
    nat.call("u3d_se_bwd_reduce", 0, _stream(U.DEV), _p(dd), _p(yd), _p(gc), _p(a), _p(ws), N, V, C, mode, _p(dls), _p(acc_gc), _p(acc_ws))
    ds = None
    if cse is not None:
        dz2, dz1, ds = (torch.empty((N, C), device=U.DEV) for _ in range(3))
        dw1, dw2 = torch.empty((C, C), device=U.DEV), torch.empty((C, C), device=U.DEV)
        db1, db2 = torch.empty(C, device=U.DEV), torch.empty(C, device=U.DEV)
        nat.call("u3d_se_gate_bwd", 0, _stream(U.DEV), _p(acc_gc), _p(gc), _p(h), _p(s), _p(w1), _p(w2), N, C, C, float(V), _p(dz2),
                 _p(dz1), _p(ds), _p(dw1), _p(db1), _p(dw2), _p(db2))
        for got, ref in ((dw1, cse.fc1.weight.grad), (db1, cse.fc1.bias.grad), (dw2, cse.fc2.weight.grad), (db2, cse.fc2.bias.grad)):
            assert U.relerr(got, ref) < 1e-4
    if sse is not None:
        assert U.relerr(acc_ws[:C].cpu().float(), sse.conv.weight.grad.view(C)) < 1e-4
        assert U.relerr(acc_ws[C:].cpu().float(), sse.conv.bias.grad) < 1e-4
    md = torch.empty_like(yd)
    nat.call("u3d_se_bwd_apply", 0, _stream(U.DEV), _p(dd), _p(yd), _p(gc), _p(a), _p(ws), _p(dls), _p(ds), N, V, C, mode,
             0 if signed else 1, _p(md))
    if signed:  # the block removes its own non-linearity afterwards (engine._block_bwd)
        nat.call("u3d_act_bwd", 0, _stream(U.DEV), _p(md), _p(yd), md.numel(), 3, 0.0, _p(md))
    assert U.relerr(U.ncdhw(md), pre.grad) < 1e-4
