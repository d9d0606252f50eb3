"""-m gpu: decoder levels that upsample n -> 2n + 1 (round 5).

`F.interpolate(x, size=skip, mode="nearest")` (buildingblocks.py:598-614) to an ODD skip size — the reference's shipped 80 x 170 x 170
patch pools 85 -> 42 and upsamples 42 -> 85 — reads src = (dst - 1) >> 1 (src(0) = 0): the exact 2x upsampling shifted by one.  Every
output voxel d >= 2 along such an axis sees exactly the shifted 2x tensor, so forward / data gradient / weight gradient of the upsampled
channels are [the sub-pixel kernels on a WINDOW, 8/27 of the multiply-adds] + [the general kernels on the slab d < 2].  Here: each of the
three decompositions through the C-ABI against F.conv3d / F.interpolate CPU autograd, for one, two and three shifted axes, and the whole
layer (skip half + residual epilogue) on top."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 2e-5


def _mods():
    import gpu_utils as U
    from pytorch3dunet_amd import _native as nat
    from pytorch3dunet_amd.engine import VSrc, _p, _stream
    from pytorch3dunet_amd._engine_base import slab_boxes

    return U, nat, VSrc, _p, _stream, slab_boxes


def _ints(vals):
    return (ctypes.c_int * len(vals))(*vals)


CASES = [
    # N, C0, C1, Cout, (D1, H1, W1), plus axes (z, y, x)
    (2, 8, 16, 32, (3, 5, 6), (0, 1, 1)),      # the shipped patch's pattern: depth exact, height and width 2n + 1
    (1, 16, 32, 32, (2, 4, 9), (1, 0, 0)),
    (1, 8, 24, 20, (3, 2, 5), (1, 1, 1)),      # all three axes, channel counts that are not multiples of 16 / 32
    (1, 4, 64, 64, (5, 10, 10), (0, 0, 1)),
    (1, 8, 16, 16, (1, 1, 1), (1, 1, 1)),      # one low-res cell: 3 x 3 x 3 output, everything is slab
]


@pytest.mark.parametrize("N,C0,C1,Cout,low_dims,plus", CASES)
def test_two_n_plus_one_levels_forward_dgrad_wgrad(N, C0, C1, Cout, low_dims, plus):
    U, nat, VSrc, _p, _stream, slab_boxes = _mods()
    lib = nat.get_lib()
    D1, H1, W1 = low_dims
    D, H, W = (2 * n + e for n, e in zip(low_dims, plus))
    Ctot = C0 + C1
    torch.manual_seed(C1 + 3 * Cout + W)
    skip = torch.randn(N, C0, D, H, W)
    low = torch.randn(N, C1, D1, H1, W1)
    w = torch.randn(Cout, Ctot, 3, 3, 3) / (27 * Ctot) ** 0.5
    ab = torch.randn(N, Ctot, 2)
    dz = torch.randn(N, Cout, D, H, W)
    # ---- reference: autograd through interpolate + cat + affine + conv
    low_l = low.clone().requires_grad_(True)
    w_l = w.clone().requires_grad_(True)
    up = F.interpolate(low_l, size=(D, H, W), mode="nearest")
    g_up = up * ab[:, C0:, 0].view(N, C1, 1, 1, 1) + ab[:, C0:, 1].view(N, C1, 1, 1, 1)
    g_up.retain_grad()
    g_skip = skip * ab[:, :C0, 0].view(N, C0, 1, 1, 1) + ab[:, :C0, 1].view(N, C0, 1, 1, 1)
    ref_up = F.conv3d(g_up, w_l[:, C0:], None, padding=1)
    pre = ref_up + F.conv3d(g_skip, w_l[:, :C0], None, padding=1)
    pre.backward(dz)
    # gradient w.r.t. the affine-transformed upsampled tensor, summed over the children of every low-res cell (what the kernels
    # produce: the GroupNorm backward multiplies by `a` afterwards)
    ups = torch.ones(N, C1, D1, H1, W1, requires_grad=True)
    F.interpolate(ups, size=(D, H, W), mode="nearest").backward(g_up.grad)
    dlow_ref = ups.grad
    dw_ref = w_l.grad

    dev = U.DEV
    wd = w.contiguous().to(dev)
    abd = ab.contiguous().to(dev)
    lowd, skipd, dzd = U.ndhwc(low), U.ndhwc(skip), U.ndhwc(dz)
    src = VSrc(skipd, lowd)
    assert src.plus == tuple(plus)
    a1 = abd[:, C0:].contiguous()             # compact affine rows of the upsampled channels (box launches)
    aff_sub = abd.view(-1)[2 * C0:]           # the same rows inside the (N, Ctot, 2) table (sub-pixel kernels: sample stride Ctot * 2)
    w1 = w[:, C0:].contiguous()
    oz, oy, ox = plus
    # ================= forward
    pk = torch.empty(lib.u3d_subpixel_packed_floats(C1, Cout), dtype=torch.float32, device=dev)
    nat.call("u3d_pack_subpixel_weights", 0, _stream(dev), _p(wd), Cout, Ctot, C0, C1, _p(pk))
    part = torch.full((N, D, H, W, Cout), float("nan"), dtype=torch.float32, device=dev)  # window + slab must cover every voxel
    nat.call("u3d_subpixel_conv_fwd_win", 0, _stream(dev), _p(lowd), _p(aff_sub), Ctot * 2, _p(pk), _p(part), N, D1, H1, W1, C1, Cout,
             _ints([D, H, W, oz, oy, ox]))
    s_up = src.up_only_struct(a1)
    wp1 = U.pack(w1, 0)
    for box in slab_boxes((D, H, W), plus, 2):
        nat.call("u3d_conv3d_box", 0, _stream(dev), ctypes.byref(s_up), _p(wp1), _p(part), N, D, H, W, Cout, _ints(box), None)
    assert U.relerr(U.ncdhw(part), ref_up.detach()) < TOL
    # the whole layer: skip half with the residual epilogue
    w0 = w[:, :C0].contiguous()
    y, _ = U.conv3d_ex(VSrc(skipd), w0, Cout, relu=1, affine=abd[:, :C0].contiguous(), residual=part, use_ws=False)
    assert U.relerr(U.ncdhw(y), F.relu(pre.detach())) < TOL
    # ================= data gradient -> low-res gradient + GroupNorm-backward sums
    pkd = torch.empty(lib.u3d_subpixel_dgrad_packed_floats(Cout, C1), dtype=torch.float32, device=dev)
    nat.call("u3d_pack_subpixel_dgrad_weights", 0, _stream(dev), _p(wd), Cout, Ctot, C0, C1, _p(pkd))
    dlow = torch.full((N, D1, H1, W1, C1), float("nan"), dtype=torch.float32, device=dev)
    gst = torch.zeros((N, C1, 2), dtype=torch.float64, device=dev)
    win9 = _ints([D, H, W, oz, oy, ox, oz, oy, ox])  # (the first counted voxel of the 2x grid is u = 1 exactly on the shifted axes)
    nat.call("u3d_subpixel_conv_dgrad_win", 0, _stream(dev), _p(dzd), _p(pkd), _p(lowd), _p(dlow), _p(gst), N, D1, H1, W1, C1, Cout, win9)
    dv = torch.full((N, D, H, W, C1), float("nan"), dtype=torch.float32, device=dev)
    s_dz = VSrc(dzd).struct()
    wpd1 = U.pack(w1, 1)
    mask = _ints([2 * oz, 2 * oy, 2 * ox])
    for box in slab_boxes((D, H, W), plus, 3):
        nat.call("u3d_conv3d_box", 0, _stream(dev), ctypes.byref(s_dz), _p(wpd1), _p(dv), N, D, H, W, C1, _ints(box), mask)
    lz, ly, lx = src.los
    nat.call("u3d_nearest_childsum_add", 0, _stream(dev), _p(dv), _p(lowd), _p(dlow), _p(gst), N, D, H, W, D1, H1, W1, C1, _p(lz),
             _p(ly), _p(lx), oz, oy, ox)
    assert U.relerr(U.ncdhw(dlow), dlow_ref) < TOL
    s_ref = torch.stack([dlow_ref.double().sum(dim=(2, 3, 4)), (dlow_ref.double() * low.double()).sum(dim=(2, 3, 4))], dim=-1)
    assert U.relerr(gst.cpu(), s_ref) < 1e-5
    # ================= weight gradient of the upsampled channels
    dw = torch.full((Cout, Ctot, 3, 3, 3), float("nan"), dtype=torch.float32, device=dev)
    boxes = slab_boxes((D, H, W), plus, 2)
    nws = max([lib.u3d_subpixel_wgrad_workspace_floats(N, D1, H1, W1, C1, Cout)] +
              [lib.u3d_wgrad_workspace_floats(N, b[3] - b[0], b[4] - b[1], b[5] - b[2], C1, Cout) for b in boxes])
    ws = torch.empty(nws, dtype=torch.float32, device=dev)
    nat.call("u3d_subpixel_conv_wgrad_win", 0, _stream(dev), _p(lowd), _p(aff_sub), Ctot * 2, _p(dzd), _p(dw.view(-1)[C0 * 27:]), Ctot, N,
             D1, H1, W1, C1, Cout, _p(ws), nws, win9)
    for box in boxes:
        tmp = torch.full((Cout, C1, 3, 3, 3), float("nan"), dtype=torch.float32, device=dev)
        nat.call("u3d_conv3d_wgrad_box", 0, _stream(dev), ctypes.byref(s_up), _p(dzd), _p(tmp), N, D, H, W, Cout, _p(ws), nws, _ints(box))
        dw[:, C0:] += tmp
    assert U.relerr(dw[:, C0:].cpu(), dw_ref[:, C0:]) < 1e-4


@pytest.mark.parametrize("patch", [(12, 42, 26), (9, 13, 11)])
def test_model_with_two_n_plus_one_levels_runs_the_windowed_kernels_and_matches(patch):
    """UNet3D on a patch whose decoder levels upsample n -> 2n + 1 along some axes (42 -> 21 -> 10 and 26 -> 13 -> 6; 9 x 13 x 11: every
    level): the step with the windowed sub-pixel + slab kernels (default) against the SAME step with U3D_SUBPIXEL_PLUS off (27-tap
    virtual-concat kernels, the path the goldens pinned until round 4) and against the torch.nn module tree on CPU; the profiler's
    entry-point table shows which kernels ran"""
    U, nat, VSrc, _p, _stream, slab_boxes = _mods()
    from pytorch3dunet_amd.unet3d.losses import BCEDiceLoss
    from pytorch3dunet_amd.unet3d.model import UNet3D

    torch.manual_seed(0)
    cpu = UNet3D(1, 1, f_maps=[16, 32, 64], num_groups=8)
    with torch.no_grad():
        for k, p in cpu.named_parameters():
            if "groupnorm" in k:
                p.add_(0.2 * torch.randn_like(p))
    x = torch.randn(2, 1, *patch)
    t = (torch.rand(2, 1, *patch) > 0.5).float()
    crit = BCEDiceLoss()
    _, lg = cpu(x, return_logits=True)
    crit(lg, t).backward()
    ref_g = torch.cat([p.grad.flatten() for p in cpu.parameters()])
    res, calls = {}, {}
    for plus in (True, False):
        dev = UNet3D(1, 1, f_maps=[16, 32, 64], num_groups=8).to(U.DEV).train()
        dev.load_state_dict(cpu.state_dict())
        eng = dev._get_engine()
        eng.subpixel_plus = plus
        assert bool(eng._subpixel_layers(patch).plus) == plus
        prof = nat.EventProfiler()
        nat.profiler = prof
        try:
            _, lgd = dev(x.to(U.DEV), return_logits=True)
            crit(lgd, t.to(U.DEV)).backward()
            torch.cuda.synchronize()
        finally:
            nat.profiler = None
        calls[plus] = set(prof.summary())
        res[plus] = (lgd.detach().cpu(), torch.cat([p.grad.flatten() for p in dev.parameters()]).cpu())
        assert U.relerr(res[plus][0], lg.detach()) < 1e-4, plus
        assert ((res[plus][1] - ref_g).norm() / ref_g.norm()).item() < 3e-3, plus
    new = {"u3d_subpixel_conv_fwd_win", "u3d_subpixel_conv_dgrad_win", "u3d_subpixel_conv_wgrad_win", "u3d_conv3d_box", "u3d_conv3d_wgrad_box",
           "u3d_nearest_childsum_add", "u3d_gn_bwd_apply_children"}
    assert new <= calls[True] and not (new & calls[False]), (sorted(calls[True]), sorted(calls[False]))
    assert U.relerr(res[True][0], res[False][0]) < 2e-5
    assert ((res[True][1] - res[False][1]).norm() / res[False][1].norm()).item() < 3e-3


@pytest.mark.timeout(600)
def test_no_kernel_of_an_n_plus_one_level_reads_memory_nobody_wrote():
    """the data gradient of an n -> 2n + 1 level fills only the slab boxes of a full-resolution scratch tensor
    (`dv`, _engine_conv._dgrad_subpixel) and relies on u3d_nearest_childsum_add reading nothing outside them — and the forward's `part`
    tensor is written by the windowed kernel plus the slab launches.  Pin both invariants: with every scratch / output buffer
    NaN-poisoned before use (U3D_POISON=1) the step must produce exactly the gradients of the unpoisoned run."""
    import json
    import os
    import subprocess
    import sys

    from conftest import ROOT

    code = r'''
import os, sys, json
sys.path.insert(0, os.path.join(os.getcwd(), "pytorch-3dunet_amd"))
import torch
from pytorch3dunet_amd.unet3d.model import UNet3D
from pytorch3dunet_amd.unet3d.losses import BCEDiceLoss
dev = torch.device("cuda", 0)
out = {}
for patch in ((12, 42, 26), (10, 21, 37)):          # 42 -> 21 -> 10 and 26 -> 13 -> 6: n -> 2n + 1 along y / x at different levels
    torch.manual_seed(0)
    m = UNet3D(1, 1, f_maps=[16, 32, 64], num_groups=8).to(dev).train()
    x = torch.randn(1, 1, *patch, device=dev); t = (torch.rand(1, 1, *patch, device=dev) > 0.5).float()
    _, lg = m(x, return_logits=True)
    BCEDiceLoss()(lg, t).backward()
    g = torch.cat([p.grad.flatten() for p in m.parameters()])
    out[str(patch)] = [bool(torch.isfinite(g).all()), float(g.double().abs().sum()), float(lg.double().abs().sum())]
print("RESULT " + json.dumps(out))
'''
    res = {}
    for poison in ("0", "1"):
        proc = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=500, cwd=ROOT,
                              env=dict(os.environ, U3D_POISON=poison))
        assert proc.returncode == 0, proc.stderr[-2000:]
        res[poison] = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    for k, (finite, gsum, lsum) in res["1"].items():
        assert finite, k
        assert (gsum, lsum) == tuple(res["0"][k][1:]), (k, res["0"][k], res["1"][k])
