"""-m gpu: volumes whose sizes are NOT multiples of the 4 x 8 x 8 tile on the PERSISTENT kernels (round 5).

The reference's shipped training patch is 80 x 170 x 170 (resources/3DUnet_confocal_boundary/train_config.yml:94; pooled levels 85,
42, 21) and its halo'd prediction input 112 x 234 x 234 (test_config.yml:37-40): until round 5 every such level fell to the generic
one-block-per-tile kernel.  Here: the ragged instantiations of `conv3d_mfma_reg_kernel` (every N-tile count, affine / no-affine /
exact-2x virtual sources, the 16-column variant) and the ragged face masks of `conv3d_wgrad_kernel<.., REG>` against F.conv3d CPU
autograd through the C-ABI, `u3d_conv3d_variant` asserting WHICH variant ran, and the generic kernel (tuning key 3 = 2: the
round-4 gate) as a second witness on the same inputs."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 2e-5


def _mods():
    import gpu_utils as U
    from pytorch3dunet_amd import _native as nat
    from pytorch3dunet_amd.engine import VSrc

    return U, nat, VSrc


def _stats(t, other=None):
    o = t.double() if other is None else other.double()
    return torch.stack([t.double().sum(dim=(2, 3, 4)), (t.double() * o).sum(dim=(2, 3, 4))], dim=-1)


RAGGED = [
    # N, Cin, Cout, D, H, W, forced N-tiles (0 = automatic)
    (2, 32, 32, 10, 21, 21, 0),    # the bottom level of the shipped patch (80x170x170 -> 10x21x21), NT = 1
    (1, 32, 64, 6, 10, 12, 2),     # NT = 2 forced on a small volume: remainders 2 / 2 / 4
    (1, 32, 96, 5, 9, 17, 3),      # NT = 3: remainders 1 / 1 / 1 (one valid voxel in the last tile of every axis)
    (1, 48, 64, 7, 15, 23, 0),     # remainders 3 / 7 / 7 (one missing), 3 chunks
    (1, 20, 24, 6, 10, 9, 0),      # channel padding inside a chunk and in the N-tile, ragged
    (1, 32, 64, 34, 66, 66, 0),    # 729 tiles: NT = 2 chosen by the launcher, blocks walk several tiles incl. ragged ones
]


@pytest.mark.parametrize("N,Cin,Cout,D,H,W,nt", RAGGED)
def test_ragged_volumes_forward_on_the_persistent_kernel(N, Cin, Cout, D, H, W, nt):
    U, nat, VSrc = _mods()
    lib = nat.get_lib()
    assert lib.u3d_conv3d_variant(N, D, H, W, Cin, Cout, 0, 0) == 2  # persistent kernel, ragged last tiles
    torch.manual_seed(Cin * 31 + Cout + D)
    x = torch.randn(N, Cin, D, H, W)
    w = torch.randn(Cout, Cin, 3, 3, 3) / (27 * Cin) ** 0.5
    ab = torch.randn(N, Cin, 2)
    aff = ab.contiguous().to(U.DEV)
    g = x * ab[:, :, 0].view(N, Cin, 1, 1, 1) + ab[:, :, 1].view(N, Cin, 1, 1, 1)
    r = torch.randn(N, Cout, D, H, W)
    ref = F.relu(F.conv3d(g, w, None, padding=1))
    ref_res = F.relu(F.conv3d(g, w, None, padding=1) + r)
    out = {}
    for gate in (0, 2):  # 0: ragged persistent kernel; 2: the round-4 gate -> generic kernel
        nat.call("u3d_set_tuning", 3, gate)
        nat.call("u3d_set_tuning", 0, nt)
        try:
            assert lib.u3d_conv3d_variant(N, D, H, W, Cin, Cout, 0, 0) == (2 if gate == 0 else 0)
            st = torch.zeros((N, Cout, 2), dtype=torch.float64, device=U.DEV)
            y = U.conv3d(VSrc(U.ndhwc(x)), w, Cout, relu=1, affine=aff, out_stats=st)
            st2 = torch.zeros((N, Cout, 2), dtype=torch.float64, device=U.DEV)
            y2, _ = U.conv3d_ex(VSrc(U.ndhwc(x)), w, Cout, relu=1, affine=aff, out_stats=st2, residual=U.ndhwc(r), use_ws=False)
            torch.cuda.synchronize()
        finally:
            nat.call("u3d_set_tuning", 3, 0)
            nat.call("u3d_set_tuning", 0, 0)
        out[gate] = (U.ncdhw(y), st.cpu(), U.ncdhw(y2), st2.cpu())
        assert U.relerr(out[gate][0], ref) < TOL and U.relerr(out[gate][2], ref_res) < TOL, gate
        assert U.relerr(out[gate][1], _stats(ref)) < 1e-5 and U.relerr(out[gate][3], _stats(ref_res)) < 1e-5, gate
    assert U.relerr(out[0][0], out[2][0]) < 1e-5


@pytest.mark.parametrize("N,Cin,Cout,D,H,W,nt", RAGGED[:5])
def test_ragged_volumes_data_and_weight_gradient(N, Cin, Cout, D, H, W, nt):
    """data gradient (no-affine instantiation, GroupNorm-backward sums against gx) and weight gradient (REG staging with ragged masks)"""
    U, nat, VSrc = _mods()
    lib = nat.get_lib()
    assert lib.u3d_conv3d_variant(N, D, H, W, Cout, Cin, 0, 0) == 2
    assert lib.u3d_conv3d_wgrad_variant(N, D, H, W, Cin, Cout, 0) & 3 == 2
    torch.manual_seed(Cin + 11 * Cout + W)
    x = torch.randn(N, Cin, D, H, W)
    w = torch.randn(Cout, Cin, 3, 3, 3) / (27 * Cin) ** 0.5
    dz = torch.randn(N, Cout, D, H, W)
    ab = torch.randn(N, Cin, 2)
    aff = ab.contiguous().to(U.DEV)
    g = x * ab[:, :, 0].view(N, Cin, 1, 1, 1) + ab[:, :, 1].view(N, Cin, 1, 1, 1)
    gl = g.clone().requires_grad_(True)
    wl = w.clone().requires_grad_(True)
    F.conv3d(gl, wl, None, padding=1).backward(dz)
    for gate in (0, 2):
        nat.call("u3d_set_tuning", 3, gate)
        nat.call("u3d_set_tuning", 0, nt)  # (ignored by the launcher when it does not divide the N-tile count)
        try:
            gst = torch.zeros((N, Cin, 2), dtype=torch.float64, device=U.DEV)
            dg = U.conv3d(VSrc(U.ndhwc(dz)), w, Cin, relu=0, mode=1, gx=VSrc(U.ndhwc(x)), gstats=gst)
            dw = U.wgrad(VSrc(U.ndhwc(x)), U.ndhwc(dz), Cout, affine=aff)
            torch.cuda.synchronize()
        finally:
            nat.call("u3d_set_tuning", 3, 0)
            nat.call("u3d_set_tuning", 0, 0)
        assert U.relerr(U.ncdhw(dg), gl.grad) < TOL, gate
        assert U.relerr(gst.cpu(), _stats(gl.grad, x)) < 1e-5, gate
        assert U.relerr(dw.cpu(), wl.grad) < 1e-4, gate


@pytest.mark.parametrize("N,Cin,Cout,D,H,W", [(2, 32, 16, 10, 21, 21), (1, 16, 8, 6, 10, 12), (1, 48, 12, 5, 17, 9)])
def test_ragged_volumes_16_column_variant(N, Cin, Cout, D, H, W):
    """<= 16 produced channels (enc0's 32 -> 16 data gradient at full resolution, 80 x 170 x 170 in the shipped config): the ragged
    16-column instantiation — forward with affine / ReLU / statistics / residual, data gradient with the GroupNorm-backward sums"""
    U, nat, VSrc = _mods()
    assert nat.get_lib().u3d_conv3d_variant(N, D, H, W, Cin, Cout, 0, 0) == 2
    torch.manual_seed(5 * Cin + Cout)
    x = torch.randn(N, Cin, D, H, W)
    w = torch.randn(Cout, Cin, 3, 3, 3) / (27 * Cin) ** 0.5
    ab = torch.randn(N, Cin, 2)
    aff = ab.contiguous().to(U.DEV)
    g = x * ab[:, :, 0].view(N, Cin, 1, 1, 1) + ab[:, :, 1].view(N, Cin, 1, 1, 1)
    r = torch.randn(N, Cout, D, H, W)
    ref = F.relu(F.conv3d(g, w, None, padding=1) + r)
    st = torch.zeros((N, Cout, 2), dtype=torch.float64, device=U.DEV)
    y, _ = U.conv3d_ex(VSrc(U.ndhwc(x)), w, Cout, relu=1, affine=aff, out_stats=st, residual=U.ndhwc(r), use_ws=False)
    assert U.relerr(U.ncdhw(y), ref) < TOL and U.relerr(st.cpu(), _stats(ref)) < 1e-5
    # data gradient of a (Cout -> 32) layer whose input has <= 16 channels
    Kd = 32
    wd = torch.randn(Kd, Cout, 3, 3, 3) / (27 * Cout) ** 0.5
    xx = torch.randn(N, Cout, D, H, W)
    dz = torch.randn(N, Kd, D, H, W)
    xl = xx.clone().requires_grad_(True)
    F.conv3d(xl, wd, None, padding=1).backward(dz)
    gst = torch.zeros((N, Cout, 2), dtype=torch.float64, device=U.DEV)
    dg = U.conv3d(VSrc(U.ndhwc(dz)), wd, Cout, relu=0, mode=1, gx=VSrc(U.ndhwc(xx)), gstats=gst)
    assert U.relerr(U.ncdhw(dg), xl.grad) < TOL and U.relerr(gst.cpu(), _stats(xl.grad, xx)) < 1e-5
    # tap-pairing weight gradient (Cin <= 16) with ragged masks
    if Cout <= 16:
        wl = torch.zeros(Kd, Cout, 3, 3, 3, requires_grad=True)
        F.conv3d(xx, wl, None, padding=1).backward(dz)
        assert nat.get_lib().u3d_conv3d_wgrad_variant(N, D, H, W, Cout, Kd, 0) == 6
        dw = U.wgrad(VSrc(U.ndhwc(xx)), U.ndhwc(dz), Kd)
        assert U.relerr(dw.cpu(), wl.grad) < 1e-4


@pytest.mark.parametrize("size,nt", [((10, 20, 20), 0), ((6, 12, 28), 2), ((2, 4, 6), 0)])
def test_ragged_volumes_exact_2x_virtual_concat(size, nt):
    """skip ++ exact-2x upsampled low-res tensor on a volume that is not a multiple of the tile (85 -> 170 in the shipped config:
    170 % 8 = 2): the VIRT ragged instantiation forward, the same source as gx of a data gradient, and the weight gradient"""
    U, nat, VSrc = _mods()
    torch.manual_seed(17)
    N, C0, C1, Cout = 2, 16, 32, 64
    D, H, W = size
    assert nat.get_lib().u3d_conv3d_variant(N, D, H, W, C0 + C1, Cout, 1, 0) == 2
    skip = torch.randn(N, C0, D, H, W)
    low = torch.randn(N, C1, D // 2, H // 2, W // 2)
    cat = torch.cat((skip, F.interpolate(low, size=size, mode="nearest")), dim=1)
    w = torch.randn(Cout, C0 + C1, 3, 3, 3) / (27 * (C0 + C1)) ** 0.5
    ab = torch.randn(N, C0 + C1, 2)
    aff = ab.contiguous().to(U.DEV)
    g = cat * ab[:, :, 0].view(N, -1, 1, 1, 1) + ab[:, :, 1].view(N, -1, 1, 1, 1)
    gl = g.clone().requires_grad_(True)
    wl = w.clone().requires_grad_(True)
    pre = F.conv3d(gl, wl, None, padding=1)
    ref = F.relu(pre)
    dz = torch.randn(N, Cout, D, H, W)
    pre.backward(dz)
    src = VSrc(U.ndhwc(skip), U.ndhwc(low))
    nat.call("u3d_set_tuning", 0, nt)
    try:
        st = torch.zeros((N, Cout, 2), dtype=torch.float64, device=U.DEV)
        y = U.conv3d(src, w, Cout, relu=1, affine=aff, out_stats=st)
        gst = torch.zeros((N, C0 + C1, 2), dtype=torch.float64, device=U.DEV)
        dg = U.conv3d(VSrc(U.ndhwc(dz)), w, C0 + C1, relu=0, mode=1, gx=src, gstats=gst)
        dw = U.wgrad(src, U.ndhwc(dz), Cout, affine=aff)
        torch.cuda.synchronize()
    finally:
        nat.call("u3d_set_tuning", 0, 0)
    assert U.relerr(U.ncdhw(y), ref) < TOL and U.relerr(st.cpu(), _stats(ref)) < 1e-5
    assert U.relerr(U.ncdhw(dg), gl.grad) < TOL
    assert U.relerr(gst.cpu(), _stats(gl.grad, cat)) < 1e-5
    assert U.relerr(dw.cpu(), wl.grad) < 1e-4


@pytest.mark.parametrize("patch", [(20, 44, 44), (12, 42, 26)])
def test_model_on_a_ragged_patch_matches_the_cpu_module_tree_and_the_generic_kernels(patch):
    """UNet3D on a patch shaped like the shipped one (no level is a multiple of the tile; 44 -> 22 -> 11: the decoders are exact 2x
    except the deepest; 42 -> 21 -> 10: 10 -> 21 goes through the general index maps): forward + BCEDice backward with the ragged
    persistent kernels vs the SAME step with the round-4 gate (generic kernels, the path the goldens pin) and vs the torch.nn module
    tree on CPU (the reference's graph; tests/test_boundary.py pins it to the goldens)"""
    U, nat, VSrc = _mods()
    from pytorch3dunet_amd.unet3d.losses import BCEDiceLoss
    from pytorch3dunet_amd.unet3d.model import UNet3D

    torch.manual_seed(0)
    cpu = UNet3D(1, 1, f_maps=[16, 32, 64], num_groups=8)
    with torch.no_grad():
        for k, p in cpu.named_parameters():
            if "groupnorm" in k:
                p.add_(0.2 * torch.randn_like(p))
    x = torch.randn(1, 1, *patch)
    t = (torch.rand(1, 1, *patch) > 0.5).float()
    crit = BCEDiceLoss()
    _, lg = cpu(x, return_logits=True)
    crit(lg, t).backward()
    ref_g = torch.cat([p.grad.flatten() for p in cpu.parameters()])
    res = {}
    for gate in (0, 2):
        nat.call("u3d_set_tuning", 3, gate)
        try:
            dev = UNet3D(1, 1, f_maps=[16, 32, 64], num_groups=8).to(U.DEV).train()
            dev.load_state_dict(cpu.state_dict())
            eng = dev._get_engine()
            eng.debug = {}
            _, lgd = dev(x.to(U.DEV), return_logits=True)
            tape = eng.debug["tape"]
            eng.debug = None
            crit(lgd, t.to(U.DEV)).backward()
            torch.cuda.synchronize()
        finally:
            nat.call("u3d_set_tuning", 3, 0)
        res[gate] = (lgd.detach().cpu(), torch.cat([p.grad.flatten() for p in dev.parameters()]).cpu())
        assert U.relerr(res[gate][0], lg.detach()) < 1e-4, gate
        # gradients: fp32 vs fp32 with different summation orders.  Without a flipped ReLU / arg-max decision the two agree to ~2e-6
        # (the second patch); every flip at the 5 x 11 x 11 level of this small net is a 1e-3 event, and WHICH pre-activations land
        # within round-off of zero changes with any 1e-7 perturbation upstream (round 5: 2.98e-3 / 2.80e-3 for the two gates on the
        # first patch, round 6 after the input-statistics kernel changed its summation order: 7.2e-3 / 3.1e-3).  So: the smoke test's
        # global bound, or — beyond it — proof that decisions are all that differs: with THIS run's masks and arg-maxes imposed the
        # float64 oracle must reproduce every gradient to 1e-4 (tests/test_gpu_model.py's decision-consistent gate)
        e_cpu = ((res[gate][1] - ref_g).norm() / ref_g.norm()).item()
        if e_cpu >= 3e-3:
            import unet3d_oracle as orc

            assert e_cpu < 2e-2, (gate, e_cpu)
            ncdhw = lambda v: v.permute(0, 4, 1, 2, 3).contiguous().cpu()  # noqa: E731
            masks = [ncdhw(r.y > 0) for r in tape.convs]
            argmax = [ncdhw(am) for (_, am, _) in tape.pools]
            sd = {k: v.detach().clone() for k, v in cpu.state_dict().items()}
            _, _, g64 = orc.forward_backward_decided(sd, x, t, masks, argmax, 8, True, True, "bce_dice")
            first_gamma = next(k for k, _ in dev.named_parameters() if k.endswith("groupnorm.weight"))
            for k, p in dev.named_parameters():
                e = orc.rel_err(p.grad.detach().cpu().double(), g64[k])
                assert e < (1e-3 if k == first_gamma else 1e-4), (gate, k, e)
            print(f"gate {gate}: rel-L2 vs the CPU module tree {e_cpu:.2e} through decision flips; decision-consistent float64 gate passed")
        del tape
    assert U.relerr(res[0][0], res[2][0]) < 2e-5
    assert ((res[0][1] - res[2][1]).norm() / res[2][1].norm()).item() < 2e-2  # (two independent sets of such decisions)
    D, H, W = patch
    lib = nat.get_lib()
    assert lib.u3d_conv3d_variant(1, D, H, W, 16, 16, 0, 0) == 2 and lib.u3d_conv3d_wgrad_variant(1, D, H, W, 16, 16, 0) == 6
