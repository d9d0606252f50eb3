"""-m gpu: every HIP kernel of the hot path, called through the C-ABI, against the CPU oracle
(torch-functional restatement, oracle/unet3d_oracle.py) on the same seeded inputs.  Tolerance: 1e-3 relative to
the tensor's max-abs (BASELINE.json north_star: "within 1e-3 rel fp32"); most kernels are held to 2e-5."""
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

    return U, nat, VSrc, _p, _stream


def test_device_is_gfx950_and_native_lib_loaded():
    U, nat, *_ = _mods()
    nat.call("u3d_check_device", 0)
    assert "gfx950" in torch.cuda.get_device_properties(0).gcnArchName


CONV_CASES = [
    # N, Cin, Cout, D, H, W, affine, relu
    (1, 16, 32, 8, 16, 16, False, 0),
    (1, 16, 32, 8, 16, 16, True, 1),
    (2, 32, 32, 4, 8, 8, True, 1),
    (1, 1, 16, 8, 16, 16, True, 1),       # first layer: C=1, scalar load path, Cout < 32
    (1, 3, 8, 5, 9, 7, True, 0),          # odd dims, odd channels
    (2, 32, 64, 9, 13, 11, True, 1),      # odd dims, partial tiles, 2 n-tiles
    (1, 96, 32, 8, 16, 16, True, 1),      # 6 chunks (dec2.conv1 channel shape)
    (1, 64, 128, 4, 8, 8, False, 0),      # 4 n-tiles
    (1, 20, 24, 6, 10, 9, True, 1),       # channel padding inside a chunk (vec path, C%4==0 but C%16!=0)
    (2, 32, 16, 8, 16, 16, True, 1),      # <= 16 output channels, aligned dims: paired-y variant
    (1, 16, 8, 4, 8, 8, True, 0),
    (1, 48, 12, 4, 16, 8, False, 1),
]


@pytest.mark.parametrize("N,Cin,Cout,D,H,W,affine,relu", CONV_CASES)
def test_conv3d_fwd(N, Cin, Cout, D, H, W, affine, relu):
    U, nat, VSrc, _p, _stream = _mods()
    torch.manual_seed(Cin * 131 + Cout)
    x = torch.randn(N, Cin, D, H, W)
    w = torch.randn(Cout, Cin, 3, 3, 3) / (27 * Cin) ** 0.5
    aff = None
    g = x
    if affine:
        ab = torch.randn(N, Cin, 2)
        aff = ab.contiguous().to(U.DEV)
        g = x * ab[:, :, 0].view(N, Cin, 1, 1, 1) + ab[:, :, 1].view(N, Cin, 1, 1, 1)
    ref = F.conv3d(g, w, None, padding=1)
    if relu:
        ref = F.relu(ref)
    src = VSrc(U.ndhwc(x))
    st = torch.zeros((N, Cout, 2), dtype=torch.float64, device=U.DEV)
    y = U.conv3d(src, w, Cout, relu=relu, affine=aff, out_stats=st)
    got = U.ncdhw(y)
    assert U.relerr(got, ref) < TOL
    # fused epilogue statistics = per-(n,channel) sum / sum of squares of the written output
    s_ref = torch.stack([ref.double().sum(dim=(2, 3, 4)), (ref.double() ** 2).sum(dim=(2, 3, 4))], dim=-1)
    assert U.relerr(st.cpu(), s_ref) < 1e-5
    # cross-check the naive device kernel (used for the full-size checks)
    yn = U.conv3d_naive(src, w, Cout, relu=relu, affine=aff)
    assert U.relerr(U.ncdhw(yn), ref) < TOL


def test_conv3d_fwd_nt2_many_tiles():
    """>= 512 tiles with 64 output channels selects the BN=64 (NT=2) kernel"""
    U, nat, VSrc, _p, _stream = _mods()
    torch.manual_seed(5)
    N, Cin, Cout, D, H, W = 1, 32, 64, 32, 64, 64
    x = torch.randn(N, Cin, D, H, W)
    w = torch.randn(Cout, Cin, 3, 3, 3) / (27 * Cin) ** 0.5
    ref = F.relu(F.conv3d(x, w, None, padding=1))
    y = U.conv3d(VSrc(U.ndhwc(x)), w, Cout, relu=1)
    assert U.relerr(U.ncdhw(y), ref) < TOL


def test_conv3d_fwd_nt3_many_tiles():
    """96 output channels (3 N-tiles) with >= 512 tiles selects the BN=96 (NT=3) kernel — the dgrad shape of the
    96->32 decoder conv"""
    U, nat, VSrc, _p, _stream = _mods()
    torch.manual_seed(6)
    N, Cin, Cout, D, H, W = 1, 32, 96, 32, 64, 64
    x = torch.randn(N, Cin, D, H, W)
    w = torch.randn(Cout, Cin, 3, 3, 3) / (27 * Cin) ** 0.5
    ref = F.conv3d(x, w, None, padding=1)
    st = torch.zeros((N, Cout, 2), dtype=torch.float64, device=U.DEV)
    y = U.conv3d(VSrc(U.ndhwc(x)), w, Cout, relu=0, out_stats=st)
    assert U.relerr(U.ncdhw(y), ref) < TOL
    s_ref = torch.stack([ref.double().sum(dim=(2, 3, 4)), (ref.double() ** 2).sum(dim=(2, 3, 4))], dim=-1)
    assert U.relerr(st.cpu(), s_ref) < 1e-5


@pytest.mark.parametrize("forced_nt", [1, 2, 3])
def test_conv3d_forced_ntile_variants_agree(forced_nt):
    """u3d_set_tuning(0, NT) forces the N-tiles-per-block variant; every variant gives the same result on a
    6-chunk virtual-concat source (the prefetch pipeline crosses the skip -> upsampled source switch)"""
    U, nat, VSrc, _p, _stream = _mods()
    torch.manual_seed(12)
    N, C0, C1, Cout = 1, 32, 64, 192
    D, H, W = 8, 16, 16
    skip = torch.randn(N, C0, D, H, W)
    low = torch.randn(N, C1, D // 2, H // 2, W // 2)
    cat = torch.cat((skip, F.interpolate(low, size=(D, H, W), mode="nearest")), dim=1)
    w = torch.randn(Cout, C0 + C1, 3, 3, 3) / (27 * (C0 + C1)) ** 0.5
    ab = torch.randn(N, C0 + C1, 2)
    g = cat * ab[:, :, 0].view(N, -1, 1, 1, 1) + ab[:, :, 1].view(N, -1, 1, 1, 1)
    ref = F.relu(F.conv3d(g, w, None, padding=1))
    src = VSrc(U.ndhwc(skip), U.ndhwc(low))
    nat.call("u3d_set_tuning", 0, forced_nt)
    try:
        y = U.conv3d(src, w, Cout, relu=1, affine=ab.contiguous().to(U.DEV))
    finally:
        nat.call("u3d_set_tuning", 0, 0)
    assert U.relerr(U.ncdhw(y), ref) < TOL


@pytest.mark.parametrize("split", [1, 3, 16])
def test_conv3d_wgrad_split_override_crosses_samples(split):
    """u3d_set_tuning(1, S) forces the split-K count: S=3 over 16 tiles makes a block walk 6 tiles across the
    sample boundary (per-tile GroupNorm affine reload, double-buffered staging), S=16 is one tile per block"""
    U, nat, VSrc, _p, _stream = _mods()
    torch.manual_seed(13)
    N, Cin, Cout, D, H, W = 2, 32, 32, 4, 16, 16
    x = torch.randn(N, Cin, D, H, W)
    dz = torch.randn(N, Cout, D, H, W)
    ab = torch.randn(N, Cin, 2)
    g = x * ab[:, :, 0].view(N, Cin, 1, 1, 1) + ab[:, :, 1].view(N, Cin, 1, 1, 1)
    wl = torch.zeros(Cout, Cin, 3, 3, 3, requires_grad=True)
    F.conv3d(g, wl, None, padding=1).backward(dz)
    nat.call("u3d_set_tuning", 1, split)
    try:
        dw = U.wgrad(VSrc(U.ndhwc(x)), U.ndhwc(dz), Cout, affine=ab.contiguous().to(U.DEV))
    finally:
        nat.call("u3d_set_tuning", 1, 0)
    assert U.relerr(dw.cpu(), wl.grad) < 1e-4


@pytest.mark.parametrize("size", [(8, 16, 16), (9, 13, 11)])
def test_conv3d_virtual_concat_upsample(size):
    """skip (full-res) ++ nearest-upsampled low-res tensor, never materialised (buildingblocks.py:491,:614)"""
    U, nat, VSrc, _p, _stream = _mods()
    torch.manual_seed(11)
    N, C0, C1, Cout = 2, 8, 16, 32
    D, H, W = size
    D1, H1, W1 = D // 2, H // 2, W // 2
    skip = torch.randn(N, C0, D, H, W)
    low = torch.randn(N, C1, D1, H1, W1)
    cat = torch.cat((skip, F.interpolate(low, size=size, mode="nearest")), dim=1)
    w = torch.randn(Cout, C0 + C1, 3, 3, 3) / (27 * (C0 + C1)) ** 0.5
    ab = torch.randn(N, C0 + C1, 2)
    g = cat * ab[:, :, 0].view(N, -1, 1, 1, 1) + ab[:, :, 1].view(N, -1, 1, 1, 1)
    ref = F.relu(F.conv3d(g, w, None, padding=1))
    src = VSrc(U.ndhwc(skip), U.ndhwc(low))
    y = U.conv3d(src, w, Cout, relu=1, affine=ab.contiguous().to(U.DEV))
    assert U.relerr(U.ncdhw(y), ref) < TOL
    # channel statistics of the virtual tensor
    st = U.chan_stats(src)
    s_ref = torch.stack([cat.double().sum(dim=(2, 3, 4)), (cat.double() ** 2).sum(dim=(2, 3, 4))], dim=-1)
    assert U.relerr(st.cpu(), s_ref) < 1e-5
    # weight gradient through the same virtual source
    dz = torch.randn(N, Cout, D, H, W)
    gl = g.clone().requires_grad_(False)
    wl = w.clone().requires_grad_(True)
    F.conv3d(gl, wl, None, padding=1).backward(dz)
    dw = U.wgrad(src, U.ndhwc(dz), Cout, affine=ab.contiguous().to(U.DEV))
    assert U.relerr(dw.cpu(), wl.grad) < 1e-4


SPLITK_CASES = [
    # N, C0, C1 (nearest-upsampled half of a virtual concat), Cout, D, H, W, residual
    (2, 128, 0, 128, 8, 16, 16, False),   # the bench workload's bottom level: 16 tiles x 4 channel blocks, 8 chunks
    (1, 64, 0, 256, 4, 8, 16, True),      # residual + ReLU in the reduce kernel
    (1, 40, 0, 36, 5, 9, 7, False),       # ragged tiles, 3 chunks with a partial last one, Cout not a multiple of 32
    (1, 32, 64, 32, 4, 8, 8, False),      # virtual concat source
    (2, 272, 0, 64, 4, 8, 8, False),      # 17 chunks > 16 splits: uneven runs
]


@pytest.mark.parametrize("N,C0,C1,Cout,D,H,W,res", SPLITK_CASES)
def test_conv3d_splitk_small_volumes(N, C0, C1, Cout, D, H, W, res):
    """u3d_conv3d_ex on bottom-of-the-U shapes: the channel reduction is split over blocks and summed by a second
    kernel that owns the epilogue — same outputs and statistics as the one-kernel path and as torch"""
    U, nat, VSrc, _p, _stream = _mods()
    torch.manual_seed(C0 + 3 * Cout + D)
    Cin = C0 + C1
    x0 = torch.randn(N, C0, D, H, W)
    x1 = torch.randn(N, C1, D // 2, H // 2, W // 2) if C1 else None
    x = x0 if x1 is None else torch.cat((x0, F.interpolate(x1, size=(D, H, W), mode="nearest")), dim=1)
    w = torch.randn(Cout, Cin, 3, 3, 3) / (27 * Cin) ** 0.5
    ab = torch.randn(N, Cin, 2)
    g = x * ab[:, :, 0].view(N, Cin, 1, 1, 1) + ab[:, :, 1].view(N, Cin, 1, 1, 1)
    r = torch.randn(N, Cout, D, H, W) if res else None
    ref = F.conv3d(g, w, None, padding=1)
    ref = F.relu(ref + r) if res else F.relu(ref)
    src = VSrc(U.ndhwc(x0), U.ndhwc(x1) if x1 is not None else None)
    aff = ab.contiguous().to(U.DEV)
    rd = U.ndhwc(r) if res else None
    st = torch.zeros((N, Cout, 2), dtype=torch.float64, device=U.DEV)
    y, need = U.conv3d_ex(src, w, Cout, relu=1, affine=aff, out_stats=st, residual=rd)
    assert need > 0, "shape is expected to take the split-K path"
    assert U.relerr(U.ncdhw(y), ref) < TOL
    s_ref = torch.stack([ref.double().sum(dim=(2, 3, 4)), (ref.double() ** 2).sum(dim=(2, 3, 4))], dim=-1)
    assert U.relerr(st.cpu(), s_ref) < 1e-5
    # against the one-kernel path (no workspace): identical decisions, sums differ only by fp32 association
    st1 = torch.zeros_like(st)
    y1, _ = U.conv3d_ex(src, w, Cout, relu=1, affine=aff, out_stats=st1, residual=rd, use_ws=False)
    assert U.relerr(y, y1) < 1e-5
    # run-to-run reproducible: the runs are added in a fixed order
    y2, _ = U.conv3d_ex(src, w, Cout, relu=1, affine=aff, out_stats=torch.zeros_like(st), residual=rd)
    assert torch.equal(y, y2)
    # data gradient through the same path, with the GroupNorm-backward sums against a (virtual) gx
    dz = torch.randn(N, Cout, D, H, W)
    xl = x.clone().requires_grad_(True)
    F.conv3d(xl, w, None, padding=1).backward(dz)
    gst = torch.zeros((N, Cin, 2), dtype=torch.float64, device=U.DEV)
    dg, need_d = U.conv3d_ex(VSrc(U.ndhwc(dz)), w, Cin, relu=0, mode=1, gx=src, gstats=gst)
    assert U.relerr(U.ncdhw(dg), xl.grad) < TOL
    sg = torch.stack([xl.grad.double().sum(dim=(2, 3, 4)), (xl.grad.double() * x.double()).sum(dim=(2, 3, 4))], dim=-1)
    assert U.relerr(gst.cpu(), sg) < 1e-5


def test_conv3d_ex_without_workspace_and_large_volumes_do_not_split():
    U, nat, VSrc, _p, _stream = _mods()
    lib = nat.get_lib()
    assert lib.u3d_conv3d_workspace_floats(2, 64, 128, 128, 32, 32) == 0
    assert lib.u3d_conv3d_workspace_floats(2, 8, 16, 16, 16, 128) == 0       # a single chunk: nothing to split
    assert lib.u3d_conv3d_workspace_floats(2, 8, 16, 16, 128, 128) == 8 * 2 * 8 * 16 * 16 * 128
    assert lib.u3d_conv3d_workspace_floats(0, 8, 16, 16, 128, 128) == 0


SUBPIX_CASES = [
    # N, C0 (skip), C1 (upsampled), Cout, D1, H1, W1, affine
    (2, 32, 64, 32, 4, 8, 8, True),     # the bench workload's dec2.c1 channel split on a small volume
    (1, 16, 24, 64, 4, 4, 8, True),     # partial last chunk (24 = 16 + 8), two output blocks
    (1, 8, 16, 20, 3, 5, 6, False),     # ragged low-res tiles, Cout not a multiple of 32, no affine
    (1, 4, 48, 32, 2, 4, 16, True),     # three chunks
]


@pytest.mark.parametrize("N,C0,C1,Cout,D1,H1,W1,affine", SUBPIX_CASES)
def test_subpixel_conv_of_upsampled_half(N, C0, C1, Cout, D1, H1, W1, affine):
    """conv3d(cat(skip, nearest2x(low))) = conv3d(skip, w[:, :C0]) + [8 parity-class 2x2x2 convolutions of low with
    pre-summed taps]; the second term comes from u3d_subpixel_conv_fwd, the sum from u3d_conv3d_residual's epilogue"""
    U, nat, VSrc, _p, _stream = _mods()
    torch.manual_seed(C1 + 5 * Cout + D1)
    D, H, W = 2 * D1, 2 * H1, 2 * W1
    Ctot = C0 + C1
    skip = torch.randn(N, C0, D, H, W)
    low = torch.randn(N, C1, D1, H1, W1)
    w = torch.randn(Cout, Ctot, 3, 3, 3) / (27 * Ctot) ** 0.5
    ab = torch.randn(N, Ctot, 2) if affine else torch.stack([torch.ones(N, Ctot), torch.zeros(N, Ctot)], dim=-1)
    cat = torch.cat((skip, F.interpolate(low, size=(D, H, W), mode="nearest")), dim=1)
    g = cat * ab[:, :, 0].view(N, Ctot, 1, 1, 1) + ab[:, :, 1].view(N, Ctot, 1, 1, 1)
    ref_up = F.conv3d(g[:, C0:], w[:, C0:].contiguous(), None, padding=1)
    ref = F.relu(F.conv3d(g, w, None, padding=1))
    lib = nat.get_lib()
    wd = w.contiguous().to(U.DEV)
    abd = ab.contiguous().to(U.DEV)
    npk = lib.u3d_subpixel_packed_floats(C1, Cout)
    pk = torch.empty(npk, dtype=torch.float32, device=U.DEV)
    nat.call("u3d_pack_subpixel_weights", 0, _stream(U.DEV), _p(wd), Cout, Ctot, C0, C1, _p(pk))
    lowd = U.ndhwc(low)
    part = torch.full((N, D, H, W, Cout), float("nan"), dtype=torch.float32, device=U.DEV)  # every element must be written
    aff_sub = abd.view(-1)[2 * C0:] if affine else None   # rows C0.. of sample 0; sample stride stays Ctot*2
    nat.call("u3d_subpixel_conv_fwd", 0, _stream(U.DEV), _p(lowd), _p(aff_sub), Ctot * 2, _p(pk), _p(part), N, D1, H1, W1, C1,
             Cout, None, 0)
    assert U.relerr(U.ncdhw(part), ref_up) < TOL
    # with the scratch buffer these small grids split the channel reduction over blocks (fixed-order sum of the runs)
    need = lib.u3d_subpixel_fwd_workspace_floats(N, D1, H1, W1, C1, Cout)
    assert (need > 0) == (C1 > 16)
    if need:
        ws = torch.empty(need, dtype=torch.float32, device=U.DEV)
        part2 = torch.full_like(part, float("nan"))
        nat.call("u3d_subpixel_conv_fwd", 0, _stream(U.DEV), _p(lowd), _p(aff_sub), Ctot * 2, _p(pk), _p(part2), N, D1, H1, W1, C1,
                 Cout, _p(ws), need)
        assert U.relerr(U.ncdhw(part2), ref_up) < TOL
        assert U.relerr(part2, part) < 1e-5
    # skip half + residual epilogue = the whole layer
    w0 = w[:, :C0].contiguous()
    a0 = abd[:, :C0].contiguous() if affine else None
    st = torch.zeros((N, Cout, 2), dtype=torch.float64, device=U.DEV)
    y, _ = U.conv3d_ex(VSrc(U.ndhwc(skip)), w0, Cout, relu=1, affine=a0, out_stats=st, residual=part, use_ws=False)
    assert U.relerr(U.ncdhw(y), ref) < TOL
    s_ref = torch.stack([ref.double().sum(dim=(2, 3, 4)), (ref.double() ** 2).sum(dim=(2, 3, 4))], dim=-1)
    assert U.relerr(st.cpu(), s_ref) < 1e-5


@pytest.mark.parametrize("N,C1,Cout,D1,H1,W1", [(2, 64, 32, 4, 8, 8), (1, 128, 64, 2, 4, 8), (1, 24, 20, 3, 5, 6),
                                                 (1, 96, 48, 2, 4, 16)])
def test_subpixel_conv_dgrad_to_low_res(N, C1, Cout, D1, H1, W1):
    """d/d(low) of conv3d(nearest2x(low), w) in one pass over dz (4x4x4 taps at stride 2 with pre-summed weights) =
    the reference's conv data gradient followed by the sum over the children of every low-res voxel"""
    U, nat, VSrc, _p, _stream = _mods()
    torch.manual_seed(C1 + 3 * Cout + W1)
    D, H, W = 2 * D1, 2 * H1, 2 * W1
    C0 = 8
    Ctot = C0 + C1
    w = torch.randn(Cout, Ctot, 3, 3, 3) / (27 * Ctot) ** 0.5
    low = torch.randn(N, C1, D1, H1, W1)
    dz = torch.randn(N, Cout, D, H, W)
    gl = low.clone().requires_grad_(True)
    F.conv3d(F.interpolate(gl, size=(D, H, W), mode="nearest"), w[:, C0:].contiguous(), None, padding=1).backward(dz)
    ref = gl.grad
    lib = nat.get_lib()
    wd = w.contiguous().to(U.DEV)
    pk = torch.empty(lib.u3d_subpixel_dgrad_packed_floats(Cout, C1), dtype=torch.float32, device=U.DEV)
    nat.call("u3d_pack_subpixel_dgrad_weights", 0, _stream(U.DEV), _p(wd), Cout, Ctot, C0, C1, _p(pk))
    lowd = U.ndhwc(low)
    out = torch.full((N, D1, H1, W1, C1), float("nan"), dtype=torch.float32, device=U.DEV)
    gst = torch.zeros((N, C1, 2), dtype=torch.float64, device=U.DEV)
    nat.call("u3d_subpixel_conv_dgrad", 0, _stream(U.DEV), _p(U.ndhwc(dz)), _p(pk), _p(lowd), _p(out), _p(gst), N, D1, H1, W1, C1,
             Cout)
    assert U.relerr(U.ncdhw(out), ref) < TOL
    s_ref = torch.stack([ref.double().sum(dim=(2, 3, 4)), (ref.double() * low.double()).sum(dim=(2, 3, 4))], dim=-1)
    assert U.relerr(gst.cpu(), s_ref) < 1e-5
    # replica rows of the sums (round 6): same gradient bit for bit, rows add up to the plain table
    out2 = torch.empty_like(out)
    rows = torch.zeros((8, N, C1, 2), dtype=torch.float64, device=U.DEV)
    nat.call("u3d_subpixel_conv_dgrad_reps", 0, _stream(U.DEV), _p(U.ndhwc(dz)), _p(pk), _p(lowd), _p(out2), _p(rows), N, D1, H1, W1, C1,
             Cout, 8)
    assert torch.equal(out, out2) and torch.allclose(rows.sum(0), gst, rtol=1e-12, atol=1e-9)


@pytest.fixture
def _tuning_key_22(request):
    from pytorch3dunet_amd import _native as nat
    nat.call("u3d_set_tuning", 22, request.param)
    yield request.param
    nat.call("u3d_set_tuning", 22, 0)


@pytest.mark.parametrize("_tuning_key_22", [0, 1, 2], indirect=True, ids=["reduce-default", "reduce-per-output", "reduce-read-once"])
@pytest.mark.parametrize("N,C0,C1,Cout,D1,H1,W1,affine", [(2, 32, 64, 32, 4, 8, 8, True), (1, 16, 40, 48, 2, 4, 8, True),
                                                           (1, 8, 24, 20, 3, 5, 6, False), (2, 4, 32, 32, 2, 4, 16, True),
                                                           (1, 8, 128, 64, 2, 4, 8, True)])
def test_subpixel_conv_wgrad_and_strided_skip_half(N, C0, C1, Cout, D1, H1, W1, affine, _tuning_key_22):
    """weight gradient of conv3d(cat(skip, nearest2x(low))): upsampled channels from the 64 (class, tap-half) matrices over
    the low-res grid (u3d_subpixel_conv_wgrad), skip channels from u3d_conv3d_wgrad_strided — both write their channel slice
    of ONE (Cout, C0+C1, 3,3,3) gradient.  Both reductions of the 64 matrices (key 22: one thread per output / every partial read
    once, round 6; the last case has the 256 blocks from which the default picks the latter)"""
    U, nat, VSrc, _p, _stream = _mods()
    torch.manual_seed(C1 + 7 * Cout + H1)
    D, H, W = 2 * D1, 2 * H1, 2 * W1
    Ctot = C0 + C1
    skip = torch.randn(N, C0, D, H, W)
    low = torch.randn(N, C1, D1, H1, W1)
    dz = torch.randn(N, Cout, D, H, W)
    ab = torch.randn(N, Ctot, 2) if affine else torch.stack([torch.ones(N, Ctot), torch.zeros(N, Ctot)], dim=-1)
    cat = torch.cat((skip, F.interpolate(low, size=(D, H, W), mode="nearest")), dim=1)
    g = cat * ab[:, :, 0].view(N, Ctot, 1, 1, 1) + ab[:, :, 1].view(N, Ctot, 1, 1, 1)
    wl = torch.zeros(Cout, Ctot, 3, 3, 3, requires_grad=True)
    F.conv3d(g, wl, None, padding=1).backward(dz)
    ref = wl.grad
    lib = nat.get_lib()
    abd = ab.contiguous().to(U.DEV)
    dzd = U.ndhwc(dz)
    dw = torch.full((Cout, Ctot, 3, 3, 3), float("nan"), dtype=torch.float32, device=U.DEV)
    nws = max(lib.u3d_subpixel_wgrad_workspace_floats(N, D1, H1, W1, C1, Cout), lib.u3d_wgrad_workspace_floats(N, D, H, W, C0, Cout))
    ws = torch.empty(nws, dtype=torch.float32, device=U.DEV)
    aff_sub = abd.view(-1)[2 * C0:] if affine else None
    nat.call("u3d_subpixel_conv_wgrad", 0, _stream(U.DEV), _p(U.ndhwc(low)), _p(aff_sub), Ctot * 2, _p(dzd), _p(dw.view(-1)[C0 * 27:]),
             Ctot, N, D1, H1, W1, C1, Cout, _p(ws), nws)
    a0 = abd[:, :C0].contiguous() if affine else None
    s0 = VSrc(U.ndhwc(skip)).struct(a0)
    nat.call("u3d_conv3d_wgrad_strided", 0, _stream(U.DEV), ctypes.byref(s0), _p(dzd), _p(dw), Ctot, N, D, H, W, Cout, _p(ws), nws)
    got = dw.cpu()
    assert torch.isfinite(got).all()
    assert U.relerr(got[:, C0:], ref[:, C0:]) < 1e-4
    assert U.relerr(got[:, :C0], ref[:, :C0]) < 1e-4


DGRAD_CASES = [(1, 16, 32, 8, 16, 16), (2, 32, 64, 9, 13, 11), (1, 1, 16, 8, 16, 16), (1, 96, 32, 4, 8, 8), (1, 3, 8, 5, 9, 7),
               # <= 16 output channels of the data gradient, aligned dims: the paired-y variant (several tiles / samples)
               (2, 16, 32, 8, 16, 32), (1, 8, 16, 4, 8, 8), (1, 12, 24, 8, 24, 16)]


@pytest.mark.parametrize("N,Cin,Cout,D,H,W", DGRAD_CASES)
def test_conv3d_dgrad_and_groupnorm_reductions(N, Cin, Cout, D, H, W):
    """data gradient = the same kernel on dz with mode-1 packed weights; its epilogue accumulates
    (sum dg, sum dg*x) — the two reductions GroupNorm backward needs"""
    U, nat, VSrc, _p, _stream = _mods()
    torch.manual_seed(Cin + 7 * Cout)
    x = torch.randn(N, Cin, D, H, W)
    w = torch.randn(Cout, Cin, 3, 3, 3) / (27 * Cin) ** 0.5
    dz = torch.randn(N, Cout, D, H, W)
    xl = x.clone().requires_grad_(True)
    F.conv3d(xl, w, None, padding=1).backward(dz)
    ref = xl.grad
    gst = torch.zeros((N, Cin, 2), dtype=torch.float64, device=U.DEV)
    dg = U.conv3d(VSrc(U.ndhwc(dz)), w, Cin, relu=0, mode=1, gx=VSrc(U.ndhwc(x)), gstats=gst)
    assert U.relerr(U.ncdhw(dg), ref) < TOL
    s_ref = torch.stack([ref.double().sum(dim=(2, 3, 4)), (ref.double() * x.double()).sum(dim=(2, 3, 4))], dim=-1)
    assert U.relerr(gst.cpu(), s_ref) < 1e-5
    dn = U.conv3d_naive(VSrc(U.ndhwc(dz)), w, Cin, flip=1)
    assert U.relerr(U.ncdhw(dn), ref) < TOL


@pytest.mark.parametrize("N,Cin,Cout,D,H,W,res", [(2, 32, 16, 8, 16, 32, False), (1, 16, 8, 4, 8, 8, True), (1, 48, 12, 4, 16, 8, False)])
def test_narrow_output_variants_agree(N, Cin, Cout, D, H, W, res):
    """<= 16 produced channels on aligned dims: the 16-column variant (v_mfma_f32_16x16x4_f32, round 4, default), the paired-y
    variant (key 4 = 2) and the padded 32-column kernel (key 4 = 1) compute the same convolution, statistics and GroupNorm-backward
    sums (fp32 summation orders differ), forward (affine, ReLU, statistics[, residual]) and data gradient (gx sums)"""
    U, nat, VSrc, _p, _stream = _mods()
    torch.manual_seed(3 * Cin + Cout)
    x = torch.randn(N, Cin, D, H, W)
    w = torch.randn(Cout, Cin, 3, 3, 3) / (27 * Cin) ** 0.5
    ab = torch.randn(N, Cin, 2)
    aff = ab.contiguous().to(U.DEV)
    g = x * ab[:, :, 0].view(N, Cin, 1, 1, 1) + ab[:, :, 1].view(N, Cin, 1, 1, 1)
    r = torch.randn(N, Cout, D, H, W) if res else None
    ref = F.conv3d(g, w, None, padding=1) + (r if res else 0.0)
    ref = F.relu(ref)
    # data gradient of a (Cout -> Cin') layer whose INPUT has <= 16 channels: dz (N, Kd, ...) -> dx (N, Cout, ...)
    Kd = 32
    wd = torch.randn(Kd, Cout, 3, 3, 3) / (27 * Cout) ** 0.5
    xx = torch.randn(N, Cout, D, H, W)
    dz = torch.randn(N, Kd, D, H, W)
    xl = xx.clone().requires_grad_(True)
    F.conv3d(xl, wd, None, padding=1).backward(dz)
    dref = xl.grad
    out = {}
    for key in (0, 2, 1):
        nat.call("u3d_set_tuning", 4, key)
        try:
            st = torch.zeros((N, Cout, 2), dtype=torch.float64, device=U.DEV)
            if res:
                y, _ = U.conv3d_ex(VSrc(U.ndhwc(x)), w, Cout, relu=1, affine=aff, out_stats=st, residual=U.ndhwc(r))
            else:
                y = U.conv3d(VSrc(U.ndhwc(x)), w, Cout, relu=1, affine=aff, out_stats=st)
            gst = torch.zeros((N, Cout, 2), dtype=torch.float64, device=U.DEV)
            dg = U.conv3d(VSrc(U.ndhwc(dz)), wd, Cout, relu=0, mode=1, gx=VSrc(U.ndhwc(xx)), gstats=gst)
            torch.cuda.synchronize()
        finally:
            nat.call("u3d_set_tuning", 4, 0)
        out[key] = (U.ncdhw(y), st.cpu(), U.ncdhw(dg), gst.cpu())
        assert U.relerr(out[key][0], ref) < TOL and U.relerr(out[key][2], dref) < TOL, key
        s_ref = torch.stack([ref.double().sum(dim=(2, 3, 4)), (ref.double() ** 2).sum(dim=(2, 3, 4))], dim=-1)
        g_ref = torch.stack([dref.double().sum(dim=(2, 3, 4)), (dref.double() * xx.double()).sum(dim=(2, 3, 4))], dim=-1)
        assert U.relerr(out[key][1], s_ref) < 1e-5 and U.relerr(out[key][3], g_ref) < 1e-5, key
    for key in (2, 1):
        assert U.relerr(out[key][0], out[0][0]) < 1e-5 and U.relerr(out[key][2], out[0][2]) < 1e-5


WGRAD_CASES = [(1, 16, 32, 8, 16, 16, False), (2, 32, 64, 9, 13, 11, True), (1, 1, 16, 8, 16, 16, True),
               (1, 96, 32, 4, 8, 8, True), (1, 3, 8, 5, 9, 7, True), (1, 64, 128, 4, 8, 8, False),
               (1, 32, 32, 16, 32, 32, True),
               # Cin <= 16 with aligned dims: the tap-pairing variant (two taps per MFMA)
               (2, 16, 32, 8, 16, 16, True), (1, 8, 16, 4, 8, 16, True), (1, 12, 20, 4, 8, 8, True), (2, 4, 8, 2, 8, 8, False)]


@pytest.mark.parametrize("N,Cin,Cout,D,H,W,affine", WGRAD_CASES)
def test_conv3d_wgrad(N, Cin, Cout, D, H, W, affine):
    U, nat, VSrc, _p, _stream = _mods()
    torch.manual_seed(Cin * 3 + Cout)
    x = torch.randn(N, Cin, D, H, W)
    dz = torch.randn(N, Cout, D, H, W)
    aff, g = None, x
    if affine:
        ab = torch.randn(N, Cin, 2)
        aff = ab.contiguous().to(U.DEV)
        g = x * ab[:, :, 0].view(N, Cin, 1, 1, 1) + ab[:, :, 1].view(N, Cin, 1, 1, 1)
    wl = torch.zeros(Cout, Cin, 3, 3, 3, requires_grad=True)
    F.conv3d(g, wl, None, padding=1).backward(dz)
    dw = U.wgrad(VSrc(U.ndhwc(x)), U.ndhwc(dz), Cout, affine=aff)
    assert U.relerr(dw.cpu(), wl.grad) < 1e-4


@pytest.mark.parametrize("N,C,G,size", [(2, 16, 8, (6, 10, 9)), (1, 1, 1, (8, 16, 16)), (2, 96, 8, (4, 8, 8)), (1, 384, 8, (2, 4, 4)), (2, 6, 1, (5, 7, 3))])
def test_groupnorm_forward_stats_and_affine(N, C, G, size):
    U, nat, VSrc, _p, _stream = _mods()
    torch.manual_seed(C)
    x = torch.randn(N, C, *size) * 1.7 + 0.6
    gamma, beta = torch.randn(C), torch.randn(C)
    src = VSrc(U.ndhwc(x))
    st = U.chan_stats(src)
    V = size[0] * size[1] * size[2]
    aff, mr = U.gn_finalize(st, C, 1.0, None, 0, 0.0, N, G, V, gamma.to(U.DEV), beta.to(U.DEV))
    ref = F.group_norm(x, G, gamma, beta, 1e-5)
    a = aff.cpu()
    got = x * a[:, :, 0].view(N, C, 1, 1, 1) + a[:, :, 1].view(N, C, 1, 1, 1)
    assert U.relerr(got, ref) < TOL
    xg = x.view(N, G, -1).double()
    assert U.relerr(mr.cpu()[:, :, 0], xg.mean(-1)) < 1e-5
    assert U.relerr(mr.cpu()[:, :, 1], 1.0 / torch.sqrt(xg.var(-1, unbiased=False) + 1e-5)) < 1e-5


@pytest.mark.parametrize("N,C,G,size,relu", [(2, 16, 8, (6, 10, 9), 1), (1, 4, 1, (8, 8, 8), 0), (2, 96, 8, (4, 8, 8), 1), (1, 6, 2, (3, 5, 7), 1)])
def test_groupnorm_backward(N, C, G, size, relu):
    """dx = p*dg + q*x + r with the finalize kernel's coefficients == autograd of F.group_norm (+ReLU of the producer)"""
    U, nat, VSrc, _p, _stream = _mods()
    torch.manual_seed(C + 1)
    pre = torch.randn(N, C, *size)
    x = F.relu(pre) if relu else pre
    gamma, beta = torch.randn(C), torch.randn(C)
    dg = torch.randn(N, C, *size)
    pl = pre.clone().requires_grad_(True)
    gl = gamma.clone().requires_grad_(True)
    bl = beta.clone().requires_grad_(True)
    xin = F.relu(pl) if relu else pl
    F.group_norm(xin, G, gl, bl, 1e-5).backward(dg)
    V = size[0] * size[1] * size[2]
    # forward stats on device
    src = VSrc(U.ndhwc(x))
    st = U.chan_stats(src)
    aff, mr = U.gn_finalize(st, C, 1.0, None, 0, 0.0, N, G, V, gamma.to(U.DEV), beta.to(U.DEV))
    gst = torch.stack([dg.double().sum(dim=(2, 3, 4)), (dg.double() * x.double()).sum(dim=(2, 3, 4))], dim=-1).contiguous().to(U.DEV)
    dgam = torch.empty(C, device=U.DEV)
    dbet = torch.empty(C, device=U.DEV)
    coef = torch.empty((N, 3, C), device=U.DEV)
    gamma_d = gamma.to(U.DEV)
    nat.call("u3d_gn_bwd_finalize", 0, _stream(U.DEV), _p(gst), _p(mr), _p(gamma_d), N, C, G, float(V), _p(dgam), _p(dbet), _p(coef))
    assert U.relerr(dgam.cpu(), gl.grad) < 1e-4
    assert U.relerr(dbet.cpu(), bl.grad) < 1e-4
    xd, dgd = U.ndhwc(x), U.ndhwc(dg)
    out = torch.empty_like(xd)
    nat.call("u3d_gn_bwd_apply", 0, _stream(U.DEV), _p(dgd), C, 0, _p(xd), C, _p(coef), C, V, N, relu, _p(out))
    assert U.relerr(U.ncdhw(out), pl.grad) < 1e-4


@pytest.mark.parametrize("size", [(8, 12, 16), (9, 13, 11)])
def test_groupnorm_backward_concat_split(size):
    """GroupNorm backward over the virtual concat: skip half (plain affine map) + upsampled half (children sum)"""
    U, nat, VSrc, _p, _stream = _mods()
    torch.manual_seed(3)
    N, C0, C1, G = 2, 8, 16, 4
    D, H, W = size
    D1, H1, W1 = D // 2, H // 2, W // 2
    skip = torch.randn(N, C0, D, H, W, requires_grad=True)
    lowpre = torch.randn(N, C1, D1, H1, W1, requires_grad=True)
    low = F.relu(lowpre)
    gamma, beta = torch.randn(C0 + C1), torch.randn(C0 + C1)
    cat = torch.cat((skip, F.interpolate(low, size=size, mode="nearest")), dim=1)
    dg = torch.randn(N, C0 + C1, D, H, W)
    F.group_norm(cat, G, gamma, beta, 1e-5).backward(dg)
    V = D * H * W
    src = VSrc(U.ndhwc(skip.detach()), U.ndhwc(low.detach()))
    st = U.chan_stats(src)
    C = C0 + C1
    aff, mr = U.gn_finalize(st, C, 1.0, None, 0, 0.0, N, G, V, gamma.to(U.DEV), beta.to(U.DEV))
    catd = cat.detach()
    gst = torch.stack([dg.double().sum(dim=(2, 3, 4)), (dg.double() * catd.double()).sum(dim=(2, 3, 4))], dim=-1).contiguous().to(U.DEV)
    dgam, dbet = torch.empty(C, device=U.DEV), torch.empty(C, device=U.DEV)
    coef = torch.empty((N, 3, C), device=U.DEV)
    gamma_d = gamma.to(U.DEV)
    nat.call("u3d_gn_bwd_finalize", 0, _stream(U.DEV), _p(gst), _p(mr), _p(gamma_d), N, C, G, float(V), _p(dgam), _p(dbet), _p(coef))
    dgd = U.ndhwc(dg)
    sg = torch.empty((N, D, H, W, C0), device=U.DEV)
    nat.call("u3d_gn_bwd_apply", 0, _stream(U.DEV), _p(dgd), C, 0, _p(src.t0), C0, _p(coef), C, V, N, 0, _p(sg))
    assert U.relerr(U.ncdhw(sg), skip.grad) < 1e-4
    dzl = torch.empty_like(src.t1)
    lz, ly, lx = src.los
    nat.call("u3d_gn_bwd_apply_up", 0, _stream(U.DEV), _p(dgd), C, C0, _p(src.t1), C1, _p(coef), C, N, D, H, W, D1, H1, W1,
             _p(lz), _p(ly), _p(lx), 1, _p(dzl))
    assert U.relerr(U.ncdhw(dzl), lowpre.grad) < 1e-4


@pytest.mark.parametrize("N,C0,C1,G,Cs", [(2, 8, 16, 4, 8), (1, 32, 64, 8, 32), (2, 96, 0, 8, 32), (2, 128, 256, 8, 128)])
def test_groupnorm_finalize_split_forms_equal_the_plain_ones_bit_for_bit(N, C0, C1, G, Cs):
    """u3d_gn_finalize_split / u3d_gn_bwd_finalize_split (round 6: the compact half tables of a virtual-concat layer written by the
    finalize launches themselves instead of strided-copy / cat / scale launches around them): same `affine`, `mean_rstd`, dgamma, dbeta,
    `coef` as the plain entry points, and the compact tables are exactly the rows / the (1, 8, 8)-scaled rows of those"""
    U, nat, VSrc, _p, _stream = _mods()
    torch.manual_seed(C0 + C1)
    C, V = C0 + C1, 4096.0
    dev = U.DEV
    st0 = (torch.randn(N, C0, 2, dtype=torch.float64, device=dev) * 50).abs_() + torch.tensor([0.0, 4000.0], dtype=torch.float64, device=dev)
    st1 = ((torch.randn(N, max(C1, 1), 2, dtype=torch.float64, device=dev) * 5).abs_() + torch.tensor([0.0, 500.0], dtype=torch.float64, device=dev))
    gamma, beta = torch.randn(C, device=dev), torch.randn(C, device=dev)
    aff, mr = U.gn_finalize(st0, C0, 1.0, st1 if C1 else None, C1, 8.0, N, G, V, gamma, beta)
    aff2, mr2 = torch.empty_like(aff), torch.empty_like(mr)
    lo, hi = torch.empty((N, Cs, 2), device=dev), torch.empty((N, C - Cs, 2), device=dev)
    nat.call("u3d_gn_finalize_split", 0, _stream(dev), _p(st0), C0, 1.0, _p(st1 if C1 else None), C1, 8.0, N, G, V, _p(gamma), _p(beta), 1e-5,
             _p(aff2), _p(mr2), Cs, _p(lo), _p(hi))
    assert torch.equal(aff, aff2) and torch.equal(mr, mr2)
    assert torch.equal(lo, aff[:, :Cs]) and torch.equal(hi, aff[:, Cs:])
    if C1 == 0:
        return
    g0 = torch.randn(N, C0, 2, dtype=torch.float64, device=dev)
    g1 = torch.randn(N, C1, 2, dtype=torch.float64, device=dev)
    gcat = torch.cat((g0, g1), dim=1).contiguous()
    outs = []
    for split in (False, True):
        dgam, dbet, coef = torch.empty(C, device=dev), torch.empty(C, device=dev), torch.empty((N, 3, C), device=dev)
        chi = torch.empty((N, 3, C1), device=dev)
        if split:
            assert nat.get_lib().u3d_gn_bwd_finalize_split_supported(N, C, G) == 1
            nat.call("u3d_gn_bwd_finalize_split", 0, _stream(dev), _p(g0), C0, _p(g1), C1, _p(mr), _p(gamma), N, G, V, _p(dgam), _p(dbet),
                     _p(coef), 8.0, _p(chi))
        else:
            nat.call("u3d_gn_bwd_finalize", 0, _stream(dev), _p(gcat), _p(mr), _p(gamma), N, C, G, V, _p(dgam), _p(dbet), _p(coef))
        outs.append((dgam, dbet, coef, chi))
    for a, b in zip(outs[0][:3], outs[1][:3]):
        assert torch.equal(a, b)
    scale = torch.tensor([1.0, 8.0, 8.0], device=dev).view(1, 3, 1)
    assert torch.equal(outs[1][3], outs[0][2][:, :, C0:] * scale)


@pytest.mark.parametrize("N,Cin,Cout,D,H,W,C0,C1,G", [(1, 32, 64, 8, 16, 16, 32, 0, 8), (2, 16, 32, 5, 9, 7, 16, 0, 4),
                                                      (1, 32, 32, 8, 16, 16, 32, 64, 8), (2, 64, 32, 4, 8, 8, 64, 128, 8)])
def test_wgrad_job_equals_wgrad_then_groupnorm_backward_finalize_bit_for_bit(N, Cin, Cout, D, H, W, C0, C1, G):
    """u3d_conv3d_wgrad_job (round 6): the weight gradient and the GroupNorm-backward reduction of the layer's input in the reduce
    launch's extra block are those of u3d_conv3d_wgrad_strided + u3d_gn_bwd_finalize[_split], bit for bit; C1 > 0: the source holds
    the C0 skip channels of a (C0 + C1)-channel weight and the sums arrive as two tables"""
    U, nat, VSrc, _p, _stream = _mods()
    torch.manual_seed(Cin + Cout + C1)
    dev = U.DEV
    assert Cin == C0
    C, V = C0 + C1, float(D * H * W)
    x = U.ndhwc(torch.randn(N, Cin, D, H, W))
    dz = U.ndhwc(torch.randn(N, Cout, D, H, W))
    aff = torch.randn(N, Cin, 2, device=dev)
    gamma = torch.randn(C, device=dev)
    mr = torch.rand(N, G, 2, device=dev) + 0.5
    g0 = torch.randn(N, C0, 2, dtype=torch.float64, device=dev) * 100
    g1 = torch.randn(N, max(C1, 1), 2, dtype=torch.float64, device=dev) * 100
    g1r = torch.stack((g1 * 0.5, g1 * 0.25, g1 * 0.25)).contiguous()  # (exact binary fractions: the rows sum to g1 bit for bit)
    lib = nat.get_lib()
    assert lib.u3d_conv3d_wgrad_job_supported(N, C, G) == 1
    n = lib.u3d_wgrad_workspace_floats(N, D, H, W, Cin, Cout)
    ws = torch.empty(n, device=dev)
    s = VSrc(x).struct(aff)
    outs = []
    for fused in (False, True):
        dw = torch.full((Cout, C, 27), 7.0, device=dev)
        dgam, dbet, coef = torch.empty(C, device=dev), torch.empty(C, device=dev), torch.empty((N, 3, C), device=dev)
        chi = torch.zeros((N, 3, max(C1, 1)), device=dev)
        if fused:
            job = nat.U3DGnBwdJob()
            job.gstats_lo, job.gstats_hi, job.C0, job.C1 = _p(g0), _p(g1r) if C1 else None, C0, C1
            job.reps_hi = 3 if C1 else 1  # (the upper table as three replica rows that add up to g1)
            job.hi_scale, job.coef_hi = (8.0, _p(chi)) if C1 else (1.0, None)
            job.mean_rstd, job.gamma, job.dgamma, job.dbeta, job.coef = _p(mr), _p(gamma), _p(dgam), _p(dbet), _p(coef)
            job.count, job.N, job.G = V, N, G
            nat.call("u3d_conv3d_wgrad_job", 0, _stream(dev), ctypes.byref(s), _p(dz), _p(dw), C, N, D, H, W, Cout, _p(ws), n,
                     ctypes.byref(job))
        else:
            nat.call("u3d_conv3d_wgrad_strided", 0, _stream(dev), ctypes.byref(s), _p(dz), _p(dw), C, N, D, H, W, Cout, _p(ws), n)
            if C1:
                nat.call("u3d_gn_bwd_finalize_split", 0, _stream(dev), _p(g0), C0, _p(g1), C1, _p(mr), _p(gamma), N, G, V, _p(dgam),
                         _p(dbet), _p(coef), 8.0, _p(chi))
            else:
                nat.call("u3d_gn_bwd_finalize", 0, _stream(dev), _p(g0), _p(mr), _p(gamma), N, C, G, V, _p(dgam), _p(dbet), _p(coef))
        outs.append((dw, dgam, dbet, coef, chi))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert bool((outs[1][0].view(Cout, C, 27)[:, C0:] == 7.0).all())  # the slice of the other channels is left alone
    # job == NULL with stride 0 is the plain weight gradient
    dw0 = torch.empty((Cout, Cin, 27), device=dev)
    nat.call("u3d_conv3d_wgrad_job", 0, _stream(dev), ctypes.byref(s), _p(dz), _p(dw0), 0, N, D, H, W, Cout, _p(ws), n, None)
    assert torch.equal(dw0, outs[0][0].view(Cout, C, 27)[:, :C0])
    # a reduction that does not fit the block's LDS is refused, not truncated
    assert lib.u3d_conv3d_wgrad_job_supported(4, 1024, 8) == 0


@pytest.mark.parametrize("N,C,K,D,H,W,G,b16", [(1, 64, 64, 8, 16, 16, 8, 0), (1, 64, 64, 8, 16, 16, 8, 1), (2, 32, 96, 5, 9, 7, 4, 0),
                                               (1, 128, 128, 4, 8, 8, 8, 1), (1, 1024, 1024, 4, 4, 4, 8, 1), (1, 64, 128, 16, 32, 32, 8, 1),
                                               (1, 256, 256, 8, 8, 8, 8, 1), (1, 256, 256, 4, 8, 8, 8, 0)])
def test_bf16_wgrad_job_equals_wgrad_then_groupnorm_backward_finalize_bit_for_bit(N, C, K, D, H, W, G, b16):
    """u3d_conv3d_wgrad_bf16_job / _b16_job (round 6): dw and the GroupNorm-backward tables of the layer's input from the extra block of
    the split reduction are those of the plain entry point + u3d_gn_bwd_finalize, bit for bit — for the block-per-row and the flat reduce
    kernel; a shape with ONE split (the main kernel writes dw, no reduce launch) says so and refuses a job"""
    U, nat, VSrc, _p, _stream = _mods()
    torch.manual_seed(C + K + b16)
    dev = U.DEV
    V = float(D * H * W)
    dt = torch.bfloat16 if b16 else torch.float32
    x = U.ndhwc(torch.randn(N, C, D, H, W)).to(dt)
    dz = U.ndhwc(torch.randn(N, K, D, H, W)).to(dt)
    aff = torch.randn(N, C, 2, device=dev)
    gamma = torch.randn(C, device=dev)
    mr = torch.rand(N, G, 2, device=dev) + 0.5
    g0 = torch.randn(N, C, 2, dtype=torch.float64, device=dev) * 100
    lib = nat.get_lib()
    n = lib.u3d_wgrad_bf16_workspace_floats(N, D, H, W, C, K)
    ws = torch.empty(n, device=dev)
    sfx = "_b16" if b16 else ""
    dw_plain = torch.empty((K, C, 27), device=dev)
    nat.call("u3d_conv3d_wgrad_bf16" + sfx, 0, _stream(dev), _p(x), _p(aff), _p(dz), _p(dw_plain), N, D, H, W, C, K, _p(ws), n)
    dgam0, dbet0, coef0 = torch.empty(C, device=dev), torch.empty(C, device=dev), torch.empty((N, 3, C), device=dev)
    nat.call("u3d_gn_bwd_finalize", 0, _stream(dev), _p(g0), _p(mr), _p(gamma), N, C, G, V, _p(dgam0), _p(dbet0), _p(coef0))
    dw = torch.full((K, C, 27), 7.0, device=dev)
    dgam, dbet, coef = torch.full((C,), 7.0, device=dev), torch.full((C,), 7.0, device=dev), torch.full((N, 3, C), 7.0, device=dev)
    job = nat.U3DGnBwdJob()
    job.gstats_lo, job.gstats_hi, job.C0, job.C1, job.hi_scale, job.coef_hi = _p(g0), None, C, 0, 1.0, None
    job.mean_rstd, job.gamma, job.dgamma, job.dbeta, job.coef = _p(mr), _p(gamma), _p(dgam), _p(dbet), _p(coef)
    job.count, job.N, job.G = V, N, G
    ok = lib.u3d_conv3d_wgrad_bf16_job_supported(N, D, H, W, C, K, b16, _p(dw), N, C, G)
    one_split = b16 == 1 and (C, K, D) in ((1024, 1024, 4), (128, 128, 4))  # one tile per (32 x 64)-channel pair
    if one_split:
        assert ok == 0  # conv3d_wgrad_b16v2_kernel writes dw itself: no reduce launch
        with pytest.raises(nat.U3DError):
            nat.call("u3d_conv3d_wgrad_bf16" + sfx + "_job", 0, _stream(dev), _p(x), _p(aff), _p(dz), _p(dw), N, D, H, W, C, K, _p(ws), n,
                     ctypes.byref(job))
        nat.call("u3d_conv3d_wgrad_bf16" + sfx + "_job", 0, _stream(dev), _p(x), _p(aff), _p(dz), _p(dw), N, D, H, W, C, K, _p(ws), n, None)
        assert torch.equal(dw, dw_plain)
        return
    assert ok == 1
    nat.call("u3d_conv3d_wgrad_bf16" + sfx + "_job", 0, _stream(dev), _p(x), _p(aff), _p(dz), _p(dw), N, D, H, W, C, K, _p(ws), n,
             ctypes.byref(job))
    assert torch.equal(dw, dw_plain) and torch.equal(dgam, dgam0) and torch.equal(dbet, dbet0) and torch.equal(coef, coef0)


@pytest.mark.parametrize("N,Cin,Cout,D,H,W,reps", [(2, 32, 32, 8, 16, 16, 8), (1, 16, 64, 8, 16, 32, 4), (2, 32, 64, 9, 13, 11, 8)])
def test_statistics_replica_rows_sum_to_the_plain_tables(N, Cin, Cout, D, H, W, reps):
    """u3d_conv3d_ex_reps (round 6): the persistent kernels' blocks spread the per-sample flush of their f64 sums over `reps` rows of the
    table; the rows sum to what u3d_conv3d_ex writes, the outputs are bit-identical, u3d_gn_finalize_reps / a weight-gradient job with
    reps_lo read the rows and give what the plain entry points give on the folded table"""
    U, nat, VSrc, _p, _stream = _mods()
    torch.manual_seed(N * 100 + Cin + Cout + D)
    dev = U.DEV
    x = U.ndhwc(torch.randn(N, Cin, D, H, W))
    w = torch.randn(Cout, Cin, 3, 3, 3) / (27 * Cin) ** 0.5
    aff = torch.randn(N, Cin, 2, device=dev)
    wp = U.pack(w.to(dev), 0)
    s = VSrc(x).struct(aff)
    y0, y1 = torch.empty((N, D, H, W, Cout), device=dev), torch.empty((N, D, H, W, Cout), device=dev)
    st = torch.zeros((N, Cout, 2), dtype=torch.float64, device=dev)
    str_ = torch.zeros((reps, N, Cout, 2), dtype=torch.float64, device=dev)
    nat.call("u3d_conv3d_ex", 0, _stream(dev), ctypes.byref(s), _p(wp), _p(y0), N, D, H, W, Cout, 1, _p(st), None, None, None, None, 0)
    nat.call("u3d_conv3d_ex_reps", 0, _stream(dev), ctypes.byref(s), _p(wp), _p(y1), N, D, H, W, Cout, 1, _p(str_), None, None, None, None, 0, reps)
    assert torch.equal(y0, y1)
    assert torch.allclose(str_.sum(0), st, rtol=1e-12, atol=1e-9)
    persistent = nat.get_lib().u3d_conv3d_variant(N, D, H, W, Cin, Cout, 0, 0) in (1, 2)
    if persistent:
        assert int((str_.abs().sum(dim=(1, 2, 3)) > 0).sum()) > 1, "the persistent kernel must use more than one row"
    # forward finalize on the rows == on the folded table
    G = 8
    gamma, beta = torch.randn(Cout, device=dev), torch.randn(Cout, device=dev)
    V = float(D * H * W)
    folded = str_.sum(0).contiguous()
    a0, m0 = U.gn_finalize(folded, Cout, 1.0, None, 0, 0.0, N, G, V, gamma, beta)
    a1, m1 = torch.empty_like(a0), torch.empty_like(m0)
    nat.call("u3d_gn_finalize_reps", 0, _stream(dev), _p(str_), Cout, 1.0, reps, None, 0, 0.0, 1, N, G, V, _p(gamma), _p(beta), 1e-5,
             _p(a1), _p(m1), 0, None, None)
    assert torch.allclose(a0, a1, rtol=1e-6, atol=1e-6) and torch.allclose(m0, m1, rtol=1e-6, atol=1e-7)
    # data gradient role: GroupNorm-backward sums in rows, consumed by the weight-gradient job
    dz = U.ndhwc(torch.randn(N, Cout, D, H, W))
    wpd = U.pack(w.to(dev), 1)
    s_dz, s_x = VSrc(dz).struct(), VSrc(x).struct()
    dg0, dg1 = torch.empty_like(x), torch.empty_like(x)
    g_plain = torch.zeros((N, Cin, 2), dtype=torch.float64, device=dev)
    g_rows = torch.zeros((reps, N, Cin, 2), dtype=torch.float64, device=dev)
    nat.call("u3d_conv3d_ex", 0, _stream(dev), ctypes.byref(s_dz), _p(wpd), _p(dg0), N, D, H, W, Cin, 0, None, ctypes.byref(s_x), _p(g_plain),
             None, None, 0)
    nat.call("u3d_conv3d_ex_reps", 0, _stream(dev), ctypes.byref(s_dz), _p(wpd), _p(dg1), N, D, H, W, Cin, 0, None, ctypes.byref(s_x),
             _p(g_rows), None, None, 0, reps)
    assert torch.equal(dg0, dg1) and torch.allclose(g_rows.sum(0), g_plain, rtol=1e-12, atol=1e-9)
    Gi = 4
    gam_i, mr = torch.randn(Cin, device=dev), torch.rand(N, Gi, 2, device=dev) + 0.5
    lib = nat.get_lib()
    n = lib.u3d_wgrad_workspace_floats(N, D, H, W, Cin, Cout)
    ws = torch.empty(n, device=dev)
    outs = []
    for rows in (False, True):
        dw = torch.empty((Cout, Cin, 27), device=dev)
        dgam, dbet, coef = torch.empty(Cin, device=dev), torch.empty(Cin, device=dev), torch.empty((N, 3, Cin), device=dev)
        table = g_rows if rows else g_rows.sum(0).contiguous()
        job = nat.U3DGnBwdJob()
        job.gstats_lo, job.gstats_hi, job.C0, job.C1, job.hi_scale, job.coef_hi, job.reps_lo = _p(table), None, Cin, 0, 1.0, None, reps if rows else 1
        job.mean_rstd, job.gamma, job.dgamma, job.dbeta, job.coef = _p(mr), _p(gam_i), _p(dgam), _p(dbet), _p(coef)
        job.count, job.N, job.G = V, N, Gi
        nat.call("u3d_conv3d_wgrad_job", 0, _stream(dev), ctypes.byref(s), _p(dz), _p(dw), 0, N, D, H, W, Cout, _p(ws), n, ctypes.byref(job))
        outs.append((dw, dgam, dbet, coef))
    assert torch.equal(outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1:], outs[1][1:]):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-6)


def test_replica_rows_of_the_other_statistics_producers():
    """u3d_chan_stats_reps, u3d_conv3d_small_cin_fwd_reps, u3d_conv1x1_head_bwd_reps + u3d_cvt_f64_f32_sum (round 6): the rows sum to
    what the plain entry points accumulate, everything else they write is bit-identical"""
    U, nat, VSrc, _p, _stream = _mods()
    torch.manual_seed(5)
    dev, R = U.DEV, 8
    N, C, D, H, W = 2, 16, 16, 32, 32
    x = U.ndhwc(torch.randn(N, C, D, H, W))
    s = VSrc(x).struct()
    st = U.chan_stats(VSrc(x))
    rows = torch.zeros((R, N, C, 2), dtype=torch.float64, device=dev)
    nat.call("u3d_chan_stats_reps", 0, _stream(dev), ctypes.byref(s), N, D, H, W, _p(rows), R)
    assert torch.allclose(rows.sum(0), st, rtol=1e-12, atol=1e-9) and int((rows.abs().sum(dim=(1, 2, 3)) > 0).sum()) > 1
    # first-layer forward
    Cin, Cout = 1, 16
    x1 = U.ndhwc(torch.randn(N, Cin, D, H, W))
    w = torch.randn(Cout, Cin, 3, 3, 3, device=dev)
    aff = torch.randn(N, Cin, 2, device=dev)
    y0, y1 = torch.empty((N, D, H, W, Cout), device=dev), torch.empty((N, D, H, W, Cout), device=dev)
    s0 = torch.zeros((N, Cout, 2), dtype=torch.float64, device=dev)
    s1 = torch.zeros((R, N, Cout, 2), dtype=torch.float64, device=dev)
    nat.call("u3d_conv3d_small_cin_fwd", 0, _stream(dev), _p(x1), _p(aff), _p(w), _p(y0), N, D, H, W, Cin, Cout, 1, _p(s0))
    nat.call("u3d_conv3d_small_cin_fwd_reps", 0, _stream(dev), _p(x1), _p(aff), _p(w), _p(y1), N, D, H, W, Cin, Cout, 1, _p(s1), R)
    assert torch.equal(y0, y1) and torch.allclose(s1.sum(0), s0, rtol=1e-12, atol=1e-9)
    # head backward
    Cf, Co, V = 32, 1, D * H * W
    hx = torch.randn(N, V, Cf, device=dev)
    hw = torch.randn(Co, Cf, device=dev)
    dl = torch.randn(N, Co, V, device=dev)
    d0, d1 = torch.empty_like(hx), torch.empty_like(hx)
    a0 = torch.zeros(Co * Cf + Co, dtype=torch.float64, device=dev)
    a1 = torch.zeros((R, Co * Cf + Co), dtype=torch.float64, device=dev)
    nat.call("u3d_conv1x1_head_bwd", 0, _stream(dev), _p(dl), _p(hx), _p(hw), N, V, Cf, Co, 1, _p(d0), _p(a0))
    nat.call("u3d_conv1x1_head_bwd_reps", 0, _stream(dev), _p(dl), _p(hx), _p(hw), N, V, Cf, Co, 1, _p(d1), _p(a1), R)
    assert torch.equal(d0, d1) and torch.allclose(a1.sum(0), a0, rtol=1e-12, atol=1e-9)
    g0, g1 = torch.empty(Co * Cf + Co, device=dev), torch.empty(Co * Cf + Co, device=dev)
    nat.call("u3d_cvt_f64_f32", 0, _stream(dev), _p(a0), _p(g0), Co * Cf + Co)
    nat.call("u3d_cvt_f64_f32_sum", 0, _stream(dev), _p(a1), _p(g1), Co * Cf + Co, R)
    assert torch.allclose(g0, g1, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("N,C,size", [(2, 8, (8, 12, 16)), (1, 5, (9, 13, 11)), (1, 32, (4, 6, 2)), (2, 64, (8, 16, 16))])
def test_maxpool_forward_backward_merge(N, C, size):
    U, nat, VSrc, _p, _stream = _mods()
    torch.manual_seed(C)
    D, H, W = size
    pre = torch.randn(N, C, D, H, W, requires_grad=True)
    e = F.relu(pre)  # post-ReLU: exact-zero ties exercise the first-max rule
    pooled = F.max_pool3d(e, 2)
    dpool = torch.randn_like(pooled)
    skipg = torch.randn(N, C, D, H, W)
    (pooled * dpool).sum().backward(retain_graph=True)
    (e * skipg).sum().backward()
    ed = U.ndhwc(e.detach())
    D2, H2, W2 = D // 2, H // 2, W // 2
    out = torch.empty((N, D2, H2, W2, C), device=U.DEV)
    am = torch.empty(out.shape, dtype=torch.uint8, device=U.DEV)
    st = torch.zeros((N, C, 2), dtype=torch.float64, device=U.DEV)
    nat.call("u3d_maxpool2_fwd", 0, _stream(U.DEV), _p(ed), N, D, H, W, C, _p(out), _p(am), _p(st))
    assert torch.equal(U.ncdhw(out), pooled.detach())
    pd = pooled.detach().double()
    assert U.relerr(st.cpu(), torch.stack([pd.sum(dim=(2, 3, 4)), (pd ** 2).sum(dim=(2, 3, 4))], dim=-1)) < 1e-6
    res = torch.empty_like(ed)
    dpool_d, skipg_d = U.ndhwc(dpool), U.ndhwc(skipg)  # keep device tensors alive across the async launch
    nat.call("u3d_maxpool2_bwd_merge", 0, _stream(U.DEV), _p(dpool_d), None, _p(am), None, _p(skipg_d), _p(ed),
             N, D, H, W, C, 1, _p(res))
    assert U.relerr(U.ncdhw(res), pre.grad) < 1e-6


@pytest.mark.parametrize("N,Cin,Cout,act", [(2, 32, 1, 1), (1, 8, 3, 2), (1, 16, 2, 0),
                                            # wide heads (round 5): > 16 outputs run in tiles of 16, softmax as a second pass
                                            (2, 32, 24, 2), (1, 16, 17, 1), (1, 64, 40, 0), (1, 6, 33, 2),
                                            # more outputs than inputs on the narrow kernels (the reduction rows overlapped until round 5)
                                            (1, 6, 12, 1), (2, 5, 16, 2)])
def test_head_forward_backward(N, Cin, Cout, act):
    U, nat, VSrc, _p, _stream = _mods()
    torch.manual_seed(Cin + Cout)
    D, H, W = 5, 6, 7
    V = D * H * W
    pre = torch.randn(N, Cin, D, H, W, requires_grad=True)
    x = F.relu(pre)
    w = torch.randn(Cout, Cin, 1, 1, 1, requires_grad=True)
    b = torch.randn(Cout, requires_grad=True)
    logits = F.conv3d(x, w, b)
    probs = torch.sigmoid(logits) if act == 1 else (torch.softmax(logits, 1) if act == 2 else logits)
    dl = torch.randn_like(logits)
    logits.backward(dl)
    xd = U.ndhwc(x.detach())
    wd, bd, dld = w.detach().to(U.DEV), b.detach().to(U.DEV), dl.to(U.DEV)  # keep alive across the async launches
    lg = torch.empty((N, Cout, D, H, W), device=U.DEV)
    pr = torch.empty_like(lg)
    nat.call("u3d_conv1x1_head_fwd", 0, _stream(U.DEV), _p(xd), _p(wd), _p(bd), N, V, Cin,
             Cout, act, _p(lg), _p(pr) if act else None)
    assert U.relerr(lg.cpu(), logits.detach()) < TOL
    if act:
        assert U.relerr(pr.cpu(), probs.detach()) < TOL
    dx = torch.empty_like(xd)
    acc = torch.zeros(Cout * Cin + Cout, dtype=torch.float64, device=U.DEV)
    nat.call("u3d_conv1x1_head_bwd", 0, _stream(U.DEV), _p(dld), _p(xd), _p(wd), N, V, Cin, Cout, 1,
             _p(dx), _p(acc))
    assert U.relerr(U.ncdhw(dx), pre.grad) < TOL
    f32 = torch.empty(Cout * Cin + Cout, device=U.DEV)
    nat.call("u3d_cvt_f64_f32", 0, _stream(U.DEV), _p(acc), _p(f32), acc.numel())
    assert U.relerr(f32[: Cout * Cin].cpu().view(Cout, Cin), w.grad.view(Cout, Cin)) < 1e-5
    assert U.relerr(f32[Cout * Cin:].cpu(), b.grad) < 1e-5


def test_layout_transposes_roundtrip():
    U, nat, VSrc, _p, _stream = _mods()
    x = torch.randn(2, 5, 7, 9, 11)
    xd = x.to(U.DEV)
    V = 7 * 9 * 11
    y = torch.empty((2, 7, 9, 11, 5), device=U.DEV)
    nat.call("u3d_ncdhw_to_ndhwc", 0, _stream(U.DEV), _p(xd), _p(y), 2, 5, V)
    assert torch.equal(y.cpu(), x.permute(0, 2, 3, 4, 1).contiguous())
    z = torch.empty_like(xd)
    nat.call("u3d_ndhwc_to_ncdhw", 0, _stream(U.DEV), _p(y), _p(z), 2, 5, V)
    assert torch.equal(z.cpu(), x)


def test_error_convention():
    """bad arguments come back as a negative code + message, never an abort (include/u3d.h conventions)"""
    U, nat, VSrc, _p, _stream = _mods()
    lib = nat.get_lib()
    rc = lib.u3d_pack_weights(0, None, None, 4, 4, 0, None)
    assert rc == -1 and b"u3d_pack_weights" in lib.u3d_last_error()
    x = torch.zeros((1, 2, 2, 2, 4), device=U.DEV)
    with pytest.raises(nat.U3DError):
        nat.call("u3d_gn_finalize", 0, _stream(U.DEV), _p(x), 5, 1.0, None, 0, 0.0, 1, 2, 8.0, _p(x), _p(x), 1e-5, _p(x), _p(x))


@pytest.mark.parametrize("N,Cin,Cout,D,H,W", [(2, 1, 16, 8, 16, 16), (1, 3, 8, 5, 9, 7), (1, 2, 32, 9, 13, 11), (1, 4, 12, 4, 8, 8), (1, 1, 6, 6, 7, 5),
                                              (1, 3, 16, 8, 8, 24)])
def test_small_cin_first_layer_kernels(N, Cin, Cout, D, H, W):
    """dedicated first-layer kernels: forward == conv3d(GN-affine(x)); backward yields dw and the GroupNorm
    reductions (sum dg, sum dg*x) WITHOUT computing dg — compare with autograd of the conv"""
    U, nat, VSrc, _p, _stream = _mods()
    torch.manual_seed(Cin * 17 + Cout)
    x = torch.randn(N, Cin, D, H, W)
    w = torch.randn(Cout, Cin, 3, 3, 3) / (27 * Cin) ** 0.5
    ab = torch.randn(N, Cin, 2)
    xl = x.clone().requires_grad_(True)
    wl = w.clone().requires_grad_(True)
    g = xl * ab[:, :, 0].view(N, Cin, 1, 1, 1) + ab[:, :, 1].view(N, Cin, 1, 1, 1)
    g.retain_grad()
    z = F.conv3d(g, wl, None, padding=1)
    dz = torch.randn_like(z)
    z.backward(dz)
    xd, abd, wd, dzd = U.ndhwc(x), ab.contiguous().to(U.DEV), w.contiguous().to(U.DEV), U.ndhwc(dz)
    y = torch.empty((N, D, H, W, Cout), device=U.DEV)
    yst = torch.zeros((N, Cout, 2), dtype=torch.float64, device=U.DEV)
    nat.call("u3d_conv3d_small_cin_fwd", 0, _stream(U.DEV), _p(xd), _p(abd), _p(wd), _p(y), N, D, H, W, Cin, Cout, 1, _p(yst))
    assert U.relerr(U.ncdhw(y), F.relu(z.detach())) < TOL
    yr = F.relu(z.detach()).double()
    assert U.relerr(yst.cpu(), torch.stack([yr.sum(dim=(2, 3, 4)), (yr * yr).sum(dim=(2, 3, 4))], dim=-1)) < 1e-5
    # <= 16 output channels in whole quads run on the matrix pipe (round 4); the direct kernel (u3d_set_tuning key 13 = 1) must agree
    # with it like two fp32 summation orders, and both are run-to-run reproducible
    y2, yst2 = torch.empty_like(y), torch.zeros_like(yst)
    nat.call("u3d_set_tuning", 13, 1)
    try:
        nat.call("u3d_conv3d_small_cin_fwd", 0, _stream(U.DEV), _p(xd), _p(abd), _p(wd), _p(y2), N, D, H, W, Cin, Cout, 1, _p(yst2))
    finally:
        nat.call("u3d_set_tuning", 13, 0)
    assert U.relerr(y2, y) < 1e-5 and U.relerr(yst2, yst) < 1e-6
    y3 = torch.empty_like(y)
    nat.call("u3d_conv3d_small_cin_fwd", 0, _stream(U.DEV), _p(xd), _p(abd), _p(wd), _p(y3), N, D, H, W, Cin, Cout, 1, None)
    assert torch.equal(y3, y)
    n = nat.get_lib().u3d_small_cin_bwd_workspace_floats(N, D, H, W, Cin, Cout)
    ws = torch.empty(n, device=U.DEV)
    dw = torch.empty((Cout, Cin, 3, 3, 3), device=U.DEV)
    gst = torch.zeros((N, Cin, 2), dtype=torch.float64, device=U.DEV)
    nat.call("u3d_conv3d_small_cin_bwd", 0, _stream(U.DEV), _p(xd), _p(abd), _p(dzd), _p(wd), _p(dw), _p(gst), N, D, H, W, Cin,
             Cout, _p(ws), n)
    assert U.relerr(dw.cpu(), wl.grad) < 1e-4
    dg = g.grad.double()
    s_ref = torch.stack([dg.sum(dim=(2, 3, 4)), (dg * x.double()).sum(dim=(2, 3, 4))], dim=-1)
    assert U.relerr(gst.cpu(), s_ref) < 1e-4
    # round 6: 16 output channels on whole tiles stage the dz tile through LDS (16-byte loads at constant offsets); the first form
    # (u3d_set_tuning key 20 = -1: 4-byte operand loads) runs the same MFMA sequence per tile over another block count: dw agrees like two
    # orders of summing the per-block partials
    dw2, gst2 = torch.empty_like(dw), torch.zeros_like(gst)
    nat.call("u3d_set_tuning", 20, -1)
    try:
        nat.call("u3d_conv3d_small_cin_bwd", 0, _stream(U.DEV), _p(xd), _p(abd), _p(dzd), _p(wd), _p(dw2), _p(gst2), N, D, H, W, Cin,
                 Cout, _p(ws), n)
    finally:
        nat.call("u3d_set_tuning", 20, 0)
    assert U.relerr(dw2, dw) < 2e-6 and U.relerr(gst2, gst) < 1e-9


@pytest.mark.parametrize("Cout,Cin,C0", [(32, 16, 0), (16, 32, 0), (32, 96, 32), (36, 20, 8), (256, 128, 0), (64, 192, 64), (8, 12, 4)])
def test_cell_packer_writes_the_element_wise_packers_images_bit_for_bit(Cout, Cin, C0):
    """u3d_pack_weights_batch_cells (round 6: contiguous runs of the reference layout -> LDS -> 16-byte fragment stores) against
    u3d_pack_weights_batch (one strided 4-byte gather per element) for every image kind the executor packs: forward / data-gradient
    images of a whole weight (incl. the paired-y and 16-column images of <= 16 produced channels and cells that overhang the channel
    counts), of the channel slice [0, C0) of a decoder's first conv, and the pre-summed sub-pixel images of its slice [C0, Cin)"""
    U, nat, VSrc, _p, _stream = _mods()
    lib = nat.get_lib()
    torch.manual_seed(Cout * 1000 + Cin)
    w = torch.randn(Cout, Cin, 3, 3, 3, device=U.DEV)
    jobs = [(w.data_ptr(), Cin, 0, 0, lib.u3d_packed_weight_floats(Cin, Cout, 0)), (w.data_ptr(), Cin, 1, 0, lib.u3d_packed_weight_floats(Cin, Cout, 1))]
    if C0:
        C1 = Cin - C0
        up = w.data_ptr() + C0 * 27 * 4
        jobs += [(w.data_ptr(), C0, 0, Cin, lib.u3d_packed_weight_floats(C0, Cout, 0)), (w.data_ptr(), C0, 1, Cin, lib.u3d_packed_weight_floats(C0, Cout, 1)),
                 (up, C1, 2, Cin, lib.u3d_subpixel_packed_floats(C1, Cout)), (up, C1, 3, Cin, lib.u3d_subpixel_dgrad_packed_floats(Cout, C1)),
                 (up, C1, 0, Cin, lib.u3d_packed_weight_floats(C1, Cout, 0)), (up, C1, 1, Cin, lib.u3d_packed_weight_floats(C1, Cout, 1))]
    outs = {}
    for cells in (False, True):
        descs = (nat.U3DPackDesc * len(jobs))()
        bufs, first = [], 0
        for i, (wptr, ci, mode, cstride, n) in enumerate(jobs):
            buf = torch.full((n,), float("nan"), device=U.DEV)
            bufs.append(buf)
            descs[i].w, descs[i].packed, descs[i].first = wptr, buf.data_ptr(), first
            descs[i].Cout, descs[i].Cin, descs[i].mode, descs[i].cin_stride = Cout, ci, mode, cstride
            nb = lib.u3d_pack_weights_cells_blocks(wptr, ci, Cout, mode, cstride)
            assert nb > 0
            first += nb if cells else n
        table = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(U.DEV)
        nat.call("u3d_pack_weights_batch_cells" if cells else "u3d_pack_weights_batch", 0, _stream(U.DEV), _p(table), len(jobs), first)
        torch.cuda.synchronize()
        outs[cells] = bufs
    for i, (a, b) in enumerate(zip(outs[False], outs[True])):
        assert not torch.isnan(b).any(), (i, jobs[i][1:4])
        assert torch.equal(a, b), (i, jobs[i][1:4], int((a != b).sum()))
    # not eligible: unaligned base / channel counts that are not multiples of 4 -> 0 blocks (the executor keeps those on the old kernel)
    assert lib.u3d_pack_weights_cells_blocks(w.data_ptr() + 4, Cin, Cout, 0, 0) == 0
    assert lib.u3d_pack_weights_cells_blocks(w.data_ptr(), 6, Cout, 0, 0) == 0 and lib.u3d_pack_weights_cells_blocks(w.data_ptr(), Cin, Cout, 0, 6) == 0


@pytest.mark.parametrize("N,Cin,Cout,act,V", [(2, 32, 1, 1, 64 * 128), (1, 64, 2, 2, 4099), (1, 32, 2, 0, 777), (3, 64, 1, 1, 65)])
def test_head_forward_rows_kernel_equals_the_vectorised_kernel_bit_for_bit(N, Cin, Cout, act, V):
    """round 6: 32 / 64-channel heads with 1 or 2 outputs run one voxel per lane through LDS rows (u3d_set_tuning key 21 = 1 selects the
    vectorised kernel); same products, same summation order -> identical logits and probabilities, ragged last rounds included"""
    U, nat, VSrc, _p, _stream = _mods()
    torch.manual_seed(Cin + Cout)
    x = torch.randn(N, V, Cin, device=U.DEV)
    w, b = torch.randn(Cout, Cin, device=U.DEV), torch.randn(Cout, device=U.DEV)
    outs = []
    for key in (1, 0):
        nat.call("u3d_set_tuning", 21, key)
        try:
            lg = torch.full((N, Cout, V), float("nan"), device=U.DEV)
            pr = torch.full((N, Cout, V), float("nan"), device=U.DEV) if act else None
            nat.call("u3d_conv1x1_head_fwd", 0, _stream(U.DEV), _p(x), _p(w), _p(b), N, V, Cin, Cout, act, _p(lg), _p(pr))
            torch.cuda.synchronize()
        finally:
            nat.call("u3d_set_tuning", 21, 0)
        outs.append((lg, pr))
    assert torch.equal(outs[0][0], outs[1][0])
    if act:
        assert torch.equal(outs[0][1], outs[1][1])
    ref = torch.einsum("nvc,oc->nov", x.double(), w.double()) + b.double().view(1, -1, 1)
    assert U.relerr(outs[1][0], ref) < 1e-5
