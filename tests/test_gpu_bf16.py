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


@pytest.mark.parametrize("shape,C,K", [((1, 5, 10, 10), 256, 128), ((2, 4, 9, 7), 128, 64)])
def test_conv3d_bf16_split_k_at_the_bottom_of_the_u(shape, C, K):
    """few tiles, many channels: the channel reduction is split over blocks (workspace) and reduced in a fixed order by the
    kernel that owns the epilogue — same results as the unsplit launch, run-to-run identical, residual / ReLU / statistics"""
    N, D, H, W = shape
    lib = nat.get_lib()
    need = lib.u3d_conv3d_bf16_workspace_floats(N, D, H, W, C, K)
    assert need > 0
    torch.manual_seed(5)
    x = torch.randn(N, C, D, H, W)
    w = torch.randn(K, C, 3, 3, 3) / (27 * C) ** 0.5
    res = torch.randn(N, K, D, H, W)
    xd, resd, wp = U.ndhwc(x), U.ndhwc(res), pack_bf16(w, 0)
    ws = torch.empty(need, dtype=torch.float32, device=U.DEV)
    outs = []
    for use_ws in (True, True, False):
        y = torch.empty((N, D, H, W, K), dtype=torch.float32, device=U.DEV)
        st = torch.zeros((N, K, 2), dtype=torch.float64, device=U.DEV)
        nat.call("u3d_conv3d_bf16_ex", 0, _stream(U.DEV), _p(xd), None, _p(wp), _p(y), N, D, H, W, C, K, 1, _p(st), None, None,
                 _p(resd), _p(ws) if use_ws else None, need if use_ws else 0)
        torch.cuda.synchronize()
        outs.append((U.ncdhw(y), st.cpu()))
    ref = (F.conv3d(bf16_round(x).double(), bf16_round(w).double(), None, padding=1) + res.double()).clamp_min(0)
    scale = ref.abs().max().item()
    assert torch.equal(outs[0][0], outs[1][0])
    for y, st in outs:
        assert (y.double() - ref).abs().max().item() < 1e-3 * scale
        assert torch.allclose(st[..., 0], y.double().sum(dim=(2, 3, 4)), rtol=1e-5, atol=1e-4)
        assert torch.allclose(st[..., 1], (y.double() ** 2).sum(dim=(2, 3, 4)), rtol=1e-5, atol=1e-4)


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
    ((1, 8, 24, 40), 96, 32),     # Cout % 64 == 32 (config 2's dec2.c1): 64-column blocks, upper half read as zero
    ((2, 6, 9, 17), 32, 96),      # ... with a whole block before the half one
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


def test_batched_bf16_weight_pack_equals_the_per_layer_pack():
    """u3d_pack_weights_bf16_batch (one launch, LDS transposition, HBM rate) writes bit for bit the images of
    u3d_pack_weights_bf16, for both modes, incl. the zeroed prefetch tail"""
    import ctypes

    lib = nat.get_lib()
    torch.manual_seed(12)
    shapes = [(64, 32), (32, 64), (128, 128), (96, 160), (512, 256)]  # (Cout, Cin)
    ws = [torch.randn(co, ci, 3, 3, 3, device=U.DEV) for co, ci in shapes]
    jobs = [(w, mode) for w in ws for mode in (0, 1)]
    descs = (nat.U3DPackDesc * len(jobs))()
    outs, first = [], 0
    for i, (w, mode) in enumerate(jobs):
        co, ci = w.shape[:2]
        n = lib.u3d_packed_weight_bf16_elems(ci, co, mode)
        buf = torch.full((n,), float("nan"), dtype=torch.bfloat16, device=U.DEV)
        outs.append(buf)
        descs[i].w, descs[i].packed, descs[i].first = w.data_ptr(), buf.data_ptr(), first
        descs[i].Cout, descs[i].Cin, descs[i].mode, descs[i].cin_stride = co, ci, mode, 0
        first += lib.u3d_pack_weights_bf16_blocks(ci, co, mode)
    table = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(U.DEV)
    nat.call("u3d_pack_weights_bf16_batch", 0, _stream(U.DEV), _p(table), len(jobs), first)
    for (w, mode), got in zip(jobs, outs):
        co, ci = w.shape[:2]
        ref = torch.full_like(got, float("nan"))
        nat.call("u3d_pack_weights_bf16", 0, _stream(U.DEV), _p(w), co, ci, mode, _p(ref))
        torch.cuda.synchronize()
        assert torch.equal(got.view(torch.int16), ref.view(torch.int16)), (co, ci, mode)


def test_both_images_from_one_read_equal_the_two_single_mode_packs():
    """round 6: mode 6 of u3d_pack_weights_bf16_batch — the forward and the data-gradient image of a weight from ONE read of it — writes bit
    for bit what modes 0 and 1 (u3d_pack_weights_bf16) write, the second image right behind the first, both prefetch tails zeroed; mixed in
    one launch with single-mode descriptors; channel counts that are not multiples of 32 are not eligible"""
    lib = nat.get_lib()
    torch.manual_seed(13)
    shapes = [(64, 32), (32, 64), (128, 128), (96, 160), (512, 256), (1024, 1024)]  # (Cout, Cin)
    ws = [torch.randn(co, ci, 3, 3, 3, device=U.DEV) for co, ci in shapes]
    extra = torch.randn(64, 64, 3, 3, 3, device=U.DEV)
    descs = (nat.U3DPackDesc * (len(ws) + 1))()
    outs, first = [], 0
    for i, w in enumerate(ws):
        co, ci = w.shape[:2]
        n0, n1 = lib.u3d_packed_weight_bf16_elems(ci, co, 0), lib.u3d_packed_weight_bf16_elems(ci, co, 1)
        buf = torch.full((n0 + n1 + 64,), float("nan"), dtype=torch.bfloat16, device=U.DEV)
        outs.append((buf, n0, n1))
        descs[i].w, descs[i].packed, descs[i].first = w.data_ptr(), buf.data_ptr(), first
        descs[i].Cout, descs[i].Cin, descs[i].mode, descs[i].cin_stride = co, ci, 6, 0
        assert lib.u3d_pack_weights_bf16_blocks(ci, co, 6) == (ci // 32) * (co // 32) + 2
        first += lib.u3d_pack_weights_bf16_blocks(ci, co, 6)
    nx = lib.u3d_packed_weight_bf16_elems(64, 64, 1)
    bx = torch.full((nx,), float("nan"), dtype=torch.bfloat16, device=U.DEV)
    i = len(ws)
    descs[i].w, descs[i].packed, descs[i].first = extra.data_ptr(), bx.data_ptr(), first
    descs[i].Cout, descs[i].Cin, descs[i].mode, descs[i].cin_stride = 64, 64, 1, 0
    first += lib.u3d_pack_weights_bf16_blocks(64, 64, 1)
    table = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(U.DEV)
    nat.call("u3d_pack_weights_bf16_batch", 0, _stream(U.DEV), _p(table), len(ws) + 1, first)
    for w, (buf, n0, n1) in zip(ws, outs):
        co, ci = w.shape[:2]
        r0, r1 = torch.empty(n0, dtype=torch.bfloat16, device=U.DEV), torch.empty(n1, dtype=torch.bfloat16, device=U.DEV)
        nat.call("u3d_pack_weights_bf16", 0, _stream(U.DEV), _p(w), co, ci, 0, _p(r0))
        nat.call("u3d_pack_weights_bf16", 0, _stream(U.DEV), _p(w), co, ci, 1, _p(r1))
        torch.cuda.synchronize()
        assert torch.equal(buf[:n0].view(torch.int16), r0.view(torch.int16)), (co, ci, 0)
        assert torch.equal(buf[n0:n0 + n1].view(torch.int16), r1.view(torch.int16)), (co, ci, 1)
        assert bool(torch.isnan(buf[n0 + n1:].float()).all())  # nothing written past the second image
    rx = torch.empty(nx, dtype=torch.bfloat16, device=U.DEV)
    nat.call("u3d_pack_weights_bf16", 0, _stream(U.DEV), _p(extra), 64, 64, 1, _p(rx))
    assert torch.equal(bx.view(torch.int16), rx.view(torch.int16))
    assert lib.u3d_pack_weights_bf16_blocks(48, 64, 6) == 0 and lib.u3d_pack_weights_bf16_blocks(64, 16, 6) == 0


def test_batched_pack_of_the_transposed_convolution_images_equals_the_per_weight_pack():
    """round 5: modes 4 / 5 of u3d_pack_weights_bf16_batch — the space-to-depth (T8) images of ConvTranspose3d weights (Cl, Cs, 3,3,3)
    ride in the model's one pack launch; bit for bit the images of u3d_pack_convtr3d_t8 (forward / data gradient), mixed with 3x3x3
    images in one descriptor table, incl. the zeroed prefetch tails"""
    lib = nat.get_lib()
    torch.manual_seed(21)
    t8 = [torch.randn(cl, cs, 3, 3, 3, device=U.DEV) for cl, cs in [(64, 32), (128, 64), (256, 128), (96, 160)]]
    w3 = torch.randn(64, 32, 3, 3, 3, device=U.DEV)
    jobs = [(t8[0], 4), (w3, 0), (t8[1], 5), (t8[1], 4), (t8[2], 4), (t8[2], 5), (w3, 1), (t8[3], 4), (t8[3], 5), (t8[0], 5)]
    assert lib.u3d_pack_weights_bf16_blocks(64, 24, 4) == 0  # Cs % 32 != 0: not batchable (falls back to the per-weight kernel)
    descs = (nat.U3DPackDesc * len(jobs))()
    outs, first = [], 0
    for i, (w, mode) in enumerate(jobs):
        if mode >= 4:
            ci, co = w.shape[:2]  # desc.Cin = Cl, desc.Cout = Cs
            n = lib.u3d_convtr3d_t8_packed_elems(ci, co, mode - 4)
        else:
            co, ci = w.shape[:2]
            n = lib.u3d_packed_weight_bf16_elems(ci, co, mode)
        buf = torch.full((n,), float("nan"), dtype=torch.bfloat16, device=U.DEV)
        outs.append(buf)
        descs[i].w, descs[i].packed, descs[i].first = w.data_ptr(), buf.data_ptr(), first
        descs[i].Cout, descs[i].Cin, descs[i].mode, descs[i].cin_stride = co, ci, mode, 0
        blocks = lib.u3d_pack_weights_bf16_blocks(ci, co, mode)
        assert blocks > 0
        first += blocks
    table = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(U.DEV)
    nat.call("u3d_pack_weights_bf16_batch", 0, _stream(U.DEV), _p(table), len(jobs), first)
    for (w, mode), got in zip(jobs, outs):
        ref = torch.full_like(got, float("nan"))
        if mode >= 4:
            nat.call("u3d_pack_convtr3d_t8", 0, _stream(U.DEV), _p(w), w.shape[0], w.shape[1], mode - 4, _p(ref))
        else:
            nat.call("u3d_pack_weights_bf16", 0, _stream(U.DEV), _p(w), w.shape[0], w.shape[1], mode, _p(ref))
        torch.cuda.synchronize()
        assert torch.equal(got.view(torch.int16), ref.view(torch.int16)), (tuple(w.shape), mode)


# ---- model level ------------------------------------------------------------------------------------------------------
MODEL_CASES = [
    # config 4's model family at reduced width / size: every 3x3x3 conv has channel counts that are multiples of 32
    (dict(name="ResidualUNet3D", in_channels=1, out_channels=1, f_maps=[32, 64, 128], num_groups=8), (1, 1, 16, 32, 32)),
    (dict(name="ResidualUNet3D", in_channels=2, out_channels=2, f_maps=[64, 128], num_groups=8, final_sigmoid=False), (2, 2, 9, 13, 21)),
    # UNet3D: every conv whose channel counts fit runs bf16 — the decoders' first convs on a MATERIALISED concat since round 4 (the
    # engine's _cat_bf16); the first layer stays fp32
    (dict(name="UNet3D", in_channels=1, out_channels=1, f_maps=32, num_levels=3, num_groups=8), (1, 1, 16, 32, 32)),
]


def _prep(cfg, shape, **extra):
    from pytorch3dunet_amd.unet3d.model import get_model

    torch.manual_seed(99)
    extra.setdefault("activation_dtype", "fp32")  # this file pins the bf16-OPERAND kernels (fp32 tensors in HBM); bf16 storage: test_gpu_b16.py
    model = get_model(dict(cfg, **extra))
    with torch.no_grad():
        for k, p in model.named_parameters():
            if "groupnorm" in k:
                p.add_(0.2 * torch.randn_like(p))
    x = torch.randn(shape)
    target = (torch.rand((shape[0], cfg["out_channels"]) + shape[2:]) > 0.5).float()
    return model, x, target


def _step(model, x, target, loss_name):
    from conftest import loss_by_name

    model = model.to(U.DEV).train()
    prof = nat.EventProfiler()
    nat.profiler = prof
    try:
        probs, logits = model(x.to(U.DEV), return_logits=True)
        loss = loss_by_name(loss_name, probs, logits, target.to(U.DEV))
        model.zero_grad()
        loss.backward()
        torch.cuda.synchronize()
    finally:
        nat.profiler = None
    return logits.detach().cpu(), loss.item(), {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters()}, set(prof.summary())


BF16_LOGITS_TOL = 3e-2   # stated bf16 tolerance against the fp32 reference path: logits within 3 % of their range
BF16_GRAD_TOL = 0.15     # ... and the global relative L2 distance of all parameter gradients within 15 % (measured 0.7-10 %
#                          on these small networks; the CPU emulation of the same operand rounding is equally far from fp32)


@pytest.mark.parametrize("cfg,shape", MODEL_CASES)
def test_model_bf16_against_bf16_operand_oracle_and_fp32_oracle(cfg, shape):
    """compute_dtype='bf16' (BASELINE config 4's "bf16 compute"): (1) against the oracle with the same operand rounding restated
    (bf16 operands, wide accumulation).  The kernels themselves reproduce that arithmetic to 4e-7 (kernel tests above), but a
    NETWORK of them is chaotic in the last bf16 bit: a 1e-6 difference in a layer's input flips a few operand roundings in the
    next (measured layer by layer: 9e-5 -> 2e-3 over ten layers), so the network-level gate is "closer to the emulation than the
    emulation is to fp32"; (2) within the STATED bf16 tolerance of the plain fp32 oracle = the reference path."""
    import unet3d_oracle as orc
    from conftest import diag

    loss_name = "bce_dice" if cfg.get("final_sigmoid", True) else "probs_sum"
    model, x, target = _prep(cfg, shape, compute_dtype="bf16")
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    G, fs = cfg["num_groups"], cfg.get("final_sigmoid", True)
    _, l32, _, g32 = orc.forward_backward(sd, x, target, G, fs, True, loss_name)
    orc.BF16_OPERANDS = True
    try:
        _, l16, _, g16 = orc.forward_backward(sd, x, target, G, fs, True, loss_name)
    finally:
        orc.BF16_OPERANDS = False
    logits, loss, grads, names = _step(model, x, target, loss_name)
    assert "u3d_conv3d_bf16_ex" in names and "u3d_conv3d_wgrad_bf16_job" in names, names
    if cfg["name"] == "UNet3D":  # the decoders' concat was written out and no sub-pixel (fp32) kernel ran on it
        assert "u3d_nearest_cat_fwd" in names and not any(n.startswith("u3d_subpixel") for n in names), names
    keys = list(g32)
    cat = lambda d: torch.cat([d[k].flatten().double() for k in keys])  # noqa: E731
    ours, r16, r32 = cat(grads), cat(g16), cat(g32)
    e_l16, e_l32 = orc.rel_err(logits, l16), orc.rel_err(logits, l32)
    e_g16 = ((ours - r16).norm() / r16.norm()).item()
    e_g32 = ((ours - r32).norm() / r32.norm()).item()
    e_or = ((r16 - r32).norm() / r32.norm()).item()
    rec = dict(test="bf16_model", cfg=str(cfg), logits_vs_bf16_oracle=e_l16, logits_vs_fp32_oracle=e_l32, grad_l2_vs_bf16_oracle=e_g16,
               grad_l2_vs_fp32_oracle=e_g32, bf16_oracle_vs_fp32_oracle_grad_l2=e_or)
    diag(**rec)
    print(rec)
    e_l_or = orc.rel_err(l16, l32)
    assert e_l16 < 0.75 * e_l_or and e_g16 < 0.75 * e_or, (rec, e_l_or)
    assert e_l32 < BF16_LOGITS_TOL and e_g32 < BF16_GRAD_TOL, rec


@pytest.mark.parametrize("bf16", [False, True])
def test_activation_checkpointing_gives_bitwise_identical_gradients(bf16):
    """checkpoint_encoders=True re-runs the encoder blocks' forward in backward instead of keeping their intermediates
    (BASELINE config 4): same kernels on the same inputs -> every parameter gradient identical bit for bit, less memory."""
    cfg = dict(name="ResidualUNet3D", in_channels=1, out_channels=1, f_maps=[32, 64, 128], num_groups=8)
    shape = (1, 1, 24, 48, 48)
    res = {}
    for ck in (False, True):
        model, x, target = _prep(cfg, shape, compute_dtype="bf16" if bf16 else "fp32", checkpoint_encoders=ck)
        assert model._get_engine().checkpoint_encoders == ck and model._get_engine().bf16 == bf16
        import gc

        gc.collect()  # the previous iteration's model / graph (reference cycles) must not be freed INSIDE the measured step
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        logits, loss, grads, _ = _step(model, x, target, "bce_dice")
        res[ck] = (logits, loss, grads, torch.cuda.max_memory_allocated() - base)
        del model
    (l0, loss0, g0, m0), (l1, loss1, g1, m1) = res[False], res[True]
    assert torch.equal(l0, l1) and loss0 == loss1
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k
    print(f"peak step memory: {m0 / 2**20:.1f} MiB without, {m1 / 2**20:.1f} MiB with encoder checkpointing")
    from conftest import diag

    diag(test="checkpoint_peak_memory", bf16=bf16, peak_mib_without=m0 / 2**20, peak_mib_with=m1 / 2**20)
    # the tape is released block by block during backward (engine.Tape.lean): the peak must drop by a real fraction of the
    # step's working set, not by the 4 % of round 2 (one autograd node then kept every activation until its backward returned)
    assert m1 < 0.95 * m0, (m0, m1)  # (this toy net is dominated by weights and scratch; config 4 itself: tools/model_bench.py)


def test_checkpointing_releases_the_tape_during_backward_and_refuses_a_second_walk():
    cfg = dict(name="ResidualUNet3D", in_channels=1, out_channels=1, f_maps=[32, 64], num_groups=8)
    model, x, target = _prep(cfg, (1, 1, 16, 32, 32), checkpoint_encoders=True)
    model = model.to(U.DEV).train()
    assert model._get_engine().lean_tape
    _, logits = model(x.to(U.DEV), return_logits=True)
    loss = logits.square().mean()
    loss.backward(retain_graph=True)  # (autograd keeps ITS buffers; ours are released block by block)
    g1 = [p.grad.clone() for p in model.parameters()]
    with pytest.raises(RuntimeError, match="second time"):
        loss.backward()
    # a fresh forward works as before and reproduces the gradients bit for bit
    model.zero_grad()
    _, logits = model(x.to(U.DEV), return_logits=True)
    logits.square().mean().backward()
    for p, a in zip(model.parameters(), g1):
        assert torch.equal(p.grad, a)


# ---- ConvTranspose3d(k3, s2, p1) in space-to-depth form on the bf16 kernels (u3d_convtr3d_*_t8) --------------------------------
def _t8_to_t(t8, Cs, Dt, Ht, Wt):
    """(N,D1,H1,W1,8*Cs) device tensor -> (N,Cs,Dt,Ht,Wt) cpu: t[2i + p] = T8[i][p]"""
    N, D1, H1, W1, _ = t8.shape
    v = t8.cpu().view(N, D1, H1, W1, 2, 2, 2, Cs).permute(0, 7, 1, 4, 2, 5, 3, 6).reshape(N, Cs, 2 * D1, 2 * H1, 2 * W1)
    return v[:, :, :Dt, :Ht, :Wt].contiguous()


def _t_to_t8(t, D1, H1, W1):
    """(N,Cs,Dt,Ht,Wt) cpu -> (N,D1,H1,W1,8*Cs) device (entries outside the (2n-1) grid zero)"""
    N, Cs, Dt, Ht, Wt = t.shape
    full = torch.zeros(N, Cs, 2 * D1, 2 * H1, 2 * W1)
    full[:, :, :Dt, :Ht, :Wt] = t
    v = full.view(N, Cs, D1, 2, H1, 2, W1, 2).permute(0, 2, 4, 6, 3, 5, 7, 1).reshape(N, D1, H1, W1, 8 * Cs)
    return v.contiguous().to(U.DEV)


@pytest.mark.parametrize("shape,Cl,Cs", [((1, 5, 10, 10), 64, 32), ((2, 4, 9, 7), 32, 8), ((1, 16, 24, 40), 32, 16), ((1, 40, 48, 48), 32, 16),
                                         ((1, 6, 10, 12), 128, 64),   # (Cs = 64: a block / chunk = ONE output parity, dead taps skipped)
                                         ((1, 3, 5, 6), 256, 128)])    # (64 chunks on 8 tile blocks: the split-K data gradient)
def test_convtranspose3d_space_to_depth_bf16(shape, Cl, Cs):
    """forward, data gradient (with the ReLU mask of x) and weight gradient of nn.ConvTranspose3d(Cl, Cs, 3, stride=2, padding=1,
    bias=False) against torch with the operands rounded to bf16 the same way"""
    N, D1, H1, W1 = shape
    Dt, Ht, Wt = 2 * D1 - 1, 2 * H1 - 1, 2 * W1 - 1
    lib = nat.get_lib()
    assert lib.u3d_convtr3d_t8_supported(Cl, Cs) == 1
    torch.manual_seed(7)
    x = torch.relu(torch.randn(N, Cl, D1, H1, W1))  # post-ReLU like the real input (exact zeros exercise the mask)
    w = torch.randn(Cl, Cs, 3, 3, 3) / (27 * Cl / 8) ** 0.5
    dt = torch.randn(N, Cs, Dt, Ht, Wt)
    xr = bf16_round(x).double().requires_grad_(True)
    wr = bf16_round(w).double().requires_grad_(True)
    t_ref = F.conv_transpose3d(xr, wr, None, stride=2, padding=1)
    xd, wd = U.ndhwc(x), w.contiguous().to(U.DEV)

    def pack(mode):
        buf = torch.empty(lib.u3d_convtr3d_t8_packed_elems(Cl, Cs, mode), dtype=torch.bfloat16, device=U.DEV)
        nat.call("u3d_pack_convtr3d_t8", 0, _stream(U.DEV), _p(wd), Cl, Cs, mode, _p(buf))
        return buf

    # forward
    t8 = torch.empty((N, D1, H1, W1, 8 * Cs), dtype=torch.float32, device=U.DEV)
    pk0 = pack(0)
    nat.call("u3d_convtr3d_fwd_t8", 0, _stream(U.DEV), _p(xd), _p(pk0), _p(t8), N, D1, H1, W1, Cl, Cs)
    torch.cuda.synchronize()
    t = _t8_to_t(t8, Cs, Dt, Ht, Wt)
    scale = t_ref.abs().max().item()
    assert (t.double() - t_ref.detach()).abs().max().item() < 1e-3 * scale
    # gradients: operands (dt, w) and (x, dt) rounded
    dtr = bf16_round(dt).double()
    gx, gw = torch.autograd.grad(t_ref, (xr, wr), dtr)
    dt8 = _t_to_t8(dt, D1, H1, W1)
    pk1 = pack(1)
    dx = torch.empty((N, D1, H1, W1, Cl), dtype=torch.float32, device=U.DEV)
    nat.call("u3d_convtr3d_dgrad_t8", 0, _stream(U.DEV), _p(dt8), _p(pk1), _p(xd), _p(dx), N, D1, H1, W1, Cl, Cs)
    need = lib.u3d_convtr3d_wgrad_t8_workspace_floats(N, D1, H1, W1, Cl, Cs)
    ws = torch.empty(need, dtype=torch.float32, device=U.DEV)
    dw = torch.full((Cl, Cs, 3, 3, 3), float("nan"), dtype=torch.float32, device=U.DEV)
    nat.call("u3d_convtr3d_wgrad_t8", 0, _stream(U.DEV), _p(xd), _p(dt8), _p(dw), N, D1, H1, W1, Cl, Cs, _p(ws), need)
    torch.cuda.synchronize()
    gx_masked = gx * (x > 0)
    assert (U.ncdhw(dx).double() - gx_masked).abs().max().item() < 1e-3 * gx.abs().max().item()
    # the split-K entry point: same result up to the fp32 summation order (its scratch is 0 floats where the rule does not split)
    nsk = lib.u3d_convtr3d_dgrad_t8_workspace_floats(N, D1, H1, W1, Cl, Cs)
    wsk = torch.empty(max(nsk, 4), dtype=torch.float32, device=U.DEV)
    dx2 = torch.full((N, D1, H1, W1, Cl), float("nan"), dtype=torch.float32, device=U.DEV)
    nat.call("u3d_convtr3d_dgrad_t8_ex", 0, _stream(U.DEV), _p(dt8), _p(pk1), _p(xd), _p(dx2), N, D1, H1, W1, Cl, Cs, _p(wsk), nsk)
    torch.cuda.synchronize()
    assert (U.ncdhw(dx2).double() - gx_masked).abs().max().item() < 1e-3 * gx.abs().max().item()
    if nsk == 0:
        assert torch.equal(dx, dx2)
    assert torch.isfinite(dw).all()
    assert (dw.cpu().double() - gw).abs().max().item() < 1e-3 * gw.abs().max().item()


def test_nearest_resize_join_on_space_to_depth_layout():
    """u3d_nearest_add_fwd_t8 / u3d_nearest_sum_bwd_t8 == the plain-layout kernels on the de-interleaved tensor"""
    from pytorch3dunet_amd.engine import _maps

    torch.manual_seed(8)
    N, D1, H1, W1, C = 2, 3, 5, 4, 8
    Dt, Ht, Wt = 2 * D1 - 1, 2 * H1 - 1, 2 * W1 - 1
    D, H, W = 2 * D1, 2 * H1 + 1, 2 * W1  # skip sizes: even and odd
    t = torch.randn(N, C, Dt, Ht, Wt)
    skip = torch.randn(N, C, D, H, W)
    t8 = _t_to_t8(t, D1, H1, W1)
    (mz, lz), (my, ly), (mx, lx) = _maps(U.DEV, Dt, D), _maps(U.DEV, Ht, H), _maps(U.DEV, Wt, W)
    skd, td = U.ndhwc(skip), U.ndhwc(t)
    outs = []
    for name, src in (("u3d_nearest_add_fwd", td), ("u3d_nearest_add_fwd_t8", t8)):
        out = torch.empty_like(skd)
        st = torch.zeros((N, C, 2), dtype=torch.float64, device=U.DEV)
        nat.call(name, 0, _stream(U.DEV), _p(skd), _p(src), _p(mz), _p(my), _p(mx), N, D, H, W, Dt, Ht, Wt, C, _p(out), _p(st))
        outs.append((out, st))
    torch.cuda.synchronize()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.allclose(outs[0][1], outs[1][1])
    assert torch.allclose(U.ncdhw(outs[0][0]), skip + F.interpolate(t, size=(D, H, W), mode="nearest"))
    dj = torch.randn(N, D, H, W, C, device=U.DEV)
    dt_plain = torch.empty((N, Dt, Ht, Wt, C), dtype=torch.float32, device=U.DEV)
    dt8 = torch.full((N, D1, H1, W1, 8 * C), float("nan"), dtype=torch.float32, device=U.DEV)
    nat.call("u3d_nearest_sum_bwd", 0, _stream(U.DEV), _p(dj), _p(lz), _p(ly), _p(lx), N, D, H, W, Dt, Ht, Wt, C, _p(dt_plain))
    nat.call("u3d_nearest_sum_bwd_t8", 0, _stream(U.DEV), _p(dj), _p(lz), _p(ly), _p(lx), N, D, H, W, Dt, Ht, Wt, C, _p(dt8))
    torch.cuda.synchronize()
    assert torch.isfinite(dt8).all()
    assert torch.equal(_t8_to_t(dt8, C, Dt, Ht, Wt), U.ncdhw(dt_plain))
    # entries outside the (2n-1) grid are zero
    full = dt8.cpu().view(N, D1, H1, W1, 2, 2, 2, C)
    assert float(full[:, -1, :, :, 1].abs().max()) == 0.0 and float(full[:, :, :, -1, :, :, 1].abs().max()) == 0.0
