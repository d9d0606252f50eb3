"""-m gpu: bf16 ACTIVATION STORAGE (`activation_dtype: bf16`; the `_b16` entry points of include/u3d.h, BASELINE config 4).

Kernel level: every `_b16` entry point against the fp32 entry point of the same name on the same (bf16-representable) inputs — the
arithmetic is shared, so the outputs must be the fp32 results rounded to nearest even, bit for bit, and the fused statistics must
be the sums of the values as stored.  Model level: the residual net with bf16 storage against the oracle's storage emulation
(oracle.BF16_STORAGE: forward tensors and gradient tensors rounded where the product stores them)."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

import gpu_utils as U
from conftest import diag
from pytorch3dunet_amd import _native as nat
from pytorch3dunet_amd.engine import _maps, _p, _stream

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
lib = None


def r16(t):
    return t.to(BF).to(torch.float32)


def dev(t):
    return t.contiguous().to(U.DEV)


def call(name, *a):
    nat.call(name, 0, _stream(U.DEV), *a)


_KEEP = []


def b16(t):
    """bf16 copy that stays alive until the test module is done: _p() hands the kernels a raw pointer, a temporary would be
    recycled by the caching allocator before the (asynchronous) launch reads it"""
    if t is None:
        return None
    _KEEP.append(t.to(BF))
    return _KEEP[-1]


def same_bits(a, b):
    return torch.equal(a.view(torch.int16), b.view(torch.int16))


@pytest.mark.parametrize("shape,C,K,res,ks", [((1, 32, 64, 64), 64, 64, True, False), ((2, 5, 9, 11), 32, 96, False, False),
                                              ((1, 4, 8, 8), 512, 512, True, True),
                                              # >= 512 blocks of 8 z-planes: the taller tile of the bf16-storage kernel, full and
                                              # ragged in z (20 = 2 x 8 + 4), one and two 64-channel output blocks
                                              ((1, 64, 64, 64), 64, 64, True, False), ((1, 20, 80, 208), 64, 128, True, False)])
def test_conv3d_bf16_b16_forward_and_data_gradient(shape, C, K, res, ks):
    N, D, H, W = shape
    torch.manual_seed(1)
    x = dev(r16(torch.randn(N, D, H, W, C)))
    aff = dev(torch.stack((1.0 + 0.3 * torch.randn(N, C), 0.2 * torch.randn(N, C)), dim=-1))
    w = dev(torch.randn(K, C, 3, 3, 3) / (27 * C) ** 0.5)
    residual = dev(r16(torch.randn(N, D, H, W, K))) if res else None
    L = nat.get_lib()
    pk = torch.empty(L.u3d_packed_weight_bf16_elems(C, K, 0), dtype=BF, device=U.DEV)
    call("u3d_pack_weights_bf16", _p(w), K, C, 0, _p(pk))
    need = L.u3d_conv3d_bf16_workspace_floats(N, D, H, W, C, K)
    assert (need > 0) == ks
    ws = torch.empty(max(need, 4), dtype=torch.float32, device=U.DEV)
    y32 = torch.empty((N, D, H, W, K), dtype=torch.float32, device=U.DEV)
    st32 = torch.zeros((N, K, 2), dtype=torch.float64, device=U.DEV)
    call("u3d_conv3d_bf16_ex", _p(x), _p(aff), _p(pk), _p(y32), N, D, H, W, C, K, 1, _p(st32), None, None, _p(residual), _p(ws), need)
    y16 = torch.full((N, D, H, W, K), float("nan"), dtype=BF, device=U.DEV)
    st16 = torch.zeros((N, K, 2), dtype=torch.float64, device=U.DEV)
    xb, rb = x.to(BF), (residual.to(BF) if res else None)
    call("u3d_conv3d_bf16_ex_b16", _p(xb), _p(aff), _p(pk), _p(y16), N, D, H, W, C, K, 1, _p(st16), None, None, _p(rb), _p(ws), need)
    torch.cuda.synchronize()
    assert same_bits(y16, y32.to(BF))
    yd = y16.double()
    want = torch.stack((yd.sum(dim=(1, 2, 3)), (yd * yd).sum(dim=(1, 2, 3))), dim=-1)  # statistics of the STORED tensor
    assert torch.allclose(st16, want, rtol=1e-6, atol=1e-6)
    # data gradient role: dz -> dg with the GroupNorm-backward sums against gx
    dz = dev(r16(torch.randn(N, D, H, W, K)))
    pk1 = torch.empty(L.u3d_packed_weight_bf16_elems(C, K, 1), dtype=BF, device=U.DEV)
    call("u3d_pack_weights_bf16", _p(w), K, C, 1, _p(pk1))
    need1 = L.u3d_conv3d_bf16_workspace_floats(N, D, H, W, K, C)
    ws1 = torch.empty(max(need1, 4), dtype=torch.float32, device=U.DEV)
    dg32 = torch.empty((N, D, H, W, C), dtype=torch.float32, device=U.DEV)
    g32 = torch.zeros((N, C, 2), dtype=torch.float64, device=U.DEV)
    call("u3d_conv3d_bf16_ex", _p(dz), None, _p(pk1), _p(dg32), N, D, H, W, K, C, 0, None, _p(x), _p(g32), None, _p(ws1), need1)
    dg16 = torch.full((N, D, H, W, C), float("nan"), dtype=BF, device=U.DEV)
    g16 = torch.zeros((N, C, 2), dtype=torch.float64, device=U.DEV)
    call("u3d_conv3d_bf16_ex_b16", _p(b16(dz)), None, _p(pk1), _p(dg16), N, D, H, W, K, C, 0, None, _p(xb), _p(g16), None, _p(ws1), need1)
    torch.cuda.synchronize()
    assert same_bits(dg16, dg32.to(BF))
    dgd = dg16.double()
    want = torch.stack((dgd.sum(dim=(1, 2, 3)), (dgd * x.double()).sum(dim=(1, 2, 3))), dim=-1)
    assert torch.allclose(g16, want, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("shape,C,K", [((1, 5, 10, 10), 128, 128), ((2, 5, 10, 20), 64, 192), ((1, 9, 17, 10), 128, 64),
                                       ((1, 10, 20, 20), 256, 128)])
def test_conv3d_bf16_b16_flat_tile_of_the_small_wide_levels(shape, C, K):
    """Round 5: the 5 x 10 x 10 FLAT tile (GEMM rows = the tile's voxels in raster order; config 4's 10 x 20 x 20 and 5 x 10 x 10 levels)
    against the 4 x 8 x 8 tiling it replaces there (u3d_set_tuning key 16 = 1).  Both run the same (chunk, tap) sequence per output
    element, so with the SAME split count (key 16 = that count) the two must agree bit for bit — forward (affine, residual, ReLU,
    statistics) and data-gradient role (GroupNorm-backward sums); with its own split count the flat plan differs by fp32 round-off of
    the split sums only.  Full, N = 2, and ragged (9 x 17 x 10: partial tiles along z and y) volumes."""
    N, D, H, W = shape
    torch.manual_seed(5)
    L = nat.get_lib()
    x = dev(r16(torch.randn(N, D, H, W, C))).to(BF)
    aff = dev(torch.stack((1.0 + 0.3 * torch.randn(N, C), 0.2 * torch.randn(N, C)), dim=-1))
    w = dev(torch.randn(K, C, 3, 3, 3) / (27 * C) ** 0.5)
    res = dev(r16(torch.randn(N, D, H, W, K))).to(BF)
    dz = dev(r16(torch.randn(N, D, H, W, K))).to(BF)
    pk = torch.empty(L.u3d_packed_weight_bf16_elems(C, K, 0), dtype=BF, device=U.DEV)
    call("u3d_pack_weights_bf16", _p(w), K, C, 0, _p(pk))
    pk1 = torch.empty(L.u3d_packed_weight_bf16_elems(C, K, 1), dtype=BF, device=U.DEV)
    call("u3d_pack_weights_bf16", _p(w), K, C, 1, _p(pk1))

    def variant(c, k):
        v = L.u3d_conv3d_bf16_tile_variant(N, D, H, W, c, k, 1)
        return v >> 16, (v >> 8) & 255

    def run(role):  # 0: forward (affine, residual, ReLU, statistics), 1: data gradient (GroupNorm-backward sums against gx)
        c, k = (C, K) if role == 0 else (K, C)
        need = L.u3d_conv3d_bf16_workspace_floats(N, D, H, W, c, k)
        assert need > 0
        ws = torch.empty(need, dtype=torch.float32, device=U.DEV)
        y = torch.full((N, D, H, W, k), float("nan"), dtype=BF, device=U.DEV)
        st = torch.zeros((N, k, 2), dtype=torch.float64, device=U.DEV)
        if role == 0:
            call("u3d_conv3d_bf16_ex_b16", _p(x), _p(aff), _p(pk), _p(y), N, D, H, W, C, K, 1, _p(st), None, None, _p(res), _p(ws), need)
        else:
            call("u3d_conv3d_bf16_ex_b16", _p(dz), None, _p(pk1), _p(y), N, D, H, W, K, C, 0, None, _p(x), _p(st), None, _p(ws), need)
        torch.cuda.synchronize()
        return y, st

    auto = None
    for role in (0, 1):
        c, k = (C, K) if role == 0 else (K, C)
        try:
            assert variant(c, k)[1] == 5, variant(c, k)  # the flat tile is what runs by default
            auto_r = run(role)
            nat.call("u3d_set_tuning", 16, 1)
            ks, planes = variant(c, k)
            assert planes == 4 and ks > 1
            old = run(role)
            nat.call("u3d_set_tuning", 16, ks)
            assert variant(c, k) == (ks, 5)
            same = run(role)
        finally:
            nat.call("u3d_set_tuning", 16, 0)
        assert same_bits(same[0], old[0]) and torch.allclose(same[1], old[1], rtol=1e-9, atol=1e-9), role
        assert not torch.isnan(auto_r[0].float()).any()
        assert torch.allclose(auto_r[0].float(), old[0].float(), rtol=1e-2, atol=1e-3)
        assert torch.allclose(auto_r[1], old[1], rtol=1e-3, atol=1e-2)
        auto = auto or auto_r
    ref = F.conv3d(U.ncdhw((x.float() * aff[:, :, 0].view(N, 1, 1, 1, C) + aff[:, :, 1].view(N, 1, 1, 1, C)).cpu()), w.cpu(), None, padding=1)
    ref = torch.relu(ref + U.ncdhw(res.float().cpu()))
    err = (U.ncdhw(auto[0].float().cpu()) - ref).norm() / ref.norm()
    diag(kind="bf16_flat_tile", shape=list(shape), C=C, K=K, rel_l2_vs_fp32=float(err))
    assert err < 1e-2


@pytest.mark.parametrize("with_affine", [True, False])
@pytest.mark.parametrize("shape,C,K,blocks", [((1, 8, 16, 16), 64, 64, 0), ((2, 5, 9, 19), 32, 128, 0),
                                              ((1, 6, 24, 50), 32, 64, 0),    # interior, border and ragged tiles, one per block
                                              ((2, 6, 24, 50), 32, 64, 5),    # ... walked by 5 blocks: tile loop across samples
                                              ((1, 7, 26, 66), 64, 128, 12),
                                              ((1, 12, 40, 40), 64, 64, 7),    # (config 4's 40-wide level: the 8-wide tile by default)
                                              # ONE split (a block owns all tiles of its channel pair — config 4's 1024-channel level): the
                                              # round-4 kernel writes dw itself, no workspace pass; same bits as through the reduction
                                              ((2, 5, 9, 19), 32, 128, 1), ((1, 7, 26, 66), 64, 128, 1)])
# (Cout % 64 == 32 runs the round-3 kernel with a masked upper half in both storage modes: tests/test_gpu_bf16.py::test_conv3d_wgrad_bf16)
def test_conv3d_wgrad_bf16_b16(shape, C, K, blocks, with_affine):
    """bf16 storage.  The round-3 kernel (key 7 = 1) and the round-4 kernel on 2 x 8 x 16 tiles (key 7 = 16) reproduce the
    fp32-storage kernel fed with bf16-representable values bit for bit (same operands, same MFMA order per accumulator); the
    4 x 8 x 8 tile (key 7 = 8) sums the same products in another order — fp32 rounding of a sum, far inside 1e-4 of the
    gradient's scale; the default is one of the two (u3d_conv3d_wgrad_bf16_b16_variant)."""
    N, D, H, W = shape
    torch.manual_seed(2)
    x = dev(r16(torch.randn(N, D, H, W, C)))
    dz = dev(r16(torch.randn(N, D, H, W, K)))
    aff = dev(torch.stack((1.0 + 0.3 * torch.randn(N, C), 0.2 * torch.randn(N, C)), dim=-1)) if with_affine else None
    L = nat.get_lib()
    nat.call("u3d_set_tuning", 8, blocks)
    out = {}
    try:
        need = L.u3d_wgrad_bf16_workspace_floats(N, D, H, W, C, K)
        ws = torch.empty(need, dtype=torch.float32, device=U.DEV)
        ref = torch.full((K, C, 3, 3, 3), float("nan"), device=U.DEV)
        call("u3d_conv3d_wgrad_bf16", _p(x), _p(aff), _p(dz), _p(ref), N, D, H, W, C, K, _p(ws), need)
        default = L.u3d_conv3d_wgrad_bf16_b16_variant(N, D, H, W, C, K)
        assert default in (8, 16)
        for key in (0, 1, 16, 8):
            nat.call("u3d_set_tuning", 7, key)
            assert L.u3d_conv3d_wgrad_bf16_b16_variant(N, D, H, W, C, K) == {0: default, 1: 0}.get(key, key)
            out[key] = torch.full((K, C, 3, 3, 3), float("nan"), device=U.DEV)
            call("u3d_conv3d_wgrad_bf16_b16", _p(b16(x)), _p(aff), _p(b16(dz)), _p(out[key]), N, D, H, W, C, K, _p(ws), need)
        torch.cuda.synchronize()
    finally:
        nat.call("u3d_set_tuning", 7, 0)
        nat.call("u3d_set_tuning", 8, 0)
    assert torch.isfinite(ref).all()
    assert torch.equal(ref, out[1])
    assert torch.equal(ref, out[16])
    scale = float(ref.abs().max())
    assert float((out[8] - ref).abs().max()) < 1e-4 * scale
    assert torch.equal(out[0], out[default])


def test_transposed_convolution_t8_b16():
    N, D1, H1, W1, Cl, Cs = 1, 4, 6, 5, 64, 32
    torch.manual_seed(3)
    L = nat.get_lib()
    x = dev(r16(torch.relu(torch.randn(N, D1, H1, W1, Cl))))
    w = dev(torch.randn(Cl, Cs, 3, 3, 3) / (27 * Cl / 8) ** 0.5)
    dt8 = dev(r16(torch.randn(N, D1, H1, W1, 8 * Cs)))
    pk = [torch.empty(L.u3d_convtr3d_t8_packed_elems(Cl, Cs, m), dtype=BF, device=U.DEV) for m in (0, 1)]
    for m in (0, 1):
        call("u3d_pack_convtr3d_t8", _p(w), Cl, Cs, m, _p(pk[m]))
    t32 = torch.empty((N, D1, H1, W1, 8 * Cs), device=U.DEV)
    t16 = torch.empty((N, D1, H1, W1, 8 * Cs), dtype=BF, device=U.DEV)
    call("u3d_convtr3d_fwd_t8", _p(x), _p(pk[0]), _p(t32), N, D1, H1, W1, Cl, Cs)
    call("u3d_convtr3d_fwd_t8_b16", _p(b16(x)), _p(pk[0]), _p(t16), N, D1, H1, W1, Cl, Cs)
    dx32 = torch.empty_like(x)
    dx16 = torch.empty(x.shape, dtype=BF, device=U.DEV)
    call("u3d_convtr3d_dgrad_t8", _p(dt8), _p(pk[1]), _p(x), _p(dx32), N, D1, H1, W1, Cl, Cs)
    call("u3d_convtr3d_dgrad_t8_b16", _p(b16(dt8)), _p(pk[1]), _p(b16(x)), _p(dx16), N, D1, H1, W1, Cl, Cs)
    need = L.u3d_convtr3d_wgrad_t8_workspace_floats(N, D1, H1, W1, Cl, Cs)
    ws = torch.empty(need, device=U.DEV)
    dwa, dwb = torch.empty_like(w), torch.empty_like(w)
    call("u3d_convtr3d_wgrad_t8", _p(x), _p(dt8), _p(dwa), N, D1, H1, W1, Cl, Cs, _p(ws), need)
    call("u3d_convtr3d_wgrad_t8_b16", _p(b16(x)), _p(b16(dt8)), _p(dwb), N, D1, H1, W1, Cl, Cs, _p(ws), need)
    # round 5: the bf16-storage weight gradient runs conv3d_wgrad_t8v2_kernel (4 x 8 x 8 tiles: the voxels are summed in another order
    # than the fp32-storage kernel's 2 x 8 x 16 tiles — fp32 rounding of a sum); key 7 = 1 selects the round-3 kernel, bit-identical
    dwc = torch.full_like(w, float("nan"))
    nat.call("u3d_set_tuning", 7, 1)
    try:
        call("u3d_convtr3d_wgrad_t8_b16", _p(b16(x)), _p(b16(dt8)), _p(dwc), N, D1, H1, W1, Cl, Cs, _p(ws), need)
        torch.cuda.synchronize()
    finally:
        nat.call("u3d_set_tuning", 7, 0)
    assert same_bits(t16, t32.to(BF)) and same_bits(dx16, dx32.to(BF)) and torch.equal(dwa, dwc)
    assert torch.isfinite(dwb).all() and float((dwb - dwa).abs().max()) < 1e-4 * float(dwa.abs().max())


@pytest.mark.parametrize("dims,Cl,Cs", [((1, 5, 10, 10), 128, 64), ((1, 9, 17, 10), 64, 32)])
def test_transposed_convolution_data_gradient_on_the_flat_tile(dims, Cl, Cs):
    """u3d_convtr3d_dgrad_t8_b16_ex on the flat 5 x 10 x 10 tile (round 5, config 4's two bottom levels) against the 4 x 8 x 8 tiling
    (u3d_set_tuning key 16 = 1): bit-identical with the same split count (key 16 = that count), fp32 round-off of the split sums apart
    with its own; the ReLU mask of x is applied by the fixed-order reduction in both."""
    N, D1, H1, W1 = dims
    torch.manual_seed(6)
    L = nat.get_lib()
    x = dev(r16(torch.relu(torch.randn(N, D1, H1, W1, Cl)))).to(BF)
    w = dev(torch.randn(Cl, Cs, 3, 3, 3) / (27 * Cl / 8) ** 0.5)
    dt8 = dev(r16(torch.randn(N, D1, H1, W1, 8 * Cs))).to(BF)
    pk = torch.empty(L.u3d_convtr3d_t8_packed_elems(Cl, Cs, 1), dtype=BF, device=U.DEV)
    call("u3d_pack_convtr3d_t8", _p(w), Cl, Cs, 1, _p(pk))

    def run():
        need = L.u3d_convtr3d_dgrad_t8_workspace_floats(N, D1, H1, W1, Cl, Cs)
        assert need > 0
        ws = torch.empty(need, device=U.DEV)
        dx = torch.full(x.shape, float("nan"), dtype=BF, device=U.DEV)
        call("u3d_convtr3d_dgrad_t8_b16_ex", _p(dt8), _p(pk), _p(x), _p(dx), N, D1, H1, W1, Cl, Cs, _p(ws), need)
        torch.cuda.synchronize()
        return dx, need // (N * D1 * H1 * W1 * Cl)

    try:
        auto, _ = run()
        nat.call("u3d_set_tuning", 16, 1)
        old, ks = run()
        assert ks > 1
        nat.call("u3d_set_tuning", 16, ks)
        same, ks2 = run()
    finally:
        nat.call("u3d_set_tuning", 16, 0)
    assert ks2 == ks and same_bits(same, old)
    assert not torch.isnan(auto.float()).any() and torch.allclose(auto.float(), old.float(), rtol=1e-2, atol=1e-3)
    assert bool((auto.float()[x.float() <= 0] == 0).all())


def test_transposed_convolution_forward_on_the_flat_tile():
    """u3d_convtr3d_fwd_t8_b16_ex: on a small low-res grid with many channels (config 4's bottom level) the forward runs the flat
    5 x 10 x 10 tile with a split channel reduction (scratch from u3d_convtr3d_fwd_t8_workspace_floats); against the plain entry point
    (4 x 8 x 8 tiles, unsplit): the same products summed in another grouping — fp32 round-off before the bf16 rounding — and against the
    fp32 reference of the transposed convolution in space-to-depth form."""
    N, D1, H1, W1, Cl, Cs = 1, 5, 10, 10, 128, 64
    torch.manual_seed(8)
    L = nat.get_lib()
    x = dev(r16(torch.relu(torch.randn(N, D1, H1, W1, Cl)))).to(BF)
    w = dev(torch.randn(Cl, Cs, 3, 3, 3) / (27 * Cl / 8) ** 0.5)
    pk = torch.empty(L.u3d_convtr3d_t8_packed_elems(Cl, Cs, 0), dtype=BF, device=U.DEV)
    call("u3d_pack_convtr3d_t8", _p(w), Cl, Cs, 0, _p(pk))
    need = L.u3d_convtr3d_fwd_t8_workspace_floats(N, D1, H1, W1, Cl, Cs)
    assert need > 0 and need % (N * D1 * H1 * W1 * 8 * Cs) == 0
    assert L.u3d_convtr3d_fwd_t8_workspace_floats(1, 20, 40, 40, 256, 128) == 0  # (config 4's upper levels keep the plain plan)
    ws = torch.empty(need, device=U.DEV)
    plain = torch.full((N, D1, H1, W1, 8 * Cs), float("nan"), dtype=BF, device=U.DEV)
    flat = torch.full_like(plain, float("nan"))
    call("u3d_convtr3d_fwd_t8_b16", _p(x), _p(pk), _p(plain), N, D1, H1, W1, Cl, Cs)
    call("u3d_convtr3d_fwd_t8_b16_ex", _p(x), _p(pk), _p(flat), N, D1, H1, W1, Cl, Cs, _p(ws), need)
    nows = torch.full_like(plain, float("nan"))
    call("u3d_convtr3d_fwd_t8_b16_ex", _p(x), _p(pk), _p(nows), N, D1, H1, W1, Cl, Cs, None, 0)  # no scratch: the plain plan
    torch.cuda.synchronize()
    assert same_bits(nows, plain)
    assert not torch.isnan(flat.float()).any() and torch.allclose(flat.float(), plain.float(), rtol=1e-2, atol=1e-3)
    # reference: ConvTranspose3d(k3, s2, p1) on the bf16-rounded operands, T8[i][parity*Cs + c] = t[2i + parity]; output_padding = 1 supplies
    # the 2n-th plane / row / column the space-to-depth form computes as well (the join never reads it)
    full = F.conv_transpose3d(U.ncdhw(x.float()), r16(w.cpu()), stride=2, padding=1, output_padding=1)
    ref = full.view(N, Cs, D1, 2, H1, 2, W1, 2).permute(0, 2, 4, 6, 3, 5, 7, 1).reshape(N, D1, H1, W1, 8 * Cs)
    err = (flat.float().cpu() - ref).norm() / ref.norm()
    assert err < 1e-2, err


def test_bandwidth_kernels_b16():
    torch.manual_seed(4)
    N, D, H, W, C = 2, 6, 9, 10, 64
    V = D * H * W
    # GroupNorm backward apply (+ add, + ReLU mask)
    dg, x, add = (dev(r16(torch.randn(N, D, H, W, C))) for _ in range(3))
    coef = dev(torch.randn(N, 3, C))
    for addt, mask in ((None, 1), (add, 0)):
        o32 = torch.empty_like(x)
        o16 = torch.empty(x.shape, dtype=BF, device=U.DEV)
        if addt is None:
            call("u3d_gn_bwd_apply", _p(dg), C, 0, _p(x), C, _p(coef), C, V, N, mask, _p(o32))
        else:
            call("u3d_gn_bwd_apply_add", _p(dg), C, 0, _p(x), C, _p(coef), C, V, N, mask, _p(addt), _p(o32))
        call("u3d_gn_bwd_apply_b16", _p(b16(dg)), C, 0, _p(b16(x)), C, _p(coef), C, V, N, mask,
             _p(b16(addt)) if addt is not None else None, _p(o16))
        torch.cuda.synchronize()
        assert same_bits(o16, o32.to(BF))
    # max-pool forward (values and arg-max bytes) and its backward merge with a skip gradient and the ReLU mask
    p32 = torch.empty((N, D // 2, H // 2, W // 2, C), device=U.DEV)
    a32 = torch.empty(p32.shape, dtype=torch.uint8, device=U.DEV)
    p16 = torch.empty(p32.shape, dtype=BF, device=U.DEV)
    a16 = torch.empty_like(a32)
    call("u3d_maxpool2_fwd", _p(x), N, D, H, W, C, _p(p32), _p(a32), None)
    call("u3d_maxpool2_fwd_b16", _p(b16(x)), N, D, H, W, C, _p(p16), _p(a16))
    dpool = dev(r16(torch.randn(p32.shape)))
    skip = dev(r16(torch.randn(N, D, H, W, C)))
    m32 = torch.empty_like(x)
    m16 = torch.empty(x.shape, dtype=BF, device=U.DEV)
    call("u3d_maxpool2_bwd_merge", _p(dpool), _p(p32), _p(a32), None, _p(skip), _p(x), N, D, H, W, C, 1, _p(m32))
    call("u3d_maxpool2_bwd_merge_b16", _p(b16(dpool)), _p(p16), _p(a16), None, _p(b16(skip)), _p(b16(x)), N, D, H, W, C, 1, _p(m16))
    torch.cuda.synchronize()
    assert same_bits(p16, p32.to(BF)) and torch.equal(a16, a32) and same_bits(m16, m32.to(BF))
    # nearest resize + join on the space-to-depth layout, and its backward children sums
    D1, H1, W1 = 3, 5, 5
    Dt, Ht, Wt = 2 * D1 - 1, 2 * H1 - 1, 2 * W1 - 1
    (mz, lz), (my, ly), (mx, lx) = _maps(U.DEV, Dt, D), _maps(U.DEV, Ht, H), _maps(U.DEV, Wt, W)
    t8 = dev(r16(torch.randn(N, D1, H1, W1, 8 * C)))
    j32 = torch.empty_like(x)
    j16 = torch.empty(x.shape, dtype=BF, device=U.DEV)
    s32, s16 = (torch.zeros((N, C, 2), dtype=torch.float64, device=U.DEV) for _ in range(2))
    call("u3d_nearest_add_fwd_t8", _p(skip), _p(t8), _p(mz), _p(my), _p(mx), N, D, H, W, Dt, Ht, Wt, C, _p(j32), _p(s32))
    call("u3d_nearest_add_fwd_t8_b16", _p(b16(skip)), _p(b16(t8)), _p(mz), _p(my), _p(mx), N, D, H, W, Dt, Ht, Wt, C, _p(j16), _p(s16))
    d32 = torch.empty_like(t8)
    d16 = torch.empty(t8.shape, dtype=BF, device=U.DEV)
    call("u3d_nearest_sum_bwd_t8", _p(dg), _p(lz), _p(ly), _p(lx), N, D, H, W, Dt, Ht, Wt, C, _p(d32))
    call("u3d_nearest_sum_bwd_t8_b16", _p(b16(dg)), _p(lz), _p(ly), _p(lx), N, D, H, W, Dt, Ht, Wt, C, _p(d16))
    torch.cuda.synchronize()
    assert same_bits(j16, j32.to(BF)) and same_bits(d16, d32.to(BF))
    jd = j16.double()
    assert torch.allclose(s16, torch.stack((jd.sum(dim=(1, 2, 3)), (jd * jd).sum(dim=(1, 2, 3))), dim=-1), rtol=1e-6, atol=1e-6)


def test_conv1x1_and_head_b16():
    torch.manual_seed(5)
    N, V, Cin, Cout = 2, 700, 64, 128
    x = dev(r16(torch.randn(N, V, Cin)))
    w, b = dev(torch.randn(Cout, Cin) / 8), dev(torch.randn(Cout))
    y32 = torch.empty((N, V, Cout), device=U.DEV)
    y16 = torch.empty((N, V, Cout), dtype=BF, device=U.DEV)
    s32, s16 = (torch.zeros((N, Cout, 2), dtype=torch.float64, device=U.DEV) for _ in range(2))
    call("u3d_conv1x1_fwd", _p(x), _p(w), _p(b), _p(y32), N, V, Cin, Cout, _p(s32))
    call("u3d_conv1x1_fwd_b16", _p(b16(x)), 0, _p(w), _p(b), _p(y16), N, V, Cin, Cout, _p(s16))
    # the first block: fp32 network input (one channel) feeding a bf16 tensor
    x1 = dev(torch.randn(N, V, 1))
    w1, b1 = dev(torch.randn(Cout, 1)), dev(torch.randn(Cout))
    z32 = torch.empty((N, V, Cout), device=U.DEV)
    z16 = torch.empty((N, V, Cout), dtype=BF, device=U.DEV)
    call("u3d_conv1x1_fwd", _p(x1), _p(w1), _p(b1), _p(z32), N, V, 1, Cout, None)
    call("u3d_conv1x1_fwd_b16", _p(x1), 1, _p(w1), _p(b1), _p(z16), N, V, 1, Cout, None)
    dy = dev(r16(torch.randn(N, V, Cout)))
    dx32 = torch.empty_like(x)
    dx16 = torch.empty(x.shape, dtype=BF, device=U.DEV)
    a32, a16_ = (torch.zeros(Cout * Cin + Cout, dtype=torch.float64, device=U.DEV) for _ in range(2))
    call("u3d_conv1x1_bwd", _p(dy), _p(x), _p(w), N, V, Cin, Cout, _p(dx32), _p(a32))
    call("u3d_conv1x1_bwd_b16", _p(b16(dy)), _p(b16(x)), 0, _p(w), N, V, Cin, Cout, _p(dx16), _p(a16_))
    b32, b16_ = (torch.zeros(Cout + Cout, dtype=torch.float64, device=U.DEV) for _ in range(2))
    call("u3d_conv1x1_bwd", _p(dy), _p(x1), _p(w1), N, V, 1, Cout, None, _p(b32))
    call("u3d_conv1x1_bwd_b16", _p(b16(dy)), _p(x1), 1, _p(w1), N, V, 1, Cout, None, _p(b16_))
    torch.cuda.synchronize()
    assert same_bits(y16, y32.to(BF)) and same_bits(dx16, dx32.to(BF))
    # the first block (Cin <= 4) has its own streaming kernels: fused multiply-add instead of the MFMA's product + bias, i.e. the
    # fp32 value may differ in its last bit and, rarely, round to the neighbouring bf16
    zd = (z16.float() - z32.to(BF).float()).abs()
    assert float((zd > 0).float().mean()) < 1e-3 and float((zd / z32.abs().clamp_min(1e-3)).max()) < 2.0 ** -7
    yd = y16.double()
    assert torch.allclose(s16, torch.stack((yd.sum(dim=1), (yd * yd).sum(dim=1)), dim=-1), rtol=1e-6, atol=1e-6)
    assert torch.allclose(a16_, a32, rtol=1e-9, atol=1e-9) and torch.allclose(b16_, b32, rtol=1e-4, atol=1e-4)  # (fp32 partial sums in another order)
    # head
    Co = 2
    wh, bh = dev(torch.randn(Co, Cin) / 8), dev(torch.randn(Co))
    l32, p32, l16, p16 = (torch.empty((N, Co, V), device=U.DEV) for _ in range(4))
    call("u3d_conv1x1_head_fwd", _p(x), _p(wh), _p(bh), N, V, Cin, Co, 2, _p(l32), _p(p32))
    call("u3d_conv1x1_head_fwd_b16", _p(b16(x)), _p(wh), _p(bh), N, V, Cin, Co, 2, _p(l16), _p(p16))
    dl = dev(torch.randn(N, Co, V))
    hx32 = torch.empty_like(x)
    hx16 = torch.empty(x.shape, dtype=BF, device=U.DEV)
    h32, h16 = (torch.zeros(Co * Cin + Co, dtype=torch.float64, device=U.DEV) for _ in range(2))
    call("u3d_conv1x1_head_bwd", _p(dl), _p(x), _p(wh), N, V, Cin, Co, 1, _p(hx32), _p(h32))
    call("u3d_conv1x1_head_bwd_b16", _p(dl), _p(b16(x)), _p(wh), N, V, Cin, Co, 1, _p(hx16), _p(h16))
    torch.cuda.synchronize()
    assert torch.equal(l16, l32) and torch.equal(p16, p32) and same_bits(hx16, hx32.to(BF)) and torch.allclose(h16, h32, rtol=1e-12)
    # one / two outputs from 32 / 64 channels take the row-per-lane kernel (a wave stages 64 voxels in LDS, each lane owns one):
    # same summation order as the shuffle kernel, so the fp32 entry point is still matched bit for bit; V = 700 leaves a partial round
    for ci, co, act in ((64, 1, 1), (32, 2, 2), (32, 1, 0), (64, 1, 0)):
        xs = dev(r16(torch.randn(N, V, ci)))
        w2, b2 = dev(torch.randn(co, ci) / 8), dev(torch.randn(co))
        a32, q32, a16, q16 = (torch.full((N, co, V), float("nan"), device=U.DEV) for _ in range(4))
        call("u3d_conv1x1_head_fwd", _p(xs), _p(w2), _p(b2), N, V, ci, co, act, _p(a32), _p(q32) if act else None)
        call("u3d_conv1x1_head_fwd_b16", _p(b16(xs)), _p(w2), _p(b2), N, V, ci, co, act, _p(a16), _p(q16) if act else None)
        torch.cuda.synchronize()
        assert torch.equal(a16, a32), (ci, co, act, float((a16 - a32).abs().max()))
        if act:
            assert torch.equal(q16, q32), (ci, co, act)


@pytest.mark.parametrize("dims,Cin,Cout", [((2, 3, 7, 11), 64, 128), ((1, 8, 16, 16), 128, 64), ((1, 2, 5, 9), 256, 512),
                                           ((2, 1, 3, 50), 512, 64), ((1, 16, 32, 32), 64, 64),
                                           ((1, 5, 10, 10), 512, 1024)])  # (K = 1024 in the data gradient: fragments from LDS)
def test_conv1x1_on_the_bf16_matrix_pipe(dims, Cin, Cout):
    """ResNetBlock.conv1 under bf16 storage (u3d_conv1x1_*_mfma_b16): bf16 x bf16 products are exact in fp32, so the only freedom
    against a float64 evaluation with the bf16-rounded weight is the fp32 accumulation order — a last-bit difference before the
    final rounding to bf16"""
    torch.manual_seed(11)
    lib = nat.get_lib()
    assert lib.u3d_conv1x1_mfma_b16_supported(Cin, Cout) == 1
    assert lib.u3d_conv1x1_mfma_b16_supported(1, Cout) == 0 and lib.u3d_conv1x1_mfma_b16_supported(2048, 64) == 0
    N, D, H, W = dims
    V = D * H * W
    x = dev(r16(torch.randn(N, V, Cin)))
    dy = dev(r16(torch.randn(N, V, Cout)))
    w, b = dev(torch.randn(Cout, Cin) / 8), dev(torch.randn(Cout))
    if Cin == 512:  # a parameter inside a flat buffer need not be 16-byte aligned
        wbuf = torch.empty(Cout * Cin + 1, device=U.DEV)
        wbuf[1:].copy_(w.view(-1))
        w = wbuf[1:].view(Cout, Cin)
        assert w.data_ptr() % 16 == 4
    xb, dyb = b16(x), b16(dy)
    wr = r16(w).double()
    y_ref = x.double() @ wr.t() + b.double()
    dx_ref = dy.double() @ wr
    dw_ref = torch.einsum("nvo,nvi->oi", dy.double(), x.double())
    db_ref = dy.double().sum(dim=(0, 1))
    outs = []
    need = lib.u3d_conv1x1_bwd_mfma_b16_workspace_floats(N, D, H, W, Cin, Cout)
    assert need > 0
    for _ in range(2):
        y = torch.empty((N, V, Cout), dtype=BF, device=U.DEV)
        st = torch.zeros((N, Cout, 2), dtype=torch.float64, device=U.DEV)
        dx = torch.empty((N, V, Cin), dtype=BF, device=U.DEV)
        dw = torch.full((Cout, Cin), float("nan"), device=U.DEV)
        db = torch.full((Cout,), float("nan"), device=U.DEV)
        ws = torch.empty(need, device=U.DEV)
        call("u3d_conv1x1_fwd_mfma_b16", _p(xb), _p(w), _p(b), _p(y), N, V, Cin, Cout, _p(st))
        call("u3d_conv1x1_bwd_mfma_b16", _p(dyb), _p(xb), _p(w), N, D, H, W, Cin, Cout, _p(dx), _p(dw), _p(db), _p(ws), need)
        torch.cuda.synchronize()
        outs.append((y, st, dx, dw, db))
    for a_, b_ in zip(*outs):  # fixed-order reductions (the statistics are float64 atomics of per-block sums: last bits may move)
        if a_.dtype == torch.float64:
            assert torch.allclose(a_, b_, rtol=1e-12, atol=1e-9)
        else:
            assert torch.equal(a_, b_)
    y, st, dx, dw, db = outs[0]
    for got, ref in ((y, y_ref), (dx, dx_ref)):
        d = (got.double() - r16(ref.float()).double()).abs()
        assert float((d > 0).double().mean()) < 2e-3, float((d > 0).double().mean())
        assert float((d / ref.abs().clamp_min(1e-2)).max()) < 2.0 ** -7
    yd = y.double()
    assert torch.allclose(st, torch.stack((yd.sum(dim=1), (yd * yd).sum(dim=1)), dim=-1), rtol=1e-5, atol=1e-3)
    assert torch.allclose(dw.double(), dw_ref, rtol=1e-4, atol=1e-4 * float(dw_ref.abs().max()))
    assert torch.allclose(db.double(), db_ref, rtol=1e-4, atol=1e-4 * float(db_ref.abs().max()))
    # the data gradient is optional (a block whose input needs no gradient)
    dw2 = torch.empty_like(dw)
    call("u3d_conv1x1_bwd_mfma_b16", _p(dyb), _p(xb), _p(w), N, D, H, W, Cin, Cout, None, _p(dw2), _p(db), _p(ws), need)
    torch.cuda.synchronize()
    assert torch.equal(dw2, dw)
    with pytest.raises(nat.U3DError, match="workspace"):
        call("u3d_conv1x1_bwd_mfma_b16", _p(dyb), _p(xb), _p(w), N, D, H, W, Cin, Cout, None, _p(dw2), _p(db), _p(ws), 16)


# ---- model level -----------------------------------------------------------------------------------------------------------------
CFG = dict(name="ResidualUNet3D", in_channels=1, out_channels=1, f_maps=[64, 128, 256], num_groups=8, final_sigmoid=True)


def _prep(extra, shape=(1, 1, 16, 32, 32)):
    from pytorch3dunet_amd.unet3d.model import get_model

    torch.manual_seed(21)
    model = get_model(dict(CFG, **extra))
    with torch.no_grad():
        for k, p in model.named_parameters():
            if "groupnorm" in k:
                p.add_(0.2 * torch.randn_like(p))
    x = torch.randn(shape)
    t = (torch.rand(shape) > 0.5).float()
    return model, x, t


def _step(model, x, t):
    import unet3d_oracle as orc

    model = model.to(U.DEV).train()
    prof = nat.EventProfiler()
    nat.profiler = prof
    try:
        _, logits = model(x.to(U.DEV), return_logits=True)
        loss = orc.bce_dice_loss(logits, t.to(U.DEV))
        model.zero_grad()
        loss.backward()
        torch.cuda.synchronize()
    finally:
        nat.profiler = None
    return logits.detach().cpu(), loss.item(), {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters()}, set(prof.summary())


@pytest.mark.parametrize("net", ["ResidualUNet3D", "ResidualUNetSE3D"])
def test_model_with_bf16_activation_storage_against_the_storage_emulation(net):
    """(ResidualUNetSE3D since round 4: the squeeze-and-excitation gates read and write bf16 block outputs, `u3d_se_*_b16`)"""
    import unet3d_oracle as orc

    model, x, t = _prep(dict(name=net, compute_dtype="bf16", activation_dtype="bf16"))
    assert model._get_engine().act_bf16
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    _, l32, _, g32 = orc.forward_backward(sd, x, t, 8, True, True, "bce_dice")
    orc.BF16_OPERANDS = orc.BF16_STORAGE = True
    try:
        _, lem, _, gem = orc.forward_backward(sd, x, t, 8, True, True, "bce_dice")
    finally:
        orc.BF16_OPERANDS = orc.BF16_STORAGE = False
    logits, loss, grads, names = _step(model, x, t)
    b16 = {n for n in names if n.endswith("_b16") or n.endswith("_b16_ex") or n.endswith("_b16_job")}
    assert {"u3d_conv3d_bf16_ex_b16", "u3d_conv3d_wgrad_bf16_b16_job", "u3d_convtr3d_fwd_t8_b16", "u3d_convtr3d_dgrad_t8_b16_ex",
            "u3d_convtr3d_wgrad_t8_b16", "u3d_conv1x1_fwd_b16", "u3d_conv1x1_bwd_b16", "u3d_maxpool2_fwd_b16", "u3d_maxpool2_bwd_merge_b16",
            "u3d_nearest_add_fwd_t8_b16", "u3d_nearest_sum_bwd_t8_b16", "u3d_gn_bwd_apply_b16", "u3d_conv1x1_head_fwd_b16",
            "u3d_conv1x1_head_bwd_b16"} <= b16, names
    if net == "ResidualUNetSE3D":
        assert {"u3d_se_apply_fwd_b16", "u3d_se_bwd_reduce_b16", "u3d_se_bwd_apply_b16"} <= b16 and not ({"u3d_se_apply_fwd", "u3d_se_bwd_apply"} & names), names
    # nothing of the fp32-storage family may have run beside them
    assert not ({"u3d_conv3d_bf16_ex", "u3d_conv3d_wgrad_bf16", "u3d_conv3d_wgrad_bf16_job", "u3d_gn_bwd_apply", "u3d_gn_bwd_apply_add", "u3d_maxpool2_fwd",
                 "u3d_conv1x1_fwd", "u3d_conv3d", "u3d_conv3d_ex", "u3d_conv3d_wgrad"} & names), names
    keys = list(g32)
    cat = lambda d: torch.cat([d[k].flatten().double() for k in keys])  # noqa: E731
    ours, em, ref = cat(grads), cat(gem), cat(g32)
    rec = dict(test="bf16_storage_model", net=net, logits_vs_emu=orc.rel_err(logits, lem), logits_vs_fp32=orc.rel_err(logits, l32),
               emu_vs_fp32_logits=orc.rel_err(lem, l32), grad_vs_emu=((ours - em).norm() / em.norm()).item(),
               grad_vs_fp32=((ours - ref).norm() / ref.norm()).item(), emu_vs_fp32_grad=((em - ref).norm() / ref.norm()).item())
    diag(**rec)
    print(rec)
    assert torch.isfinite(ours).all()
    # closer to the emulation of its own arithmetic than that emulation is to fp32, and not farther from fp32 than it
    assert rec["logits_vs_emu"] < 0.75 * rec["emu_vs_fp32_logits"] and rec["grad_vs_emu"] < rec["emu_vs_fp32_grad"], rec
    assert rec["logits_vs_fp32"] < 1.25 * rec["emu_vs_fp32_logits"] + 1e-3 and rec["grad_vs_fp32"] < 1.15 * rec["emu_vs_fp32_grad"] + 1e-3, rec


def test_bf16_storage_is_reproducible_composes_with_checkpointing_and_graphs_and_halves_the_tape():
    res = {}
    # (ckpt2: `checkpoint_encoders: 2` — only the two highest-resolution encoder levels are recomputed, round 5)
    for tag, extra in (("plain", {}), ("ckpt", dict(checkpoint_encoders=True)), ("ckpt2", dict(checkpoint_encoders=2)), ("graph", dict(hip_graph=True))):
        import gc

        model, x, t = _prep(dict(compute_dtype="bf16", activation_dtype="bf16", **extra), shape=(1, 1, 24, 48, 48))
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        out = _step(model, x, t)
        res[tag] = out + (torch.cuda.max_memory_allocated() - base,)
        if tag == "plain":
            again = _step(model, x, t)
            assert torch.equal(out[0], again[0]) and all(torch.equal(out[2][k], again[2][k]) for k in out[2])  # run-to-run identical
        del model
    # level-limited checkpointing sits between the two in memory (the deep levels' tape is small but not nothing)
    assert res["ckpt"][4] <= res["ckpt2"][4] < res["plain"][4], {k: v[4] for k, v in res.items()}
    for tag in ("ckpt", "ckpt2", "graph"):
        assert torch.equal(res["plain"][0], res[tag][0]), tag
        for k in res["plain"][2]:
            assert torch.equal(res["plain"][2][k], res[tag][2][k]), (tag, k)
    import gc

    model, x, t = _prep(dict(compute_dtype="bf16", activation_dtype="fp32"), shape=(1, 1, 24, 48, 48))
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    _step(model, x, t)
    m32 = torch.cuda.max_memory_allocated() - base
    diag(test="bf16_storage_peak", fp32_storage_mib=m32 / 2**20, bf16_storage_mib=res["plain"][4] / 2**20, with_ckpt_mib=res["ckpt"][4] / 2**20)
    # (0.75 until the weight gradient's fp32 split-K scratch — shared by all layers, not tape — was sized for the larger of its two tile
    # shapes: 48 instead of 32 MB of this small step's 242 MB peak; config 4 at its full shape: 4.66 vs 7.48 GB, profiles/r04_cfg4_model_bench.jsonl / r03_f32act_cfg4_model_bench.jsonl)
    assert res["plain"][4] < 0.78 * m32, (res["plain"][4], m32)


def test_activation_bf16_falls_back_with_a_warning_outside_its_envelope():
    from pytorch3dunet_amd.unet3d.model import get_model

    m = get_model(dict(CFG, f_maps=[32, 64], compute_dtype="bf16", activation_dtype="bf16")).to(U.DEV)
    with pytest.warns(UserWarning, match="activation_dtype bf16 requested"):
        eng = m._get_engine()
    assert eng.bf16 and not eng.act_bf16
    import warnings

    m2 = get_model(dict(CFG, compute_dtype="fp32", activation_dtype="bf16")).to(U.DEV)
    assert not m2._get_engine().act_bf16  # (bf16 storage belongs to the bf16 compute path)
    # the default is 'auto': bf16 storage when the model qualifies, silently fp32 when it does not
    assert get_model(dict(CFG, compute_dtype="bf16")).to(U.DEV)._get_engine().act_bf16
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert not get_model(dict(CFG, f_maps=[32, 64], compute_dtype="bf16")).to(U.DEV)._get_engine().act_bf16
    assert not get_model(dict(CFG, compute_dtype="bf16", activation_dtype="fp32")).to(U.DEV)._get_engine().act_bf16


# Bands of the trajectory test below, 3x the measured figures (profiles/r04_parity_diag.jsonl, test "bf16_adam_trajectory": worst step
# deviation 0.83 %, mean of the last five steps 0.33 % apart, update rel-L2 0.39), relative to the fp32 run's loss at the same step.
TRAJ_STEP_BAND = 0.025  # every step: |loss_bf16 - loss_fp32| <= 2.5 % of loss_fp32
TRAJ_TAIL_BAND = 0.01   # mean of the last 5 steps: within 1 %
TRAJ_MIN_DROP = 0.30    # both runs must have learnt: mean(last 5) < 0.30 * loss(step 0) (the host's fp32 module tree: 0.17)


@pytest.mark.timeout(900)
def test_bf16_with_bf16_storage_trains_like_fp32_over_30_adam_steps():
    """nothing showed that `compute_dtype: bf16` + `activation_dtype: bf16` still TRAINS.  A mid-size residual
    net (ResidualUNet3D 64/128/256: the bf16-storage envelope; reference buildingblocks.py:230-288, model.py:193-234) is trained
    for 30 Adam steps (utils.py:307-309 with the shipped config's lr 2e-4, weight decay 1e-5; on the host's module tree the curve
    falls smoothly 1.17 -> 0.19) on a learnable
    synthetic task — the target is a threshold of the locally averaged input, 6 batches cycled — once on the default fp32 path
    and once in bf16 with bf16 storage, from the same initial weights.  Gates: both loss curves fall, they stay within a stated
    band of each other at every step, and the parameter UPDATE of the bf16 run is reported against the fp32 run's."""
    from pytorch3dunet_amd.unet3d.losses import BCEDiceLoss
    from pytorch3dunet_amd.unet3d.model import get_model

    shape, steps = (2, 1, 16, 32, 32), 30
    g = torch.Generator().manual_seed(2024)
    batches = []
    for _ in range(6):
        x = torch.randn(shape, generator=g)
        batches.append((x, (F.avg_pool3d(x, 3, stride=1, padding=1) > 0.05).float()))
    torch.manual_seed(7)
    init = get_model(dict(CFG)).state_dict()
    crit = BCEDiceLoss()
    runs = {}
    for tag, extra in (("fp32", {}), ("bf16", dict(compute_dtype="bf16", activation_dtype="bf16"))):
        model = get_model(dict(CFG, **extra))
        model.load_state_dict(init)
        model = model.to(U.DEV).train()
        eng = model._get_engine()
        assert eng.bf16 == (tag == "bf16") and eng.act_bf16 == (tag == "bf16")
        opt = torch.optim.Adam(model.parameters(), lr=2e-4, weight_decay=1e-5)  # resources/3DUnet_confocal_boundary/train_config.yml:31-35
        losses = []
        n0 = nat.launch_count
        for i in range(steps):
            x, t = batches[i % len(batches)]
            _, logits = model(x.to(U.DEV), return_logits=True)
            loss = crit(logits, t.to(U.DEV))
            losses.append(loss.item())
            opt.zero_grad()
            loss.backward()
            opt.step()
        assert nat.launch_count > n0
        runs[tag] = (losses, {k: v.detach().cpu().double() for k, v in model.state_dict().items()})
        del model, opt
    l32, l16 = runs["fp32"][0], runs["bf16"][0]
    num = den = 0.0
    for k, p0 in init.items():
        u32, u16 = runs["fp32"][1][k] - p0.double(), runs["bf16"][1][k] - p0.double()
        num += (u16 - u32).pow(2).sum().item()
        den += u32.pow(2).sum().item()
    upd_rel = (num / den) ** 0.5
    step_dev = max(abs(a - b) / b for a, b in zip(l16, l32))
    tail32, tail16 = sum(l32[-5:]) / 5, sum(l16[-5:]) / 5
    diag(test="bf16_adam_trajectory", losses_fp32=l32, losses_bf16=l16, worst_step_rel_dev=step_dev, tail_fp32=tail32, tail_bf16=tail16,
         update_rel_l2_bf16_vs_fp32=upd_rel)
    print(f"fp32 {l32[0]:.4f} -> {tail32:.4f}; bf16 {l16[0]:.4f} -> {tail16:.4f}; worst step deviation {step_dev:.3%}; "
          f"update rel-L2 (bf16 vs fp32) {upd_rel:.3f}")
    assert all(map(lambda v: v == v and v < 10, l16))  # finite
    assert tail32 < TRAJ_MIN_DROP * l32[0] and tail16 < TRAJ_MIN_DROP * l16[0], (l32, l16)
    assert step_dev < TRAJ_STEP_BAND, (step_dev, l32, l16)
    assert abs(tail16 - tail32) < TRAJ_TAIL_BAND * tail32, (tail16, tail32)
    assert upd_rel < 1.0, upd_rel  # reported above; Adam moves round-off-level coordinates by +-lr in arbitrary directions (cf. the fp32 test)
