"""CPU: the algebra behind csrc/u3d_subpix.hip, restated in oracle/subpixel_oracle.py, against the reference's own
formulation (F.interpolate(nearest) -> nn.Conv3d(k=3, padding=1), buildingblocks.py:614,:56) and its autograd."""
import pytest
import torch
import torch.nn.functional as F

import subpixel_oracle as so


@pytest.mark.parametrize("N,C1,K,size", [(2, 3, 4, (2, 3, 4)), (1, 5, 2, (1, 1, 1)), (1, 2, 3, (3, 2, 5))])
def test_subpixel_forward_dgrad_wgrad_equal_the_reference_formulation(N, C1, K, size):
    torch.manual_seed(sum(size) + C1)
    D1, H1, W1 = size
    low = torch.randn(N, C1, D1, H1, W1, dtype=torch.float64)
    w = torch.randn(K, C1, 3, 3, 3, dtype=torch.float64)
    dz = torch.randn(N, K, 2 * D1, 2 * H1, 2 * W1, dtype=torch.float64)
    ll = low.clone().requires_grad_(True)
    wl = w.clone().requires_grad_(True)
    y = F.conv3d(F.interpolate(ll, scale_factor=2, mode="nearest"), wl, None, padding=1)
    y.backward(dz)
    assert torch.allclose(so.forward(low, w), y.detach(), atol=1e-12)
    assert torch.allclose(so.dgrad_low(dz, w), ll.grad, atol=1e-12)
    assert torch.allclose(so.wgrad(low, dz), wl.grad, atol=1e-11)


def test_presummed_kernels_preserve_the_tap_mass():
    """every original tap lands in exactly one (tap half) slot of every parity class: 8 classes x 27 taps in total"""
    w = torch.ones(1, 1, 3, 3, 3, dtype=torch.float64)
    ks = so.presum_weights(w)
    assert len(ks) == 8
    for k in ks.values():
        assert k.sum().item() == 27.0


def test_engine_selects_subpixel_layers_only_for_exact_2x_levels():
    """host logic (no GPU): which decoder first convs take the sub-pixel path depends on the input size — every level
    whose skip is exactly twice the low-res tensor, with channel counts divisible by 4"""
    from pytorch3dunet_amd.engine import UNet3DEngine
    from pytorch3dunet_amd.unet3d.model import UNet3D

    model = UNet3D(in_channels=1, out_channels=1, f_maps=32, layer_order="gcr", num_groups=8, final_sigmoid=True)
    eng = UNet3DEngine(model)
    sub = eng._subpixel_layers((64, 128, 128))
    w = {id(d.basic_module.SingleConv1.conv.weight): d.basic_module.SingleConv1.conv.weight for d in model.decoders}
    assert set(sub) == set(w)
    assert sorted(sub.values()) == [(32, 64), (64, 128), (128, 256)]  # (skip channels, upsampled channels)
    assert sub.plus == frozenset()
    # 20 -> 10 -> 5 -> 2: the deepest level upsamples 2 -> 5 = 2n + 1 (round 5: sub-pixel kernels on a shifted window + the general
    # kernels on the boundary slab), the other two are exact
    sub = eng._subpixel_layers((20, 40, 40))
    assert sorted(sub.values()) == [(32, 64), (64, 128), (128, 256)]
    assert sub.plus == {id(model.decoders[0].basic_module.SingleConv1.conv.weight)}
    # the shipped patch (resources/3DUnet_confocal_boundary/train_config.yml:94): 170 -> 85 -> 42 -> 21, the middle decoder is 42 -> 85
    sub = eng._subpixel_layers((80, 170, 170))
    assert len(sub) == 3 and sub.plus == {id(model.decoders[1].basic_module.SingleConv1.conv.weight)}
    # odd input: every level is n -> 2n + 1 along some axis
    sub = eng._subpixel_layers((9, 13, 11))
    assert len(sub) == 3 and sub.plus == set(w)
    eng.subpixel_plus = False   # U3D_SUBPIXEL_PLUS=0: exact levels only (rounds 1-4)
    assert sorted(eng._subpixel_layers((20, 40, 40)).values()) == [(32, 64), (64, 128)]
    assert eng._subpixel_layers((9, 13, 11)) == {}
    eng.subpixel = False
    assert eng._subpixel_layers((64, 128, 128)) == {}
