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
