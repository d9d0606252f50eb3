"""pytorch3dunet_amd.optim.FusedAdam: torch.optim.Adam's update (reference create_optimizer, utils.py:246-316; trainer.py:246) — on CPU
tensors through torch's own functional form, on a HIP device as ONE kernel launch (csrc/u3d_optim.hip)."""
import pytest
import torch

from conftest import ROOT  # noqa: F401  (sets sys.path)


def _models(dev):
    torch.manual_seed(0)
    a = torch.nn.Sequential(torch.nn.Conv3d(1, 5, 3), torch.nn.GroupNorm(1, 5), torch.nn.Conv3d(5, 3, 1)).to(dev)
    import copy

    return a, copy.deepcopy(a)


def _run(dev, steps=5, wd=1e-5, tol=0.0):
    from pytorch3dunet_amd.optim import FusedAdam, as_fused

    a, b = _models(dev)
    oa = torch.optim.Adam(a.parameters(), lr=2e-3, betas=(0.9, 0.999), weight_decay=wd)
    ob = FusedAdam(b.parameters(), lr=2e-3, betas=(0.9, 0.999), weight_decay=wd)
    g = torch.Generator().manual_seed(1)
    for it in range(steps):
        x = torch.randn(2, 1, 6, 6, 6, generator=g).to(dev)
        for m, o in ((a, oa), (b, ob)):
            o.zero_grad(set_to_none=True)
            m(x).square().mean().backward()
            o.step()
        if it == 2:
            # state_dicts are interchangeable with torch.optim.Adam's (the reference checkpoints `optimizer.state_dict()`, trainer.py:385-403)
            sd = ob.state_dict()
            ob = FusedAdam(b.parameters(), lr=1.0)
            ob.load_state_dict(sd)
            oa2 = torch.optim.Adam(a.parameters(), lr=1.0)
            oa2.load_state_dict(oa.state_dict())
            oa = oa2
    for p, q in zip(a.parameters(), b.parameters()):
        if tol == 0.0:
            assert torch.equal(p, q)
        else:
            assert (p - q).abs().max().item() <= tol * p.abs().max().item(), (p - q).abs().max().item()
    # as_fused: an existing torch.optim.Adam (what create_optimizer builds) continues as a FusedAdam with the same state
    of = as_fused(oa)
    assert type(of).__name__ == "FusedAdam" and len(of.state) == len(oa.state)
    assert of.param_groups[0]["lr"] == oa.param_groups[0]["lr"] and of.param_groups[0]["weight_decay"] == oa.param_groups[0]["weight_decay"]
    assert as_fused(torch.optim.SGD(a.parameters(), lr=0.1)).__class__ is torch.optim.SGD
    return a, b


def test_fused_adam_on_cpu_tensors_is_torch_adam():
    _run(torch.device("cpu"), tol=0.0)
    with pytest.raises(ValueError):
        from pytorch3dunet_amd.optim import FusedAdam

        FusedAdam([torch.nn.Parameter(torch.zeros(3))], amsgrad=True)


@pytest.mark.gpu
@pytest.mark.parametrize("wd", [0.0, 1e-5])
def test_fused_adam_kernel_against_torch_adam(wd):
    """one launch for all parameters (odd sizes: 135-element weights, 5-element GroupNorm vectors -> scalar tails, unaligned views)"""
    from pytorch3dunet_amd import _native as nat

    n0 = nat.launch_count
    _run(torch.device("cuda", 0), steps=6, wd=wd, tol=2e-6)
    assert nat.launch_count - n0 == 6  # one u3d_adam_step per optimizer step


@pytest.mark.gpu
def test_fused_adam_on_the_native_models_gradient_views():
    """UNet3D on the native executor: every `.grad` is a view into ONE flat buffer (odd offsets: unaligned views, 1-element parameters).
    Three steps, torch.optim.Adam and FusedAdam side by side from the SAME parameters each step (the second model is re-synchronised
    after every comparison: two free-running trajectories cannot be compared — Adam turns a 1-ulp parameter difference, through one
    flipped ReLU and a 0.3 % change of a small gradient, into a full-size update of that element; measured 1.2e-4 after two steps)."""
    import copy

    from pytorch3dunet_amd.optim import FusedAdam
    from pytorch3dunet_amd.unet3d.losses import BCEDiceLoss
    from pytorch3dunet_amd.unet3d.model import UNet3D

    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    base = UNet3D(1, 1, f_maps=16, num_groups=8)
    with torch.no_grad():
        for k, p in base.named_parameters():
            if "groupnorm" in k:
                p.add_(0.2 * torch.randn_like(p))
    x = torch.randn(2, 1, 16, 32, 32, device=dev)
    t = (torch.rand(2, 1, 16, 32, 32, device=dev) > 0.5).float()
    ma, mb = copy.deepcopy(base).to(dev).train(), copy.deepcopy(base).to(dev).train()
    oa = torch.optim.Adam(ma.parameters(), lr=2e-4, weight_decay=1e-5)
    ob = FusedAdam(mb.parameters(), lr=2e-4, weight_decay=1e-5)
    crit = BCEDiceLoss()
    for it in range(3):
        for m, o in ((ma, oa), (mb, ob)):
            _, lg = m(x, return_logits=True)
            loss = crit(lg, t)
            o.zero_grad(set_to_none=True)
            loss.backward()
            o.step()
        for (k, p), q in zip(ma.named_parameters(), mb.parameters()):
            assert (p - q).abs().max().item() <= 3e-6 * max(p.abs().max().item(), 1e-2), (it, k)
            sa, sb = oa.state[p], ob.state[q]
            assert float(sa["step"]) == float(sb["step"]) == it + 1
            for key in ("exp_avg", "exp_avg_sq"):
                assert (sa[key] - sb[key]).abs().max().item() <= 3e-6 * max(sa[key].abs().max().item(), 1e-30), (it, k, key)
        with torch.no_grad():  # same starting point for the next step (states differ by round-off only)
            for p, q in zip(ma.parameters(), mb.parameters()):
                q.copy_(p)
                for key in ("exp_avg", "exp_avg_sq"):
                    ob.state[q][key].copy_(oa.state[p][key])
