"""Generate the golden vectors in tests/golden/*.npz from the LIVE reference (wolny/pytorch-3dunet 1.9.6 imported
from /root/reference through oracle/ref_import.py) — run in the build container only:

    python tests/golden/make_golden.py

Each fixture stores the synthetic input, the full state_dict, the reference's (probs, logits), the loss value and
every parameter gradient after loss.backward() on the CPU path (fp32, torch 2.10.0+rocm7.0 CPU operators).
`big` fixtures store seeds + strided samples instead of full tensors (the state_dict is re-created by seeding;
the generator asserts that our module tree reproduces the reference's seeded initialisation bit-for-bit)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "pytorch-3dunet_amd"))

from ref_import import import_reference  # noqa: E402

CASES = {
    # name: (model config, input shape, loss, full?)
    "g1_unet3d_small": (dict(name="UNet3D", in_channels=1, out_channels=1, f_maps=[8, 16, 32], num_groups=4,
                             final_sigmoid=True), (1, 1, 16, 24, 24), "bce_dice", True),
    "g2_unet3d_multi_odd": (dict(name="UNet3D", in_channels=2, out_channels=3, f_maps=[8, 16], num_groups=2,
                                 final_sigmoid=False), (2, 2, 9, 13, 11), "probs_sum", True),
    "g3_unet3d_regression": (dict(name="UNet3D", in_channels=3, out_channels=2, f_maps=[8, 16, 32], num_groups=8,
                                  is_segmentation=False), (1, 3, 12, 20, 17), "mse", True),
    "g4_unet3d_f16_cfg1": (dict(name="UNet3D", in_channels=1, out_channels=1, f_maps=16, num_groups=8,
                                final_sigmoid=True), (1, 1, 32, 64, 64), "bce_dice", False),
    "g5_resunet3d_small": (dict(name="ResidualUNet3D", in_channels=1, out_channels=1, f_maps=[8, 16, 32], num_groups=4,
                                final_sigmoid=True), (1, 1, 16, 24, 24), "bce_dice", True),
    "g6_resunet3d_multi_odd": (dict(name="ResidualUNet3D", in_channels=2, out_channels=3, f_maps=[8, 16, 24], num_groups=2,
                                    final_sigmoid=False), (2, 2, 9, 13, 11), "probs_sum", True),
    "g7_resunetse3d_small": (dict(name="ResidualUNetSE3D", in_channels=1, out_channels=1, f_maps=[8, 16, 32], num_groups=4,
                                  final_sigmoid=True), (1, 1, 16, 24, 24), "bce_dice", True),
    "g8_resunetse3d_multi_odd": (dict(name="ResidualUNetSE3D", in_channels=3, out_channels=2, f_maps=[8, 16], num_groups=2,
                                      final_sigmoid=False), (2, 3, 9, 13, 11), "probs_sum", True),
    # ---- BASELINE.json configurations at full channel width (round 2): sampled `big` fixtures
    # config 2 exactly: UNet3D f_maps=32, per-GPU batch 2x1x64x128x128, BCEDiceLoss
    "g9_unet3d_f32_cfg2": (dict(name="UNet3D", in_channels=1, out_channels=1, f_maps=32, num_groups=8,
                                final_sigmoid=True), (2, 1, 64, 128, 128), "bce_dice", False),
    # config 4's channel ladder 64..1024 over 5 levels (ResidualUNet3D f_maps=64), reduced spatial size
    "g10_resunet3d_f64_ladder": (dict(name="ResidualUNet3D", in_channels=1, out_channels=1, f_maps=64, num_groups=8,
                                      final_sigmoid=True), (1, 1, 32, 64, 64), "bce_dice", False),
    # config 5's ladder: ResidualUNetSE3D, 3 input channels, f_maps=64, 5 levels, non-cubic patch
    "g11_resunetse3d_in3_ladder": (dict(name="ResidualUNetSE3D", in_channels=3, out_channels=1, f_maps=64, num_groups=8,
                                        final_sigmoid=True), (1, 3, 16, 32, 48), "bce_dice", False),
    # ---- round 4: BASELINE config 4 AT THE SHAPE IT IS BENCHMARKED ON (ResidualUNet3D f_maps=64, 1x1x80x160x160; tools/model_bench.py).
    # 11 TFLOP per step on the host: fp32 only (the float64 twin needs > 60 GB) -> no ref_err / grad64 samples in this fixture
    "g12_resunet3d_f64_cfg4_fullsize": (dict(name="ResidualUNet3D", in_channels=1, out_channels=1, f_maps=64, num_groups=8,
                                             final_sigmoid=True), (1, 1, 80, 160, 160), "bce_dice", False),
    # ---- round 6: the shape the reference SHIPS for training (resources/3DUnet_confocal_boundary/train_config.yml:94, patch 80x170x170,
    # UNet3D f_maps=32): the only shape where the ragged tiles and the n -> 2n + 1 decoder levels (170 -> 85 -> 42 -> 21) all run at full width
    "g13_unet3d_f32_shipped_80x170x170": (dict(name="UNet3D", in_channels=1, out_channels=1, f_maps=32, num_groups=8,
                                               final_sigmoid=True), (1, 1, 80, 170, 170), "bce_dice", False),
}
NO_F64 = {"g12_resunet3d_f64_cfg4_fullsize"}
SAMPLE = 97  # stride of the samples kept for `big` fixtures
SAMPLES = {"g10_resunet3d_f64_ladder": 499, "g11_resunetse3d_in3_ladder": 499, "g12_resunet3d_f64_cfg4_fullsize": 499}  # >100 M parameters: sparser samples


def loss_fn(ref_losses, name, probs, logits, target):
    if name == "bce_dice":
        return ref_losses.BCEDiceLoss()(logits, target)  # losses.py:187-201, the loss of BASELINE config 2
    if name == "mse":
        return torch.nn.functional.mse_loss(logits, target)
    if name == "probs_sum":
        return (probs * target).sum() + 0.5 * (logits * logits).mean()
    raise ValueError(name)


def main():
    import importlib

    ref_model = import_reference()
    ref_losses = importlib.import_module("pytorch3dunet.unet3d.losses")
    from pytorch3dunet_amd.unet3d import model as mine

    torch.set_num_threads(8)
    only = [a.split("=", 1)[1].split(",") for a in sys.argv if a.startswith("--only=")]
    for seed, (name, (cfg, shape, loss_name, full)) in enumerate(CASES.items()):  # seed = position: append only
        if only and name not in only[0]:
            continue
        torch.manual_seed(100 + seed)
        model = ref_model.get_model(dict(cfg))
        torch.manual_seed(100 + seed)
        ours = mine.get_model(dict(cfg))
        same_init = all(torch.equal(a, b) for a, b in zip(model.state_dict().values(), ours.state_dict().values()))
        assert same_init, "seeded initialisation differs from the reference"
        # perturb GroupNorm affine params away from (1,0) so their gradients/paths are exercised
        g = torch.Generator().manual_seed(200 + seed)
        with torch.no_grad():
            for k, p in model.named_parameters():
                if "groupnorm" in k:
                    p.add_(0.2 * torch.randn(p.shape, generator=g))
        model.train()
        x = torch.randn(shape, generator=g)
        n_out = cfg["out_channels"]
        tshape = (shape[0], n_out) + shape[2:]
        target = (torch.rand(tshape, generator=g) > 0.5).float()
        probs, logits = model(x, return_logits=True)
        loss = loss_fn(ref_losses, loss_name, probs, logits, target)
        model.zero_grad()
        loss.backward()
        # the same step in float64 (reference model .double()): how far the reference's OWN fp32 CPU arithmetic is
        # from exact.  Some gradients are cancellation-dominated (e.g. the first GroupNorm's gamma: the next
        # GroupNorm makes the loss almost scale-invariant) and carry >1e-3 relative noise in the reference itself;
        # parity tests use tol = max(1e-3 * absmax, 3 * ref_err) per parameter.
        has64 = name not in NO_F64
        model64, ref_err, l64 = model, {}, None
        if has64:
            model64 = ref_model.get_model(dict(cfg)).double()
            model64.load_state_dict({k: v.double() for k, v in model.state_dict().items()})
            model64.train()
            p64, l64 = model64(x.double(), return_logits=True)
            loss_fn(ref_losses, loss_name, p64, l64, target.double()).backward()
            ref_err = {k: (p.grad.double() - q.grad).abs().max().item()
                       for (k, p), (_, q) in zip(model.named_parameters(), model64.named_parameters())}
        out = {"cfg": np.array(repr(cfg)), "loss_name": np.array(loss_name), "seed": np.array(100 + seed),
               "pert_seed": np.array(200 + seed), "x_shape": np.array(shape), "loss": loss.detach().numpy(),
               "full": np.array(full)}
        if full:
            out["x"] = x.numpy()
            out["target"] = target.numpy()
            out["probs"] = probs.detach().numpy()
            out["logits"] = logits.detach().numpy()
            for k, p in model.named_parameters():
                out["sd/" + k] = p.detach().numpy()
                out["grad/" + k] = p.grad.numpy()
        else:
            S = SAMPLES.get(name, SAMPLE)
            out["sample"] = np.array(S)
            out["probs_s"] = probs.detach().flatten()[::SAMPLE].numpy()
            out["logits_s"] = logits.detach().flatten()[::SAMPLE].numpy()
            out["logits_absmax"] = logits.detach().abs().max().numpy()
            for (k, p), (_, q) in zip(model.named_parameters(), model64.named_parameters()):
                out["grad_s/" + k] = p.grad.flatten()[::S].numpy()
                out["grad_norm/" + k] = p.grad.norm().numpy()
                out["grad_absmax/" + k] = p.grad.abs().max().numpy()
                if name not in ("g4_unet3d_f16_cfg1",) and has64:  # round-2 fixtures also carry the float64 reference's samples
                    out["grad64_s/" + k] = q.grad.flatten()[::S].numpy()
            # global relative L2 distance of the reference's fp32 gradient from its own float64 gradient
            if has64:
                num = sum((p.grad.double() - q.grad).pow(2).sum().item()
                          for (_, p), (_, q) in zip(model.named_parameters(), model64.named_parameters()))
                den = sum(q.grad.pow(2).sum().item() for _, q in model64.named_parameters())
                out["ref_grad_rel_l2"] = np.array((num / den) ** 0.5)
        for k, v in ref_err.items():
            out["ref_err/" + k] = np.array(v)
        if has64:
            out["logits_ref_err"] = np.array((logits.detach().double() - l64.detach()).abs().max().item())
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{name}: loss={loss.item():.6f} -> {os.path.getsize(path) / 1024:.0f} KiB")


def make_loss_golden():
    """l1_losses.npz: values and dlogits of the reference's BCEDiceLoss / DiceLoss / BCEWithLogitsLoss
    (pytorch3dunet/unet3d/losses.py) on small seeded (logits, target) pairs, incl. the clamp(min=eps) corner."""
    import importlib

    import_reference()
    L = importlib.import_module("pytorch3dunet.unet3d.losses")
    g = torch.Generator().manual_seed(4242)
    out = {}
    cases = {
        "a": (2.0 * torch.randn((2, 3, 5, 6, 7), generator=g), None),
        "b": (3.0 * torch.randn((1, 1, 9, 13, 11), generator=g), None),
        "c": (torch.randn((3, 2, 4, 4, 8), generator=g), None),
        # channel 1 has an empty target and p ~ 0: sum(p^2)+sum(t^2) < 1e-6 -> the clamped branch of the Dice
        "z": (torch.cat([torch.randn((1, 1, 4, 4, 4), generator=g), torch.full((1, 1, 4, 4, 4), -12.0)], dim=1), "zero1"),
    }
    crits = {
        "bcedice": lambda: L.BCEDiceLoss(),
        "bcedice_a05": lambda: L.BCEDiceLoss(alpha=0.5),
        "dice": lambda: L.DiceLoss(),
        "bce": lambda: torch.nn.BCEWithLogitsLoss(),
    }
    for cname, (logits, special) in cases.items():
        target = (torch.rand(logits.shape, generator=g) > 0.5).float()
        if special == "zero1":
            target[:, 1] = 0.0
        out[f"{cname}/logits"] = logits.numpy()
        out[f"{cname}/target"] = target.numpy()
        items = dict(crits)
        if logits.shape[1] == 3:
            items["dice_w"] = lambda: L.DiceLoss(weight=torch.tensor([0.2, 0.3, 0.5]))
        for lname, mk in items.items():
            x = logits.clone().requires_grad_(True)
            val = mk()(x, target)
            (1.7 * val).backward()  # non-unit upstream gradient
            out[f"{cname}/{lname}/loss"] = val.detach().numpy()
            out[f"{cname}/{lname}/dlogits"] = x.grad.numpy()
    path = os.path.join(HERE, "l1_losses.npz")
    np.savez_compressed(path, **out)
    print(f"l1_losses: {len(out)} arrays -> {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    if "--losses-only" not in sys.argv:
        main()
    if not any(a.startswith("--only=") for a in sys.argv):
        make_loss_golden()
