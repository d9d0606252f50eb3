import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "pytorch-3dunet_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with `-m gpu` through gpurun")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def diag(**kw):
    """append one JSON line of observed error figures to gpurun_out/parity_diag.jsonl (evidence for the tolerances
    written in the tests; gpurun merges gpurun_out/ back)"""
    import json

    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "parity_diag.jsonl"), "a") as fh:
            fh.write(json.dumps(kw) + "\n")
    except OSError:
        pass


class Golden:
    """One fixture written by tests/golden/make_golden.py from the live reference."""

    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
        self.name = name
        self.cfg = eval(str(z["cfg"]))  # noqa: S307 - a dict literal we wrote ourselves
        self.loss_name = str(z["loss_name"])
        self.full = bool(z["full"])
        self.seed = int(z["seed"])
        self.pert_seed = int(z["pert_seed"])
        self.x_shape = tuple(int(v) for v in z["x_shape"])
        self.loss = float(z["loss"])
        self.sample = int(z["sample"]) if "sample" in z.files else 97  # stride of the gradient samples of `big` fixtures
        self.z = z

    def tensor(self, key):
        return torch.from_numpy(np.array(self.z[key]))

    def group(self, prefix):
        return {k[len(prefix):]: torch.from_numpy(np.array(self.z[k])) for k in self.z.files if k.startswith(prefix)}

    def inputs(self):
        """(x, target): stored for `full` fixtures, regenerated from the recorded seeds otherwise."""
        if self.full:
            return self.tensor("x"), self.tensor("target")
        # must replay make_golden.py's generator draws: first the GroupNorm perturbations, then x, then target
        model = self.build_model(perturb=False)
        g = torch.Generator().manual_seed(self.pert_seed)
        for k, p in model.named_parameters():
            if "groupnorm" in k:
                torch.randn(p.shape, generator=g)
        x = torch.randn(self.x_shape, generator=g)
        tshape = (self.x_shape[0], self.cfg["out_channels"]) + self.x_shape[2:]
        target = (torch.rand(tshape, generator=g) > 0.5).float()
        return x, target

    def build_model(self, perturb=True):
        """our module tree with the fixture's parameters (full: stored state_dict; big: seeded init)."""
        from pytorch3dunet_amd.unet3d.model import get_model

        torch.manual_seed(self.seed)
        model = get_model(dict(self.cfg))
        if self.full:
            model.load_state_dict(self.group("sd/"), strict=True)
        elif perturb:
            g = torch.Generator().manual_seed(self.pert_seed)
            with torch.no_grad():
                for k, p in model.named_parameters():
                    if "groupnorm" in k:
                        p.add_(0.2 * torch.randn(p.shape, generator=g))
        return model


GOLDEN_NAMES = ["g1_unet3d_small", "g2_unet3d_multi_odd", "g3_unet3d_regression", "g4_unet3d_f16_cfg1",
                "g5_resunet3d_small", "g6_resunet3d_multi_odd", "g7_resunetse3d_small", "g8_resunetse3d_multi_odd",
                # BASELINE.json configurations at full channel width (sampled fixtures): config 2 itself, config 4's and 5's ladders
                "g9_unet3d_f32_cfg2", "g10_resunet3d_f64_ladder", "g11_resunetse3d_in3_ladder",
                # the patch shape the reference SHIPS for training (80x170x170, resources/3DUnet_confocal_boundary/train_config.yml:94):
                # ragged tiles + n -> 2n + 1 decoder levels at full width
                "g13_unet3d_f32_shipped_80x170x170"]


@pytest.fixture(params=GOLDEN_NAMES)
def golden(request):
    return Golden(request.param)


def loss_by_name(name, probs, logits, target):
    from unet3d_oracle import bce_dice_loss

    if name == "bce_dice":
        return bce_dice_loss(logits, target)
    if name == "mse":
        return torch.nn.functional.mse_loss(logits, target)
    if name == "probs_sum":
        return (probs * target).sum() + 0.5 * (logits * logits).mean()
    raise ValueError(name)
