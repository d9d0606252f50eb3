"""fp32 vs fp32_split on one model / input: per-parameter gradient differences (largest first) and the logits difference."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "pytorch-3dunet_amd"))
from pytorch3dunet_amd.unet3d import model as M  # noqa: E402
from pytorch3dunet_amd.unet3d.losses import BCEDiceLoss  # noqa: E402

name, shape = sys.argv[1], tuple(int(v) for v in sys.argv[2].split(","))
kw = dict(in_channels=shape[1], out_channels=1, f_maps=int(sys.argv[3]), num_groups=8, final_sigmoid=True)
dev = torch.device("cuda", 0)
x = torch.randn(shape, generator=torch.Generator().manual_seed(0)).to(dev)
t = (torch.rand((shape[0], 1) + shape[2:], generator=torch.Generator().manual_seed(1)) > 0.5).float().to(dev)
res = {}
for mode in ("fp32", "fp32_split"):
    torch.manual_seed(0)
    m = getattr(M, name)(compute_dtype=mode, **kw).to(dev).train()
    p, l = m(x, return_logits=True)
    BCEDiceLoss()(l, t).backward()
    res[mode] = (l.detach().double(), {k: v.grad.double().clone() for k, v in m.named_parameters()})
la, ga = res["fp32"]
lb, gb = res["fp32_split"]
print("logits rel l2", float((la - lb).norm() / la.norm()))
rows = sorted(((float((ga[k] - gb[k]).norm() / ga[k].norm().clamp_min(1e-30)), k, float(ga[k].norm())) for k in ga), reverse=True)
for r in rows[:12]:
    print(f"{r[0]:.3e}  {r[1]}  |g|={r[2]:.3e}")
tot = torch.cat([(ga[k] - gb[k]).flatten() for k in ga]).norm() / torch.cat([ga[k].flatten() for k in ga]).norm()
print("global grad rel l2", float(tot))
