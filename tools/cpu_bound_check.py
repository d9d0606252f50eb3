"""Is the training step launch-bound?  Times the host side of the bench step (loop without a final sync) against the
device side (with sync), with and without the HIP-event profiler."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-3dunet_amd"))
import torch  # noqa: E402

from pytorch3dunet_amd import _native as nat  # noqa: E402
from pytorch3dunet_amd.unet3d.losses import BCEDiceLoss  # noqa: E402
from pytorch3dunet_amd.unet3d.model import get_model  # noqa: E402

dev = torch.device("cuda", 0)
model = get_model({"name": "UNet3D", "in_channels": 1, "out_channels": 1, "f_maps": 32, "layer_order": "gcr", "num_groups": 8,
                   "final_sigmoid": True, "is_segmentation": True}).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-4)
x = torch.randn(2, 1, 64, 128, 128, device=dev)
tgt = (torch.rand(2, 1, 64, 128, 128, device=dev) > 0.5).float()
crit = BCEDiceLoss()


def step():
    probs, logits = model(x, return_logits=True)
    loss = crit(logits, tgt)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
for label, prof in (("no profiler", False), ("HIP-event profiler", True)):
    nat.profiler = nat.EventProfiler() if prof else None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    nat.profiler = None
    print(f"{label}: host {1e2 * (t1 - t0):.2f} ms/step, host+device {1e2 * (t2 - t0):.2f} ms/step")
