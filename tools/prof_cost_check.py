import os, sys, time
sys.path.insert(0, "pytorch-3dunet_amd")
import torch
from pytorch3dunet_amd import _native as nat
from pytorch3dunet_amd.unet3d.losses import BCEDiceLoss
from pytorch3dunet_amd.unet3d.model import UNet3D
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = UNet3D(in_channels=1, out_channels=1, f_maps=32, num_groups=8).to(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=2e-4, weight_decay=1e-5)
x = torch.randn(2, 1, 64, 128, 128, device=dev); t = (torch.rand_like(x) > 0.5).float()
crit = BCEDiceLoss()
def step():
    p, l = model(x, return_logits=True); loss = crit(l, t); opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
for mode in ("off", "dominant", "off", "all_flops"):
    nat.profiler = None if mode == "off" else nat.EventProfiler(flops_only=True, only={"u3d_conv3d", "u3d_conv3d_ex"} if mode == "dominant" else None)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    n = len(nat.profiler.records) if nat.profiler else 0
    nat.profiler = None
    print(f"{mode}: host {1e3*(t1-t0)/10:.2f} ms/step, step {1e3*(t2-t0)/10:.2f} ms, events pairs/step {n/10:.0f}", flush=True)
