"""pick the thread count for bench.py's cpu_baseline leg: time the CPU oracle on a small patch at several counts"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "pytorch-3dunet_amd"))
import torch
import unet3d_oracle as orc
from pytorch3dunet_amd.unet3d.model import UNet3D
torch.manual_seed(0)
m = UNet3D(1, 1, f_maps=32)
sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
x = torch.randn(1, 1, 32, 64, 64); t = (torch.rand(1, 1, 32, 64, 64) > 0.5).float()
print("cpu_count", os.cpu_count())
for n in [8, 16, 32, 64, 128, 256]:
    if n > (os.cpu_count() or 1): break
    torch.set_num_threads(n)
    orc.forward_backward(sd, x, t, 8)
    t0 = time.perf_counter(); orc.forward_backward(sd, x, t, 8); dt = time.perf_counter() - t0
    print(f"threads {n}: {dt:.3f} s / iter (1x1x32x64x64)", flush=True)
