#!/usr/bin/env python
"""Times the segmentation head's forward / backward kernels on the headline shape (1 x 64 x 128 x 128 voxels, 32 -> 1 channels):
    python tools/head_bench.py [--cin 32] [--cout 1] [--tune 22:512]"""
import argparse
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pytorch-3dunet_amd"))
import torch
from pytorch3dunet_amd import _native as nat
from pytorch3dunet_amd.engine import _p, _stream

ap = argparse.ArgumentParser()
ap.add_argument("--cin", type=int, default=32)
ap.add_argument("--cout", type=int, default=1)
ap.add_argument("--size", type=int, nargs=3, default=[64, 128, 128])
ap.add_argument("--tune", default="")
ap.add_argument("--no-acc", action="store_true")
ap.add_argument("--copy", action="store_true", help="time dx.copy_(x) instead (what a plain read + write stream of the same bytes takes)")
ap.add_argument("--flush", action="store_true", help="sweep 1 GB between the timed launches (nothing of x / dx left in L2 / Infinity Cache)")
ap.add_argument("--fresh", action="store_true", help="x written by a kernel right before the timed launch")
a = ap.parse_args()
for kv in a.tune.split(","):
    if kv:
        nat.call("u3d_set_tuning", int(kv.split(":")[0]), int(kv.split(":")[1]))
dev = torch.device("cuda:0")
V = a.size[0] * a.size[1] * a.size[2]
x = torch.randn(1, V, a.cin, device=dev)
w = torch.randn(a.cout, a.cin, device=dev)
dl = torch.randn(1, a.cout, V, device=dev)
dx = torch.empty_like(x)
acc = torch.zeros(a.cout * a.cin + a.cout, dtype=torch.float64, device=dev)
def run():
    if a.copy:
        dx.copy_(x)
        return
    nat.call("u3d_conv1x1_head_bwd", 0, _stream(dev), _p(dl), _p(x), _p(w), 1, V, a.cin, a.cout, 1, _p(dx), None if a.no_acc else _p(acc))
for _ in range(5):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
if a.flush or a.fresh:
    big = torch.zeros(256 * 1024 * 1024, device=dev)
    tot = 0.0
    for _ in range(20):
        if a.flush:
            big.add_(1.0)
        if a.fresh:
            x.mul_(1.0)
        e0.record()
        run()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    us = tot * 1000 / 20
else:
    e0.record()
    for _ in range(50):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 50
print(f"{'copy' if a.copy else 'head_bwd'} cin={a.cin} cout={a.cout} tune={a.tune!r} no_acc={a.no_acc} flush={a.flush} fresh={a.fresh}: {us:.1f} us  ({(2 * x.numel() * 4 + dl.numel() * 4) / us / 1e6:.2f} TB/s)")
