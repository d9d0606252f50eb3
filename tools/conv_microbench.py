"""time u3d_conv3d (and wgrad) on single layer shapes, optionally under the timing-only ablation masks"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("pytorch-3dunet_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
from pytorch3dunet_amd import _native as nat
from pytorch3dunet_amd.engine import VSrc, _p, _stream
import gpu_utils as U

dev = U.DEV

def time_conv(N, Cin, Cout, D, H, W, abl=0, iters=5, wgrad=False):
    x = torch.randn(N, D, H, W, Cin, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, 3, device=dev) / (27 * Cin) ** 0.5
    aff = torch.randn(N, Cin, 2, device=dev)
    src = VSrc(x)
    flops = 54.0 * Cin * Cout * N * D * H * W
    nat.call("u3d_set_tuning", 1, abl)
    if wgrad:
        dz = torch.randn(N, D, H, W, Cout, device=dev)
        fn = lambda: U.wgrad(src, dz, Cout, affine=aff)
    else:
        wp = U.pack(w, 0)
        y = torch.empty((N, D, H, W, Cout), device=dev)
        st = torch.zeros((N, Cout, 2), dtype=torch.float64, device=dev)
        s = src.struct(aff)
        fn = lambda: nat.call("u3d_conv3d", 0, _stream(dev), ctypes.byref(s), _p(wp), _p(y), N, D, H, W, Cout, 1, _p(st), None, None)
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    nat.call("u3d_set_tuning", 1, 0)
    return ms, flops / ms / 1e9

if __name__ == "__main__":
    shapes = [(1, 96, 32, 64, 128, 128), (1, 32, 32, 64, 128, 128), (1, 16, 32, 64, 128, 128)]
    for sh in shapes:
        for abl in (0, 1, 2, 4, 8, 7, 15, 0):
            ms, tf = time_conv(*sh, abl=abl)
            print(f"conv {sh} abl={abl:2d}: {ms:8.3f} ms  {tf:7.1f} TF", flush=True)
    for sh in shapes[:2]:
        ms, tf = time_conv(*sh, wgrad=True)
        print(f"wgrad {sh}: {ms:8.3f} ms  {tf:7.1f} TF", flush=True)
