"""Isolated timing of the first-layer kernels (csrc/u3d_smallc.hip) and a few bandwidth kernels at the bench workload's shapes.
    python tools/small_bench.py
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("pytorch-3dunet_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch  # noqa: E402

from pytorch3dunet_amd import _native as nat  # noqa: E402
from pytorch3dunet_amd.engine import _p, _stream  # noqa: E402

dev = torch.device("cuda", 0)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    N, D, H, W, Cin, Cout = 2, 64, 128, 128, 1, 16
    x = torch.randn(N, D, H, W, Cin, device=dev)
    aff = torch.randn(N, Cin, 2, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, 3, device=dev)
    y = torch.empty(N, D, H, W, Cout, device=dev)
    st = torch.zeros(N, Cout, 2, dtype=torch.float64, device=dev)
    dz = torch.randn(N, D, H, W, Cout, device=dev)
    dw = torch.empty_like(w)
    gst = torch.zeros(N, Cin, 2, dtype=torch.float64, device=dev)
    lib = nat.get_lib()
    for key, vals in ((19, (0, 512, 1024, 4096, 8192)), (20, (0, 512, 2048, 4096))):
        for v in vals:
            nat.call("u3d_set_tuning", key, v)
            if key == 19:
                for stats in (True, False):
                    us = timeit(lambda: nat.call("u3d_conv3d_small_cin_fwd", 0, _stream(dev), _p(x), _p(aff), _p(w), _p(y), N, D, H, W, Cin, Cout, 1,
                                                 _p(st) if stats else None))
                    print(f"small_fwd  blocks={v or 2048:5d} stats={int(stats)}: {us:7.1f} us  ({(y.numel() + x.numel()) * 4 / us / 1e6:.2f} TB/s)")
            else:
                need = lib.u3d_small_cin_bwd_workspace_floats(N, D, H, W, Cin, Cout)
                ws = torch.empty(need, device=dev)
                us = timeit(lambda: nat.call("u3d_conv3d_small_cin_bwd", 0, _stream(dev), _p(x), _p(aff), _p(dz), _p(w), _p(dw), _p(gst), N, D, H, W,
                                             Cin, Cout, _p(ws), need))
                print(f"small_bwd  blocks={v or 1024:5d}: {us:7.1f} us  ({(dz.numel() + x.numel()) * 4 / us / 1e6:.2f} TB/s)")
        nat.call("u3d_set_tuning", key, 0)


if __name__ == "__main__":
    main()
