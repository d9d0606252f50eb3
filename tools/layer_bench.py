"""Per-layer timing of the three conv kernels (fwd / dgrad / wgrad) on the 14 SingleConv shapes of the bench workload
(UNet3D f_maps=32, per-GPU batch 2x1x64x128x128), each launched in isolation through the C-ABI with HIP events.

    python tools/layer_bench.py [--batch 2] [--iters 5] [--only fwd,dgrad,wgrad]
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("pytorch-3dunet_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch  # noqa: E402

from pytorch3dunet_amd import _native as nat  # noqa: E402
from pytorch3dunet_amd.engine import VSrc, _p, _stream  # noqa: E402
import gpu_utils as U  # noqa: E402

dev = U.DEV

# (name, C0, C1 (virtual upsampled half), Cout, level)
LAYERS = [
    ("enc0.c2", 16, 0, 32, 0), ("enc1.c1", 32, 0, 32, 1), ("enc1.c2", 32, 0, 64, 1), ("enc2.c1", 64, 0, 64, 2),
    ("enc2.c2", 64, 0, 128, 2), ("enc3.c1", 128, 0, 128, 3), ("enc3.c2", 128, 0, 256, 3),
    ("dec0.c1", 128, 256, 128, 2), ("dec0.c2", 128, 0, 128, 2), ("dec1.c1", 64, 128, 64, 1), ("dec1.c2", 64, 0, 64, 1),
    ("dec2.c1", 32, 64, 32, 0), ("dec2.c2", 32, 0, 32, 0),
]


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--only", default="fwd,dgrad,wgrad")
    ap.add_argument("--patch", default="64,128,128")
    ap.add_argument("--layers", default="", help="comma-separated layer names (default: all)")
    ap.add_argument("--no-stats", action="store_true", help="forward without the output statistics (what the f64 atomics of the epilogue cost)")
    args = ap.parse_args()
    only = args.only.split(",")
    if os.environ.get("U3D_LAYERS"):  # custom shapes: name:C0:C1:Cout:level;...
        LAYERS[:] = [(a, int(b), int(c), int(d), int(e)) for a, b, c, d, e in (x.split(":") for x in os.environ["U3D_LAYERS"].split(";"))]
    N = args.batch
    D0, H0, W0 = (int(v) for v in args.patch.split(","))
    tot = {k: [0.0, 0.0] for k in only}
    lib = nat.get_lib()
    for kv in os.environ.get("U3D_TUNE", "").split(","):  # e.g. U3D_TUNE=5:24 (start-phase stagger of 24k cycles)
        if ":" in kv:
            nat.call("u3d_set_tuning", int(kv.split(":")[0]), int(kv.split(":")[1]))
    # warm the clocks
    a = torch.randn(4096, 4096, device=dev)
    for _ in range(20):
        a @ a
    torch.cuda.synchronize()
    for name, C0, C1, Cout, lvl in LAYERS:
        if args.layers and name not in args.layers.split(","):
            continue
        D, H, W = D0 >> lvl, H0 >> lvl, W0 >> lvl
        Cin = C0 + C1
        t0 = torch.randn(N, D, H, W, C0, device=dev)
        t1 = torch.randn(N, D // 2, H // 2, W // 2, C1, device=dev) if C1 else None
        src = VSrc(t0, t1)
        aff = torch.randn(N, Cin, 2, device=dev)
        w = torch.randn(Cout, Cin, 3, 3, 3, device=dev) / (27 * Cin) ** 0.5
        flops = 54.0 * Cin * Cout * N * D * H * W
        kind = 0 if t1 is None else (1 if (D, H, W) == (2 * (D // 2), 2 * (H // 2), 2 * (W // 2)) else 2)
        kn_ = lib.u3d_conv3d_workspace_floats(N, D, H, W, Cin, Cout)
        var = (lib.u3d_conv3d_variant(N, D, H, W, Cin, Cout, kind, 1 if kn_ else 0), lib.u3d_conv3d_variant(N, D, H, W, Cout, Cin, 0, 1),
               lib.u3d_conv3d_wgrad_variant(N, D, H, W, Cin, Cout, kind))
        line = f"{name:8s} {Cin:3d}->{Cout:3d} @{D}x{H}x{W} variants {var[0]}/{var[1]}/{var[2]}: "
        if "fwd" in only:
            wp = U.pack(w, 0)
            y = torch.empty((N, D, H, W, Cout), device=dev)
            st = torch.zeros((N, Cout, 2), dtype=torch.float64, device=dev)
            s = src.struct(aff)
            kn = lib.u3d_conv3d_workspace_floats(N, D, H, W, Cin, Cout)
            kws = torch.empty(kn, device=dev) if kn else None
            ms = timeit(lambda: nat.call("u3d_conv3d_ex", 0, _stream(dev), ctypes.byref(s), _p(wp), _p(y), N, D, H, W, Cout, 1,
                                         None if args.no_stats else _p(st), None, None, None, _p(kws), kn), args.iters)
            line += f"fwd {ms:7.3f} ms {flops / ms / 1e9:6.1f} TF | "
            tot["fwd"][0] += ms
            tot["fwd"][1] += flops
        dz = torch.randn(N, D, H, W, Cout, device=dev)
        if "dgrad" in only:
            wpd = U.pack(w, 1)
            dg = torch.empty((N, D, H, W, Cin), device=dev)
            gst = torch.zeros((N, Cin, 2), dtype=torch.float64, device=dev)
            s_dz = VSrc(dz).struct()
            s_x = src.struct()
            kn = lib.u3d_conv3d_workspace_floats(N, D, H, W, Cout, Cin)
            kws = torch.empty(kn, device=dev) if kn else None
            ms = timeit(lambda: nat.call("u3d_conv3d_ex", 0, _stream(dev), ctypes.byref(s_dz), _p(wpd), _p(dg), N, D, H, W, Cin, 0,
                                         None, ctypes.byref(s_x), _p(gst), None, _p(kws), kn), args.iters)
            line += f"dgrad {ms:7.3f} ms {flops / ms / 1e9:6.1f} TF | "
            tot["dgrad"][0] += ms
            tot["dgrad"][1] += flops
        if "wgrad" in only:
            nws = lib.u3d_wgrad_workspace_floats(N, D, H, W, Cin, Cout)
            ws = torch.empty(nws, device=dev)
            dw = torch.empty((Cout, Cin, 3, 3, 3), device=dev)
            s = src.struct(aff)
            ms = timeit(lambda: nat.call("u3d_conv3d_wgrad", 0, _stream(dev), ctypes.byref(s), _p(dz), _p(dw), N, D, H, W, Cout,
                                         _p(ws), nws), args.iters)
            line += f"wgrad {ms:7.3f} ms {flops / ms / 1e9:6.1f} TF"
            tot["wgrad"][0] += ms
            tot["wgrad"][1] += flops
        print(line, flush=True)
    allms = 0.0
    allfl = 0.0
    for k, (ms, fl) in tot.items():
        print(f"total {k}: {ms:.3f} ms, {fl / ms / 1e9:.1f} TF")
        allms += ms
        allfl += fl
    print(f"total: {allms:.3f} ms/step (batch {N}), {allfl / allms / 1e9:.1f} TF, conv-only bound {N / allms * 1e3:.1f} patches/s")


if __name__ == "__main__":
    main()
