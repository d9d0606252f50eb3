"""Decoder first-conv forward, two ways: the one-kernel virtual-concat convolution (u3d_conv3d_ex) against the split
skip-half convolution + sub-pixel convolution of the upsampled half (u3d_subpixel_conv_fwd + residual epilogue).

    python tools/subpix_bench.py [--batch 2] [--iters 10]
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
from layer_bench import timeit  # noqa: E402

dev = U.DEV
LAYERS = [("dec0.c1", 128, 256, 128, 2), ("dec1.c1", 64, 128, 64, 1), ("dec2.c1", 32, 64, 32, 0)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    N = args.batch
    lib = nat.get_lib()
    a = torch.randn(4096, 4096, device=dev)
    for _ in range(20):
        a @ a
    torch.cuda.synchronize()
    for name, C0, C1, Cout, lvl in LAYERS:
        D, H, W = 64 >> lvl, 128 >> lvl, 128 >> lvl
        D1, H1, W1 = D // 2, H // 2, W // 2
        Cin = C0 + C1
        t0 = torch.randn(N, D, H, W, C0, device=dev)
        t1 = torch.randn(N, D1, H1, W1, C1, device=dev)
        aff = torch.randn(N, Cin, 2, device=dev)
        w = torch.randn(Cout, Cin, 3, 3, 3, device=dev) / (27 * Cin) ** 0.5
        y = torch.empty((N, D, H, W, Cout), device=dev)
        st = torch.zeros((N, Cout, 2), dtype=torch.float64, device=dev)
        flops = 54.0 * Cin * Cout * N * D * H * W
        # (a) one kernel over the virtual concat
        src = VSrc(t0, t1)
        s = src.struct(aff)
        wp = U.pack(w, 0)
        ms_a = timeit(lambda: nat.call("u3d_conv3d_ex", 0, _stream(dev), ctypes.byref(s), _p(wp), _p(y), N, D, H, W, Cout, 1,
                                       _p(st), None, None, None, None, 0), args.iters)
        ya = y.clone()
        # (b) sub-pixel convolution of the upsampled half + skip half with the residual epilogue
        pk = torch.empty(lib.u3d_subpixel_packed_floats(C1, Cout), device=dev)
        nat.call("u3d_pack_subpixel_weights", 0, _stream(dev), _p(w), Cout, Cin, C0, C1, _p(pk))
        part = torch.empty((N, D, H, W, Cout), device=dev)
        aff_sub = aff.view(-1)[2 * C0:]
        kn = lib.u3d_subpixel_fwd_workspace_floats(N, D1, H1, W1, C1, Cout)
        kws = torch.empty(kn, device=dev) if kn else None
        ms_b1 = timeit(lambda: nat.call("u3d_subpixel_conv_fwd", 0, _stream(dev), _p(t1), _p(aff_sub), Cin * 2, _p(pk), _p(part),
                                        N, D1, H1, W1, C1, Cout, _p(kws), kn), args.iters)
        w0 = U.pack(w[:, :C0].contiguous(), 0)
        a0 = aff[:, :C0].contiguous()
        s0 = VSrc(t0).struct(a0)
        ms_b2 = timeit(lambda: nat.call("u3d_conv3d_ex", 0, _stream(dev), ctypes.byref(s0), _p(w0), _p(y), N, D, H, W, Cout, 1,
                                        _p(st), None, None, _p(part), None, 0), args.iters)
        # (c) data gradient of the upsampled half at low resolution vs the full virtual data gradient
        dz = torch.randn(N, D, H, W, Cout, device=dev)
        pkd = torch.empty(lib.u3d_subpixel_dgrad_packed_floats(Cout, C1), device=dev)
        nat.call("u3d_pack_subpixel_dgrad_weights", 0, _stream(dev), _p(w), Cout, Cin, C0, C1, _p(pkd))
        dlow = torch.empty((N, D1, H1, W1, C1), device=dev)
        gst = torch.zeros((N, C1, 2), dtype=torch.float64, device=dev)
        ms_c = timeit(lambda: nat.call("u3d_subpixel_conv_dgrad", 0, _stream(dev), _p(dz), _p(pkd), _p(t1), _p(dlow), _p(gst), N, D1,
                                       H1, W1, C1, Cout), args.iters)
        f_dg = 2.0 * 64 * C1 * Cout * N * D1 * H1 * W1
        print(f"{name} dgrad(low) {ms_c:.3f} ms ({f_dg / ms_c / 1e9:.1f} TF executed)", flush=True)
        # (d) weight gradient: full virtual source vs skip slice + sub-pixel slice
        nws = max(lib.u3d_wgrad_workspace_floats(N, D, H, W, Cin, Cout), lib.u3d_subpixel_wgrad_workspace_floats(N, D1, H1, W1, C1, Cout))
        ws = torch.empty(nws, device=dev)
        dw = torch.empty((Cout, Cin, 3, 3, 3), device=dev)
        ms_w = timeit(lambda: nat.call("u3d_conv3d_wgrad", 0, _stream(dev), ctypes.byref(s), _p(dz), _p(dw), N, D, H, W, Cout, _p(ws),
                                       nws), args.iters)
        dw_ref = dw.clone()
        ms_w1 = timeit(lambda: nat.call("u3d_subpixel_conv_wgrad", 0, _stream(dev), _p(t1), _p(aff_sub), Cin * 2, _p(dz),
                                        _p(dw.view(-1)[C0 * 27:]), Cin, N, D1, H1, W1, C1, Cout, _p(ws), nws), args.iters)
        ms_w0 = timeit(lambda: nat.call("u3d_conv3d_wgrad_strided", 0, _stream(dev), ctypes.byref(s0), _p(dz), _p(dw), Cin, N, D, H, W,
                                        Cout, _p(ws), nws), args.iters)
        werr = ((dw - dw_ref).abs().max() / dw_ref.abs().max()).item()
        print(f"{name} wgrad: one kernel {ms_w:.3f} ms | subpixel {ms_w1:.3f} ms ({f_dg / ms_w1 / 1e9:.1f} TF executed) + skip slice "
              f"{ms_w0:.3f} ms = {ms_w1 + ms_w0:.3f} ms, max rel diff {werr:.2e}", flush=True)
        err = ((y - ya).abs().max() / ya.abs().max()).item()
        f_sub = 2.0 * 64 * C1 * Cout * N * D1 * H1 * W1  # executed: 8 classes x 8 taps per low-res voxel
        print(f"{name} {C0}+{C1}->{Cout}: one kernel {ms_a:.3f} ms ({flops / ms_a / 1e9:.1f} TF algorithmic) | subpixel {ms_b1:.3f} ms "
              f"({f_sub / ms_b1 / 1e9:.1f} TF executed) + skip half {ms_b2:.3f} ms = {ms_b1 + ms_b2:.3f} ms "
              f"({flops / (ms_b1 + ms_b2) / 1e9:.1f} TF algorithmic), max rel diff {err:.2e}", flush=True)


if __name__ == "__main__":
    main()
