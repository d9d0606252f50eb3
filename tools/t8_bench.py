"""Transposed convolution (ConvTranspose3d k3 s2 p1, space-to-depth form) of config 4's four decoder levels in bf16 storage:
forward / data gradient / weight gradient, ms and TFLOP/s of the 27-tap (minimal) multiply-adds.

    python tools/t8_bench.py [--iters 20]           # A/B: U3D_TUNE=10:2 runs the 2x2x2 kernels without zero-block skipping
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pytorch-3dunet_amd"))
from pytorch3dunet_amd import _native as nat  # noqa: E402
from pytorch3dunet_amd.engine import _p, _stream  # noqa: E402

BF = torch.bfloat16


def timeit(fn, iters):
    for _ in range(3):
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
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--patch", default="80,160,160")
    ap.add_argument("--fmaps", type=int, default=64)
    ap.add_argument("--levels", type=int, default=5)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = nat.get_lib()
    D0, H0, W0 = (int(v) for v in args.patch.split(","))
    S = _stream(dev)
    N = 1
    tot = [0.0, 0.0, 0.0]
    for lvl in range(args.levels - 1, 0, -1):  # decoder input level lvl -> output level lvl - 1
        Cl, Cs = args.fmaps << lvl, args.fmaps << (lvl - 1)
        D1, H1, W1 = D0 >> lvl, H0 >> lvl, W0 >> lvl
        x = torch.relu(torch.randn(N, D1, H1, W1, Cl, device=dev)).to(BF)
        w = torch.randn(Cl, Cs, 3, 3, 3, device=dev) / (27 * Cl / 8) ** 0.5
        t8 = torch.empty(N, D1, H1, W1, 8 * Cs, dtype=BF, device=dev)
        dt8 = torch.randn(N, D1, H1, W1, 8 * Cs, device=dev).to(BF)
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        pk = []
        for mode in (0, 1):
            b = torch.empty(lib.u3d_convtr3d_t8_packed_elems(Cl, Cs, mode), dtype=BF, device=dev)
            nat.call("u3d_pack_convtr3d_t8", 0, S, _p(w), Cl, Cs, mode, _p(b))
            pk.append(b)
        need = lib.u3d_convtr3d_wgrad_t8_workspace_floats(N, D1, H1, W1, Cl, Cs)
        ws = torch.empty(max(need, 4), device=dev)
        nsd = 0 if os.environ.get("U3D_T8_NOSPLIT") else lib.u3d_convtr3d_dgrad_t8_workspace_floats(N, D1, H1, W1, Cl, Cs)  # (A/B: unsplit)
        wsd = torch.empty(max(nsd, 4), device=dev)
        flops = 2.0 * 27 * Cl * Cs * N * D1 * H1 * W1
        nsf = lib.u3d_convtr3d_fwd_t8_workspace_floats(N, D1, H1, W1, Cl, Cs)  # (> 0: the flat tile with a split reduction, as the engine calls it)
        wsf = torch.empty(max(nsf, 4), device=dev)
        ms = (timeit(lambda: nat.call("u3d_convtr3d_fwd_t8_b16_ex", 0, S, _p(x), _p(pk[0]), _p(t8), N, D1, H1, W1, Cl, Cs, _p(wsf), nsf), args.iters),
              timeit(lambda: nat.call("u3d_convtr3d_dgrad_t8_b16_ex", 0, S, _p(dt8), _p(pk[1]), _p(x), _p(dx), N, D1, H1, W1, Cl, Cs, _p(wsd), nsd),
                     args.iters),
              timeit(lambda: nat.call("u3d_convtr3d_wgrad_t8_b16", 0, S, _p(x), _p(dt8), _p(dw), N, D1, H1, W1, Cl, Cs, _p(ws), need), args.iters))
        tot = [a + b for a, b in zip(tot, ms)]
        print(f"{Cl:4d}->{Cs:4d} @{D1}x{H1}x{W1}: fwd {ms[0]:6.3f} ms ({flops / ms[0] / 1e9:6.1f} TF)  dgrad {ms[1]:6.3f} ms ({flops / ms[1] / 1e9:6.1f} TF)  "
              f"wgrad {ms[2]:6.3f} ms ({flops / ms[2] / 1e9:6.1f} TF)", flush=True)
    print(f"sum: fwd {tot[0]:.3f}  dgrad {tot[1]:.3f}  wgrad {tot[2]:.3f} ms")


if __name__ == "__main__":
    main()
