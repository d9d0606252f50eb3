"""Per-layer timing of the bf16-operand kernels with fp32 versus bf16 ACTIVATION STORAGE (the `_b16` entry points) on BASELINE
config 4's square layers: forward, data gradient, weight gradient.

    python tools/b16_bench.py [--patch 80,160,160 --fmaps 64 --levels 5 --iters 5]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("pytorch-3dunet_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch  # noqa: E402

from pytorch3dunet_amd import _native as nat  # noqa: E402
from pytorch3dunet_amd.engine import _p, _stream  # noqa: E402

dev = torch.device("cuda", 0)
BF = torch.bfloat16


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
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--patch", default="80,160,160")
    ap.add_argument("--fmaps", type=int, default=64)
    ap.add_argument("--levels", type=int, default=5)
    args = ap.parse_args()
    D0, H0, W0 = (int(v) for v in args.patch.split(","))
    lib = nat.get_lib()
    a = torch.randn(4096, 4096, device=dev)
    for _ in range(20):
        a @ a
    torch.cuda.synchronize()
    N = 1
    tot = {}
    for lvl in range(args.levels):
        C = args.fmaps << lvl
        D, H, W = D0 >> lvl, H0 >> lvl, W0 >> lvl
        x = torch.randn(N, D, H, W, C, device=dev)
        dz = torch.randn(N, D, H, W, C, device=dev)
        res = torch.randn(N, D, H, W, C, device=dev)
        aff = torch.randn(N, C, 2, device=dev)
        w = torch.randn(C, C, 3, 3, 3, device=dev) / (27 * C) ** 0.5
        xb, dzb, resb = x.to(BF), dz.to(BF), res.to(BF)
        y, yb = torch.empty_like(x), torch.empty_like(xb)
        st = torch.zeros((N, C, 2), dtype=torch.float64, device=dev)
        flops = 54.0 * C * C * N * D * H * W
        pk = []
        for mode in (0, 1):
            t = torch.empty(lib.u3d_packed_weight_bf16_elems(C, C, mode), dtype=BF, device=dev)
            nat.call("u3d_pack_weights_bf16", 0, _stream(dev), _p(w), C, C, mode, _p(t))
            pk.append(t)
        nk = lib.u3d_conv3d_bf16_workspace_floats(N, D, H, W, C, C)
        wsk = torch.empty(max(nk, 4), device=dev)
        nw = lib.u3d_wgrad_bf16_workspace_floats(N, D, H, W, C, C)
        wsw = torch.empty(max(nw, 4), device=dev)
        dw = torch.empty_like(w)
        S = _stream(dev)
        cases = {
            "fwd": (lambda: nat.call("u3d_conv3d_bf16_ex", 0, S, _p(x), _p(aff), _p(pk[0]), _p(y), N, D, H, W, C, C, 1, _p(st), None, None,
                                     _p(res), _p(wsk), nk),
                    lambda: nat.call("u3d_conv3d_bf16_ex_b16", 0, S, _p(xb), _p(aff), _p(pk[0]), _p(yb), N, D, H, W, C, C, 1, _p(st), None,
                                     None, _p(resb), _p(wsk), nk)),
            "dgrad": (lambda: nat.call("u3d_conv3d_bf16_ex", 0, S, _p(dz), None, _p(pk[1]), _p(y), N, D, H, W, C, C, 0, None, _p(x), _p(st),
                                       None, _p(wsk), nk),
                      lambda: nat.call("u3d_conv3d_bf16_ex_b16", 0, S, _p(dzb), None, _p(pk[1]), _p(yb), N, D, H, W, C, C, 0, None, _p(xb),
                                       _p(st), None, _p(wsk), nk)),
            "wgrad": (lambda: nat.call("u3d_conv3d_wgrad_bf16", 0, S, _p(x), _p(aff), _p(dz), _p(dw), N, D, H, W, C, C, _p(wsw), nw),
                      lambda: nat.call("u3d_conv3d_wgrad_bf16_b16", 0, S, _p(xb), _p(aff), _p(dzb), _p(dw), N, D, H, W, C, C, _p(wsw), nw)),
        }
        for name, (f32, b16) in cases.items():
            if os.environ.get("U3D_BENCH_TRACE"):
                print(f"L{lvl} {name} ...", flush=True)
            m32, m16 = timeit(f32, args.iters), timeit(b16, args.iters)
            tot[name] = tuple(a_ + b_ for a_, b_ in zip(tot.get(name, (0.0, 0.0)), (m32, m16)))
            print(f"L{lvl} {name:5s} {C:4d}->{C:4d} @{D}x{H}x{W}: fp32 storage {m32:6.3f} ms ({flops / m32 / 1e9:6.1f} TF)   bf16 storage "
                  f"{m16:6.3f} ms ({flops / m16 / 1e9:6.1f} TF = {flops / m16 / 1e9 / 2500.0:.2f} of the bf16 peak)   {m32 / m16:4.2f}x", flush=True)
    for name, (m32, m16) in tot.items():
        print(f"sum {name}: fp32 storage {m32:.3f} ms, bf16 storage {m16:.3f} ms ({m32 / m16:.2f}x)")


if __name__ == "__main__":
    main()
