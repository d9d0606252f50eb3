"""Per-layer timing of the bf16-operand convolution (u3d_conv3d_bf16, csrc/u3d_bf16.hip) beside the fp32-MFMA one
(u3d_conv3d_ex) on the 3x3x3 layer shapes of BASELINE config 4 (ResidualUNet3D f_maps=64, 1x80x160x160): forward
(affine + ReLU + statistics) and data gradient (GroupNorm-backward sums), each launched in isolation with HIP events.

    python tools/bf16_bench.py [--iters 5] [--patch 80,160,160] [--fmaps 64] [--levels 5]
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
PEAK_BF16, PEAK_F32 = 2500.0, 157.3  # dense TFLOP/s, MI355X_MICROARCH.md


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
    ap.add_argument("--batch", type=int, default=1)
    args = ap.parse_args()
    D0, H0, W0 = (int(v) for v in args.patch.split(","))
    N = args.batch
    lib = nat.get_lib()
    for kv in os.environ.get("U3D_TUNE", "").split(","):
        if ":" in kv:
            nat.call("u3d_set_tuning", int(kv.split(":")[0]), int(kv.split(":")[1]))
    a = torch.randn(4096, 4096, device=dev)
    for _ in range(20):
        a @ a
    torch.cuda.synchronize()
    tot = {"f32": 0.0, "bf16": 0.0, "flops": 0.0}
    for lvl in range(args.levels):
        C = args.fmaps << lvl
        D, H, W = D0 >> lvl, H0 >> lvl, W0 >> lvl
        x = torch.randn(N, D, H, W, C, device=dev)
        aff = torch.randn(N, C, 2, device=dev)
        w = torch.randn(C, C, 3, 3, 3, device=dev) / (27 * C) ** 0.5
        flops = 54.0 * C * C * N * D * H * W
        y = torch.empty((N, D, H, W, C), device=dev)
        st = torch.zeros((N, C, 2), dtype=torch.float64, device=dev)
        src = VSrc(x)
        for mode, label in ((0, "fwd  "), (1, "dgrad")):
            wp32 = U.pack(w, mode)
            n16 = lib.u3d_packed_weight_bf16_elems(C, C, mode)
            wp16 = torch.empty(n16, dtype=torch.bfloat16, device=dev)
            nat.call("u3d_pack_weights_bf16", 0, _stream(dev), _p(w), C, C, mode, _p(wp16))
            n16k = lib.u3d_conv3d_bf16_workspace_floats(N, D, H, W, C, C)
            ws16k = torch.empty(max(n16k, 4), device=dev)
            need = lib.u3d_conv3d_workspace_floats(N, D, H, W, C, C)
            ws = torch.empty(max(need, 4), device=dev)
            s = src.struct(aff if mode == 0 else None)
            gxs = src.struct()
            if mode == 0:
                f32 = lambda: nat.call("u3d_conv3d_ex", 0, _stream(dev), ctypes.byref(s), _p(wp32), _p(y), N, D, H, W, C, 1,  # noqa: E731
                                       _p(st), None, None, None, _p(ws), need)
                b16 = lambda: nat.call("u3d_conv3d_bf16_ex", 0, _stream(dev), _p(x), _p(aff), _p(wp16), _p(y), N, D, H, W, C, C, 1,  # noqa: E731
                                       _p(st), None, None, None, _p(ws16k), n16k)
            else:
                f32 = lambda: nat.call("u3d_conv3d_ex", 0, _stream(dev), ctypes.byref(s), _p(wp32), _p(y), N, D, H, W, C, 0,  # noqa: E731
                                       None, ctypes.byref(gxs), _p(st), None, _p(ws), need)
                b16 = lambda: nat.call("u3d_conv3d_bf16_ex", 0, _stream(dev), _p(x), None, _p(wp16), _p(y), N, D, H, W, C, C, 0,  # noqa: E731
                                       None, _p(x), _p(st), None, _p(ws16k), n16k)
            m32, m16 = timeit(f32, args.iters), timeit(b16, args.iters)
            tot["f32"] += m32
            tot["bf16"] += m16
            tot["flops"] += flops
            # the split-fp32 kernel (fp32-grade results on the bf16 pipe: 6 bf16 MFMAs per fp32 multiply-add)
            n3 = lib.u3d_packed_weight_f32s_elems(C, C, mode)
            wp3 = torch.empty(n3, dtype=torch.bfloat16, device=dev)
            nat.call("u3d_pack_weights_f32s", 0, _stream(dev), _p(w), C, C, mode, C, 0, _p(wp3))
            if mode == 0:
                s3 = lambda: nat.call("u3d_conv3d_f32s", 0, _stream(dev), _p(x), _p(aff), _p(wp3), _p(y), N, D, H, W, C, C, 1,  # noqa: E731
                                      _p(st), None, None, None, _p(ws16k), n16k)
            else:
                s3 = lambda: nat.call("u3d_conv3d_f32s", 0, _stream(dev), _p(x), None, _p(wp3), _p(y), N, D, H, W, C, C, 0,  # noqa: E731
                                      None, _p(x), _p(st), None, _p(ws16k), n16k)
            m3 = timeit(s3, args.iters)
            tot["f32s"] = tot.get("f32s", 0.0) + m3
            tot["f32_fd"] = tot.get("f32_fd", 0.0) + m32
            print(f"L{lvl} {label} split-fp32 {m3:7.3f} ms ({flops / m3 / 1e9:6.1f} TF fp32-equivalent = {6 * flops / m3 / 1e9 / PEAK_BF16:.2f} "
                  f"of bf16 peak executed)   vs fp32 MFMA {m32 / m3:4.2f}x", flush=True)
            # algorithmic HBM bytes: input + output once (fp32), + gx re-read for the data gradient
            hbm = (2 + (mode == 1)) * 4.0 * C * N * D * H * W
            print(f"L{lvl} {label} {C:4d}->{C:4d} @{D}x{H}x{W}: fp32 {m32:7.3f} ms ({flops / m32 / 1e9:6.1f} TF)   "
                  f"bf16 {m16:7.3f} ms ({flops / m16 / 1e9:6.1f} TF = {flops / m16 / 1e9 / PEAK_BF16:.2f} of bf16 peak; "
                  f"{hbm / m16 / 1e6:5.0f} GB/s algorithmic)   speed-up {m32 / m16:4.2f}x", flush=True)
        # weight gradient (the bf16 kernel needs Cin % 32 == 0 and Cout % 64 == 0)
        if C % 64 != 0:
            continue
        dz = torch.randn(N, D, H, W, C, device=dev)
        dw = torch.empty((C, C, 3, 3, 3), device=dev)
        n32 = lib.u3d_wgrad_workspace_floats(N, D, H, W, C, C)
        n16 = lib.u3d_wgrad_bf16_workspace_floats(N, D, H, W, C, C)
        ws32 = torch.empty(max(n32, 4), device=dev)
        ws16 = torch.empty(max(n16, 4), device=dev)
        sa = src.struct(aff)
        f32 = lambda: nat.call("u3d_conv3d_wgrad", 0, _stream(dev), ctypes.byref(sa), _p(dz), _p(dw), N, D, H, W, C, _p(ws32), n32)  # noqa: E731
        b16 = lambda: nat.call("u3d_conv3d_wgrad_bf16", 0, _stream(dev), _p(x), _p(aff), _p(dz), _p(dw), N, D, H, W, C, C, _p(ws16), n16)  # noqa: E731
        m32, m16 = timeit(f32, args.iters), timeit(b16, args.iters)
        tot["f32"] += m32
        tot["bf16"] += m16
        tot["flops"] += flops
        print(f"L{lvl} wgrad {C:4d}->{C:4d} @{D}x{H}x{W}: fp32 {m32:7.3f} ms ({flops / m32 / 1e9:6.1f} TF)   "
              f"bf16 {m16:7.3f} ms ({flops / m16 / 1e9:6.1f} TF = {flops / m16 / 1e9 / PEAK_BF16:.2f} of bf16 peak; "
              f"{8.0 * C * N * D * H * W / m16 / 1e6:5.0f} GB/s algorithmic)   speed-up {m32 / m16:4.2f}x", flush=True)
    if "f32s" in tot:
        print(f"fwd+dgrad: fp32 MFMA {tot['f32_fd']:.2f} ms, split-fp32 {tot['f32s']:.2f} ms, speed-up {tot['f32_fd'] / tot['f32s']:.2f}x")
    print(f"sum: fp32 {tot['f32']:.2f} ms ({tot['flops'] / tot['f32'] / 1e9:.1f} TF), bf16 {tot['bf16']:.2f} ms "
          f"({tot['flops'] / tot['bf16'] / 1e9:.1f} TF), speed-up {tot['f32'] / tot['bf16']:.2f}x")


if __name__ == "__main__":
    main()
