"""Error of the split-fp32 convolution vs the fp32-MFMA kernel as the contraction grows (both against float64): random-sign
operands and all-positive operands (the latter exposes a rounding BIAS in the accumulation as a drift linear in K)."""
import ctypes
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "pytorch-3dunet_amd"))
import gpu_utils as U  # noqa: E402
from pytorch3dunet_amd import _native as nat  # noqa: E402
from pytorch3dunet_amd.engine import VSrc, _p, _stream  # noqa: E402

dev = U.DEV
for positive in (False, True):
    for C in (16, 64, 256, 1024):
        K = 64
        N, D, H, W = 1, 6, 10, 10
        torch.manual_seed(C)
        x = torch.randn(N, C, D, H, W)
        w = torch.randn(K, C, 3, 3, 3) / (27 * C) ** 0.5
        if positive:
            x, w = x.abs(), w.abs()
        ref = F.conv3d(x.double(), w.double(), None, padding=1)
        xd = U.ndhwc(x)
        n = nat.get_lib().u3d_packed_weight_f32s_elems(C, K, 0)
        wp = torch.empty(n, dtype=torch.bfloat16, device=dev)
        wd = w.to(dev)
        nat.call("u3d_pack_weights_f32s", 0, _stream(dev), _p(wd), K, C, 0, C, 0, _p(wp))
        y = torch.empty((N, D, H, W, K), device=dev)
        nat.call("u3d_conv3d_f32s", 0, _stream(dev), _p(xd), None, _p(wp), _p(y), N, D, H, W, C, K, 0, None, None, None, None, None, 0)
        y32 = torch.empty_like(y)
        s = VSrc(xd).struct(None)
        nat.call("u3d_conv3d_ex", 0, _stream(dev), ctypes.byref(s), _p(U.pack(wd, 0)), _p(y32), N, D, H, W, K, 0, None, None, None, None, None, 0)
        torch.cuda.synchronize()
        ycpu = F.conv3d(x, w, None, padding=1).double()
        out = {}
        for name, t in (("split", U.ncdhw(y).double()), ("f32mfma", U.ncdhw(y32).double()), ("cpu_f32", ycpu)):
            e = t - ref
            out[name] = (float(e.norm() / ref.norm()), float(e.mean() / ref.abs().mean()))
        print(f"positive={positive} C={C:5d}  " + "  ".join(f"{k}: l2 {v[0]:.2e} bias {v[1]:+.2e}" for k, v in out.items()), flush=True)
