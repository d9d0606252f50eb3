"""Developer aid: forward / data gradient / weight gradient of one layer shape under the ragged persistent kernels (tuning key 3 = 0) and
the round-4 gate (key 3 = 2), against F.conv3d CPU autograd.  python tools/diag_ragged.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("tests", "pytorch-3dunet_amd", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, d))
import torch
import torch.nn.functional as F
import gpu_utils as U
from pytorch3dunet_amd import _native as nat
from pytorch3dunet_amd.engine import VSrc

CASES = [(1, 8, 8, 12, 20, 17, True), (1, 8, 8, 12, 20, 17, False), (1, 8, 32, 12, 20, 17, True), (1, 16, 8, 12, 20, 17, True),
         (1, 8, 8, 12, 24, 17, True), (1, 8, 8, 12, 20, 16, True), (1, 8, 8, 4, 20, 17, True), (1, 32, 8, 12, 20, 17, True), (1, 4, 8, 12, 20, 17, True)]
for (N, Cin, Cout, D, H, W, affine) in CASES:
    torch.manual_seed(1)
    x = torch.randn(N, Cin, D, H, W)
    w = torch.randn(Cout, Cin, 3, 3, 3) / (27 * Cin) ** 0.5
    dz = torch.randn(N, Cout, D, H, W)
    ab = torch.randn(N, Cin, 2)
    aff = ab.contiguous().to(U.DEV) if affine else None
    g = x * ab[:, :, 0].view(N, Cin, 1, 1, 1) + ab[:, :, 1].view(N, Cin, 1, 1, 1) if affine else x
    gl = g.clone().requires_grad_(True)
    wl = w.clone().requires_grad_(True)
    y = F.conv3d(gl, wl, None, padding=1)
    y.backward(dz)
    for gate in (0, 2):
        nat.call("u3d_set_tuning", 3, gate)
        st = torch.zeros((N, Cout, 2), dtype=torch.float64, device=U.DEV)
        yo = U.conv3d(VSrc(U.ndhwc(x)), w, Cout, relu=0, affine=aff, out_stats=st)
        gst = torch.zeros((N, Cin, 2), dtype=torch.float64, device=U.DEV)
        dg = U.conv3d(VSrc(U.ndhwc(dz)), w, Cin, relu=0, mode=1, gx=VSrc(U.ndhwc(x)), gstats=gst)
        dw = U.wgrad(VSrc(U.ndhwc(x)), U.ndhwc(dz), Cout, affine=aff)
        torch.cuda.synchronize()
        s_ref = torch.stack([gl.grad.double().sum(dim=(2, 3, 4)), (gl.grad.double() * x.double()).sum(dim=(2, 3, 4))], dim=-1)
        print((N, Cin, Cout, D, H, W, affine), "gate", gate, "fwd %.1e dgrad %.1e gstats %.1e wgrad %.1e" % (
            U.relerr(U.ncdhw(yo), y.detach()), U.relerr(U.ncdhw(dg), gl.grad), U.relerr(gst.cpu(), s_ref), U.relerr(dw.cpu(), wl.grad)),
            "variants", nat.get_lib().u3d_conv3d_variant(N, D, H, W, Cin, Cout, 0, 0), nat.get_lib().u3d_conv3d_wgrad_variant(N, D, H, W, Cin, Cout, 0))
nat.call("u3d_set_tuning", 3, 0)
