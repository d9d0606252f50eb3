"""helpers for the -m gpu tests: thin callers of the C-ABI on torch CUDA tensors"""
import ctypes

import torch

from pytorch3dunet_amd import _native as nat
from pytorch3dunet_amd.engine import VSrc, _p, _stream

DEV = torch.device("cuda", 0)


def ndhwc(x):  # (N,C,D,H,W) cpu -> (N,D,H,W,C) gpu
    return x.permute(0, 2, 3, 4, 1).contiguous().to(DEV)


def ncdhw(x):  # (N,D,H,W,C) gpu -> (N,C,D,H,W) cpu
    return x.permute(0, 4, 1, 2, 3).contiguous().cpu()


def pack(w, mode):
    Cout, Cin = w.shape[:2]
    n = nat.get_lib().u3d_packed_weight_floats(Cin, Cout, mode)
    out = torch.empty(n, dtype=torch.float32, device=DEV)
    wd = w.contiguous().to(DEV)
    nat.call("u3d_pack_weights", 0, _stream(DEV), _p(wd), Cout, Cin, mode, _p(out))
    return out


def conv3d(src: VSrc, w, Cout, relu=0, affine=None, mode=0, out_stats=None, gx: VSrc = None, gstats=None):
    wp = pack(w, mode)
    y = torch.empty((src.N, src.D, src.H, src.W, Cout), dtype=torch.float32, device=DEV)
    s = src.struct(affine)
    gs = gx.struct() if gx is not None else None
    nat.call("u3d_conv3d", 0, _stream(DEV), ctypes.byref(s), _p(wp), _p(y), src.N, src.D, src.H, src.W, Cout, relu,
             _p(out_stats), ctypes.byref(gs) if gs is not None else None, _p(gstats))
    return y


def conv3d_ex(src: VSrc, w, Cout, relu=0, affine=None, mode=0, out_stats=None, gx: VSrc = None, gstats=None, residual=None,
              use_ws=True):
    """u3d_conv3d_ex with the scratch buffer the library asks for (split-K on small volumes); returns (y, ws_floats)"""
    wp = pack(w, mode)
    y = torch.empty((src.N, src.D, src.H, src.W, Cout), dtype=torch.float32, device=DEV)
    s = src.struct(affine)
    gs = gx.struct() if gx is not None else None
    need = nat.get_lib().u3d_conv3d_workspace_floats(src.N, src.D, src.H, src.W, src.C, Cout) if use_ws else 0
    ws = torch.empty(need, dtype=torch.float32, device=DEV) if need > 0 else None
    nat.call("u3d_conv3d_ex", 0, _stream(DEV), ctypes.byref(s), _p(wp), _p(y), src.N, src.D, src.H, src.W, Cout, relu,
             _p(out_stats), ctypes.byref(gs) if gs is not None else None, _p(gstats), _p(residual), _p(ws), need)
    return y, need


def conv3d_naive(src: VSrc, w, Cout, relu=0, affine=None, flip=0):
    y = torch.empty((src.N, src.D, src.H, src.W, Cout), dtype=torch.float32, device=DEV)
    s = src.struct(affine)
    wd = w.contiguous().to(DEV)
    nat.call("u3d_conv3d_naive", 0, _stream(DEV), ctypes.byref(s), _p(wd), _p(y), src.N, src.D, src.H, src.W, src.C, Cout,
             relu, flip)
    return y


def wgrad(src: VSrc, dz, Cout, affine=None):
    lib = nat.get_lib()
    n = lib.u3d_wgrad_workspace_floats(src.N, src.D, src.H, src.W, src.C, Cout)
    ws = torch.empty(n, dtype=torch.float32, device=DEV)
    dw = torch.empty((Cout, src.C, 3, 3, 3), dtype=torch.float32, device=DEV)
    s = src.struct(affine)
    nat.call("u3d_conv3d_wgrad", 0, _stream(DEV), ctypes.byref(s), _p(dz), _p(dw), src.N, src.D, src.H, src.W, Cout, _p(ws), n)
    return dw


def chan_stats(src: VSrc):
    st = torch.zeros((src.N, src.C, 2), dtype=torch.float64, device=DEV)
    s = src.struct()
    nat.call("u3d_chan_stats", 0, _stream(DEV), ctypes.byref(s), src.N, src.D, src.H, src.W, _p(st))
    return st


def gn_finalize(st0, C0, sc0, st1, C1, sc1, N, G, count, gamma, beta, eps=1e-5):
    C = C0 + C1
    aff = torch.empty((N, C, 2), dtype=torch.float32, device=DEV)
    mr = torch.empty((N, G, 2), dtype=torch.float32, device=DEV)
    nat.call("u3d_gn_finalize", 0, _stream(DEV), _p(st0), C0, sc0, _p(st1), C1, sc1, N, G, float(count), _p(gamma), _p(beta),
             eps, _p(aff), _p(mr))
    return aff, mr


def relerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    d = b.abs().max().item()
    return (a - b).abs().max().item() / (d if d > 0 else 1.0)
