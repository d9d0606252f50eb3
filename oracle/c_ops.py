"""TEST INFRASTRUCTURE ONLY — ctypes front-end of oracle/ref_ops.c (plain-C operator restatement)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libref_ops.so")
_lib = None


def build():
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.c_void_p)


def conv3d_fwd(x, w):
    N, C, D, H, W = x.shape
    K = w.shape[0]
    x, xp = _f(x)
    w, wp = _f(w)
    y = np.empty((N, K, D, H, W), np.float32)
    lib().ref_conv3d_fwd(xp, wp, y.ctypes.data_as(ctypes.c_void_p), N, C, K, D, H, W)
    return y


def conv3d_dgrad(dy, w):
    N, K, D, H, W = dy.shape
    C = w.shape[1]
    dy, dp = _f(dy)
    w, wp = _f(w)
    dx = np.empty((N, C, D, H, W), np.float32)
    lib().ref_conv3d_dgrad(dp, wp, dx.ctypes.data_as(ctypes.c_void_p), N, C, K, D, H, W)
    return dx


def conv3d_wgrad(x, dy):
    N, C, D, H, W = x.shape
    K = dy.shape[1]
    x, xp = _f(x)
    dy, dp = _f(dy)
    dw = np.empty((K, C, 3, 3, 3), np.float32)
    lib().ref_conv3d_wgrad(xp, dp, dw.ctypes.data_as(ctypes.c_void_p), N, C, K, D, H, W)
    return dw


def groupnorm_fwd(x, gamma, beta, G, eps=1e-5):
    N, C = x.shape[:2]
    V = int(np.prod(x.shape[2:]))
    x, xp = _f(x)
    g, gp = _f(gamma)
    b, bp = _f(beta)
    y = np.empty_like(x)
    mean = np.empty((N, G), np.float32)
    rstd = np.empty((N, G), np.float32)
    lib().ref_groupnorm_fwd(xp, gp, bp, y.ctypes.data_as(ctypes.c_void_p), mean.ctypes.data_as(ctypes.c_void_p),
                            rstd.ctypes.data_as(ctypes.c_void_p), N, C, G, ctypes.c_size_t(V), ctypes.c_float(eps))
    return y, mean, rstd


def groupnorm_bwd(dy, x, mean, rstd, gamma, G):
    N, C = x.shape[:2]
    V = int(np.prod(x.shape[2:]))
    dy, dp = _f(dy)
    x, xp = _f(x)
    m, mp = _f(mean)
    r, rp = _f(rstd)
    g, gp = _f(gamma)
    dx = np.empty_like(x)
    dg = np.empty(C, np.float32)
    db = np.empty(C, np.float32)
    lib().ref_groupnorm_bwd(dp, xp, mp, rp, gp, dx.ctypes.data_as(ctypes.c_void_p), dg.ctypes.data_as(ctypes.c_void_p),
                            db.ctypes.data_as(ctypes.c_void_p), N, C, G, ctypes.c_size_t(V))
    return dx, dg, db


def maxpool2_fwd(x):
    N, C, D, H, W = x.shape
    x, xp = _f(x)
    y = np.empty((N, C, D // 2, H // 2, W // 2), np.float32)
    idx = np.empty(y.shape, np.uint8)
    lib().ref_maxpool2_fwd(xp, y.ctypes.data_as(ctypes.c_void_p), idx.ctypes.data_as(ctypes.c_void_p), N, C, D, H, W)
    return y, idx


def upsample_nearest(x, size):
    N, C, D1, H1, W1 = x.shape
    D, H, W = size
    x, xp = _f(x)
    y = np.empty((N, C, D, H, W), np.float32)
    lib().ref_upsample_nearest(xp, y.ctypes.data_as(ctypes.c_void_p), N, C, D1, H1, W1, D, H, W)
    return y


def batchnorm_fwd(x, gamma, beta, running_mean, running_var, training=True, eps=1e-5, momentum=0.1):
    """returns (y, running_mean', running_var') — the running estimates are copies, updated like nn.BatchNorm3d does"""
    N, C = x.shape[:2]
    V = int(np.prod(x.shape[2:]))
    x, xp = _f(x)
    g, gp = _f(gamma)
    b, bp = _f(beta)
    rm, rmp = _f(np.array(running_mean, dtype=np.float32, copy=True))
    rv, rvp = _f(np.array(running_var, dtype=np.float32, copy=True))
    y = np.empty_like(x)
    lib().ref_batchnorm_fwd(xp, gp, bp, rmp, rvp, y.ctypes.data_as(ctypes.c_void_p), N, C, ctypes.c_size_t(V), ctypes.c_float(eps),
                            ctypes.c_float(momentum), 1 if training else 0)
    return y, rm, rv


def _resize(fn, x, size):
    N, C, D1, H1, W1 = x.shape
    D, H, W = size
    x, xp = _f(x)
    y = np.empty((N, C, D, H, W), np.float32)
    getattr(lib(), fn)(xp, y.ctypes.data_as(ctypes.c_void_p), N, C, D1, H1, W1, D, H, W)
    return y


def upsample_trilinear(x, size):
    return _resize("ref_upsample_trilinear", x, size)


def upsample_area(x, size):
    return _resize("ref_upsample_area", x, size)


def conv_transpose3d_fwd(x, w):
    N, Cin, D1, H1, W1 = x.shape
    Cout = w.shape[1]
    x, xp = _f(x)
    w, wp = _f(w)
    y = np.empty((N, Cout, 2 * D1 - 1, 2 * H1 - 1, 2 * W1 - 1), np.float32)
    lib().ref_conv_transpose3d_fwd(xp, wp, y.ctypes.data_as(ctypes.c_void_p), N, Cin, Cout, D1, H1, W1)
    return y
