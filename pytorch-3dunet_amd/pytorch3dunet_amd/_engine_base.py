"""Executor plumbing shared by every part of the native executor (engine.py): buffer helpers, the virtual-concat source
descriptor, the layer-order grammar (reference buildingblocks.py:10-96), the tape records, the zeroed statistics pool and the
per-backward context.  Split out of engine.py in round 4; nothing here launches model kernels."""
from __future__ import annotations

import copy
import ctypes
import dataclasses
import os
import threading
from dataclasses import dataclass, field
from typing import List, Optional

import torch
import torch.nn.functional as F

from . import _native as nat
from ._native import U3DSrc

__all__ = [
    "ACT_ELU",
    "ACT_LEAKY",
    "ACT_NONE",
    "ACT_RELU",
    "CkptRec",
    "ConvRec",
    "F",
    "LayerSpec",
    "List",
    "Optional",
    "ResRec",
    "StaleParameters",
    "slab_boxes",
    "Tape",
    "U3DSrc",
    "UpRec",
    "VSrc",
    "_ACTS",
    "_ALWAYS_REPACK",
    "_BwdCtx",
    "_F32",
    "_LEAF_TYPES",
    "_MAP_CACHE",
    "_PIndex",
    "_POISON",
    "_RECORD_TYPES",
    "_RESAMPLE_CACHE",
    "_Ref",
    "_SIDE_STREAMS",
    "_StatPool",
    "_side_stream",
    "_empty",
    "_empty_like",
    "_maps",
    "_p",
    "_resample_tables",
    "_stream",
    "_walk",
    "copy",
    "ctypes",
    "dataclass",
    "dataclasses",
    "field",
    "layer_spec",
    "module_params",
    "nat",
    "nearest_map_host",
    "os",
    "parse_order",
    "resample_tables_host",
    "stash_tape",
    "threading",
    "torch",
    "unstash_tape",
]


_F32 = torch.float32


_POISON = os.environ.get("U3D_POISON", "0") == "1"  # debugging: every scratch / output buffer starts as NaN (or 0xFF bytes), so that
                                                     # a kernel reading memory nobody wrote shows up as NaN instead of stale values


def _empty(*size, **kw):
    t = torch.empty(*size, **kw)
    if _POISON:
        t.fill_(float("nan")) if t.is_floating_point() else t.fill_(-1 if t.dtype != torch.uint8 else 255)
    return t


def _empty_like(x, **kw):
    t = torch.empty_like(x, **kw)
    if _POISON:
        t.fill_(float("nan")) if t.is_floating_point() else t.fill_(-1 if t.dtype != torch.uint8 else 255)
    return t


def _p(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(dev: torch.device):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


# ---------------------------------------------------------------------------------------------------------
# nearest-neighbour index maps (F.interpolate(mode="nearest"), buildingblocks.py:614)
_MAP_CACHE: dict = {}


def nearest_map_host(n_in: int, n_out: int) -> torch.Tensor:
    """src index for every dst index, exactly as ATen computes it:
    src = min(floor(dst * float32(n_in / n_out)), n_in - 1)  (identity / >>1 special cases included).
    Obtained by running the 1-D CPU operator itself on an index ramp, so there is no formula drift."""
    ramp = torch.arange(n_in, dtype=torch.float32).view(1, 1, n_in)
    out = torch.nn.functional.interpolate(ramp, size=n_out, mode="nearest")
    return out.view(-1).to(torch.int32)


def _maps(dev: torch.device, n_in: int, n_out: int):
    """(map[n_out], lo[n_in+1]) device int32 tensors; children of low-res i are [lo[i], lo[i+1])."""
    key = (str(dev), n_in, n_out)
    hit = _MAP_CACHE.get(key)
    if hit is None:
        m = nearest_map_host(n_in, n_out)
        lo = torch.searchsorted(m.to(torch.int64), torch.arange(n_in + 1, dtype=torch.int64)).to(torch.int32)
        hit = (m.to(dev), lo.to(dev))
        _MAP_CACHE[key] = hit
    return hit


def module_params(module) -> list:
    """`list(module.parameters())` that also works on an nn.DataParallel replica: replicate() empties `_parameters` and keeps
    the broadcast copies (non-leaf tensors that require grad) as plain attributes + `_former_parameters`, in the same
    registration order (torch/nn/parallel/replicate.py).  Same module pre-order as nn.Module.parameters()."""
    out, seen = [], set()
    for mod in module.modules():
        # a replica's `_parameters` holds only the None entries (e.g. bias=False); the live copies are in `_former_parameters`
        for p in list(mod._parameters.values()) + list((getattr(mod, "_former_parameters", None) or {}).values()):
            if p is not None and id(p) not in seen:
                seen.add(id(p))
                out.append(p)
    return out


class StaleParameters(KeyError):
    """a module of the tree holds a parameter OBJECT this executor was not built with (`module.weight = nn.Parameter(...)`, weight
    surgery): the model rebuilds its executor and runs the forward again (unet3d/model.py)"""


class _PIndex(dict):
    """parameter object id -> position in `engine.params`; a miss means the module tree changed under the executor"""

    def __missing__(self, key):
        raise StaleParameters("u3d: a parameter object of the module tree is not one this executor was built with")


class _Ref:
    """placeholder of a tensor inside a stashed tape: index into ctx.saved_tensors, or into engine.params"""

    __slots__ = ("i", "param")

    def __init__(self, i, param):
        self.i, self.param = i, param


_LEAF_TYPES = (type(None), int, float, str, bool)
_RECORD_TYPES: set = set()  # dataclasses of the tape + VSrc, filled in below their definitions (cheaper than dataclasses.is_dataclass)


def _walk(obj, fn):
    """rebuild the tape's object graph (dataclasses, VSrc, lists/tuples/dicts) with every leaf mapped through fn;
    nn.Modules, numbers and strings stay as they are"""
    t = type(obj)
    if t in _LEAF_TYPES:
        return obj
    if t is _Ref or isinstance(obj, torch.Tensor):
        return fn(obj)
    if t is list or t is tuple:
        return t(_walk(o, fn) for o in obj)
    if t is dict:
        return {k: _walk(v, fn) for k, v in obj.items()}
    if t in _RECORD_TYPES or dataclasses.is_dataclass(obj):
        new = copy.copy(obj)
        for k, v in vars(obj).items():
            setattr(new, k, _walk(v, fn))
        return new
    return obj


def stash_tape(tape, pindex):
    """(skeleton, tensors): the tape with every activation replaced by a placeholder, and the activations as a flat list
    for ctx.save_for_backward — autograd then owns their lifetime exactly as it does for stock modules: released after
    backward unless retain_graph=True, 'backward through the graph a second time' raised by autograd itself, in-place
    modification detected by the version counters.  Parameters are referenced by position, not saved."""
    bag, slot = [], {}

    def put(t):
        pi = pindex.get(id(t))
        if pi is not None:
            return _Ref(pi, True)
        i = slot.get(id(t))
        if i is None:
            i = slot[id(t)] = len(bag)
            bag.append(t)
        return _Ref(i, False)

    return _walk(tape, put), bag


def unstash_tape(skel, saved, params):
    return _walk(skel, lambda r: params[r.i] if r.param else saved[r.i])


def resample_tables_host(mode: str, n_in: int, n_out: int):
    """Per-dimension tables of F.interpolate(mode='trilinear' | 'area') for one (n_in -> n_out >= n_in) axis, computed with
    ATen's own float32 formulas (UpSample.h area_pixel_compute_source_index, align_corners=False, scale = in/out because
    the reference passes `size`; AdaptiveAveragePooling start/end indices): idx (n_out,2) int32 source samples, wt (n_out,2)
    float32 weights, rng (n_in,2) int32 = [lo, hi) outputs touching each input (the adjoint gathers over them)."""
    assert n_out >= n_in >= 1, "decoders only upsample"
    o = torch.arange(n_out)
    if mode == "trilinear":
        scale = torch.tensor(float(n_in), dtype=torch.float32) / torch.tensor(float(n_out), dtype=torch.float32)
        src = (scale * (o.to(torch.float32) + 0.5) - 0.5).clamp_min(0.0)
        i0 = src.to(torch.int64)
        i1 = i0 + (i0 < n_in - 1).to(torch.int64)
        w1 = src - i0.to(torch.float32)
        w0 = 1.0 - w1
    elif mode == "area":
        start = (o * n_in) // n_out
        end = ((o + 1) * n_in + n_out - 1) // n_out
        ln = end - start
        assert int(ln.max()) <= 2 and int(ln.min()) >= 1
        i0, i1 = start, end - 1
        w0 = torch.where(ln == 1, torch.tensor(1.0), torch.tensor(0.5))
        w1 = torch.where(ln == 1, torch.tensor(0.0), torch.tensor(0.5))
    else:
        raise ValueError(mode)
    idx = torch.stack((i0, i1), dim=1).to(torch.int32).contiguous()
    wt = torch.stack((w0, w1), dim=1).to(torch.float32).contiguous()
    rng = torch.zeros((n_in, 2), dtype=torch.int32)
    for i in range(n_in):
        hit = ((i0 == i) | (i1 == i)).nonzero().flatten()
        if hit.numel():
            rng[i, 0], rng[i, 1] = int(hit[0]), int(hit[-1]) + 1
    return idx, wt, rng


_RESAMPLE_CACHE: dict = {}


def _resample_tables(dev: torch.device, mode: str, n_in: int, n_out: int):
    key = (str(dev), mode, n_in, n_out)
    t = _RESAMPLE_CACHE.get(key)
    if t is None:
        t = _RESAMPLE_CACHE[key] = tuple(a.to(dev) for a in resample_tables_host(mode, n_in, n_out))
    return t


class VSrc:
    """A (virtual) NDHWC activation: full-res tensor t0 (N,D,H,W,C0) [+ low-res t1 (N,D1,H1,W1,C1) read through
    nearest maps = the never-materialised torch.cat((skip, interpolate(x)), dim=1)]."""

    def __init__(self, t0: torch.Tensor, t1: Optional[torch.Tensor] = None):
        self.t0 = t0
        self.t1 = t1
        self.N, self.D, self.H, self.W, self.C0 = t0.shape
        self.C1 = 0
        self.maps = None
        self.los = None
        if t1 is not None:
            _, self.D1, self.H1, self.W1, self.C1 = t1.shape
            dev = t0.device
            mz, lz = _maps(dev, self.D1, self.D)
            my, ly = _maps(dev, self.H1, self.H)
            mx, lx = _maps(dev, self.W1, self.W)
            self.maps = (mz, my, mx)
            self.los = (lz, ly, lx)

    @property
    def C(self):
        return self.C0 + self.C1

    @property
    def exact2x(self):
        return self.t1 is not None and self.D == 2 * self.D1 and self.H == 2 * self.H1 and self.W == 2 * self.W1

    def struct(self, affine: Optional[torch.Tensor] = None) -> U3DSrc:
        s = U3DSrc()
        s.p0 = self.t0.data_ptr()
        s.C0 = self.C0
        s.C1 = self.C1
        s.affine = affine.data_ptr() if affine is not None else None
        if self.t1 is not None:
            s.p1 = self.t1.data_ptr()
            s.zmap, s.ymap, s.xmap = (m.data_ptr() for m in self.maps)
            s.D1, s.H1, s.W1 = self.D1, self.H1, self.W1
        return s


    def up_only_struct(self, affine: Optional[torch.Tensor] = None) -> U3DSrc:
        """the UPSAMPLED half alone as a source (C0 = 0; p0 only has to be readable): the box launches of a level that upsamples
        n -> 2n + 1 convolve just those channels.  `affine`: the compact (N, C1, 2) rows of these channels."""
        s = U3DSrc()
        s.p0 = self.t1.data_ptr()
        s.C0 = 0
        s.C1 = self.C1
        s.affine = affine.data_ptr() if affine is not None else None
        s.p1 = self.t1.data_ptr()
        s.zmap, s.ymap, s.xmap = (m.data_ptr() for m in self.maps)
        s.D1, s.H1, s.W1 = self.D1, self.H1, self.W1
        return s

    @property
    def plus(self):
        """per axis: 0 if the low-res half is upsampled by exactly 2, 1 if n -> 2n + 1 (nearest: src = (dst - 1) >> 1, src(0) = 0), None
        for any other ratio (tests/test_boundary.py::test_nearest_maps_match_interpolate pins the closed form on ATen's operator)"""
        if self.t1 is None:
            return None
        e = tuple(full - 2 * low for full, low in zip((self.D, self.H, self.W), (self.D1, self.H1, self.W1)))
        return e if all(v in (0, 1) for v in e) else None


def slab_boxes(dims, plus, t):
    """disjoint boxes (z0, y0, x0, z1, y1, x1) covering {voxels with coordinate < t along at least one axis a with plus[a]} — the
    near-boundary slab of a level that upsamples n -> 2n + 1 (t = 2: the outputs the sub-pixel window cannot produce; t = 3: the source
    voxels their data gradient reaches)"""
    boxes = []
    lo = [0, 0, 0]
    for a in range(3):
        if not plus[a]:
            continue
        b0, b1 = list(lo), list(dims)
        b1[a] = min(t, dims[a])
        if all(x < y for x, y in zip(b0, b1)):
            boxes.append(tuple(b0) + tuple(b1))
        lo[a] = min(t, dims[a])  # the later boxes start beyond this axis' slab
    return boxes


# ---------------------------------------------------------------------------------------------------------
ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_ELU = 0, 1, 2, 3  # activation codes of include/u3d.h (u3d_act_fwd)


@dataclass(frozen=True)
class LayerSpec:
    """one SingleConv order string (create_conv, buildingblocks.py:10-96) as the executor runs it"""

    norm: Optional[str]   # 'g' GroupNorm, 'b' BatchNorm3d, None: no norm -> the conv has a bias (:54-55)
    pre: bool             # the norm acts on the conv INPUT ('gc…', 'bc…')
    act: int              # non-linearity of the layer output
    slope: float
    inner: int            # non-linearity between the conv and a TRAILING norm ('crg', the reference docstring's example)
    islope: float
    drop: Optional[str]   # 'd' nn.Dropout / 'D' nn.Dropout2d (per-(n, channel) on 5-D inputs) as the LAST operation


_ACTS = {"r": (ACT_RELU, 0.0), "l": (ACT_LEAKY, 0.01), "e": (ACT_ELU, 0.0)}  # nn defaults (:47-51)


def layer_spec(order: str) -> Optional[LayerSpec]:
    """Native grammar:  [g|b] c [r|l|e] [d|D]   |   c [r|l|e] (g|b) [d|D]   |   c (g|b) [r|l|e] [d|D]   |   c [r|l|e] [d|D].
    Anything else (two norms, dropout in the middle of a layer, ELU before a dropout, …) runs the module tree."""
    if not order or any(ch not in "gbcrledD" for ch in order) or order.count("c") != 1:
        return None
    drop = None
    if order[-1] in "dD":
        drop, order = order[-1], order[:-1]
    if any(ch in "dD" for ch in order) or not order:
        return None
    norms = [ch for ch in order if ch in "gb"]
    acts = [ch for ch in order if ch in "rle"]
    if len(norms) > 1 or len(acts) > 1:
        return None
    norm = norms[0] if norms else None
    a, sl = _ACTS[acts[0]] if acts else (ACT_NONE, 0.0)
    if drop and a == ACT_ELU:
        return None  # the consumers remove f through the layer OUTPUT, which the dropout rescales: exact for ReLU / LeakyReLU only
    ci = order.index("c")
    if norm is None:
        return LayerSpec(None, False, a, sl, ACT_NONE, 0.0, drop) if order in ("c", "c" + "".join(acts)) else None
    ni = order.index(norm)
    if ni < ci:  # pre-norm: N c [A]
        return LayerSpec(norm, True, a, sl, ACT_NONE, 0.0, drop) if order == norm + "c" + "".join(acts) else None
    if order == "c" + norm + "".join(acts):  # post-norm: c N [A]
        return LayerSpec(norm, False, a, sl, ACT_NONE, 0.0, drop)
    if acts and order == "c" + acts[0] + norm:  # c A N: the non-linearity sits inside
        return LayerSpec(norm, False, ACT_NONE, 0.0, a, sl, drop)
    return None


def parse_order(order: str):
    """(conv input has no norm of its own, act, slope) of a natively executable order, else None — see layer_spec"""
    sp = layer_spec(order)
    return None if sp is None else (not sp.pre, sp.act, sp.slope)


@dataclass
class ConvRec:
    """what one SingleConv ('gcr': GroupNorm -> Conv3d -> ReLU, buildingblocks.py:99-135) saves for backward"""

    name: str
    src: VSrc
    affine: torch.Tensor
    mean_rstd: torch.Tensor
    y: torch.Tensor
    gn_w: torch.Tensor
    conv_w: torch.Tensor
    G: int
    idx_gw: int = -1  # indices into the flat parameter list
    idx_gb: int = -1
    idx_w: int = -1
    small: bool = False  # ran through the small-Cin (first layer) kernels
    sub: Optional[tuple] = None  # (C0, C1): the upsampled half ran as a sub-pixel convolution (csrc/u3d_subpix.hip)
    pre_norm: bool = True        # GroupNorm on the conv input ('gc…'); False: `affine` is the identity table
    post: Optional[tuple] = None  # post-norm order ('cg…'): (z = [f_inner](conv output), its GroupNorm affine table, f_inner, slope); y = f(a*z + b)
    norm: Optional[str] = "g"    # 'g' GroupNorm, 'b' BatchNorm3d (mean_rstd is (C,2)), None: conv bias (idx_gb = its index)
    bn_training: bool = True     # BatchNorm normalised with batch statistics (else: running statistics, constants in backward)
    drop: Optional[tuple] = None  # trailing dropout: ('d', mask NDHWC) or ('D', (N,C,2) table (mask, 0))
    # sub-pixel decoder layers: compact (N,C0,2) / (N,C1,2) copies of `affine`'s rows for the skip / upsampled half, written by the
    # GroupNorm finalize itself (u3d_gn_finalize_split) — the kernels that read ONE half as a plain tensor take these
    affine_lo: Optional[torch.Tensor] = None
    affine_hi: Optional[torch.Tensor] = None


@dataclass
class Tape:
    convs: List[ConvRec] = field(default_factory=list)
    pools: list = field(default_factory=list)  # (pooled, argmax, e_in) per encoder level > 0
    head_x: Optional[torch.Tensor] = None
    dims: tuple = ()
    x0: Optional[torch.Tensor] = None
    blocks: list = field(default_factory=list)  # residual executor: ResRec per block (encoders, then decoders)
    ups: list = field(default_factory=list)     # residual executor: UpRec per decoder
    cats: dict = field(default_factory=dict)    # DoubleConv executor, bf16 mode: decoder index -> the VIRTUAL source whose concat was materialised
    lean: bool = False      # memory-lean mode (checkpoint_encoders): backward releases every block's tensors as soon as it is done
    consumed: bool = False  # ... so the tape can be walked only once
    bwd_pool: Optional[object] = None  # DoubleConv executor: the backward pass's zeroed scratch, carved from the forward's pool (one fill launch)


class _StatPool:
    """one zero-filled double buffer per pass, handed out in slices (a single memset per forward/backward)"""

    def __init__(self, dev, doubles: int):
        self.buf = torch.zeros(max(doubles, 2), dtype=torch.float64, device=dev)
        self.off = 0

    def take(self, n: int) -> torch.Tensor:
        if self.off + n > self.buf.numel():
            # (recomputed blocks of the activation-checkpointing path are not known when the pool is sized) — a fresh zeroed
            # chunk; slices handed out earlier keep the old buffer alive
            self.buf = torch.zeros(max(n, 1 << 16), dtype=torch.float64, device=self.buf.device)
            self.off = 0
        s = self.buf[self.off : self.off + n]
        self.off += n
        return s

    def carve(self, n: int) -> "_StatPool":
        """a pool of its own over the next n zeroed doubles of this one (the backward pass's scratch inside the forward's fill launch)"""
        sub = _StatPool.__new__(_StatPool)
        sub.buf = self.take(n)
        sub.off = 0
        return sub


_SIDE_STREAMS: dict = {}


def _side_stream(dev):
    """the process-wide second HIP stream of a device (weight packing beside the first layer; side-stream weight gradients)"""
    key = (dev.type, dev.index)
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(dev)
    return st


class _BwdCtx:
    """per-backward scratch shared by the helper methods: zeroed double pool, wgrad workspace, flat gradient buffer, and the
    side stream on which the weight gradients of SMALL layers run concurrently with their data gradients"""

    # Layers with at most this many voxels (N*D*H*W) issue their weight gradient — independent of the data gradient, both
    # only read dz — on a second HIP stream.  Measured on the bench workload (profiles/r01v_side_stream_sweep.txt): 0 (off)
    # 80.1 patches/s, 32 k voxels (the levels that cannot fill 256 CUs) 80.0, every layer 82.0 (+2.4 %: tails of one
    # kernel filled by the other).  Default OFF: concurrent kernels make every per-kernel duration (HIP events, rocprofv3)
    # read longer, which would blur the roofline evidence for +2.4 %; export U3D_SIDE_VOXELS=4000000 to trade that.
    SIDE_MAX_VOXELS = int(os.environ.get("U3D_SIDE_VOXELS", 0))

    def __init__(self, dev, pool, ws, flat, engine):
        self.dev, self.pool, self.ws, self.flat = dev, pool, ws, flat
        self._e = engine
        self.side = None
        self.ws_side = None
        self.side_used = False
        self.coef_hi = None  # set by ConvLayers._norm_bwd_finalize for a sub-pixel layer: (N,3,C1) table (p, 8q, 8r) of the upsampled channels

    def gview(self, idx):
        e = self._e
        return self.flat[e.poffs[idx] : e.poffs[idx] + e.params[idx].numel()]

    def ensure_ws(self, floats):
        """the shared scratch buffer, grown on demand (kernels already queued on this stream keep using the old block: the
        caching allocator only hands it out again to later work of the same stream)"""
        if self.ws.numel() < floats:
            self.ws = _empty(int(floats), dtype=_F32, device=self.dev)
        return self.ws

    def side_stream(self, ws_floats):
        if self.side is None:
            self.side = _side_stream(self.dev)
        if self.ws_side is None or self.ws_side.numel() < ws_floats:
            if self.ws_side is not None:
                self.join()  # the old workspace may still be in use on the side stream
            self.ws_side = _empty(max(int(ws_floats), 4), dtype=_F32, device=self.dev)
        return self.side

    def join(self):
        """make the caller's stream wait for every weight gradient issued on the side stream"""
        if self.side_used:
            torch.cuda.current_stream(self.dev).wait_stream(self.side)
            self.side_used = False


_ALWAYS_REPACK = os.environ.get("U3D_ALWAYS_REPACK", "0") == "1"




@dataclass
class ResRec:
    """what one ResNetBlock (buildingblocks.py:230-288, any native order) saves for backward"""

    name: str
    x_in: torch.Tensor           # block input (pooled tensor / network input / joined decoder tensor)
    r: torch.Tensor              # `residual` = conv1(x_in) (or x_in itself for nn.Identity)
    rec2: ConvRec                # conv2: SingleConv(order) on r
    rec3: ConvRec                # conv3: SingleConv(order without r/l/e) on conv2's output; rec3.y = f(conv3 + r) = block output
    conv1: Optional[torch.nn.Module]  # the 1x1x1 conv with bias, None for nn.Identity
    se: Optional[dict] = None    # ResNetBlockSE: gate tensors saved by _se_fwd (the block output is se["out"])


@dataclass
class CkptRec:
    """an encoder block under activation checkpointing: only its input is kept, `_block_fwd` is re-run in backward"""

    name: str
    bm: torch.nn.Module
    x_in: torch.Tensor
    out: torch.Tensor  # the block output (alive anyway: skip connection / pool input); rewritten in place by the recomputation


@dataclass
class UpRec:
    """TransposeConvUpsampling + summation joining of one decoder (buildingblocks.py:617-664, :493)"""

    x_low: torch.Tensor
    weight: torch.Tensor  # (Cin, Cout, 3, 3, 3)
    los: tuple            # children tables of the nearest resize (2n-1 -> skip size)
    tdims: tuple          # (Dt, Ht, Wt)
    t8: bool = False      # ran in space-to-depth form on the bf16 kernels (csrc/u3d_bf16.hip)
    concat: Optional[tuple] = None  # explicit upsample='deconv' on a residual net: concat joining, (Cs skip, Ct upsampled) channels


_RECORD_TYPES.update({VSrc, ConvRec, Tape, ResRec, CkptRec, UpRec})
