"""Fused MI355X executor for the UNet3D family (DoubleConv blocks, 'gcr' order, max-pool down,
nearest-upsample + concat up, 1x1x1 head) — the hot path of pytorch3dunet/unet3d/model.py:123-149 and
buildingblocks.py:138-227,380-384,482-493 of the reference, run as hand-written gfx950 HIP kernels through the
C-ABI of include/u3d.h.

Design:
  * activations live in HBM as NDHWC fp32 torch tensors; skip tensors are never copied (virtual concat),
    the upsampled tensor is never materialised (nearest index maps), GroupNorm apply is fused into the conv
    A-tile load, ReLU and the next GroupNorm's statistics into the conv epilogue;
  * GroupNorm backward is `dx = p*dg + q*x + r` with per-(n,channel) coefficients, fused with the ReLU mask,
    the upsample-backward reduction and the max-pool scatter;
  * parameter gradients are written straight into one flat buffer whose [decoders|head] and [encoders] halves
    are all-reduced (RCCL) asynchronously while the encoder backward is still running (parallel.py).

The whole model is ONE torch.autograd.Function: PyTorch is used for memory (caching allocator), streams and
torch.distributed only.
"""
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


@dataclass
class Tape:
    convs: List[ConvRec] = field(default_factory=list)
    pools: list = field(default_factory=list)  # (pooled, argmax, e_in) per encoder level > 0
    head_x: Optional[torch.Tensor] = None
    dims: tuple = ()
    x0: Optional[torch.Tensor] = None
    blocks: list = field(default_factory=list)  # residual executor: ResRec per block (encoders, then decoders)
    ups: list = field(default_factory=list)     # residual executor: UpRec per decoder
    lean: bool = False      # memory-lean mode (checkpoint_encoders): backward releases every block's tensors as soon as it is done
    consumed: bool = False  # ... so the tape can be walked only once


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


_SIDE_STREAMS: dict = {}


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
            key = (self.dev.type, self.dev.index)
            st = _SIDE_STREAMS.get(key)
            if st is None:
                st = _SIDE_STREAMS[key] = torch.cuda.Stream(self.dev)
            self.side = st
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


class UNet3DEngine:
    """Executes the forward / backward of a UNet3D-family model natively.  Built once per model by
    `pytorch3dunet_amd.unet3d.model.AbstractUNet`; holds no tensors between calls except caches keyed on
    parameter versions (packed weights) and index maps."""

    def __init__(self, model):
        self.model = model
        self._pack_cache: dict = {}
        self.grad_sync = None  # set by parallel.GradSync (RCCL all-reduce overlapped with the encoder backward)
        self.debug = None  # dict -> backward stores clones of per-layer dz / dg (tools/gpu_layer_diag.py)
        self.fused_stats = True
        self.small_cin = True  # dedicated kernels for the in_channels<=4 first layer
        self.overlap_small_wgrad = True  # weight gradients of small layers on a second HIP stream (see _BwdCtx)
        # decoder first convs over an exact-2x upsampling: sub-pixel convolution of the upsampled half (csrc/u3d_subpix.hip)
        self.subpixel = os.environ.get("U3D_SUBPIXEL", "1") != "0"
        # opt-in (BASELINE config 4): bf16 MFMA operands with fp32 accumulation for the 3x3x3 convolutions whose channel
        # counts allow it (csrc/u3d_bf16.hip), fp32 master weights / activations / statistics; and recomputation of the
        # encoder blocks in backward instead of keeping their intermediates.  Set through the model
        # (`compute_dtype: bf16`, `checkpoint_encoders: true` in the YAML's model section, or U3D_BF16=1 / U3D_CHECKPOINT=1).
        self.bf16 = bool(getattr(model, "compute_bf16", False))
        # opt-in `compute_dtype: fp32_split`: FP32-grade convolutions on the bf16 matrix pipe — every fp32 operand split exactly
        # into three bf16 values, six partial products per multiply accumulated in fp32 (csrc/u3d_bf16.hip, u3d_conv3d_f32s);
        # forward and data gradients only, weight gradients stay on the fp32 MFMA kernels
        self.split = bool(getattr(model, "compute_split", False)) and not self.bf16
        self.checkpoint_encoders = bool(getattr(model, "checkpoint_encoders", False))
        # with activation checkpointing the tape is also RELEASED block by block during backward (ResUNetEngine.backward): a feature
        # whose only purpose is memory must move the peak, and with one autograd node owning the whole tape it otherwise does not
        self.lean_tape = False  # (ResUNetEngine turns it on together with checkpoint_encoders)
        # bf16 ACTIVATION STORAGE (`activation_dtype: bf16`; ResUNetEngine decides whether the model qualifies): every NDHWC
        # activation / gradient tensor between kernels is bf16, through the `_b16` entry points of include/u3d.h
        self.act_bf16 = False
        # id(conv weight) -> (C0, C1) of every decoder first conv (static); WHICH of them take the sub-pixel path depends on
        # the input size and is per-call state (`sub` argument / ConvRec.sub), never stored on the engine: forwards at
        # different sizes, other threads and nn.DataParallel replicas must not see each other's choice
        self._sub_pairs: dict = {}
        self._lock = threading.RLock()  # host-side enqueue of one forward / backward at a time per engine
        # opt-in static-shape step runner (`hip_graph: true` in the YAML's model section or U3D_GRAPH=1): the ~70 forward and ~110
        # backward launches of a TRAINING step are captured once per input shape in two hipGraphs and replayed (GraphStep below)
        self.hip_graph = bool(getattr(model, "hip_graph", False))
        self._graph_steps: dict = {}
        self._graph_off_reason = None
        self._placed = None  # check_placement's memo
        self._salt = 0  # advanced by every training forward: see _ver
        self._const: dict = {}
        # the model-wide layer order (every SingleConv of a DoubleConv net shares it): non-linearity of the layer outputs
        spec = parse_order(getattr(model, "layer_order", "gcr")) or (False, ACT_RELU, 0.0)
        self.post_norm, self.act, self.slope = spec
        self.mask = 1 if self.act == ACT_RELU else 0  # ReLU backward is a fused mask in the consumer kernels
        self.params = module_params(model)
        self._pids = [id(p) for p in self.params]
        # where the first parameter lives (model._get_engine's sentinel reads it back without walking the module tree)
        self._first_param_owner, self._first_param_name = next(
            ((mod, name) for mod in model.modules() for name, p in mod._parameters.items() if p is self.params[0]), (None, None))
        self._pindex = _PIndex({id(p): i for i, p in enumerate(self.params)})
        self._build_layer_table(model)
        self._virtual_w = self._virtual_weights()
        # split point of the flat gradient buffer: encoders first (module order), then decoders + head
        n_enc = sum(p.numel() for p in module_params(model.encoders))
        self.n_enc_params = n_enc
        # per-level offsets inside the encoder part [enc0 | enc1 | ...]: the encoder backward walks the levels deepest first, and the
        # deepest levels hold most of the parameters (config 4: 170 of 305 MB in the last one) — their gradients are final early and
        # are handed to RCCL level by level (`_enc_bucket_plan`)
        self.enc_level_offs = [0]
        for enc in model.encoders:
            self.enc_level_offs.append(self.enc_level_offs[-1] + sum(p.numel() for p in module_params(enc)))
        assert self.enc_level_offs[-1] == n_enc
        self.n_params = sum(p.numel() for p in self.params)
        offs, o = [], 0
        for p in self.params:
            offs.append(o)
            o += p.numel()
        self.poffs = offs

    # gradient buckets smaller than this are merged with the next (shallower) encoder level's: an all-reduce costs ~20-30 us of latency
    MIN_BUCKET_FLOATS = int(os.environ.get("U3D_MIN_BUCKET_MB", "1")) * (1 << 20) // 4

    def _sync_encoder_level(self, cx, flat, level: int, pending_hi: int) -> int:
        """Called by backward when encoder level `level` is done (levels run deepest first).  Hands the gradient slice
        [offs[level], pending_hi) to the exchange once it holds MIN_BUCKET_FLOATS (or level 0 is reached) and returns the new upper end
        of the not-yet-exchanged range.  With the decoder + head bucket that makes 2 + (number of big encoder levels) collectives per
        step; the last one is followed by `finish()`."""
        lo = self.enc_level_offs[level]
        if level > 0 and pending_hi - lo < self.MIN_BUCKET_FLOATS:
            return pending_hi
        if pending_hi > lo:
            cx.join()  # (a side-stream weight gradient of this level may still be writing its slice)
            self.grad_sync.launch(flat[lo:pending_hi])
        return lo

    def _virtual_weights(self):
        """ids of the conv weights whose input is a virtual concat (decoder first convs): fp32 kernels only"""
        return {id(c1.conv.weight) for c1, _ in self.dec}

    def _build_layer_table(self, model):
        self.enc = []
        for enc in model.encoders:
            bm = enc.basic_module
            self.enc.append((enc.pooling is not None, bm.SingleConv1, bm.SingleConv2))
        self.dec = []
        self.dec_up = []  # upsample='deconv' (buildingblocks.py:445-451): the decoder's ConvTranspose3d, else None (nearest)
        for dec in model.decoders:
            bm = dec.basic_module
            self.dec.append((bm.SingleConv1, bm.SingleConv2))
            self.dec_up.append(getattr(getattr(dec.upsampling, "upsample", None), "conv_transposed", None))
        # upsample='trilinear' / 'area' (buildingblocks.py:598-614): materialised by csrc/u3d_interp.hip, then a same-size concat
        self.dec_interp = [getattr(dec.upsampling, "mode", None) if getattr(dec.upsampling, "mode", None) in ("trilinear", "area")
                           else None for dec in model.decoders]

    # -- helpers ------------------------------------------------------------------------------------
    def _ver(self, w: torch.Tensor):
        """Cache key of a packed weight image.  Autograd's version counter sees optimizer steps, load_state_dict and every other
        tracked in-place update, but NOT writes through `param.data` (EMA swaps, hand-written updates): a TRAINING forward
        therefore always repacks (weights change every step anyway: `_salt` advances), an inference forward trusts version +
        storage pointer — after `param.data` edits in eval mode call model.invalidate_native_caches() (or set U3D_ALWAYS_REPACK=1)."""
        return (w._version, w.data_ptr(), self._salt)

    def begin_forward(self, training: bool):
        # the FIRST inference forward after a training forward also repacks: weights written through `param.data` while training
        # (EMA swap before validation, trainer-side weight surgery) are then picked up without anybody calling
        # invalidate_native_caches(); later inference forwards trust version + storage pointer again
        if training or _ALWAYS_REPACK or getattr(self, "_last_training", False):
            self._salt += 1
        self._last_training = training

    def _bf16_layer(self, Cin: int, Cout: int) -> bool:
        """forward AND data gradient of a (Cin -> Cout) 3x3x3 conv can run on the bf16 kernels (both directions need the
        contraction channels % 16 and the produced channels % 32)"""
        return self.bf16 and Cin % 32 == 0 and Cout % 32 == 0

    def _split_fwd(self, Cin: int, Cout: int) -> bool:
        return self.split and Cin % 16 == 0 and Cout % 32 == 0

    def _split_dgrad(self, Cin: int, Cout: int) -> bool:
        """data gradient of a (Cin -> Cout) conv: contraction over Cout, produces Cin channels"""
        return self.split and Cout % 16 == 0 and Cin % 32 == 0

    def _packed_f32s(self, w: torch.Tensor, mode: int, dev, Cin: Optional[int] = None, ci_off: int = 0) -> torch.Tensor:
        """three-image (high / middle / low bf16) fragment image of an fp32 weight, or of its input-channel slice
        [ci_off, ci_off + Cin) (u3d_pack_weights_f32s), cached per parameter version"""
        Cout, Ct = w.shape[0], w.shape[1]
        Cin = Ct if Cin is None else Cin
        key = (id(w), 30 + mode, Cin, ci_off)
        ver = self._ver(w)
        hit = self._pack_cache.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1]
        n = nat.get_lib().u3d_packed_weight_f32s_elems(Cin, Cout, mode)
        assert n > 0
        out = hit[1] if hit is not None and hit[1].numel() == n and hit[1].device == dev else _empty(
            n, dtype=torch.bfloat16, device=dev)
        nat.call("u3d_pack_weights_f32s", dev.index, _stream(dev), _p(w.detach()), Cout, Cin, mode, Ct, ci_off, _p(out))
        self._pack_cache[key] = (ver, out)
        return out

    def _packed_bf16(self, w: torch.Tensor, mode: int, dev) -> torch.Tensor:
        """bf16 fragment image of an fp32 master weight (u3d_pack_weights_bf16), cached per parameter version"""
        key = (id(w), 20 + mode)
        ver = self._ver(w)
        hit = self._pack_cache.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1]
        Cout, Cin = w.shape[0], w.shape[1]
        n = nat.get_lib().u3d_packed_weight_bf16_elems(Cin, Cout, mode)
        out = hit[1] if hit is not None and hit[1].numel() == n and hit[1].device == dev else _empty(
            n, dtype=torch.bfloat16, device=dev)
        nat.call("u3d_pack_weights_bf16", dev.index, _stream(dev), _p(w.detach()), Cout, Cin, mode, _p(out))
        self._pack_cache[key] = (ver, out)
        return out

    def _packed_convtr(self, w: torch.Tensor, mode: int, dev) -> torch.Tensor:
        """[tap][Cin][Cout] (mode 0) / [tap][Cout][Cin] (mode 1) image of a ConvTranspose3d weight, cached per version"""
        key = (id(w), 10 + mode)
        ver = self._ver(w)
        hit = self._pack_cache.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1]
        Cin, Cout = w.shape[0], w.shape[1]
        if mode == 2:  # fragment image of the sub-pixel forward kernel
            out = _empty(nat.get_lib().u3d_convtr3d_subpixel_packed_floats(Cin, Cout), dtype=_F32, device=dev)
            nat.call("u3d_pack_convtr3d_subpixel", dev.index, _stream(dev), _p(w.detach()), Cin, Cout, _p(out))
        else:
            out = _empty(27 * Cin * Cout, dtype=_F32, device=dev)
            nat.call("u3d_pack_convtr_weights", dev.index, _stream(dev), _p(w.detach()), Cin, Cout, mode, _p(out))
        self._pack_cache[key] = (ver, out)
        return out

    def _convtr_t8(self, Cl: int, Cs: int) -> bool:
        """the transposed convolution and its gradients run in space-to-depth form on the bf16 MFMA kernels"""
        return self.bf16 and nat.get_lib().u3d_convtr3d_t8_supported(Cl, Cs) == 1

    def _packed_convtr_t8(self, w: torch.Tensor, mode: int, dev) -> torch.Tensor:
        key = (id(w), 30 + mode)
        ver = self._ver(w)
        hit = self._pack_cache.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1]
        Cl, Cs = w.shape[0], w.shape[1]
        out = _empty(nat.get_lib().u3d_convtr3d_t8_packed_elems(Cl, Cs, mode), dtype=torch.bfloat16, device=dev)
        nat.call("u3d_pack_convtr3d_t8", dev.index, _stream(dev), _p(w.detach()), Cl, Cs, mode, _p(out))
        self._pack_cache[key] = (ver, out)
        return out

    def _conv_weights(self):
        """every 3x3x3 conv weight the MFMA kernels read through a packed image"""
        out = []
        for mod in self.model.modules():
            if isinstance(mod, torch.nn.Conv3d) and mod.kernel_size == (3, 3, 3):
                out.append(mod.weight)
        return out

    # pack modes: 0 forward, 1 data gradient (u3d_pack_weights).  Layers in self._sub (sub-pixel path) use instead: 10 / 11 =
    # forward / data-gradient image of the first C0 input channels, 12 / 13 = sub-pixel forward / data-gradient image of the
    # remaining C1 — and no mode-0 / mode-1 image.
    def _pack_shape(self, w, mode):
        """(w pointer, Cin, C-ABI mode, cin_stride, floats) of one packed image"""
        lib = nat.get_lib()
        Cout, Cin = w.shape[0], w.shape[1]
        if mode >= 10:
            C0, C1 = self._sub_pairs[id(w)]
            if mode in (10, 11):
                return w.data_ptr(), C0, mode - 10, Cin, lib.u3d_packed_weight_floats(C0, Cout, mode - 10)
            if mode == 12:
                return w.data_ptr() + C0 * 27 * 4, C1, 2, Cin, lib.u3d_subpixel_packed_floats(C1, Cout)
            return w.data_ptr() + C0 * 27 * 4, C1, 3, Cin, lib.u3d_subpixel_dgrad_packed_floats(Cout, C1)
        return w.data_ptr(), Cin, mode, 0, lib.u3d_packed_weight_floats(Cin, Cout, mode)

    def _repack_bf16_all(self, dev, modes, ws):
        """bf16 fragment images of every bf16 layer whose parameter changed: ONE launch at HBM rate (u3d_pack_weights_bf16_batch)
        instead of one strided-read launch per layer and mode (36 + 36 per config-4 step, 1.0 ms -> 0.25 ms)"""
        lib = nat.get_lib()
        stale = []
        for w in ws:
            if not self._bf16_layer(w.shape[1], w.shape[0]) or id(w) in self._virtual_w or w.data_ptr() % 16 != 0:
                continue  # (the batch kernel reads 16 bytes per lane; an unaligned view is packed on demand by _packed_bf16)
            for mode in modes:
                hit = self._pack_cache.get((id(w), 20 + mode))
                if hit is None or hit[0] != self._ver(w):
                    stale.append((w, mode))
        if not stale:
            return
        key = tuple((id(w), mode, w.data_ptr()) for w, mode in stale)
        tab = getattr(self, "_pack_tables_bf16", None)
        if tab is None:
            tab = self._pack_tables_bf16 = {}
        ent = tab.get(key)
        if ent is None:
            descs = (nat.U3DPackDesc * len(stale))()
            bufs, first = [], 0
            for i, (w, mode) in enumerate(stale):
                Cout, Cin = w.shape[0], w.shape[1]
                n = lib.u3d_packed_weight_bf16_elems(Cin, Cout, mode)
                hit = self._pack_cache.get((id(w), 20 + mode))
                buf = hit[1] if hit is not None and hit[1].numel() == n and hit[1].device == dev else _empty(
                    n, dtype=torch.bfloat16, device=dev)
                bufs.append(buf)
                descs[i].w, descs[i].packed, descs[i].first = w.data_ptr(), buf.data_ptr(), first
                descs[i].Cout, descs[i].Cin, descs[i].mode, descs[i].cin_stride = Cout, Cin, mode, 0
                first += lib.u3d_pack_weights_bf16_blocks(Cin, Cout, mode)
            host = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8)
            ent = (host.to(dev), bufs, first)
            tab.clear()
            tab[key] = ent
        table, bufs, total = ent
        nat.call("u3d_pack_weights_bf16_batch", dev.index, _stream(dev), _p(table), len(stale), total)
        for (w, mode), buf in zip(stale, bufs):
            self._pack_cache[(id(w), 20 + mode)] = (self._ver(w), buf)

    def _repack_all(self, dev, modes, sub=()):
        """(Re)pack the images of ALL conv weights whose parameter changed since the last pack — one launch for the whole
        model (u3d_pack_weights_batch) instead of one per layer and mode.  The packed buffers and the device descriptor
        table are allocated once and reused (stable pointers)."""
        ws = getattr(self, "_cw", None)
        if ws is None:
            ws = self._cw = self._conv_weights()
        if self.bf16:
            self._repack_bf16_all(dev, modes, ws)
        stale = []
        for w in ws:
            if self.small_cin and w.shape[1] <= 4 and w.shape[0] <= 32:
                continue  # first layer: dedicated kernels read the reference layout
            if self._bf16_layer(w.shape[1], w.shape[0]) and id(w) not in self._virtual_w:
                continue  # bf16 fragment images are packed on demand (_packed_bf16)
            wmodes = modes
            if id(w) in sub:
                wmodes = tuple(mm + 10 for mm in modes) + tuple(mm + 12 for mm in modes)
            for mode in wmodes:
                hit = self._pack_cache.get((id(w), mode))
                if hit is None or hit[0] != self._ver(w):
                    stale.append((w, mode))
        if not stale:
            return
        lib = nat.get_lib()
        key = tuple((id(w), mode, w.data_ptr()) for w, mode in stale)
        tab = getattr(self, "_pack_tables", None)
        if tab is None:
            tab = self._pack_tables = {}
        ent = tab.get(key)
        if ent is None:
            descs = (nat.U3DPackDesc * len(stale))()
            bufs, first = [], 0
            for i, (w, mode) in enumerate(stale):
                wptr, Cin, cmode, cstride, n = self._pack_shape(w, mode)
                hit = self._pack_cache.get((id(w), mode))
                buf = hit[1] if hit is not None and hit[1].numel() == n and hit[1].device == dev else _empty(
                    n, dtype=_F32, device=dev)
                bufs.append(buf)
                descs[i].w, descs[i].packed, descs[i].first = wptr, buf.data_ptr(), first
                descs[i].Cout, descs[i].Cin, descs[i].mode, descs[i].cin_stride = w.shape[0], Cin, cmode, cstride
                first += n
            host = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8)
            ent = (host.to(dev), bufs, first)
            tab.clear()  # one live table per (set of stale weights): parameters are re-packed together every step
            tab[key] = ent
        table, bufs, total = ent
        nat.call("u3d_pack_weights_batch", dev.index, _stream(dev), _p(table), len(stale), total)
        for (w, mode), buf in zip(stale, bufs):
            self._pack_cache[(id(w), mode)] = (self._ver(w), buf)

    def graph_pins(self) -> list:
        """every lazily built device buffer a captured step may dereference (GraphStep keeps this list alive): the pack descriptor
        tables with their packed images, the packed images in `_pack_cache`, the constant tables"""
        pins = [list(getattr(self, name, {}).values()) for name in ("_pack_tables", "_pack_tables_bf16")]
        pins.append([hit[1] for hit in self._pack_cache.values()])
        pins.append(list(self._const.values()))
        return pins

    def _packed_sub(self, rec: ConvRec, mode: int, dev) -> torch.Tensor:
        """packed image of a sub-pixel layer (modes 10..13); normally current from the forward's batch pack"""
        w = rec.conv_w
        hit = self._pack_cache.get((id(w), mode))
        if hit is None or hit[0] != self._ver(w):  # e.g. a no-grad forward packed only the forward images
            wptr, Cin, cmode, cstride, n = self._pack_shape(w, mode)
            buf = _empty(n, dtype=_F32, device=dev)
            desc = (nat.U3DPackDesc * 1)()
            desc[0].w, desc[0].packed, desc[0].first = wptr, buf.data_ptr(), 0
            desc[0].Cout, desc[0].Cin, desc[0].mode, desc[0].cin_stride = w.shape[0], Cin, cmode, cstride
            table = torch.frombuffer(bytearray(bytes(desc)), dtype=torch.uint8).to(dev)
            nat.call("u3d_pack_weights_batch", dev.index, _stream(dev), _p(table), 1, n)
            hit = (self._ver(w), buf)
            self._pack_cache[(id(w), mode)] = hit
        return hit[1]

    def _packed(self, w: torch.Tensor, mode: int, dev) -> torch.Tensor:
        key = (id(w), mode)
        ver = self._ver(w)
        hit = self._pack_cache.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1]
        Cout, Cin = w.shape[0], w.shape[1]
        n = nat.get_lib().u3d_packed_weight_floats(Cin, Cout, mode)
        out = _empty(n, dtype=_F32, device=dev)
        nat.call("u3d_pack_weights", dev.index, _stream(dev), _p(w.detach()), Cout, Cin, mode, _p(out))
        self._pack_cache[key] = (ver, out)
        return out

    def _identity_affine(self, N, C, dev):
        """(N,C,2) table a = 1, b = 0: the 'GroupNorm affine' of a conv input that has no GroupNorm (post-norm orders)"""
        key = ("ida", N, C, str(dev))
        t = self._const.get(key)
        if t is None:
            t = self._const[key] = torch.tensor([1.0, 0.0], dtype=_F32, device=dev).repeat(N, C, 1).contiguous()
        return t

    def _identity_coef(self, N, C, dev):
        """(N,3,C) table p = 1, q = 0, r = 0: GroupNorm backward of 'no GroupNorm' (dx = dg)"""
        key = ("idc", N, C, str(dev))
        t = self._const.get(key)
        if t is None:
            t = self._const[key] = torch.tensor([1.0, 0.0, 0.0], dtype=_F32, device=dev).view(1, 3, 1).repeat(N, 1, C).contiguous()
        return t

    def _unact(self, dev, g, y):
        """in place: gradient w.r.t. the activated tensor y -> gradient w.r.t. its pre-activation (LeakyReLU / ELU; ReLU is
        fused into the producing kernels as a mask, 'no activation' needs nothing)"""
        if self.act in (ACT_LEAKY, ACT_ELU):
            nat.call("u3d_act_bwd", dev.index, _stream(dev), _p(g), _p(y), g.numel(), self.act, self.slope, _p(g))

    def _up_scale(self, dev):
        """(1, 8, 8) on the (p, q, r) rows of a GroupNorm-backward coefficient table: a low-res voxel stands for 8 children"""
        t = getattr(self, "_up_scale_t", None)
        if t is None or t.device != dev:
            t = self._up_scale_t = torch.tensor([1.0, 8.0, 8.0], dtype=_F32, device=dev).view(1, 3, 1)
        return t

    def _subpixel_layers(self, size):
        """decoder first convs whose low-res input is upsampled by exactly 2 in every dimension at this input size:
        {id(weight): (C0, C1)} — per-call state, handed down as the `sub` argument"""
        if not self.subpixel or any(ct is not None for ct in self.dec_up) or any(self.dec_interp):
            return {}  # (a transposed convolution yields 2n-1 voxels, resized to the skip: never an exact 2x replication)
        dims = [tuple(size)]
        for has_pool, _, _ in self.enc:
            if has_pool:
                dims.append(tuple(d // 2 for d in dims[-1]))
        out = {}
        L = len(self.enc)
        for j, (c1, _) in enumerate(self.dec):
            skip_lvl, low_lvl = L - 2 - j, L - 1 - j
            if skip_lvl < 0 or low_lvl >= len(dims):
                continue
            C0 = self.enc[skip_lvl][2].conv.out_channels
            C1 = c1.conv.in_channels - C0
            if (all(a == 2 * b for a, b in zip(dims[skip_lvl], dims[low_lvl])) and C0 > 0 and C1 > 0 and C0 % 4 == 0
                    and C1 % 4 == 0 and c1.conv.out_channels % 4 == 0):
                out[id(c1.conv.weight)] = (C0, C1)
        self._sub_pairs.update(out)
        return out

    def _stats_of(self, src: VSrc, st0, st1, pool: _StatPool, dev):
        """(stats0, C0, scale0, stats1, C1, scale1) describing the per-channel sums of a (virtual) tensor"""
        if src.t1 is None:
            if st0 is None:
                st0 = pool.take(src.N * src.C0 * 2)
                s = src.struct()
                nat.call("u3d_chan_stats", dev.index, _stream(dev), ctypes.byref(s), src.N, src.D, src.H, src.W, _p(st0))
            return st0, src.C0, 1.0, None, 0, 0.0
        if st0 is not None and st1 is not None and src.exact2x and self.fused_stats:
            # every low-res voxel is replicated exactly 8x: reuse the producer's sums
            return st0, src.C0, 1.0, st1, src.C1, 8.0
        st = pool.take(src.N * src.C * 2)
        s = src.struct()
        nat.call("u3d_chan_stats", dev.index, _stream(dev), ctypes.byref(s), src.N, src.D, src.H, src.W, _p(st))
        return st, src.C, 1.0, None, 0, 0.0

    def _norm_finalize(self, kind, mod, st0, C0, sc0, st1, C1, sc1, N, G, count, affine, dev):
        """per-(n,c) sums -> the (a, b) table the convolutions / apply passes use; returns what backward needs (mean, rstd)"""
        if kind == "g":
            mean_rstd = _empty((N, G, 2), dtype=_F32, device=dev)
            nat.call("u3d_gn_finalize", dev.index, _stream(dev), _p(st0), C0, sc0, _p(st1), C1, sc1, N, G, count,
                     _p(mod.weight.detach()), _p(mod.bias.detach()), float(mod.eps), _p(affine), _p(mean_rstd))
            return mean_rstd
        # nn.BatchNorm3d (buildingblocks.py:78-88): batch statistics + running-estimate update in training, running statistics in eval
        C = C0 + C1
        training = bool(mod.training) or mod.running_mean is None
        mean_rstd = _empty((C, 2), dtype=_F32, device=dev)
        momentum = 0.0
        rm, rv = mod.running_mean, mod.running_var
        if training and rm is not None:
            if getattr(self, "_in_recompute", False):
                rm = rv = None  # activation checkpointing re-runs this forward in backward: the estimates were updated the first time
            else:
                mod.num_batches_tracked.add_(1)  # (ATen's batch_norm does the same before the kernel)
                momentum = (1.0 / float(mod.num_batches_tracked.item())) if mod.momentum is None else float(mod.momentum)
        nat.call("u3d_bn_finalize", dev.index, _stream(dev), _p(st0), C0, sc0, _p(st1), C1, sc1, N, count, _p(mod.weight.detach()),
                 _p(mod.bias.detach()), float(mod.eps), 1 if training else 0, momentum, _p(rm), _p(rv), _p(affine), _p(mean_rstd))
        return mean_rstd

    def _norm_bwd_finalize(self, cx, rec: ConvRec, gst, N, C, count, coef):
        dev, gview = cx.dev, cx.gview
        if rec.norm == "g":
            nat.call("u3d_gn_bwd_finalize", dev.index, _stream(dev), _p(gst), _p(rec.mean_rstd), _p(rec.gn_w.detach()), N, C, rec.G,
                     count, _p(gview(rec.idx_gw)), _p(gview(rec.idx_gb)), _p(coef))
        else:
            nat.call("u3d_bn_bwd_finalize", dev.index, _stream(dev), _p(gst), _p(rec.mean_rstd), _p(rec.gn_w.detach()), N, C, count,
                     1 if rec.bn_training else 0, _p(gview(rec.idx_gw)), _p(gview(rec.idx_gb)), _p(coef))

    def _single_conv_fwd(self, sc, name, src: VSrc, st_in, pool: _StatPool, tape: Optional[Tape], want_stats=True,
                         residual: Optional[torch.Tensor] = None, sub=(), y_out: Optional[torch.Tensor] = None, act=None):
        """One SingleConv (buildingblocks.py:99-135) in any native order (parse_order): 'gcr' = GroupNorm -> Conv3d -> ReLU fully
        fused; other non-linearities / GroupNorm after the conv add one bandwidth pass (csrc/u3d_act.hip).  With `residual`:
        f(conv(GN(x)) + residual), the tail of ResNetBlock.forward (buildingblocks.py:277-288; `act` = the block's f)."""
        dev = src.t0.device
        conv = sc.conv
        spec = layer_spec(sc.order)
        gn = getattr(sc, "groupnorm", None) if spec.norm == "g" else (getattr(sc, "batchnorm", None) if spec.norm == "b" else None)
        N, D, H, W = src.N, src.D, src.H, src.W
        Ctot, Cout = src.C, conv.out_channels
        G = gn.num_groups if spec.norm == "g" else 1
        post = not spec.pre  # the conv input has no norm of its own (post-norm and norm-free layers)
        act, slope = (spec.act, spec.slope) if act is None else act
        inner, islope = spec.inner, spec.islope  # 'crg' family: non-linearity on the conv output BEFORE its norm
        assert conv.in_channels == Ctot and (gn is None or getattr(gn, "num_channels", getattr(gn, "num_features", None)) == (Cout if post else Ctot))
        relu = 1 if ((act == ACT_RELU and not post) or inner == ACT_RELU) else 0
        # `out += residual` follows the block's last GroupNorm: inside the conv epilogue for pre-norm orders, in the
        # GroupNorm-apply pass for post-norm orders
        conv_res = None if post else residual
        # the conv epilogue's statistics describe the conv OUTPUT: they are the next GroupNorm's input only when nothing
        # else transforms it (ReLU is in the epilogue); a post-norm layer needs them for its own GroupNorm
        want_stats = ((post and spec.norm is not None and inner in (ACT_NONE, ACT_RELU))
                      or (not post and want_stats and act in (ACT_NONE, ACT_RELU)))
        # ONE flag for the finalize call (batch vs running statistics) and for backward (mean / rstd functions of x vs constants): a
        # BatchNorm3d without running estimates normalises with batch statistics in eval mode too (_norm_finalize)
        bn_training = (bool(gn.training) or gn.running_mean is None) if spec.norm == "b" else True
        if post:
            affine, mean_rstd = self._identity_affine(N, Ctot, dev), None
        else:
            st0, C0, sc0, st1, C1, sc1 = st_in
            affine = _empty((N, Ctot, 2), dtype=_F32, device=dev)
            mean_rstd = self._norm_finalize(spec.norm, gn, st0, C0, sc0, st1, C1, sc1, N, G, float(D * H * W), affine, dev)
        # y_out: recomputation under activation checkpointing rewrites the (still alive) block output in place with the
        # bit-identical values instead of allocating a second copy
        b16 = src.t0.dtype == torch.bfloat16  # bf16 activation storage: only the bf16-operand branch below handles it
        if b16:
            assert self.act_bf16 and src.t1 is None and not post and self._bf16_layer(Ctot, Cout) and act == ACT_RELU, \
                "bf16 activation storage reached a layer outside its envelope"
        y = y_out if (y_out is not None and not post) else _empty((N, D, H, W, Cout), dtype=src.t0.dtype if b16 else _F32, device=dev)
        small = self.small_cin and src.t1 is None and Ctot <= 4 and Cout <= 32 and residual is None and not b16
        if small:
            # first layer of the network: K = 27*Cin is too small for the MFMA tiling (csrc/u3d_smallc.hip)
            ystats = pool.take(N * Cout * 2) if (want_stats and self.fused_stats) else None
            nat.call("u3d_conv3d_small_cin_fwd", dev.index, _stream(dev), _p(src.t0), _p(affine), _p(conv.weight.detach()),
                     _p(y), N, D, H, W, Ctot, Cout, relu, _p(ystats), flops=54.0 * Ctot * Cout * N * D * H * W)
        elif src.t1 is not None and residual is None and id(conv.weight) in sub:
            # cat(skip, nearest2x(low)): the upsampled half as 8 parity-class 2x2x2 convolutions over the low-res tensor
            # (8/27 of the multiply-adds), then the skip half, whose epilogue adds the partial sums before ReLU / statistics
            C0, C1 = sub[id(conv.weight)]
            ystats = pool.take(N * Cout * 2) if (want_stats and self.fused_stats) else None
            part = _empty((N, D, H, W, Cout), dtype=_F32, device=dev)
            D1, H1, W1 = D // 2, H // 2, W // 2
            need = nat.get_lib().u3d_subpixel_fwd_workspace_floats(N, D1, H1, W1, C1, Cout)  # split-K scratch, small levels only
            kws = _empty(need, dtype=_F32, device=dev) if need > 0 else None
            nat.call("u3d_subpixel_conv_fwd", dev.index, _stream(dev), _p(src.t1), _p(affine.view(-1)[2 * C0:]), Ctot * 2,
                     _p(self._pack_cache[(id(conv.weight), 12)][1]), _p(part), N, D1, H1, W1, C1, Cout, _p(kws), need,
                     flops=128.0 * C1 * Cout * N * D1 * H1 * W1)
            a0 = affine[:, :C0].contiguous()
            if self._split_fwd(C0, Cout):
                nat.call("u3d_conv3d_f32s", dev.index, _stream(dev), _p(src.t0), _p(a0), _p(self._packed_f32s(conv.weight, 0, dev, C0, 0)),
                         _p(y), N, D, H, W, C0, Cout, relu, _p(ystats), None, None, _p(part), None, 0,
                         flops=54.0 * C0 * Cout * N * D * H * W)
            else:
                s0 = VSrc(src.t0).struct(a0)
                nat.call("u3d_conv3d_ex", dev.index, _stream(dev), ctypes.byref(s0), _p(self._pack_cache[(id(conv.weight), 10)][1]),
                         _p(y), N, D, H, W, Cout, relu, _p(ystats), None, None, _p(part), None, 0,
                         flops=54.0 * C0 * Cout * N * D * H * W)
        elif src.t1 is None and self._split_fwd(Ctot, Cout):
            # fp32 operands split into three bf16 values each, six partial products on the bf16 MFMA pipe (csrc/u3d_bf16.hip)
            ystats = pool.take(N * Cout * 2) if (want_stats and self.fused_stats) else None
            need = nat.get_lib().u3d_conv3d_bf16_workspace_floats(N, D, H, W, Ctot, Cout)
            kws = _empty(need, dtype=_F32, device=dev) if need > 0 else None
            nat.call("u3d_conv3d_f32s", dev.index, _stream(dev), _p(src.t0), _p(affine), _p(self._packed_f32s(conv.weight, 0, dev)),
                     _p(y), N, D, H, W, Ctot, Cout, relu, _p(ystats), None, None, _p(conv_res), _p(kws), need,
                     flops=54.0 * Ctot * Cout * N * D * H * W)
        elif src.t1 is None and self._bf16_layer(Ctot, Cout):
            # bf16 MFMA operands, fp32 accumulation / epilogue (csrc/u3d_bf16.hip); with bf16 activation storage the input, the
            # output and the residual are bf16 tensors (`_b16` entry point)
            ystats = pool.take(N * Cout * 2) if (want_stats and self.fused_stats) else None
            need = nat.get_lib().u3d_conv3d_bf16_workspace_floats(N, D, H, W, Ctot, Cout)  # split-K scratch at the bottom of the U
            kws = _empty(need, dtype=_F32, device=dev) if need > 0 else None
            nat.call("u3d_conv3d_bf16_ex" + ("_b16" if b16 else ""), dev.index, _stream(dev), _p(src.t0), _p(affine),
                     _p(self._packed_bf16(conv.weight, 0, dev)), _p(y), N, D, H, W, Ctot, Cout, relu, _p(ystats), None, None,
                     _p(conv_res), _p(kws), need, flops=54.0 * Ctot * Cout * N * D * H * W)
        else:
            wp = self._packed(conv.weight, 0, dev)
            ystats = pool.take(N * Cout * 2) if (want_stats and self.fused_stats) else None
            s = src.struct(affine)
            # bottom-of-the-U shapes split the channel reduction over blocks through a scratch buffer (0 floats otherwise)
            need = nat.get_lib().u3d_conv3d_workspace_floats(N, D, H, W, Ctot, Cout)
            kws = _empty(need, dtype=_F32, device=dev) if need > 0 else None
            nat.call("u3d_conv3d_ex", dev.index, _stream(dev), ctypes.byref(s), _p(wp), _p(y), N, D, H, W, Cout, relu,
                     _p(ystats), None, None, _p(conv_res), _p(kws), need, flops=54.0 * Ctot * Cout * N * D * H * W)
        post_rec = None
        if post:
            # GroupNorm over the conv output z (statistics from the conv epilogue), then the non-linearity: y = f(a*z + b);
            # 'crg' family: z is already f_inner(conv) (ReLU in the epilogue, LeakyReLU / ELU in place here)
            if inner in (ACT_LEAKY, ACT_ELU):
                nat.call("u3d_act_fwd", dev.index, _stream(dev), _p(y), y.numel(), inner, islope, _p(y))
            z, zst = y, ystats
            aff2 = _empty((N, Cout, 2), dtype=_F32, device=dev)
            if spec.norm is None:
                # no norm: the conv's bias (buildingblocks.py:54-55) is the constant affine (1, bias)
                nat.call("u3d_bias_table", dev.index, _stream(dev), _p(conv.bias.detach()), N, Cout, _p(aff2))
            else:
                if zst is None and (spec.norm == "g" or bn_training):
                    zst = self._stats_of(VSrc(z), None, None, pool, dev)[0]
                mean_rstd = self._norm_finalize(spec.norm, gn, zst, Cout, 1.0, None, 0, 0.0, N, G, float(D * H * W), aff2, dev)
            y = y_out if y_out is not None else _empty_like(z)
            nat.call("u3d_affine_add_act_fwd", dev.index, _stream(dev), _p(z), _p(aff2), _p(residual), N, D * H * W, Cout, act,
                     slope, _p(y))
            post_rec, ystats = (z, aff2, inner, islope), None
        elif act in (ACT_LEAKY, ACT_ELU):
            nat.call("u3d_act_fwd", dev.index, _stream(dev), _p(y), y.numel(), act, slope, _p(y))
            ystats = None
        drop_rec = None
        dmod = getattr(sc, "dropout", None) if spec.drop == "d" else (getattr(sc, "dropout2d", None) if spec.drop == "D" else None)
        if dmod is not None and dmod.training and dmod.p > 0.0:
            # The MASK comes from torch's generator exactly as the reference draws it (F.dropout on an NCDHW tensor of this
            # shape / feature_dropout's (N,C,1,1,1) noise: same Philox consumption, same element order), applied natively.
            if spec.drop == "d":
                m = F.dropout(torch.ones((N, Cout, D, H, W), dtype=_F32, device=dev), dmod.p, True)
                if Cout == 1:
                    mask = m.view(N, D, H, W, 1)
                else:
                    mask = _empty((N, D, H, W, Cout), dtype=_F32, device=dev)
                    nat.call("u3d_ncdhw_to_ndhwc", dev.index, _stream(dev), _p(m), _p(mask), N, Cout, D * H * W)
                nat.call("u3d_mul", dev.index, _stream(dev), _p(y), _p(mask), y.numel(), _p(y))
                drop_rec = ("d", mask)
            else:
                m = torch.feature_dropout(torch.ones((N, Cout, 1, 1, 1), dtype=_F32, device=dev), dmod.p, True).view(N, Cout)
                table = torch.stack((m, torch.zeros_like(m)), dim=-1).contiguous()
                nat.call("u3d_affine_act_fwd", dev.index, _stream(dev), _p(y), _p(table), N, D * H * W, Cout, ACT_NONE, 0.0, _p(y))
                drop_rec = ("D", table)
            ystats = None  # the epilogue's sums describe the tensor before the dropout
        if tape is not None:
            nw = gn.weight if gn is not None else None
            tape.convs.append(
                ConvRec(name, src, affine, mean_rstd, y, nw, conv.weight, G,
                        self._pindex[id(gn.weight)] if gn is not None else -1,
                        self._pindex[id(gn.bias)] if gn is not None else self._pindex[id(conv.bias)],
                        self._pindex[id(conv.weight)], small,
                        sub.get(id(conv.weight)) if (sub and src.t1 is not None and residual is None) else None,
                        not post, post_rec, spec.norm, bn_training, drop_rec)
            )
        return y, ystats

    # -- backward building blocks (shared by the DoubleConv and the residual executors) -----------------------
    def _conv_bwd(self, cx, rec: ConvRec, dz_, need_dg=True):
        """wgrad + dgrad + GroupNorm-backward reductions of one SingleConv; returns (dg, coef)"""
        dev, pool, ws, gview = cx.dev, cx.pool, cx.ws, cx.gview
        src = rec.src
        Nn, Dd, Hh, Ww = src.N, src.D, src.H, src.W
        Cout = rec.y.shape[-1]
        if rec.drop is not None:
            # trailing dropout: the consumers already removed f through the (rescaled, sign-preserving) layer output
            kind, mask = rec.drop
            g = _empty_like(dz_)
            if kind == "d":
                nat.call("u3d_mul", dev.index, _stream(dev), _p(dz_), _p(mask), dz_.numel(), _p(g))
            else:
                nat.call("u3d_affine_act_fwd", dev.index, _stream(dev), _p(dz_), _p(mask), Nn, Dd * Hh * Ww, Cout, ACT_NONE, 0.0, _p(g))
            dz_ = g
        if rec.post is not None:
            # post-norm layer: dz_ is the gradient w.r.t. n = a*z + b (the caller removed the non-linearity): norm backward
            # over the conv output z first — sums (sum dn, sum dn*z), parameter gradients, dz = p*dn + q*z + r
            z, _, inner, islope = rec.post
            Vz = Dd * Hh * Ww
            gst2 = pool.take(Nn * Cout * 2)
            nat.call("u3d_pair_stats", dev.index, _stream(dev), _p(dz_), _p(z), Nn, Vz, Cout, _p(gst2))
            if rec.norm is None:
                # norm-free layer: n = z + bias -> dbias = sum dn, dz = dn
                nat.call("u3d_bias_grad", dev.index, _stream(dev), _p(gst2), Nn, Cout, _p(gview(rec.idx_gb)))
            else:
                coef2 = _empty((Nn, 3, Cout), dtype=_F32, device=dev)
                self._norm_bwd_finalize(cx, rec, gst2, Nn, Cout, float(Vz), coef2)
                dz_ = self._plain_apply(cx, dz_, coef2, z, 1 if inner == ACT_RELU else 0)  # ('crg': z = relu(conv), mask fused)
                if inner in (ACT_LEAKY, ACT_ELU):
                    nat.call("u3d_act_bwd", dev.index, _stream(dev), _p(dz_), _p(z), dz_.numel(), inner, islope, _p(dz_))
        if self.debug is not None:
            self.debug[rec.name + ".dz"] = dz_.clone()
        if rec.small and not need_dg:
            # one pass gives dw and the GroupNorm-backward sums; no data gradient needed (csrc/u3d_smallc.hip)
            gst = pool.take(Nn * src.C * 2)
            nat.call("u3d_conv3d_small_cin_bwd", dev.index, _stream(dev), _p(src.t0), _p(rec.affine), _p(dz_),
                     _p(rec.conv_w.detach()), _p(gview(rec.idx_w)), _p(gst), Nn, Dd, Hh, Ww, src.C, Cout, _p(ws), ws.numel(),
                     flops=2 * 54.0 * src.C * Cout * Nn * Dd * Hh * Ww)
            if not rec.pre_norm:
                return None, self._identity_coef(Nn, src.C, dev)
            coef = _empty((Nn, 3, src.C), dtype=_F32, device=dev)
            self._norm_bwd_finalize(cx, rec, gst, Nn, src.C, float(Dd * Hh * Ww), coef)
            return None, coef
        s_aff = src.struct(rec.affine)
        flops = 54.0 * src.C * Cout * Nn * Dd * Hh * Ww
        bf16 = src.t1 is None and rec.sub is None and not rec.small and self._bf16_layer(src.C, Cout)
        b16 = src.t0.dtype == torch.bfloat16  # bf16 activation storage
        assert not b16 or (bf16 and Cout % 64 == 0 and dz_.dtype == torch.bfloat16)
        if bf16 and Cout % 64 == 0:
            need = nat.get_lib().u3d_wgrad_bf16_workspace_floats(Nn, Dd, Hh, Ww, src.C, Cout)
            ws = cx.ensure_ws(need)
            nat.call("u3d_conv3d_wgrad_bf16" + ("_b16" if b16 else ""), dev.index, _stream(dev), _p(src.t0), _p(rec.affine), _p(dz_),
                     _p(gview(rec.idx_w)), Nn, Dd, Hh, Ww, src.C, Cout, _p(ws), ws.numel(), flops=flops)
        elif rec.sub is not None:
            # weight gradient in two channel slices of the same (Cout, Ctot, 27) buffer: upsampled channels from the 64
            # (parity class, tap half) matrices over the low-res grid, skip channels from the standard kernel
            C0, C1 = rec.sub
            Ct = src.C
            dwv = gview(rec.idx_w)
            nat.call("u3d_subpixel_conv_wgrad", dev.index, _stream(dev), _p(src.t1), _p(rec.affine.view(-1)[2 * C0:]), Ct * 2,
                     _p(dz_), _p(dwv[C0 * 27:]), Ct, Nn, src.D1, src.H1, src.W1, C1, Cout, _p(ws), ws.numel(),
                     flops=128.0 * C1 * Cout * Nn * src.D1 * src.H1 * src.W1)
            a0 = rec.affine[:, :C0].contiguous()
            s0 = VSrc(src.t0).struct(a0)
            nat.call("u3d_conv3d_wgrad_strided", dev.index, _stream(dev), ctypes.byref(s0), _p(dz_), _p(dwv), Ct, Nn, Dd, Hh, Ww,
                     Cout, _p(ws), ws.numel(), flops=54.0 * C0 * Cout * Nn * Dd * Hh * Ww)
        elif self.overlap_small_wgrad and Nn * Dd * Hh * Ww <= cx.SIDE_MAX_VOXELS and self.debug is None:
            # small layer: neither kernel fills the chip on its own -> weight gradient on the side stream, data gradient
            # (below) on the caller's stream; joined before anything consumes the flat gradient buffer
            need = nat.get_lib().u3d_wgrad_workspace_floats(Nn, Dd, Hh, Ww, src.C, Cout)
            side = cx.side_stream(need)
            side.wait_stream(torch.cuda.current_stream(dev))  # dz_ (and the flat buffer) are ready
            with torch.cuda.stream(side):
                nat.call("u3d_conv3d_wgrad", dev.index, _stream(dev), ctypes.byref(s_aff), _p(dz_), _p(gview(rec.idx_w)), Nn,
                         Dd, Hh, Ww, Cout, _p(cx.ws_side), cx.ws_side.numel(), flops=flops)
            dz_.record_stream(side)  # dz_ is released on the main stream while the side stream may still read it
            cx.side_used = True
        else:
            nat.call("u3d_conv3d_wgrad", dev.index, _stream(dev), ctypes.byref(s_aff), _p(dz_), _p(gview(rec.idx_w)), Nn, Dd,
                     Hh, Ww, Cout, _p(ws), ws.numel(), flops=flops)
        s_dz = VSrc(dz_).struct()
        if rec.sub is not None:
            # skip half at full resolution; upsampled half directly at LOW resolution (the children sum of the nearest
            # upsampling is folded into the 4x4x4-tap stride-2 gather).  dg = (dg_skip, dlow)
            C0, C1 = rec.sub
            dg0 = _empty((Nn, Dd, Hh, Ww, C0), dtype=_F32, device=dev)
            dlow = _empty_like(src.t1)
            gst0, gst1 = pool.take(Nn * C0 * 2), pool.take(Nn * C1 * 2)
            if self._split_dgrad(C0, Cout):
                need = nat.get_lib().u3d_conv3d_bf16_workspace_floats(Nn, Dd, Hh, Ww, Cout, C0)
                kws = cx.ensure_ws(need) if need > 0 else None
                nat.call("u3d_conv3d_f32s", dev.index, _stream(dev), _p(dz_), None, _p(self._packed_f32s(rec.conv_w, 1, dev, C0, 0)),
                         _p(dg0), Nn, Dd, Hh, Ww, Cout, C0, 0, None, _p(src.t0), _p(gst0), None, _p(kws), need,
                         flops=54.0 * C0 * Cout * Nn * Dd * Hh * Ww)
            else:
                s_x0 = VSrc(src.t0).struct()
                nat.call("u3d_conv3d_ex", dev.index, _stream(dev), ctypes.byref(s_dz), _p(self._packed_sub(rec, 11, dev)), _p(dg0),
                         Nn, Dd, Hh, Ww, C0, 0, None, ctypes.byref(s_x0), _p(gst0), None, _p(ws), ws.numel(),
                         flops=54.0 * C0 * Cout * Nn * Dd * Hh * Ww)
            nat.call("u3d_subpixel_conv_dgrad", dev.index, _stream(dev), _p(dz_), _p(self._packed_sub(rec, 13, dev)), _p(src.t1),
                     _p(dlow), _p(gst1), Nn, src.D1, src.H1, src.W1, C1, Cout,
                     flops=128.0 * C1 * Cout * Nn * src.D1 * src.H1 * src.W1)
            gst = torch.cat((gst0.view(Nn, C0, 2), gst1.view(Nn, C1, 2)), dim=1)
            dg = (dg0, dlow)
        elif src.t1 is None and not rec.small and self._split_dgrad(src.C, Cout):
            dg = _empty((Nn, Dd, Hh, Ww, src.C), dtype=_F32, device=dev)
            gst = pool.take(Nn * src.C * 2)
            need = nat.get_lib().u3d_conv3d_bf16_workspace_floats(Nn, Dd, Hh, Ww, Cout, src.C)
            kws = cx.ensure_ws(need) if need > 0 else None
            nat.call("u3d_conv3d_f32s", dev.index, _stream(dev), _p(dz_), None, _p(self._packed_f32s(rec.conv_w, 1, dev)), _p(dg),
                     Nn, Dd, Hh, Ww, Cout, src.C, 0, None, _p(src.t0), _p(gst), None, _p(kws), need, flops=flops)
        elif bf16:
            dg = _empty((Nn, Dd, Hh, Ww, src.C), dtype=dz_.dtype if b16 else _F32, device=dev)
            gst = pool.take(Nn * src.C * 2)
            need = nat.get_lib().u3d_conv3d_bf16_workspace_floats(Nn, Dd, Hh, Ww, Cout, src.C)
            kws = cx.ensure_ws(need) if need > 0 else None
            nat.call("u3d_conv3d_bf16_ex" + ("_b16" if b16 else ""), dev.index, _stream(dev), _p(dz_), None,
                     _p(self._packed_bf16(rec.conv_w, 1, dev)), _p(dg), Nn, Dd, Hh, Ww, Cout, src.C, 0, None, _p(src.t0), _p(gst),
                     None, _p(kws), need, flops=flops)
        else:
            wpd = self._packed(rec.conv_w, 1, dev)
            dg = _empty((Nn, Dd, Hh, Ww, src.C), dtype=_F32, device=dev)
            gst = pool.take(Nn * src.C * 2)
            s_x = src.struct()
            nat.call("u3d_conv3d_ex", dev.index, _stream(dev), ctypes.byref(s_dz), _p(wpd), _p(dg), Nn, Dd, Hh, Ww, src.C, 0, None,
                     ctypes.byref(s_x), _p(gst), None, _p(ws), ws.numel(), flops=flops)
        if self.debug is not None and rec.sub is None:
            self.debug[rec.name + ".dg"] = dg.clone()
        if not rec.pre_norm:
            return dg, self._identity_coef(Nn, src.C, dev)  # no GroupNorm on the conv input: dx = dg
        coef = _empty((Nn, 3, src.C), dtype=_F32, device=dev)
        self._norm_bwd_finalize(cx, rec, gst, Nn, src.C, float(Dd * Hh * Ww), coef)
        return dg, coef

    def _plain_apply(self, cx, dg, coef, x, relu_mask, add=None):
        """GroupNorm backward, elementwise part: (p*dg + q*x + r [+ add]) * (relu_mask ? x > 0 : 1)"""
        dev = cx.dev
        out = _empty_like(x)
        Nn = x.shape[0]
        C = x.shape[-1]
        if x.dtype == torch.bfloat16:  # bf16 activation storage (dg, x, add, out all bf16)
            nat.call("u3d_gn_bwd_apply_b16", dev.index, _stream(dev), _p(dg), C, 0, _p(x), C, _p(coef), C, x.numel() // (Nn * C), Nn,
                     relu_mask, _p(add), _p(out))
            return out
        if add is None:
            nat.call("u3d_gn_bwd_apply", dev.index, _stream(dev), _p(dg), C, 0, _p(x), C, _p(coef), C, x.numel() // (Nn * C), Nn,
                     relu_mask, _p(out))
        else:
            nat.call("u3d_gn_bwd_apply_add", dev.index, _stream(dev), _p(dg), C, 0, _p(x), C, _p(coef), C,
                     x.numel() // (Nn * C), Nn, relu_mask, _p(add), _p(out))
        return out

    def _wgrad_workspace(self, tape, dev):
        return _empty(max(self._wgrad_workspace_floats(tape.convs), 4), dtype=_F32, device=dev)

    def _layer_ws_floats(self, N, D, H, W, Cin, Cout, sub=None, small=False, virtual=False):
        """scratch floats one 3x3x3 layer's backward needs from the shared buffer, for the kernels it will actually run"""
        lib = nat.get_lib()
        if small:
            return lib.u3d_small_cin_bwd_workspace_floats(N, D, H, W, Cin, Cout)
        if sub is not None:  # skip slice (fp32 kernels) + sub-pixel slice
            return max(lib.u3d_wgrad_workspace_floats(N, D, H, W, sub[0], Cout),
                       lib.u3d_subpixel_wgrad_workspace_floats(N, D // 2, H // 2, W // 2, sub[1], Cout),
                       lib.u3d_conv3d_workspace_floats(N, D, H, W, Cout, sub[0]))
        if not virtual and self._bf16_layer(Cin, Cout):
            need = lib.u3d_conv3d_bf16_workspace_floats(N, D, H, W, Cout, Cin)  # data gradient: roles swapped
            wg = lib.u3d_wgrad_bf16_workspace_floats(N, D, H, W, Cin, Cout) if Cout % 64 == 0 else lib.u3d_wgrad_workspace_floats(
                N, D, H, W, Cin, Cout)
            return max(need, wg)
        return max(lib.u3d_wgrad_workspace_floats(N, D, H, W, Cin, Cout), lib.u3d_conv3d_workspace_floats(N, D, H, W, Cout, Cin))

    def _wgrad_workspace_floats(self, convs):
        """scratch floats the backward kernels of these recorded layers need (one shared buffer, sized once per backward)"""
        need = 0
        for r in convs:
            need = max(need, self._layer_ws_floats(r.src.N, r.src.D, r.src.H, r.src.W, r.src.C, r.y.shape[-1], r.sub, r.small,
                                                   r.src.t1 is not None))
        return int(need)

    # -- forward ------------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, save: bool):
        """x: (N,C,D,H,W) fp32 on a gfx950 device.  Returns (logits, probs_or_None, tape_or_None), both
        outputs in the reference's NCDHW layout."""
        m = self.model
        dev = x.device
        N, Cin, D, H, W = x.shape
        x = x.contiguous()
        if Cin == 1:
            x0 = x.view(N, D, H, W, 1)  # NCDHW == NDHWC when C == 1
        else:
            x0 = _empty((N, D, H, W, Cin), dtype=_F32, device=dev)
            nat.call("u3d_ncdhw_to_ndhwc", dev.index, _stream(dev), _p(x), _p(x0), N, Cin, D * H * W)
        tape = Tape() if save else None
        if tape is not None:
            tape.x0 = x0
            tape.dims = (N, Cin, D, H, W)
        sub = self._subpixel_layers((D, H, W))
        self._repack_all(dev, (0, 1) if save else (0,), sub)
        # stat doubles: every conv output + every GN input computed standalone; generous upper bound
        tot = 0
        for _, c1, c2 in self.enc:
            tot += 4 * N * (c1.conv.in_channels + c1.conv.out_channels + c2.conv.out_channels) * 2
        for c1, c2 in self.dec:
            tot += 4 * N * (c1.conv.in_channels + c1.conv.out_channels + c2.conv.out_channels) * 2
        pool = _StatPool(dev, tot)

        feats = []  # (tensor, stats) of every encoder output
        cur, cur_st = x0, None
        for i, (has_pool, c1, c2) in enumerate(self.enc):
            if has_pool:
                Np, Dp, Hp, Wp, Cp = cur.shape
                pooled = _empty((Np, Dp // 2, Hp // 2, Wp // 2, Cp), dtype=_F32, device=dev)
                argmax = _empty(pooled.shape, dtype=torch.uint8, device=dev)
                pst = None if self.post_norm else pool.take(Np * Cp * 2)
                nat.call("u3d_maxpool2_fwd", dev.index, _stream(dev), _p(cur), Np, Dp, Hp, Wp, Cp, _p(pooled), _p(argmax),
                         _p(pst))
                if tape is not None:
                    tape.pools.append((pooled, argmax, cur))
                cur, cur_st = pooled, pst
            src = VSrc(cur)
            stats_of = (lambda *a: None) if self.post_norm else self._stats_of  # only a GroupNorm on the conv INPUT needs them
            y1, s1 = self._single_conv_fwd(c1, f"enc{i}.c1", src, stats_of(src, cur_st, None, pool, dev), pool, tape)
            src2 = VSrc(y1)
            y2, s2 = self._single_conv_fwd(c2, f"enc{i}.c2", src2, stats_of(src2, s1, None, pool, dev), pool, tape)
            feats.append((y2, s2))
            cur, cur_st = y2, s2

        skips = feats[:-1][::-1]  # model.py:126-133
        for j, ((c1, c2), (sk, sk_st)) in enumerate(zip(self.dec, skips)):
            ct = self.dec_up[j]
            if ct is not None:
                # upsample='deconv': ConvTranspose3d(k3, s2, p1) -> 2n-1 voxels (buildingblocks.py:617-664); the nearest resize
                # to the skip's size (:650-651) and the concat are virtual, like the interpolation path
                Nl, D1, H1, W1, Cl = cur.shape
                Cs = ct.out_channels
                t = _empty((Nl, 2 * D1 - 1, 2 * H1 - 1, 2 * W1 - 1, Cs), dtype=_F32, device=dev)
                if self.subpixel and Cl % 4 == 0 and Cs % 4 == 0:
                    nat.call("u3d_convtr3d_fwd_subpixel", dev.index, _stream(dev), _p(cur), _p(self._packed_convtr(ct.weight, 2, dev)),
                             _p(t), Nl, D1, H1, W1, Cl, Cs, flops=2.0 * 27 * Cl * Cs * Nl * D1 * H1 * W1)
                else:
                    nat.call("u3d_convtr3d_fwd", dev.index, _stream(dev), _p(cur), _p(ct.weight.detach()), _p(t), Nl, D1, H1, W1, Cl,
                             Cs, _p(self._packed_convtr(ct.weight, 0, dev)), flops=2.0 * 27 * Cl * Cs * Nl * D1 * H1 * W1)
                if tape is not None:
                    tape.ups.append(UpRec(cur, ct.weight, None, tuple(t.shape[1:4])))
                cur, cur_st = t, None
            elif self.dec_interp[j] is not None:
                # F.interpolate(mode='trilinear' | 'area') to the skip's size: a real tensor (2-tap separable gather), joined by
                # a same-size virtual concat
                Nl, D1, H1, W1, Cl = cur.shape
                _, Ds, Hs, Ws, _ = sk.shape
                tabs = [_resample_tables(dev, self.dec_interp[j], a, b) for a, b in ((D1, Ds), (H1, Hs), (W1, Ws))]
                up = _empty((Nl, Ds, Hs, Ws, Cl), dtype=_F32, device=dev)
                nat.call("u3d_resample2_fwd", dev.index, _stream(dev), _p(cur), _p(tabs[0][0]), _p(tabs[1][0]), _p(tabs[2][0]),
                         _p(tabs[0][1]), _p(tabs[1][1]), _p(tabs[2][1]), Nl, D1, H1, W1, Ds, Hs, Ws, Cl, _p(up))
                if tape is not None:
                    tape.ups.append(UpRec(cur, None, tabs, (Ds, Hs, Ws)))
                cur, cur_st = up, None
            src = VSrc(sk, cur)  # skip channels first (buildingblocks.py:491)
            y1, s1 = self._single_conv_fwd(c1, f"dec{j}.c1", src, stats_of(src, sk_st, cur_st, pool, dev), pool, tape,
                                           sub=sub)
            src2 = VSrc(y1)
            y2, s2 = self._single_conv_fwd(c2, f"dec{j}.c2", src2, stats_of(src2, s1, None, pool, dev), pool, tape)
            cur, cur_st = y2, s2

        # head: 1x1x1 conv + bias + activation (model.py:141-147), NCDHW outputs
        fc = m.final_conv
        Co, Cf = fc.out_channels, fc.in_channels
        V = D * H * W
        logits = _empty((N, Co, D, H, W), dtype=_F32, device=dev)
        act = 0
        probs = None
        if m.final_activation is not None:
            act = 1 if isinstance(m.final_activation, torch.nn.Sigmoid) else 2
            probs = _empty_like(logits)
        nat.call("u3d_conv1x1_head_fwd", dev.index, _stream(dev), _p(cur), _p(fc.weight.detach()), _p(fc.bias.detach()), N, V,
                 Cf, Co, act, _p(logits), _p(probs))
        if tape is not None:
            tape.head_x = cur
            if self.debug is not None:
                self.debug["tape"] = tape
        return logits, probs, tape

    # -- backward -----------------------------------------------------------------------------------
    def backward(self, tape: Tape, dlogits: torch.Tensor, need_input_grad: bool):
        """Returns (flat_grad, dx_or_None).  flat_grad holds every parameter gradient in module order."""
        m = self.model
        dev = dlogits.device
        N, Cin, D, H, W = tape.dims
        V = D * H * W
        dlogits = dlogits.contiguous()
        flat = _empty(self.n_params, dtype=_F32, device=dev)

        def gview(idx):
            p = self.params[idx]
            return flat[self.poffs[idx] : self.poffs[idx] + p.numel()]

        # zeroed double scratch: head (dw,db) + 2 doubles per (n, channel) per conv layer
        fc = m.final_conv
        Co, Cf = fc.out_channels, fc.in_channels
        tot = Co * Cf + Co + sum(N * r.src.C * 2 for r in tape.convs)
        pool = _StatPool(dev, tot)
        ws = self._wgrad_workspace(tape, dev)

        # ---- head backward: dz of the last decoder conv (ReLU mask fused)
        hacc = pool.take(Co * Cf + Co)
        dz = _empty_like(tape.head_x)
        nat.call("u3d_conv1x1_head_bwd", dev.index, _stream(dev), _p(dlogits), _p(tape.head_x), _p(fc.weight.detach()), N, V,
                 Cf, Co, self.mask, _p(dz), _p(hacc))
        self._unact(dev, dz, tape.head_x)
        iw, ib = self._pindex[id(fc.weight)], self._pindex[id(fc.bias)]
        assert self.poffs[ib] == self.poffs[iw] + Co * Cf
        nat.call("u3d_cvt_f64_f32", dev.index, _stream(dev), _p(hacc), _p(gview(iw)), Co * Cf + Co)

        n_levels = len(self.enc)
        n_dec = len(self.dec)
        mk = self.mask  # 1: the producers' ReLU masks are applied inside the consumer kernels; else _unact afterwards
        skip_grad = {}  # encoder level -> gradient arriving through the skip connection (pre-mask)

        cx = _BwdCtx(dev, pool, ws, flat, self)

        def conv_bwd(rec: ConvRec, dz_, need_dg=True):
            return self._conv_bwd(cx, rec, dz_, need_dg)

        def plain_apply(dg, coef, x, relu_mask):
            return self._plain_apply(cx, dg, coef, x, relu_mask)

        recs = tape.convs  # order: enc0.c1, enc0.c2, enc1.c1, ..., dec0.c1, dec0.c2, ...
        enc_recs = [(recs[2 * i], recs[2 * i + 1]) for i in range(n_levels)]
        dec_recs = [(recs[2 * n_levels + 2 * j], recs[2 * n_levels + 2 * j + 1]) for j in range(n_dec)]

        # ---- decoders, last to first
        for j in range(n_dec - 1, -1, -1):
            r1, r2 = dec_recs[j]
            dg2, coef2 = conv_bwd(r2, dz)
            dz1 = plain_apply(dg2, coef2, r2.src.t0, mk)  # r2.src.t0 is r1.y (post-activation)
            self._unact(dev, dz1, r2.src.t0)
            del dg2
            dg1, coef1 = conv_bwd(r1, dz1)
            src = r1.src
            C0, C1, Ct = src.C0, src.C1, src.C
            # skip half -> gradient of the encoder feature: its GroupNorm backward (p*dg + q*e + r on the first C0 channels)
            # is evaluated inside the max-pool merge kernel of that encoder level, never written to HBM
            lvl = n_levels - 2 - j
            dzl = _empty_like(src.t1)
            if r1.sub is not None:
                dg0, dlow = dg1
                skip_grad[lvl] = (dg0, C0, coef1, Ct)
                # dlow already holds the children sums: (p*dlow + 8*(q*x + r)) * (x > 0) on the low-res producer
                coef_up = coef1[:, :, C0:] * self._up_scale(dev)
                nat.call("u3d_gn_bwd_apply", dev.index, _stream(dev), _p(dlow), C1, 0, _p(src.t1), C1, _p(coef_up), C1,
                         src.D1 * src.H1 * src.W1, src.N, mk, _p(dzl))
                del dg0, dlow
            else:
                skip_grad[lvl] = (dg1, Ct, coef1, Ct)
                # upsampled half -> low-res producer (previous decoder's conv2 or the deepest encoder), ReLU mask fused
                lz, ly, lx = src.los
                nat.call("u3d_gn_bwd_apply_up", dev.index, _stream(dev), _p(dg1), Ct, C0, _p(src.t1), C1, _p(coef1), Ct, src.N,
                         src.D, src.H, src.W, src.D1, src.H1, src.W1, _p(lz), _p(ly), _p(lx),
                         0 if (self.dec_up[j] is not None or self.dec_interp[j] is not None) else mk, _p(dzl))
            del dg1
            if self.dec_up[j] is not None:
                # dzl is the gradient of the transposed convolution's (linear) output: its two gradients, with the non-linearity
                # of the tensor it upsampled
                up = tape.ups[j]
                xl = up.x_low
                Nl, D1, H1, W1, Cl = xl.shape
                Cs = up.weight.shape[1]
                acc = pool.take(up.weight.numel())
                dxl = _empty_like(xl)
                nat.call("u3d_convtr3d_bwd", dev.index, _stream(dev), _p(dzl), _p(xl), _p(up.weight.detach()), Nl, D1, H1, W1, Cl, Cs,
                         mk, _p(dxl), _p(acc), _p(self._packed_convtr(up.weight, 1, dev)), flops=4.0 * 27 * Cl * Cs * Nl * D1 * H1 * W1)
                nat.call("u3d_cvt_f64_f32", dev.index, _stream(dev), _p(acc), _p(gview(self._pindex[id(up.weight)])),
                         up.weight.numel())
                self._unact(dev, dxl, xl)
                dzl = dxl
            elif self.dec_interp[j] is not None:
                # dzl is the gradient of the interpolated (linear) tensor: the adjoint of the gather, then the non-linearity of
                # the tensor that was upsampled
                up = tape.ups[j]
                xl = up.x_low
                Nl, D1, H1, W1, Cl = xl.shape
                Ds, Hs, Ws = up.tdims
                tz, ty, tx = up.los
                dxl = _empty_like(xl)
                nat.call("u3d_resample2_bwd", dev.index, _stream(dev), _p(dzl), _p(tz[2]), _p(ty[2]), _p(tx[2]), _p(tz[0]), _p(ty[0]),
                         _p(tx[0]), _p(tz[1]), _p(ty[1]), _p(tx[1]), Nl, D1, H1, W1, Ds, Hs, Ws, Cl, _p(dxl))
                if self.act != ACT_NONE:
                    nat.call("u3d_act_bwd", dev.index, _stream(dev), _p(dxl), _p(xl), dxl.numel(), self.act, self.slope, _p(dxl))
                dzl = dxl
            else:
                self._unact(dev, dzl, src.t1)
            dz = dzl

        # decoder + head gradients are final: start their all-reduce now, overlapped with the encoder backward
        if self.grad_sync is not None:
            cx.join()
            self.grad_sync.launch(flat[self.n_enc_params :])

        # ---- encoders, deepest to first
        dx0 = None
        pending_hi = self.n_enc_params  # upper end of the encoder gradients not yet handed to the exchange
        for i in range(n_levels - 1, -1, -1):
            r1, r2 = enc_recs[i]
            dg2, coef2 = conv_bwd(r2, dz)
            dz1 = plain_apply(dg2, coef2, r2.src.t0, mk)
            self._unact(dev, dz1, r2.src.t0)
            del dg2
            dg1, coef1 = conv_bwd(r1, dz1, need_dg=(i > 0 or need_input_grad))
            if self.grad_sync is not None:
                pending_hi = self._sync_encoder_level(cx, flat, i, pending_hi)  # this level's parameter gradients are final
            if i > 0:
                pooled, argmax, e_in = tape.pools[i - 1]
                Ne, De, He, We, Ce = e_in.shape
                out = _empty_like(e_in)
                sk = skip_grad.pop(i - 1, None)
                if sk is None:
                    nat.call("u3d_maxpool2_bwd_merge", dev.index, _stream(dev), _p(dg1), _p(pooled), _p(argmax), _p(coef1), None,
                             _p(e_in), Ne, De, He, We, Ce, mk, _p(out))
                else:
                    sdg, sCdg, scoef, sCt = sk
                    nat.call("u3d_maxpool2_bwd_merge_gn", dev.index, _stream(dev), _p(dg1), _p(pooled), _p(argmax), _p(coef1),
                             _p(sdg), sCdg, _p(scoef), sCt, _p(e_in), Ne, De, He, We, Ce, mk, _p(out))
                    del sk, sdg, scoef
                self._unact(dev, out, e_in)
                dz = out
            elif need_input_grad:
                dx0 = plain_apply(dg1, coef1, tape.x0, 0)
            del dg1

        cx.join()
        if self.grad_sync is not None:
            self.grad_sync.finish()

        dx = None
        if dx0 is not None:
            if Cin == 1:
                dx = dx0.view(N, 1, D, H, W)
            else:
                dx = _empty((N, Cin, D, H, W), dtype=_F32, device=dev)
                nat.call("u3d_ndhwc_to_ncdhw", dev.index, _stream(dev), _p(dx0), _p(dx), N, Cin, V)
        return flat, dx


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


class ResUNetEngine(UNet3DEngine):
    """Native executor of ResidualUNet3D (model.py:193-234): ResNetBlock encoders (max-pool down), decoders that upsample
    with ConvTranspose3d(k3,s2,p1) -> nearest resize -> sum with the skip, then a ResNetBlock; same head.

    The 3x3x3 convolutions (94 % of the FLOPs at BASELINE config 4) run on the same MFMA kernels as UNet3D, with the
    block's `out += residual; ReLU` fused into conv3's epilogue (u3d_conv3d_residual); GroupNorm statistics of the
    residual come out of the 1x1x1 conv's / the joining kernel's epilogue.  The 1x1x1 convolutions and the transposed
    convolution run on the FP32 vector units (csrc/u3d_res.hip)."""

    def __init__(self, model):
        super().__init__(model)
        # two non-linearities per block: conv2's own (from the order string, nn defaults: LeakyReLU 0.01) and the block's final
        # one after `out += residual` (buildingblocks.py:270-275: LeakyReLU(0.1) if 'l', ELU if 'e', else ReLU).  self.act /
        # self.slope / self.mask describe the BLOCK outputs (what pooling, joining and the head consume).
        order = getattr(model, "layer_order", "gcr")
        self.act2, self.slope2 = self.act, self.slope
        self.act, self.slope = (ACT_LEAKY, 0.1) if "l" in order else ((ACT_ELU, 0.0) if "e" in order else (ACT_RELU, 0.0))
        self.mask = 1 if self.act == ACT_RELU else 0
        self.lean_tape = self.checkpoint_encoders and os.environ.get("U3D_LEAN_TAPE", "1") != "0"
        self.adt = _F32
        if bool(getattr(model, "activation_bf16", False)):
            why = self._act_bf16_blocker(model, order)
            if why is None:
                self.act_bf16, self.adt = True, torch.bfloat16
            elif getattr(model, "activation_dtype", "bf16") != "auto":
                import warnings

                warnings.warn(f"u3d: activation_dtype bf16 requested but {why}; activations stay fp32 in HBM", stacklevel=3)

    def _act_bf16_blocker(self, model, order) -> Optional[str]:
        """why this model cannot keep its activations in bf16 (None = it can): the `_b16` entry points cover the 'gcr' residual
        net whose every 3x3x3 / transposed convolution runs on the bf16 MFMA kernels"""
        lib = nat.get_lib()
        if not self.bf16:
            return "compute_dtype is not bf16"
        if order != "gcr":
            return f"layer_order '{order}' (only 'gcr')"
        if any(self.dec_concat):
            return "explicit upsample='deconv' (concat joining)"
        for _, bm in self.enc + [(None, b) for _, b in self.dec]:
            C = bm.conv2.conv.in_channels
            if getattr(bm, "se_module", None) is not None:
                return "squeeze-and-excitation blocks"
            if C % 64 != 0:
                return f"a block of {C} channels (multiples of 64: bf16 forward, data- and weight-gradient kernels)"
        for ct, _ in self.dec:
            if lib.u3d_convtr3d_t8_supported(ct.weight.shape[0], ct.weight.shape[1]) != 1:
                return f"a transposed convolution {ct.weight.shape[0]} -> {ct.weight.shape[1]} outside the space-to-depth kernels"
        fc = model.final_conv
        g = fc.in_channels // 4
        if fc.in_channels % 4 or g & (g - 1) or g > 64 or fc.out_channels > 4:
            return f"a head {fc.in_channels} -> {fc.out_channels} outside the vector kernels"
        return None

    def _virtual_weights(self):
        return set()  # summation joining: every 3x3x3 conv reads one real tensor

    def _build_layer_table(self, model):
        self.enc = [(e.pooling is not None, e.basic_module) for e in model.encoders]
        self.dec = [(d.upsampling.upsample.conv_transposed, d.basic_module) for d in model.decoders]
        # explicit upsample='deconv' (buildingblocks.py:435-468): concat joining and a 1x1x1 conv in the block instead of the sum
        self.dec_concat = [bool(getattr(d, "concat", False)) for d in model.decoders]

    # -- forward ------------------------------------------------------------------------------------
    def _block_fwd(self, bm, name, x_in, x_st, pool, tape, dev, y_out=None):
        N, D, H, W, Cin = x_in.shape
        Cout = bm.conv2.conv.in_channels
        conv1 = None if isinstance(bm.conv1, torch.nn.Identity) else bm.conv1
        if conv1 is None:
            r, r_st = x_in, x_st
            if r_st is None:
                r_st = pool.take(N * Cout * 2)
                sx = VSrc(r).struct()
                nat.call("u3d_chan_stats", dev.index, _stream(dev), ctypes.byref(sx), N, D, H, W, _p(r_st))
        else:
            r = _empty((N, D, H, W, Cout), dtype=self.adt, device=dev)
            r_st = pool.take(N * Cout * 2)
            w1 = conv1.weight.detach().view(Cout, Cin)
            if self.act_bf16 and x_in.dtype != _F32 and nat.get_lib().u3d_conv1x1_mfma_b16_supported(Cin, Cout):
                nat.call("u3d_conv1x1_fwd_mfma_b16", dev.index, _stream(dev), _p(x_in), _p(w1), _p(conv1.bias.detach()), _p(r), N,
                         D * H * W, Cin, Cout, _p(r_st), flops=2.0 * Cin * Cout * N * D * H * W)
            elif self.act_bf16:  # (the first block reads the fp32 network input)
                nat.call("u3d_conv1x1_fwd_b16", dev.index, _stream(dev), _p(x_in), 1 if x_in.dtype == _F32 else 0, _p(w1),
                         _p(conv1.bias.detach()), _p(r), N, D * H * W, Cin, Cout, _p(r_st), flops=2.0 * Cin * Cout * N * D * H * W)
            else:
                nat.call("u3d_conv1x1_fwd", dev.index, _stream(dev), _p(x_in), _p(w1), _p(conv1.bias.detach()), _p(r), N, D * H * W,
                         Cin, Cout, _p(r_st), flops=2.0 * Cin * Cout * N * D * H * W)
        n0 = len(tape.convs) if tape is not None else 0
        src2 = VSrc(r)
        out2, st2 = self._single_conv_fwd(bm.conv2, name + ".c2", src2, (r_st, Cout, 1.0, None, 0, 0.0), pool, tape)
        src3 = VSrc(out2)
        if st2 is None and not self.post_norm:  # conv2's epilogue sums do not describe its (LeakyReLU / ELU) output
            st2 = self._stats_of(src3, None, None, pool, dev)[0]
        se_mod = getattr(bm, "se_module", None)
        y, y_st = self._single_conv_fwd(bm.conv3, name + ".c3", src3, (st2, Cout, 1.0, None, 0, 0.0), pool, tape,
                                        want_stats=se_mod is not None, residual=r, y_out=y_out if se_mod is None else None,
                                        act=(self.act, self.slope))
        se = None
        out = y
        if se_mod is not None:
            if y_st is None:
                y_st = self._stats_of(VSrc(y), None, None, pool, dev)[0]
            se = self._se_fwd(se_mod, y, y_st, dev)
            out = se["out"]
        if tape is not None:
            tape.blocks.append(ResRec(name, x_in, r, tape.convs[n0], tape.convs[n0 + 1], conv1, se))
        return out

    @staticmethod
    def _se_parts(se_mod):
        """(mode, cSE-or-None, sSE-or-None): 0 scSE, 1 cSE, 2 sSE (buildingblocks.py:298-307)"""
        if hasattr(se_mod, "cSE"):
            return 0, se_mod.cSE, se_mod.sSE
        if hasattr(se_mod, "fc1"):
            return 1, se_mod, None
        return 2, None, se_mod

    def _se_fwd(self, se_mod, y, y_st, dev):
        """squeeze-and-excitation gate on a block output (se.py:18-114): out = y * max(gc[n,c], a[n,v])"""
        N, D, H, W, C = y.shape
        V = D * H * W
        mode, cse, sse = self._se_parts(se_mod)
        st = {"mode": mode, "y": y, "cse": cse, "sse": sse, "gc": None, "a": None}
        if cse is not None:
            Cr = cse.fc1.out_features
            st["s"] = _empty((N, C), dtype=_F32, device=dev)
            st["h"] = _empty((N, Cr), dtype=_F32, device=dev)
            st["gc"] = _empty((N, C), dtype=_F32, device=dev)
            nat.call("u3d_se_gate_fwd", dev.index, _stream(dev), _p(y_st), float(V), _p(cse.fc1.weight.detach()),
                     _p(cse.fc1.bias.detach()), _p(cse.fc2.weight.detach()), _p(cse.fc2.bias.detach()), N, C, Cr, _p(st["s"]),
                     _p(st["h"]), _p(st["gc"]))
        ws = bs = None
        if sse is not None:
            ws, bs = sse.conv.weight.detach().view(C), sse.conv.bias.detach()
            st["a"] = _empty((N * V,), dtype=_F32, device=dev)
        out = _empty_like(y)
        nat.call("u3d_se_apply_fwd", dev.index, _stream(dev), _p(y), _p(st["gc"]), _p(ws), _p(bs), N, V, C, mode, _p(out),
                 _p(st["a"]))
        st["out"] = out
        return st

    def _se_bwd(self, cx, se, dout):
        """gradient of the block's pre-ReLU sum from the gradient of the gated output (masked by y > 0)"""
        dev, pool, gview = cx.dev, cx.pool, cx.gview
        y = se["y"]
        N, D, H, W, C = y.shape
        V = D * H * W
        mode, cse, sse = se["mode"], se["cse"], se["sse"]
        acc_gc = pool.take(N * C) if cse is not None else None
        acc_ws = pool.take(C + 1) if sse is not None else None
        dls = _empty((N * V,), dtype=_F32, device=dev) if sse is not None else None
        ws = sse.conv.weight.detach().view(C) if sse is not None else None
        nat.call("u3d_se_bwd_reduce", dev.index, _stream(dev), _p(dout), _p(y), _p(se["gc"]), _p(se["a"]), _p(ws), N, V, C, mode,
                 _p(dls), _p(acc_gc), _p(acc_ws))
        ds = None
        if cse is not None:
            Cr = cse.fc1.out_features
            dz2 = _empty((N, C), dtype=_F32, device=dev)
            dz1 = _empty((N, Cr), dtype=_F32, device=dev)
            ds = _empty((N, C), dtype=_F32, device=dev)
            ix = [self._pindex[id(p)] for p in (cse.fc1.weight, cse.fc1.bias, cse.fc2.weight, cse.fc2.bias)]
            nat.call("u3d_se_gate_bwd", dev.index, _stream(dev), _p(acc_gc), _p(se["gc"]), _p(se["h"]), _p(se["s"]),
                     _p(cse.fc1.weight.detach()), _p(cse.fc2.weight.detach()), N, C, Cr, float(V), _p(dz2), _p(dz1), _p(ds),
                     _p(gview(ix[0])), _p(gview(ix[1])), _p(gview(ix[2])), _p(gview(ix[3])))
        if sse is not None:
            jw, jb = self._pindex[id(sse.conv.weight)], self._pindex[id(sse.conv.bias)]
            assert self.poffs[jb] == self.poffs[jw] + C
            nat.call("u3d_cvt_f64_f32", dev.index, _stream(dev), _p(acc_ws), _p(gview(jw)), C + 1)
        m_ = _empty_like(y)
        nat.call("u3d_se_bwd_apply", dev.index, _stream(dev), _p(dout), _p(y), _p(se["gc"]), _p(se["a"]), _p(ws), _p(dls), _p(ds),
                 N, V, C, mode, self.mask, _p(m_))
        return m_

    def forward(self, x: torch.Tensor, save: bool):
        m = self.model
        dev = x.device
        N, Cin, D, H, W = x.shape
        x = x.contiguous()
        if Cin == 1:
            x0 = x.view(N, D, H, W, 1)
        else:
            x0 = _empty((N, D, H, W, Cin), dtype=_F32, device=dev)
            nat.call("u3d_ncdhw_to_ndhwc", dev.index, _stream(dev), _p(x), _p(x0), N, Cin, D * H * W)
        tape = Tape() if save else None
        if tape is not None:
            tape.x0 = x0
            tape.dims = (N, Cin, D, H, W)
            tape.blocks = []
            tape.ups = []
        self._repack_all(dev, (0, 1) if save else (0,))
        widths = [bm.conv2.conv.in_channels for _, bm in self.enc]
        pool = _StatPool(dev, 16 * N * sum(widths) * 2 + 64)

        feats = []
        cur = x0
        for i, (has_pool, bm) in enumerate(self.enc):
            if has_pool:
                Np, Dp, Hp, Wp, Cp = cur.shape
                pooled = _empty((Np, Dp // 2, Hp // 2, Wp // 2, Cp), dtype=self.adt, device=dev)
                argmax = _empty(pooled.shape, dtype=torch.uint8, device=dev)
                if self.act_bf16:
                    nat.call("u3d_maxpool2_fwd_b16", dev.index, _stream(dev), _p(cur), Np, Dp, Hp, Wp, Cp, _p(pooled), _p(argmax))
                else:
                    nat.call("u3d_maxpool2_fwd", dev.index, _stream(dev), _p(cur), Np, Dp, Hp, Wp, Cp, _p(pooled), _p(argmax),
                             None)
                if tape is not None:
                    tape.pools.append((pooled, argmax, cur))
                cur = pooled
            if tape is not None and self.checkpoint_encoders:
                # activation checkpointing of the encoder blocks (BASELINE config 4): keep the block input only
                x_in = cur
                cur = self._block_fwd(bm, f"enc{i}", cur, None, pool, None, dev)
                tape.blocks.append(CkptRec(f"enc{i}", bm, x_in, cur))
            else:
                cur = self._block_fwd(bm, f"enc{i}", cur, None, pool, tape, dev)
            feats.append(cur)

        skips = feats[:-1][::-1]
        for j, ((ct, bm), sk) in enumerate(zip(self.dec, skips)):
            Nl, D1, H1, W1, Cl = cur.shape
            _, Ds, Hs, Ws, Cs = sk.shape
            Dt, Ht, Wt = 2 * D1 - 1, 2 * H1 - 1, 2 * W1 - 1
            concat = self.dec_concat[j]
            Ct = ct.out_channels  # (== Cs for summation joining)
            t8 = self._convtr_t8(Cl, Cs) and not concat
            (mz, lz), (my, ly), (mx, lx) = _maps(dev, Dt, Ds), _maps(dev, Ht, Hs), _maps(dev, Wt, Ws)
            joined = _empty((Nl, Ds, Hs, Ws, Cs + Ct), dtype=_F32, device=dev) if concat else _empty_like(sk)
            j_st = None if concat else pool.take(Nl * Cs * 2)
            if t8:
                # bf16 mode: 2x2x2 convolution on the low-res grid into the space-to-depth layout T8[i][parity*Cs + c] = t[2i + parity];
                # the resize + join reads that layout directly
                sfx = "_b16" if self.act_bf16 else ""
                t = _empty((Nl, D1, H1, W1, 8 * Cs), dtype=self.adt, device=dev)
                nat.call("u3d_convtr3d_fwd_t8" + sfx, dev.index, _stream(dev), _p(cur), _p(self._packed_convtr_t8(ct.weight, 0, dev)),
                         _p(t), Nl, D1, H1, W1, Cl, Cs, flops=2.0 * 27 * Cl * Cs * Nl * D1 * H1 * W1)
                nat.call("u3d_nearest_add_fwd_t8" + sfx, dev.index, _stream(dev), _p(sk), _p(t), _p(mz), _p(my), _p(mx), Nl, Ds, Hs,
                         Ws, Dt, Ht, Wt, Cs, _p(joined), _p(j_st))
                del t
                if tape is not None:
                    tape.ups.append(UpRec(cur, ct.weight, (lz, ly, lx), (Dt, Ht, Wt), True))
                cur = self._block_fwd(bm, f"dec{j}", joined, j_st, pool, tape, dev)
                continue
            t = _empty((Nl, Dt, Ht, Wt, Ct), dtype=_F32, device=dev)
            if self.subpixel and Cl % 4 == 0 and Ct % 4 == 0:
                # 8 output parity classes accumulated from one staged input halo tile (csrc/u3d_subpix.hip, scheme Deconv3s2)
                nat.call("u3d_convtr3d_fwd_subpixel", dev.index, _stream(dev), _p(cur), _p(self._packed_convtr(ct.weight, 2, dev)),
                         _p(t), Nl, D1, H1, W1, Cl, Ct, flops=2.0 * 27 * Cl * Ct * Nl * D1 * H1 * W1)
            else:
                nat.call("u3d_convtr3d_fwd", dev.index, _stream(dev), _p(cur), _p(ct.weight.detach()), _p(t), Nl, D1, H1, W1, Cl,
                         Ct, _p(self._packed_convtr(ct.weight, 0, dev)), flops=2.0 * 27 * Cl * Ct * Nl * D1 * H1 * W1)
            if concat:
                nat.call("u3d_nearest_cat_fwd", dev.index, _stream(dev), _p(sk), _p(t), _p(mz), _p(my), _p(mx), Nl, Ds, Hs, Ws, Dt, Ht,
                         Wt, Cs, Ct, _p(joined))
            else:
                nat.call("u3d_nearest_add_fwd", dev.index, _stream(dev), _p(sk), _p(t), _p(mz), _p(my), _p(mx), Nl, Ds, Hs, Ws, Dt, Ht,
                         Wt, Cs, _p(joined), _p(j_st))
            del t
            if tape is not None:
                tape.ups.append(UpRec(cur, ct.weight, (lz, ly, lx), (Dt, Ht, Wt), False, (Cs, Ct) if concat else None))
            cur = self._block_fwd(bm, f"dec{j}", joined, j_st, pool, tape, dev)

        fc = m.final_conv
        Co, Cf = fc.out_channels, fc.in_channels
        V = D * H * W
        logits = _empty((N, Co, D, H, W), dtype=_F32, device=dev)
        act = 0
        probs = None
        if m.final_activation is not None:
            act = 1 if isinstance(m.final_activation, torch.nn.Sigmoid) else 2
            probs = _empty_like(logits)
        nat.call("u3d_conv1x1_head_fwd" + ("_b16" if self.act_bf16 else ""), dev.index, _stream(dev), _p(cur), _p(fc.weight.detach()),
                 _p(fc.bias.detach()), N, V, Cf, Co, act, _p(logits), _p(probs))
        if tape is not None:
            tape.head_x = cur
            if self.debug is not None:
                self.debug["tape"] = tape
        return logits, probs, tape

    # -- backward -----------------------------------------------------------------------------------
    def _block_bwd(self, cx, rec: ResRec, m_):
        """m_ = dL/d(block output); for ReLU blocks the producers already masked it by (output > 0) (self.mask), other
        non-linearities are removed here through the block's pre-gate output y = f(sum).  Returns dL/d(residual r)."""
        dev = cx.dev
        if rec.se is not None:
            m_ = self._se_bwd(cx, rec.se, m_)
        self._unact(dev, m_, rec.rec3.y)  # -> gradient of (conv3 branch + residual)
        dg3, coef3 = self._conv_bwd(cx, rec.rec3, m_)
        o2 = rec.rec3.src.t0
        dz2 = self._plain_apply(cx, dg3, coef3, o2, 1 if self.act2 == ACT_RELU else 0)  # through conv2's non-linearity
        if self.act2 in (ACT_LEAKY, ACT_ELU):
            nat.call("u3d_act_bwd", dev.index, _stream(dev), _p(dz2), _p(o2), dz2.numel(), self.act2, self.slope2, _p(dz2))
        del dg3
        dg2, coef2 = self._conv_bwd(cx, rec.rec2, dz2)
        del dz2
        # r feeds conv2's GroupNorm AND the `out += residual` shortcut; r itself is linear (no ReLU mask)
        return self._plain_apply(cx, dg2, coef2, rec.r, 0, add=m_)

    def _conv1_bwd(self, cx, rec: ResRec, dr, need_dx: bool):
        """the block's 1x1x1 conv with bias (buildingblocks.py:248-255): parameter gradients, and dL/d(block input) if wanted"""
        dev, pool, gview = cx.dev, cx.pool, cx.gview
        c1 = rec.conv1
        Cout_, Cin_ = c1.weight.shape[0], c1.weight.shape[1]
        xin = rec.x_in
        dxin = _empty(xin.shape, dtype=dr.dtype, device=dev) if need_dx else None
        jw, jb = self._pindex[id(c1.weight)], self._pindex[id(c1.bias)]
        assert self.poffs[jb] == self.poffs[jw] + Cout_ * Cin_
        if self.act_bf16 and xin.dtype != _F32 and nat.get_lib().u3d_conv1x1_mfma_b16_supported(Cin_, Cout_):
            Nn, Dd, Hh, Ww = xin.shape[:4]
            need = nat.get_lib().u3d_conv1x1_bwd_mfma_b16_workspace_floats(Nn, Dd, Hh, Ww, Cin_, Cout_)
            ws = cx.ensure_ws(need)
            nat.call("u3d_conv1x1_bwd_mfma_b16", dev.index, _stream(dev), _p(dr), _p(xin), _p(c1.weight.detach().view(Cout_, Cin_)),
                     Nn, Dd, Hh, Ww, Cin_, Cout_, _p(dxin), _p(gview(jw)), _p(gview(jb)), _p(ws), ws.numel(),
                     flops=(4.0 if need_dx else 2.0) * Cin_ * Cout_ * (xin.numel() // Cin_))
            return dxin
        acc = pool.take(Cout_ * Cin_ + Cout_)
        if self.act_bf16:
            nat.call("u3d_conv1x1_bwd_b16", dev.index, _stream(dev), _p(dr), _p(xin), 1 if xin.dtype == _F32 else 0,
                     _p(c1.weight.detach().view(Cout_, Cin_)), xin.shape[0], xin.numel() // (xin.shape[0] * Cin_), Cin_, Cout_,
                     _p(dxin), _p(acc), flops=(4.0 if need_dx else 2.0) * Cin_ * Cout_ * (xin.numel() // Cin_))
        else:
            nat.call("u3d_conv1x1_bwd", dev.index, _stream(dev), _p(dr), _p(xin), _p(c1.weight.detach().view(Cout_, Cin_)),
                     xin.shape[0], xin.numel() // (xin.shape[0] * Cin_), Cin_, Cout_, _p(dxin), _p(acc),
                     flops=(4.0 if need_dx else 2.0) * Cin_ * Cout_ * (xin.numel() // Cin_))
        nat.call("u3d_cvt_f64_f32", dev.index, _stream(dev), _p(acc), _p(gview(jw)), Cout_ * Cin_ + Cout_)
        return dxin

    def backward(self, tape: Tape, dlogits: torch.Tensor, need_input_grad: bool):
        m = self.model
        dev = dlogits.device
        N, Cin, D, H, W = tape.dims
        V = D * H * W
        dlogits = dlogits.contiguous()
        flat = _empty(self.n_params, dtype=_F32, device=dev)
        fc = m.final_conv
        Co, Cf = fc.out_channels, fc.in_channels
        tot = Co * Cf + Co + sum(N * r.src.C * 2 for r in tape.convs)
        for b in tape.blocks:
            if isinstance(b, ResRec) and b.conv1 is not None:
                tot += b.conv1.weight.numel() + b.conv1.bias.numel()
        for u in tape.ups:
            tot += u.weight.numel()
        for b in tape.blocks:
            if isinstance(b, ResRec) and b.se is not None:
                tot += (N + 1) * b.se["y"].shape[-1] + 1
        pool = _StatPool(dev, tot)
        need = self._wgrad_workspace_floats(tape.convs)
        for b in tape.blocks:
            if isinstance(b, CkptRec):  # recomputed in backward: two (C -> C) convolutions at the block's resolution
                Nb, Db, Hb, Wb, _ = b.x_in.shape
                Cb = b.bm.conv2.conv.in_channels
                need = max(need, self._layer_ws_floats(Nb, Db, Hb, Wb, Cb, Cb))
        b = u = None  # (loop variables would pin the LAST decoder block — the full-resolution one — for the whole backward)
        ws = _empty(max(int(need), 4), dtype=_F32, device=dev)
        cx = _BwdCtx(dev, pool, ws, flat, self)
        gview = cx.gview

        hacc = pool.take(Co * Cf + Co)
        dz = _empty_like(tape.head_x)
        mk = self.mask  # ReLU blocks: the consumers' backward kernels mask by (block output > 0); else _block_bwd removes f
        nat.call("u3d_conv1x1_head_bwd" + ("_b16" if self.act_bf16 else ""), dev.index, _stream(dev), _p(dlogits), _p(tape.head_x),
                 _p(fc.weight.detach()), N, V, Cf, Co, mk, _p(dz), _p(hacc))
        iw, ib = self._pindex[id(fc.weight)], self._pindex[id(fc.bias)]
        assert self.poffs[ib] == self.poffs[iw] + Co * Cf
        nat.call("u3d_cvt_f64_f32", dev.index, _stream(dev), _p(hacc), _p(gview(iw)), Co * Cf + Co)

        n_levels, n_dec = len(self.enc), len(self.dec)
        enc_blocks, dec_blocks = tape.blocks[:n_levels], tape.blocks[n_levels:]
        skip_grad = {}
        lean = tape.lean
        if lean:
            # memory-lean mode: this walk is the tape's only one — every block's activations are dropped as soon as its backward
            # is queued (the caching allocator hands the memory to the next block's temporaries in stream order), so the peak is
            # one level's working set on top of what is still to be walked, not the whole tape
            tape.consumed = True
            tape.convs, tape.blocks, tape.head_x = [], [], None
            ups, pools = tape.ups, tape.pools
            tape.ups, tape.pools = [], []
        else:
            ups, pools = tape.ups, tape.pools

        for j in range(n_dec - 1, -1, -1):
            rec, up = dec_blocks[j], ups[j]
            if lean:
                dec_blocks[j] = ups[j] = None
            dj = self._block_bwd(cx, rec, dz)  # gradient of the block's residual r (= the joined tensor when conv1 is nn.Identity)
            dz = None
            if up.concat is not None:
                # concat joining: through the block's 1x1x1 conv, then split into the skip's and the resized tensor's gradient
                Cs_, Ct_ = up.concat
                dcat = self._conv1_bwd(cx, rec, dj, True)
                d_skip = _empty(dcat.shape[:-1] + (Cs_,), dtype=_F32, device=dev)
                d_up = _empty(dcat.shape[:-1] + (Ct_,), dtype=_F32, device=dev)
                nat.call("u3d_split_channels", dev.index, _stream(dev), _p(dcat), dcat.numel() // (Cs_ + Ct_), Cs_, Ct_, _p(d_skip),
                         _p(d_up))
                skip_grad[n_levels - 2 - j] = d_skip
                dj = d_up
                del dcat
            else:
                assert rec.conv1 is None
                skip_grad[n_levels - 2 - j] = dj   # summation joining: the skip receives dj as is
            rec = None  # (lean tape: the block's activations go back to the allocator before the transposed convolution's buffers)
            xl = up.x_low
            Nl, D1, H1, W1, Cl = xl.shape
            _, Ds, Hs, Ws, Cs = dj.shape
            Dt, Ht, Wt = up.tdims
            lz, ly, lx = up.los
            if up.t8:
                sfx = "_b16" if self.act_bf16 else ""
                dt8 = _empty((Nl, D1, H1, W1, 8 * Cs), dtype=self.adt, device=dev)
                nat.call("u3d_nearest_sum_bwd_t8" + sfx, dev.index, _stream(dev), _p(dj), _p(lz), _p(ly), _p(lx), Nl, Ds, Hs, Ws, Dt, Ht,
                         Wt, Cs, _p(dt8))
                need = nat.get_lib().u3d_convtr3d_wgrad_t8_workspace_floats(Nl, D1, H1, W1, Cl, Cs)
                wsb = cx.ensure_ws(need)
                nat.call("u3d_convtr3d_wgrad_t8" + sfx, dev.index, _stream(dev), _p(xl), _p(dt8),
                         _p(gview(self._pindex[id(up.weight)])), Nl, D1, H1, W1, Cl, Cs, _p(wsb), wsb.numel(),
                         flops=2.0 * 27 * Cl * Cs * Nl * D1 * H1 * W1)
                dxl = _empty_like(xl)
                nat.call("u3d_convtr3d_dgrad_t8" + sfx, dev.index, _stream(dev), _p(dt8), _p(self._packed_convtr_t8(up.weight, 1, dev)),
                         _p(xl) if mk else None, _p(dxl), Nl, D1, H1, W1, Cl, Cs, flops=2.0 * 27 * Cl * Cs * Nl * D1 * H1 * W1)
                del dt8
                dz = dxl  # ReLU blocks: masked by (x_low > 0)
                continue
            dt = _empty((Nl, Dt, Ht, Wt, Cs), dtype=_F32, device=dev)
            nat.call("u3d_nearest_sum_bwd", dev.index, _stream(dev), _p(dj), _p(lz), _p(ly), _p(lx), Nl, Ds, Hs, Ws, Dt, Ht, Wt,
                     Cs, _p(dt))
            acc = pool.take(up.weight.numel())
            dxl = _empty_like(xl)
            nat.call("u3d_convtr3d_bwd", dev.index, _stream(dev), _p(dt), _p(xl), _p(up.weight.detach()), Nl, D1, H1, W1, Cl, Cs,
                     mk, _p(dxl), _p(acc), _p(self._packed_convtr(up.weight, 1, dev)), flops=4.0 * 27 * Cl * Cs * Nl * D1 * H1 * W1)
            nat.call("u3d_cvt_f64_f32", dev.index, _stream(dev), _p(acc), _p(gview(self._pindex[id(up.weight)])),
                     up.weight.numel())
            del dt
            dz = dxl  # ReLU blocks: masked by (x_low > 0), x_low being the output of the block below

        rec = up = None
        if self.grad_sync is not None:
            cx.join()
            self.grad_sync.launch(flat[self.n_enc_params :])

        dx0 = None
        pending_hi = self.n_enc_params
        for i in range(n_levels - 1, -1, -1):
            rec = enc_blocks[i]
            recomputed = isinstance(rec, CkptRec)
            if recomputed:
                # recompute the block's forward (bit-identical kernels, same inputs) to rebuild what backward needs
                tmp = Tape()
                fpool = _StatPool(dev, 16 * rec.x_in.shape[0] * rec.bm.conv2.conv.in_channels * 2 + 64)
                self._in_recompute = True
                try:
                    self._block_fwd(rec.bm, rec.name, rec.x_in, None, fpool, tmp, dev, y_out=rec.out)
                finally:
                    self._in_recompute = False
                cx.ensure_ws(self._wgrad_workspace_floats(tmp.convs))
                rec = tmp.blocks[0]
                del tmp, fpool
            if lean:
                enc_blocks[i] = None
            dr = self._block_bwd(cx, rec, dz)
            dz = None
            need_dx = i > 0 or need_input_grad
            if rec.conv1 is not None:
                dxin = self._conv1_bwd(cx, rec, dr, need_dx)
            else:
                dxin = dr
            if self.grad_sync is not None:
                pending_hi = self._sync_encoder_level(cx, flat, i, pending_hi)  # this level's parameter gradients are final
            if recomputed:
                cx.join()  # a side-stream weight gradient may still read the recomputed tensors released with `rec` below
            rec = None
            if i > 0:
                pooled, argmax, e_in = pools[i - 1]
                if lean:
                    pools[i - 1] = None
                Ne, De, He, We, Ce = e_in.shape
                out = _empty_like(e_in)
                nat.call("u3d_maxpool2_bwd_merge" + ("_b16" if self.act_bf16 else ""), dev.index, _stream(dev), _p(dxin), _p(pooled),
                         _p(argmax), None, _p(skip_grad.get(i - 1)), _p(e_in), Ne, De, He, We, Ce, mk, _p(out))
                skip_grad.pop(i - 1, None)
                dz = out
            elif need_input_grad:
                dx0 = dxin

        cx.join()
        if self.grad_sync is not None:
            self.grad_sync.finish()

        dx = None
        if dx0 is not None:
            if dx0.dtype != _F32:
                dx0 = dx0.to(_F32)  # (input gradients are rare; the network input and its gradient are fp32 tensors)
            if Cin == 1:
                dx = dx0.reshape(N, 1, D, H, W)
            else:
                dx = _empty((N, Cin, D, H, W), dtype=_F32, device=dev)
                nat.call("u3d_ndhwc_to_ncdhw", dev.index, _stream(dev), _p(dx0), _p(dx), N, Cin, V)
        return flat, dx


class _UNet3DFunction(torch.autograd.Function):
    """The whole encoder-decoder as one autograd node (forward = engine.forward, backward = engine.backward)."""

    @staticmethod
    def forward(ctx, engine: UNet3DEngine, grad_mode: bool, x: torch.Tensor, *params):
        # Grad mode is always off inside Function.forward, and `ctx.needs_input_grad` reports the inputs' requires_grad flags even
        # when the CALLER runs under torch.no_grad() (round 4: inference forwards therefore kept a tape, advanced the repack salt and
        # repacked every weight image each time — 6 ms per volume of BASELINE config 5).  The caller's grad mode is passed in.
        save = grad_mode and any(ctx.needs_input_grad)
        with engine._lock:
            engine.begin_forward(save)
            logits, probs, tape = engine.forward(x, save)
        ctx.engine = engine
        ctx.has_probs = probs is not None
        ctx.x_requires_grad = x.requires_grad
        ctx.skel = None
        ctx.lean_tape = None
        # parameters are referenced by position, not saved: record their versions so that an in-place update between forward and
        # backward is refused like stock autograd refuses it (backward would otherwise repack and use the NEW weights)
        ctx.pversions = [p._version for p in engine.params] if tape is not None else None
        if tape is not None and engine.lean_tape:
            # memory-lean mode (checkpoint_encoders): the tape stays a plain Python object owned by this node, so that backward can
            # release it block by block — autograd's saved-tensor slots are only freed when the whole node is done.  The price:
            # ONE backward per forward (retain_graph is refused with a clear error, see backward)
            tape.lean = True
            ctx.lean_tape = tape
            if probs is not None:
                ctx.save_for_backward(probs)
        elif tape is not None:
            ctx.skel, bag = stash_tape(tape, engine._pindex)
            ctx.save_for_backward(*([probs] if probs is not None else []), *bag)
        elif probs is not None:
            ctx.save_for_backward(probs)
        if probs is not None:
            return logits, probs
        return (logits,)

    @staticmethod
    def backward(ctx, *grads):
        engine = ctx.engine
        if ctx.skel is None and ctx.lean_tape is None:
            raise RuntimeError("u3d: no activation tape for this backward (the forward ran without any input requiring grad)")
        # a second backward without retain_graph=True raises autograd's own "backward through the graph a second time" here
        saved = ctx.saved_tensors
        probs = saved[0] if ctx.has_probs else None
        for p, v in zip(engine.params, ctx.pversions):
            if p._version != v:
                raise RuntimeError("one of the variables needed for gradient computation has been modified by an inplace operation: "
                                   f"a parameter of shape {tuple(p.shape)} is at version {p._version}, expected version {v} "
                                   "(u3d: the weights changed between this forward and its backward)")
        if ctx.lean_tape is not None:
            tape = ctx.lean_tape
            if tape.consumed:
                raise RuntimeError("u3d: backward through the graph a second time — with checkpoint_encoders the activation tape is "
                                   "released block by block DURING backward (that is where the memory saving comes from), so "
                                   "retain_graph=True is not available in this mode")
        else:
            tape = unstash_tape(ctx.skel, saved[1:] if ctx.has_probs else saved, engine.params)
        dlogits = grads[0]
        if ctx.has_probs and len(grads) > 1 and grads[1] is not None:
            # gradient flowing through the probabilities (rare: the reference's trainer takes the loss on logits,
            # trainer.py:362-365): fold it into dlogits.  Tiny (N,Cout,D,H,W) tensors.
            gp = grads[1]
            if isinstance(engine.model.final_activation, torch.nn.Sigmoid):
                extra = gp * probs * (1 - probs)
            else:
                extra = probs * (gp - (gp * probs).sum(dim=1, keepdim=True))
            dlogits = extra if dlogits is None else dlogits + extra
        if dlogits is None:
            dlogits = torch.zeros_like(probs)
        with engine._lock:
            flat, dx = engine.backward(tape, dlogits, ctx.x_requires_grad)
        del tape, saved
        out = [None, None, dx]
        for p, off in zip(engine.params, engine.poffs):
            out.append(flat[off : off + p.numel()].view(p.shape) if p.requires_grad else None)
        return tuple(out)


class GraphStep:
    """The launch sequences of ONE training step at ONE input shape, captured in two hipGraphs (forward: input -> logits /
    probabilities + the activation tape; backward: dlogits -> flat parameter gradients [+ input gradient]) and replayed with two
    `hipGraphLaunch` calls instead of ~180 ctypes calls + ~150 tensor allocations (3.3 ms of host time per step, which makes
    BASELINE config 1's shape host-bound: tools/host_bound_check.py).  The loop it serves is the reference's unchanged
    `output, loss = self._forward_pass(...); loss.backward(); optimizer.step()` (unet3d/trainer.py:231-246): the model call replays
    the forward graph, `loss.backward()` reaches `_GraphedUNet3DFunction.backward`, which replays the backward graph.

    What is static: the input / dlogits staging buffers, every activation of the tape, the flat gradient buffer and all scratch —
    one private allocator pool shared by both graphs; parameters are read through their (stable) storage pointers, and the weight
    repacking of a training forward is PART of the forward graph, so optimizer steps between replays are seen.  What the caller
    gets are fresh copies (logits, probabilities, one flat gradient buffer), so holding outputs or `.grad` across steps is as
    safe as in eager mode.  One tape per shape: a backward must follow ITS forward before the next forward of that shape (the
    reference loop does); anything else raises instead of silently using a newer tape.  (Until then the tape stays valid, so a
    second backward over a retained graph replays again, like eager mode with retain_graph=True.)"""

    def __init__(self, engine: "UNet3DEngine", x: torch.Tensor, need_dx: bool):
        dev = x.device
        self.engine = engine
        self.need_dx = need_dx
        self.gen = 0          # forwards replayed so far (the tape in the pool belongs to the latest one)
        self.static_x = torch.empty_like(x)
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            # one eager step first: every lazily built constant (index maps, identity tables, the pack descriptor table, the
            # library's function attributes) must exist before capture — host-to-device copies are illegal inside it
            self.static_x.copy_(x)
            engine.begin_forward(True)
            logits, probs, tape = engine.forward(self.static_x, True)
            engine.backward(tape, torch.zeros_like(logits), need_dx)
            del logits, probs, tape
        cur.wait_stream(side)
        torch.cuda.synchronize(dev)
        self.pool = torch.cuda.graph_pool_handle()
        self.g_fwd = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_fwd, pool=self.pool, capture_error_mode="thread_local"):
            engine.begin_forward(True)
            self.logits, self.probs, self.tape = engine.forward(self.static_x, True)
        self.static_dl = torch.zeros_like(self.logits)
        # Data parallelism (parallel.GradSync attached): RCCL launches cannot live inside a captured graph that is replayed with
        # other buckets in flight, so the backward is captured as a CHAIN of graphs cut exactly where engine.backward hands a gradient
        # bucket to RCCL ([decoders | head] first, then the encoder levels deepest first) — `_CaptureSplit.launch` ends the running
        # capture and begins the next on the same stream and pool; backward() issues the real all-reduces eagerly between the replays
        # (trainer.py:202-205 is the loop this serves: one gradient exchange per step, overlapped with the rest of the backward).
        self.g_bwds = [torch.cuda.CUDAGraph()]
        self.sync = engine.grad_sync
        self.buckets: list = []
        if self.sync is not None:
            engine.grad_sync = _CaptureSplit(self)
        try:
            # (what `torch.cuda.graph` does, by hand: its __exit__ would call capture_end() on the graph it was given, but with a split
            # the capture has moved on to a later graph by then)
            torch.cuda.synchronize(dev)
            torch.cuda.empty_cache()
            cap = torch.cuda.Stream(dev)
            with torch.cuda.stream(cap):
                self.g_bwds[0].capture_begin(pool=self.pool, capture_error_mode="thread_local")
                try:
                    self.flat, self.dx = engine.backward(self.tape, self.static_dl, need_dx)
                finally:
                    self.g_bwds[-1].capture_end()
            torch.cuda.synchronize(dev)
        finally:
            engine.grad_sync = self.sync
        # Strong references to every PRE-CAPTURE device buffer the graphs dereference: the pack descriptor
        # tables' only other owner is a one-entry dict that the next eager forward with a different stale set clears
        # (`tab.clear()` in _repack_all / _repack_bf16_all: validation between training steps does exactly that), the packed
        # images can be replaced in `_pack_cache`, constants can be rebuilt — the caching allocator (or torch.cuda.empty_cache())
        # would then hand the blocks to someone else while every later replay still reads / writes them.
        self._pins = engine.graph_pins()

    def forward(self, x: torch.Tensor):
        self.static_x.copy_(x)
        self.g_fwd.replay()
        self.gen += 1
        return self.logits.clone(), (self.probs.clone() if self.probs is not None else None)

    def backward(self, gen: int, dlogits: torch.Tensor):
        if gen != self.gen:
            raise RuntimeError("u3d hip_graph: a later forward of the same input shape has overwritten this step's activation tape "
                               "(graph mode keeps ONE tape per shape: run forward -> backward in turn, or set hip_graph: false / "
                               "U3D_GRAPH=0 for interleaved graphs)")
        self.static_dl.copy_(dlogits)
        for i, g in enumerate(self.g_bwds):
            g.replay()
            if i < len(self.buckets):
                self.sync.launch(self.buckets[i])  # final here: exchanged while the following graphs run
        if self.sync is not None:
            self.sync.finish()
        return self.flat.clone(), (self.dx.clone() if self.dx is not None else None)


class _CaptureSplit:
    """stands in for parallel.GradSync while GraphStep captures the backward: every `launch` (a gradient bucket that is final at
    that point of engine.backward) is a cut between two backward graphs; the collectives themselves are issued at replay time"""

    def __init__(self, step: "GraphStep"):
        self.step = step

    def launch(self, bucket: torch.Tensor) -> None:
        st = self.step
        st.buckets.append(bucket)
        st.g_bwds[-1].capture_end()
        st.g_bwds.append(torch.cuda.CUDAGraph())
        st.g_bwds[-1].capture_begin(pool=st.pool, capture_error_mode="thread_local")

    def finish(self) -> None:
        pass


class _GraphedUNet3DFunction(torch.autograd.Function):
    """The same autograd node as _UNet3DFunction with both directions replayed from GraphStep's hipGraphs."""

    @staticmethod
    def forward(ctx, step: GraphStep, x: torch.Tensor, *params):
        with step.engine._lock:
            logits, probs = step.forward(x)
            ctx.gen = step.gen
        # as in _UNet3DFunction: an in-place weight update between this forward and its backward is refused — the backward graph
        # would mix packed images of the old weights (data gradients) with raw reads of the new ones (1x1x1 convs, head)
        ctx.pversions = [p._version for p in step.engine.params]
        ctx.step = step
        ctx.has_probs = probs is not None
        ctx.x_requires_grad = x.requires_grad
        if probs is not None:
            ctx.save_for_backward(probs)
            return logits, probs
        return (logits,)

    @staticmethod
    def backward(ctx, *grads):
        step = ctx.step
        engine = step.engine
        probs = ctx.saved_tensors[0] if ctx.has_probs else None
        for p, v in zip(engine.params, ctx.pversions):
            if p._version != v:
                raise RuntimeError("one of the variables needed for gradient computation has been modified by an inplace operation: "
                                   f"a parameter of shape {tuple(p.shape)} is at version {p._version}, expected version {v} "
                                   "(u3d hip_graph: the weights changed between this forward and its backward)")
        dlogits = grads[0]
        if ctx.has_probs and len(grads) > 1 and grads[1] is not None:
            gp = grads[1]
            if isinstance(engine.model.final_activation, torch.nn.Sigmoid):
                extra = gp * probs * (1 - probs)
            else:
                extra = probs * (gp - (gp * probs).sum(dim=1, keepdim=True))
            dlogits = extra if dlogits is None else dlogits + extra
        if dlogits is None:
            dlogits = torch.zeros_like(probs)
        with engine._lock:
            flat, dx = step.backward(ctx.gen, dlogits)
        out = [None, dx if ctx.x_requires_grad else None]
        for p, off in zip(engine.params, engine.poffs):
            out.append(flat[off : off + p.numel()].view(p.shape) if p.requires_grad else None)
        return tuple(out)


_GRAPH_MAX_SHAPES = int(os.environ.get("U3D_GRAPH_SHAPES", 2))  # captured shapes kept per model (each pins its whole tape in HBM)


def _graph_blocker(engine: UNet3DEngine) -> Optional[str]:
    """why this model cannot be captured (None = it can).  Static per engine."""
    order = getattr(engine.model, "layer_order", "gcr")
    if any(ch in order for ch in "bdD"):
        return f"layer_order '{order}': BatchNorm reads its step counter on the host, dropout draws a fresh mask per step"
    if engine.debug is not None or nat.profiler is not None or _POISON:
        return "debug / profiler / poison mode"
    return None


def graph_step_for(engine: UNet3DEngine, x: torch.Tensor) -> Optional[GraphStep]:
    """the captured step of this input shape (captured on first use), or None when the eager path must run"""
    if not engine.hip_graph or not torch.is_grad_enabled() or torch.cuda.is_current_stream_capturing():
        return None
    if not any(p.requires_grad for p in engine.params):
        return None
    why = _graph_blocker(engine)
    if why is not None:
        if engine._graph_off_reason != why:
            engine._graph_off_reason = why
            import warnings

            warnings.warn(f"u3d: hip_graph requested but this step runs eagerly ({why})", stacklevel=4)
        return None
    # (the graphs bake the parameters' storage pointers in: first + last pointer is the cheap sentinel that check_placement uses too —
    # module.to() / load_state_dict(assign=True) move all of them, and the executor itself is rebuilt when parameter OBJECTS change)
    key = (tuple(x.shape), bool(x.requires_grad), engine.params[0].data_ptr(), engine.params[-1].data_ptr(), id(engine.grad_sync))
    step = engine._graph_steps.get(key)
    if step is None:
        while len(engine._graph_steps) >= _GRAPH_MAX_SHAPES:
            engine._graph_steps.pop(next(iter(engine._graph_steps)))  # oldest shape: its graphs and pool are released
        with engine._lock:
            step = GraphStep(engine, x.contiguous(), bool(x.requires_grad))
        engine._graph_steps[key] = step
    else:
        engine._graph_steps[key] = engine._graph_steps.pop(key)  # most recently used last
    return step


def check_placement(engine: UNet3DEngine, x: torch.Tensor):
    """The kernels read raw pointers: every parameter must be fp32 and live on the input's device (stock modules raise
    ATen's device/dtype mismatch errors in the same situations, e.g. model.half() or a model left on another GPU)."""
    first, last = engine.params[0], engine.params[-1]
    if engine._placed == (x.device, first.data_ptr(), last.data_ptr()):
        return  # every parameter was checked for this device and these storages (module.to / .half re-allocate them)
    engine._placed = None
    for p in engine.params:
        if p.device != x.device or p.dtype != torch.float32:
            raise RuntimeError(f"u3d: parameter of shape {tuple(p.shape)} is {p.dtype} on {p.device}, the input is "
                               f"{x.dtype} on {x.device} — the native gfx950 path needs fp32 parameters on the input's "
                               "device (one process per GPU: pytorch3dunet_amd.parallel.attach; or model.to(x.device))")
    engine._placed = (x.device, first.data_ptr(), last.data_ptr())


def run_model(engine: UNet3DEngine, x: torch.Tensor):
    """(probs_or_logits, logits) exactly like AbstractUNet._forward_logits (model.py:123-149)."""
    check_placement(engine, x)
    step = graph_step_for(engine, x) if engine.hip_graph else None
    if step is not None:
        outs = _GraphedUNet3DFunction.apply(step, x, *engine.params)
    else:
        outs = _UNet3DFunction.apply(engine, torch.is_grad_enabled(), x, *engine.params)
    if len(outs) == 2:
        logits, probs = outs
        return probs, logits
    return outs[0], outs[0]
