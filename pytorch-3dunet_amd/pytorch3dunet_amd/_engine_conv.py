"""One SingleConv / ResNetBlock convolution (reference buildingblocks.py:99-135) forward and backward on the native kernels: which
kernel FAMILY runs a layer (first-layer small-Cin, sub-pixel decoder, split-fp32, bf16 operands / bf16 storage, fp32 MFMA) is decided
in ONE place per direction (`_fwd_family` / `_bwd_family`) and dispatched through a table — a new family is a new entry, not a
new flag inside the layer code.  Mixin of `engine.UNet3DEngine`."""
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

from ._engine_base import *  # noqa: F401,F403  (explicit __all__: helpers, records, activation codes)

_WGRAD_JOB = os.environ.get("U3D_WGRAD_JOB", "1") != "0"  # A/B: 0 = GroupNorm-backward reductions as launches of their own


def _reps(t) -> int:
    """replica rows of a statistics table (u3d_conv3d_ex_reps): the tensor a persistent convolution wrote carries the count as an
    attribute; every other table is one row"""
    return getattr(t, "_u3d_reps", 1) if t is not None else 1


def _take_reps(pool, n: int, reps: int):
    t = pool.take(reps * n)
    if reps > 1:
        t._u3d_reps = reps
    return t


def _fold_reps(t):
    """the plain table of a replicated one (consumers without a replica-aware entry point: BatchNorm, stand-alone backward finalize)"""
    r = _reps(t)
    return t if r == 1 else t.view(r, -1).sum(0)


class _SubLayers(dict):
    """{id(conv weight): (C0, C1)} of the decoder first convs that take the sub-pixel path at one input size; `plus` = the ids whose level
    upsamples n -> 2n + 1 along some axis"""

    plus = frozenset()


@dataclass
class _ConvCall:
    """everything a forward kernel family needs to launch one convolution (filled by ConvLayers._single_conv_fwd)"""
    dev: torch.device
    conv: torch.nn.Module
    src: VSrc
    affine: torch.Tensor
    y: torch.Tensor
    N: int
    D: int
    H: int
    W: int
    Ctot: int
    Cout: int
    relu: int
    stats: bool                          # the epilogue accumulates (sum, sum of squares) of the written values
    pool: _StatPool
    residual: Optional[torch.Tensor]     # added before the ReLU inside the conv epilogue (pre-norm residual blocks)
    sub: dict                            # sub-pixel layers of this forward: id(weight) -> (C0, C1)
    b16: bool                            # bf16 activation storage
    affine_lo: Optional[torch.Tensor] = None  # sub-pixel layers: compact rows of `affine` for the skip / upsampled channels
    affine_hi: Optional[torch.Tensor] = None

    def take_stats(self, reps: int = 1):
        return _take_reps(self.pool, self.N * self.Cout * 2, reps) if self.stats else None

    @property
    def flops(self):
        return 54.0 * self.Ctot * self.Cout * self.N * self.D * self.H * self.W


@dataclass
class _BwdCall:
    """everything a backward kernel family needs for one convolution (filled by ConvLayers._conv_bwd)"""
    cx: "_BwdCtx"
    rec: ConvRec
    dz: torch.Tensor
    src: VSrc
    N: int
    D: int
    H: int
    W: int
    Cout: int
    bf16: bool   # both directions of this layer run on the bf16-operand kernels
    b16: bool    # bf16 activation storage
    job: object = None  # U3DGnBwdJob for the weight-gradient reduce launch to carry (the family that takes it sets it back to None)
    greps: int = 1      # replica rows of the GroupNorm-backward sums the fp32 data-gradient kernel writes (u3d_conv3d_ex_reps)

    @property
    def flops(self):
        return 54.0 * self.src.C * self.Cout * self.N * self.D * self.H * self.W


class ConvLayers:
    """mixin: layer-level forward / backward building blocks shared by the DoubleConv and the residual executors"""

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
        """decoder first convs whose low-res input is upsampled by exactly 2 — or, round 5, from n to 2n + 1 voxels — in every dimension at
        this input size: {id(weight): (C0, C1)} (+ `.plus`: the ids with an n -> 2n + 1 axis) — per-call state, handed down as `sub`"""
        if not self.subpixel or any(ct is not None for ct in self.dec_up) or any(self.dec_interp):
            return _SubLayers()  # (a transposed convolution yields 2n-1 voxels, resized to the skip: never an exact 2x replication)
        dims = [tuple(size)]
        for has_pool, _, _ in self.enc:
            if has_pool:
                dims.append(tuple(d // 2 for d in dims[-1]))
        out, plus = _SubLayers(), set()
        L = len(self.enc)
        for j, (c1, _) in enumerate(self.dec):
            skip_lvl, low_lvl = L - 2 - j, L - 1 - j
            if skip_lvl < 0 or low_lvl >= len(dims) or self._cat_bf16(c1):  # (bf16 mode: the concat is materialised, an ordinary layer)
                continue
            C0 = self.enc[skip_lvl][2].conv.out_channels
            C1 = c1.conv.in_channels - C0
            # every axis upsampled by exactly 2, or n -> 2n + 1 (the pooled size of an odd level; round 5: the sub-pixel kernels on a
            # shifted window + the general kernels on the near-boundary slab, _fwd_subpixel / _dgrad_subpixel / _wgrad_subpixel)
            ratio = [a - 2 * b for a, b in zip(dims[skip_lvl], dims[low_lvl])]
            if (all(r in (0, 1) for r in ratio) and (self.subpixel_plus or not any(ratio)) and C0 > 0 and C1 > 0 and C0 % 4 == 0
                    and C1 % 4 == 0 and c1.conv.out_channels % 4 == 0):
                out[id(c1.conv.weight)] = (C0, C1)
                if any(ratio):
                    plus.add(id(c1.conv.weight))
        self._sub_pairs.update(out)
        out.plus = frozenset(plus)  # the layers of this call whose packed set includes the slab images (_repack_all)
        return out

    def _stats_of(self, src: VSrc, st0, st1, pool: _StatPool, dev):
        """(stats0, C0, scale0, stats1, C1, scale1) describing the per-channel sums of a (virtual) tensor"""
        if src.t1 is None:
            if st0 is None:
                st0 = _take_reps(pool, src.N * src.C0 * 2, self.stat_reps)
                s = src.struct()
                nat.call("u3d_chan_stats_reps", dev.index, _stream(dev), ctypes.byref(s), src.N, src.D, src.H, src.W, _p(st0), _reps(st0))
            return st0, src.C0, 1.0, None, 0, 0.0
        if st0 is not None and st1 is not None and src.exact2x and self.fused_stats:
            # every low-res voxel is replicated exactly 8x: reuse the producer's sums
            return st0, src.C0, 1.0, st1, src.C1, 8.0
        plus = src.plus
        if st0 is not None and plus is not None and any(plus) and self.fused_stats and src.C1 % 4 == 0 and src.t1.dtype == _F32:
            # n -> 2n + 1 along some axes: the first low-res cell of such an axis has three children, every other cell two — the sums of
            # the upsampled half as a weighted pass over the LOW-RES tensor (round 6), the skip half from its producer
            st1w = pool.take(src.N * src.C1 * 2)
            nat.call("u3d_chan_stats_children", dev.index, _stream(dev), _p(src.t1), src.N, src.D1, src.H1, src.W1, src.C1, *plus, _p(st1w))
            return st0, src.C0, 1.0, st1w, src.C1, 1.0
        st = _take_reps(pool, src.N * src.C * 2, self.stat_reps)
        s = src.struct()
        nat.call("u3d_chan_stats_reps", dev.index, _stream(dev), ctypes.byref(s), src.N, src.D, src.H, src.W, _p(st), _reps(st))
        return st, src.C, 1.0, None, 0, 0.0

    def _norm_finalize(self, kind, mod, st0, C0, sc0, st1, C1, sc1, N, G, count, affine, dev, split=None):
        """per-(n,c) sums -> the (a, b) table the convolutions / apply passes use; returns what backward needs (mean, rstd).
        `split` = (Csplit, affine_lo, affine_hi): compact tables of the channel ranges [0, Csplit) / [Csplit, C) written in the same
        launch (GroupNorm over a virtual concat whose halves are read by different kernels; None entries are skipped)"""
        r0, r1 = _reps(st0), _reps(st1)
        if kind == "g" and (r0 > 1 or r1 > 1):
            mean_rstd = _empty((N, G, 2), dtype=_F32, device=dev)
            sp = split if split is not None else (0, None, None)
            nat.call("u3d_gn_finalize_reps", dev.index, _stream(dev), _p(st0), C0, sc0, r0, _p(st1), C1, sc1, r1, N, G, count,
                     _p(mod.weight.detach()), _p(mod.bias.detach()), float(mod.eps), _p(affine), _p(mean_rstd), sp[0], _p(sp[1]), _p(sp[2]))
            return mean_rstd
        st0, st1 = _fold_reps(st0), _fold_reps(st1)
        if kind == "g":
            mean_rstd = _empty((N, G, 2), dtype=_F32, device=dev)
            if split is not None:
                nat.call("u3d_gn_finalize_split", dev.index, _stream(dev), _p(st0), C0, sc0, _p(st1), C1, sc1, N, G, count,
                         _p(mod.weight.detach()), _p(mod.bias.detach()), float(mod.eps), _p(affine), _p(mean_rstd), split[0],
                         _p(split[1]), _p(split[2]))
                return mean_rstd
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

    def _norm_bwd_job(self, cx, rec: ConvRec, gst, N, C, count, coef):
        """The GroupNorm-backward reduction of a layer's input as a job for the launch that reduces the layer's weight gradient
        (u3d_conv3d_wgrad_job: one launch less per layer); None when it has to run on its own (_norm_bwd_finalize)."""
        if rec.norm != "g" or not _WGRAD_JOB or nat.get_lib().u3d_conv3d_wgrad_job_supported(N, C, rec.G) != 1:
            return None
        dev, gview = cx.dev, cx.gview
        cx.coef_hi = None
        job = nat.U3DGnBwdJob()
        if isinstance(gst, tuple):
            g0, g1 = gst
            C0 = g0.numel() // (2 * N * _reps(g0))
            coef_hi = _empty((N, 3, C - C0), dtype=_F32, device=dev) if not any(rec.src.plus) else None
            job.gstats_lo, job.gstats_hi, job.C0, job.C1, job.hi_scale, job.coef_hi = _p(g0), _p(g1), C0, C - C0, 8.0, _p(coef_hi)
            cx.coef_hi = coef_hi
        else:
            job.gstats_lo, job.gstats_hi, job.C0, job.C1, job.hi_scale, job.coef_hi = _p(gst), None, C, 0, 1.0, None
        job.reps_lo = _reps(gst[0] if isinstance(gst, tuple) else gst)
        job.reps_hi = _reps(gst[1]) if isinstance(gst, tuple) else 1
        job.mean_rstd, job.gamma = _p(rec.mean_rstd), _p(rec.gn_w.detach())
        job.dgamma, job.dbeta, job.coef = _p(gview(rec.idx_gw)), _p(gview(rec.idx_gb)), _p(coef)
        job.count, job.N, job.G = count, N, rec.G
        return job

    def _norm_bwd_finalize(self, cx, rec: ConvRec, gst, N, C, count, coef):
        dev, gview = cx.dev, cx.gview
        cx.coef_hi = None
        gst = tuple(_fold_reps(g) for g in gst) if isinstance(gst, tuple) else _fold_reps(gst)
        if isinstance(gst, tuple):
            # sub-pixel decoder layer: the sums of the skip / upsampled channels come from two kernels as two tables
            g0, g1 = gst
            C0 = g0.numel() // (2 * N)
            if rec.norm == "g" and nat.get_lib().u3d_gn_bwd_finalize_split_supported(N, C, rec.G) == 1:
                # ... and the low-res apply pass of an exact-2x level wants the upper channels' (p, 8q, 8r) as a compact table
                coef_hi = _empty((N, 3, C - C0), dtype=_F32, device=dev) if not any(rec.src.plus) else None
                nat.call("u3d_gn_bwd_finalize_split", dev.index, _stream(dev), _p(g0), C0, _p(g1), C - C0, _p(rec.mean_rstd),
                         _p(rec.gn_w.detach()), N, rec.G, count, _p(gview(rec.idx_gw)), _p(gview(rec.idx_gb)), _p(coef), 8.0, _p(coef_hi))
                cx.coef_hi = coef_hi
                return
            gst = torch.cat((g0.view(N, C0, 2), g1.view(N, C - C0, 2)), dim=1)
        if rec.norm == "g":
            nat.call("u3d_gn_bwd_finalize", dev.index, _stream(dev), _p(gst), _p(rec.mean_rstd), _p(rec.gn_w.detach()), N, C, rec.G,
                     count, _p(gview(rec.idx_gw)), _p(gview(rec.idx_gb)), _p(coef))
        else:
            nat.call("u3d_bn_bwd_finalize", dev.index, _stream(dev), _p(gst), _p(rec.mean_rstd), _p(rec.gn_w.detach()), N, C, count,
                     1 if rec.bn_training else 0, _p(gview(rec.idx_gw)), _p(gview(rec.idx_gb)), _p(coef))

    # ---- forward kernel families (csrc file; what selects it) -------------------------------------------------------------------
    _FWD_KERNELS = {
        "small": "_fwd_small",        # u3d_smallc.hip: first layer, Cin <= 4
        "subpixel": "_fwd_subpixel",  # u3d_subpix.hip + u3d_conv.hip: cat(skip, nearest2x(low)) on the low-res grid, 8/27 of the MACs
        "f32s": "_fwd_f32s",          # u3d_bf16.hip: compute_dtype fp32_split
        "bf16": "_fwd_bf16",          # u3d_bf16.hip: compute_dtype bf16 (fp32 or bf16 activation storage)
        "fp32": "_fwd_fp32",          # u3d_conv.hip: fp32 MFMA (persistent / generic / split-K chosen by the library)
    }

    def _fwd_family(self, c: "_ConvCall", residual) -> str:
        if self.small_cin and c.src.t1 is None and c.Ctot <= 4 and c.Cout <= 32 and residual is None and not c.b16:
            return "small"
        if c.src.t1 is not None and residual is None and id(c.conv.weight) in c.sub:
            return "subpixel"
        if c.src.t1 is None and self._split_fwd(c.Ctot, c.Cout):
            return "f32s"
        if c.src.t1 is None and self._bf16_layer(c.Ctot, c.Cout):
            return "bf16"
        return "fp32"

    def _fwd_small(self, c: "_ConvCall"):
        # first layer of the network: K = 27*Cin is too small for the MFMA tiling (csrc/u3d_smallc.hip)
        ystats = c.take_stats(self.stat_reps)
        nat.call("u3d_conv3d_small_cin_fwd_reps", c.dev.index, _stream(c.dev), _p(c.src.t0), _p(c.affine), _p(c.conv.weight.detach()),
                 _p(c.y), c.N, c.D, c.H, c.W, c.Ctot, c.Cout, c.relu, _p(ystats), _reps(ystats), flops=c.flops)
        return ystats

    def _fwd_subpixel(self, c: "_ConvCall"):
        # cat(skip, nearest2x(low)): the upsampled half as 8 parity-class 2x2x2 convolutions over the low-res tensor
        # (8/27 of the multiply-adds), then the skip half, whose epilogue adds the partial sums before ReLU / statistics
        dev, conv, src, N, D, H, W, Cout = c.dev, c.conv, c.src, c.N, c.D, c.H, c.W, c.Cout
        C0, C1 = c.sub[id(conv.weight)]
        ystats = c.take_stats(1 if self._split_fwd(C0, Cout) else self.stat_reps)
        part = _empty((N, D, H, W, Cout), dtype=_F32, device=dev)
        D1, H1, W1 = src.D1, src.H1, src.W1
        plus = src.plus
        if any(plus):
            # n -> 2n + 1 along the axes with plus = 1: the sub-pixel kernel writes the window d = u + plus, the general kernel the slab
            # d < 2 of those axes (overwriting the window's first plane, which misses the extra copy of low[0])
            win = (ctypes.c_int * 6)(D, H, W, *plus)
            nat.call("u3d_subpixel_conv_fwd_win", dev.index, _stream(dev), _p(src.t1), _p(c.affine.view(-1)[2 * C0:]), c.Ctot * 2,
                     _p(self._pack_cache[(id(conv.weight), 12)][1]), _p(part), N, D1, H1, W1, C1, Cout, win,
                     flops=128.0 * C1 * Cout * N * D1 * H1 * W1)
            s_up = src.up_only_struct(c.affine_hi if c.affine_hi is not None else c.affine[:, C0:].contiguous())
            wp1 = self._pack_cache[(id(conv.weight), 14)][1]  # 27-tap forward image of the upsampled channels (slab launches)
            for box in slab_boxes((D, H, W), plus, 2):
                nat.call("u3d_conv3d_box", dev.index, _stream(dev), ctypes.byref(s_up), _p(wp1), _p(part), N, D, H, W, Cout,
                         (ctypes.c_int * 6)(*box), None,
                         flops=54.0 * C1 * Cout * N * (box[3] - box[0]) * (box[4] - box[1]) * (box[5] - box[2]))
        else:
            need = nat.get_lib().u3d_subpixel_fwd_workspace_floats(N, D1, H1, W1, C1, Cout)  # split-K scratch, small levels only
            kws = _empty(need, dtype=_F32, device=dev) if need > 0 else None
            nat.call("u3d_subpixel_conv_fwd", dev.index, _stream(dev), _p(src.t1), _p(c.affine.view(-1)[2 * C0:]), c.Ctot * 2,
                     _p(self._pack_cache[(id(conv.weight), 12)][1]), _p(part), N, D1, H1, W1, C1, Cout, _p(kws), need,
                     flops=128.0 * C1 * Cout * N * D1 * H1 * W1)
        a0 = c.affine_lo if c.affine_lo is not None else c.affine[:, :C0].contiguous()
        if self._split_fwd(C0, Cout):
            nat.call("u3d_conv3d_f32s", dev.index, _stream(dev), _p(src.t0), _p(a0), _p(self._packed_f32s(conv.weight, 0, dev, C0, 0)),
                     _p(c.y), N, D, H, W, C0, Cout, c.relu, _p(ystats), None, None, _p(part), None, 0,
                     flops=54.0 * C0 * Cout * N * D * H * W)
        else:
            s0 = VSrc(src.t0).struct(a0)
            nat.call("u3d_conv3d_ex_reps", dev.index, _stream(dev), ctypes.byref(s0), _p(self._pack_cache[(id(conv.weight), 10)][1]),
                     _p(c.y), N, D, H, W, Cout, c.relu, _p(ystats), None, None, _p(part), None, 0, _reps(ystats),
                     flops=54.0 * C0 * Cout * N * D * H * W)
        return ystats

    def _fwd_f32s(self, c: "_ConvCall"):
        # fp32 operands split into three bf16 values each, six partial products on the bf16 MFMA pipe (csrc/u3d_bf16.hip)
        ystats = c.take_stats()
        need = nat.get_lib().u3d_conv3d_bf16_workspace_floats(c.N, c.D, c.H, c.W, c.Ctot, c.Cout)
        kws = _empty(need, dtype=_F32, device=c.dev) if need > 0 else None
        nat.call("u3d_conv3d_f32s", c.dev.index, _stream(c.dev), _p(c.src.t0), _p(c.affine), _p(self._packed_f32s(c.conv.weight, 0, c.dev)),
                 _p(c.y), c.N, c.D, c.H, c.W, c.Ctot, c.Cout, c.relu, _p(ystats), None, None, _p(c.residual), _p(kws), need, flops=c.flops)
        return ystats

    def _fwd_bf16(self, c: "_ConvCall"):
        # bf16 MFMA operands, fp32 accumulation / epilogue (csrc/u3d_bf16.hip); with bf16 activation storage the input, the
        # output and the residual are bf16 tensors (`_b16` entry point)
        ystats = c.take_stats()
        need = nat.get_lib().u3d_conv3d_bf16_workspace_floats(c.N, c.D, c.H, c.W, c.Ctot, c.Cout)  # split-K scratch at the bottom of the U
        kws = _empty(need, dtype=_F32, device=c.dev) if need > 0 else None
        nat.call("u3d_conv3d_bf16_ex" + ("_b16" if c.b16 else ""), c.dev.index, _stream(c.dev), _p(c.src.t0), _p(c.affine),
                 _p(self._packed_bf16(c.conv.weight, 0, c.dev)), _p(c.y), c.N, c.D, c.H, c.W, c.Ctot, c.Cout, c.relu, _p(ystats), None, None,
                 _p(c.residual), _p(kws), need, flops=c.flops)
        return ystats

    def _fwd_fp32(self, c: "_ConvCall"):
        wp = self._packed(c.conv.weight, 0, c.dev)
        ystats = c.take_stats(self.stat_reps)  # (replica rows: the persistent kernel's blocks spread their same-address f64 atomics)
        s = c.src.struct(c.affine)
        # bottom-of-the-U shapes split the channel reduction over blocks through a scratch buffer (0 floats otherwise)
        need = nat.get_lib().u3d_conv3d_workspace_floats(c.N, c.D, c.H, c.W, c.Ctot, c.Cout)
        kws = _empty(need, dtype=_F32, device=c.dev) if need > 0 else None
        nat.call("u3d_conv3d_ex_reps", c.dev.index, _stream(c.dev), ctypes.byref(s), _p(wp), _p(c.y), c.N, c.D, c.H, c.W, c.Cout, c.relu,
                 _p(ystats), None, None, _p(c.residual), _p(kws), need, _reps(ystats), flops=c.flops)
        return ystats

    def _single_conv_fwd(self, sc, name, src: VSrc, st_in, pool: _StatPool, tape: Optional[Tape], want_stats=True,
                         residual: Optional[torch.Tensor] = None, sub=(), y_out: Optional[torch.Tensor] = None, act=None,
                         record_only: bool = False):
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
        split = None  # sub-pixel layers: compact (N,C0,2) / (N,C1,2) copies of the affine rows, written by the finalize launch
        if post:
            affine, mean_rstd = self._identity_affine(N, Ctot, dev), None
        else:
            st0, C0, sc0, st1, C1, sc1 = st_in
            affine = _empty((N, Ctot, 2), dtype=_F32, device=dev)
            if spec.norm == "g" and src.t1 is not None and residual is None and sub and id(conv.weight) in sub:
                # sub-pixel layer: its two halves are read by different kernels as plain tensors
                Cs0, Cs1 = sub[id(conv.weight)]
                split = (Cs0, _empty((N, Cs0, 2), dtype=_F32, device=dev),
                         _empty((N, Cs1, 2), dtype=_F32, device=dev) if any(src.plus) else None)
            mean_rstd = self._norm_finalize(spec.norm, gn, st0, C0, sc0, st1, C1, sc1, N, G, float(D * H * W), affine, dev, split)
        # y_out: recomputation under activation checkpointing rewrites the (still alive) block output in place with the
        # bit-identical values instead of allocating a second copy
        b16 = src.t0.dtype == torch.bfloat16  # bf16 activation storage: only the bf16-operand branch below handles it
        if b16:
            assert self.act_bf16 and src.t1 is None and not post and self._bf16_layer(Ctot, Cout) and act == ACT_RELU, \
                "bf16 activation storage reached a layer outside its envelope"
        y = y_out if (y_out is not None and not post) else _empty((N, D, H, W, Cout), dtype=src.t0.dtype if b16 else _F32, device=dev)
        # ---- the convolution itself: ONE decision (which kernel family), then the family's launcher from the table
        call = _ConvCall(dev, conv, src, affine, y, N, D, H, W, Ctot, Cout, relu, bool(want_stats and self.fused_stats), pool, conv_res,
                         sub, b16, *(split[1:] if split is not None else (None, None)))
        family = self._fwd_family(call, residual)
        small = family == "small"
        dmod = getattr(sc, "dropout", None) if spec.drop == "d" else (getattr(sc, "dropout2d", None) if spec.drop == "D" else None)
        # record_only (recomputation under activation checkpointing, the block's LAST convolution): y_out still holds this layer's final
        # output — the block output the forward pass kept — so only what backward needs beside it is rebuilt (the GroupNorm tables of
        # the layer's input, above): no convolution launch, no in-place activation pass.  Not for post-norm orders (backward wants the
        # pre-norm tensor z) and not under dropout (its mask belongs to the record).
        if record_only and y_out is not None and not post and not (dmod is not None and dmod.training and dmod.p > 0.0):
            if tape is not None:
                nw = gn.weight if gn is not None else None
                tape.convs.append(
                    ConvRec(name, src, affine, mean_rstd, y, nw, conv.weight, G,
                            self._pindex[id(gn.weight)] if gn is not None else -1,
                            self._pindex[id(gn.bias)] if gn is not None else self._pindex[id(conv.bias)],
                            self._pindex[id(conv.weight)], small,
                            sub.get(id(conv.weight)) if (sub and src.t1 is not None and residual is None) else None,
                            not post, None, spec.norm, bn_training, None, call.affine_lo, call.affine_hi)
                )
            return y, None
        ystats = getattr(self, self._FWD_KERNELS[family])(call)
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
                        not post, post_rec, spec.norm, bn_training, drop_rec, call.affine_lo, call.affine_hi)
            )
        return y, ystats

    # ---- backward kernel families ---------------------------------------------------------------------------------------------------
    _WGRAD_KERNELS = {
        "bf16": "_wgrad_bf16",          # u3d_bf16.hip (Cin % 32 == 0 and Cout % 32 == 0: every layer the bf16 forward covers)
        "subpixel": "_wgrad_subpixel",  # u3d_subpix.hip (upsampled channels) + u3d_conv.hip strided (skip channels)
        "fp32_side": "_wgrad_fp32_side",  # u3d_conv.hip on the side stream (U3D_SIDE_VOXELS, off by default)
        "fp32": "_wgrad_fp32",          # u3d_conv.hip
    }
    _DGRAD_KERNELS = {
        "subpixel": "_dgrad_subpixel",  # skip half at full resolution + upsampled half directly at LOW resolution
        "f32s": "_dgrad_f32s",
        "bf16": "_dgrad_bf16",
        "fp32": "_dgrad_fp32",
    }

    def _wgrad_family(self, c: "_BwdCall") -> str:
        if c.bf16 and c.Cout % 32 == 0:  # (Cout % 64 == 32 since round 4: 64-column blocks with a zero upper half)
            return "bf16"
        if c.rec.sub is not None:
            return "subpixel"
        if self.overlap_small_wgrad and c.N * c.D * c.H * c.W <= c.cx.SIDE_MAX_VOXELS and self.debug is None:
            return "fp32_side"
        return "fp32"

    def _dgrad_family(self, c: "_BwdCall") -> str:
        if c.rec.sub is not None:
            return "subpixel"
        if c.src.t1 is None and not c.rec.small and self._split_dgrad(c.src.C, c.Cout):
            return "f32s"
        return "bf16" if c.bf16 else "fp32"

    def _wgrad_bf16(self, c: "_BwdCall"):
        cx, dev, src, rec = c.cx, c.cx.dev, c.src, c.rec
        need = nat.get_lib().u3d_wgrad_bf16_workspace_floats(c.N, c.D, c.H, c.W, src.C, c.Cout)
        ws = cx.ensure_ws(need)
        dw = cx.gview(rec.idx_w)
        job = c.job
        # the GroupNorm-backward reduction of the layer's input rides in the launch that adds the splits — where there is one
        if job is not None and nat.get_lib().u3d_conv3d_wgrad_bf16_job_supported(c.N, c.D, c.H, c.W, src.C, c.Cout, 1 if c.b16 else 0, _p(dw),
                                                                                  job.N, job.C0 + job.C1, job.G) != 1:
            job = None
        nat.call("u3d_conv3d_wgrad_bf16" + ("_b16" if c.b16 else "") + "_job", dev.index, _stream(dev), _p(src.t0), _p(rec.affine), _p(c.dz),
                 _p(dw), c.N, c.D, c.H, c.W, src.C, c.Cout, _p(ws), ws.numel(), ctypes.byref(job) if job is not None else None,
                 flops=c.flops)
        if job is not None:
            c.job = None

    def _wgrad_subpixel(self, c: "_BwdCall"):
        # weight gradient in two channel slices of the same (Cout, Ctot, 27) buffer: upsampled channels from the 64
        # (parity class, tap half) matrices over the low-res grid, skip channels from the standard kernel
        cx, dev, src, rec, ws = c.cx, c.cx.dev, c.src, c.rec, c.cx.ws
        C0, C1 = rec.sub
        Ct = src.C
        dwv = cx.gview(rec.idx_w)
        plus = src.plus
        if any(plus):
            # n -> 2n + 1: outputs d >= 2 of the shifted axes through the windowed sub-pixel kernel, the slab d < 2 through the general
            # kernel on disjoint boxes of dz (each into scratch, added to the same channel slice)
            win = (ctypes.c_int * 9)(c.D, c.H, c.W, *plus, *plus)
            nat.call("u3d_subpixel_conv_wgrad_win", dev.index, _stream(dev), _p(src.t1), _p(rec.affine.view(-1)[2 * C0:]), Ct * 2,
                     _p(c.dz), _p(dwv[C0 * 27:]), Ct, c.N, src.D1, src.H1, src.W1, C1, c.Cout, _p(ws), ws.numel(), win,
                     flops=128.0 * C1 * c.Cout * c.N * src.D1 * src.H1 * src.W1)
            s_up = src.up_only_struct(rec.affine_hi if rec.affine_hi is not None else rec.affine[:, C0:].contiguous())
            slice_view = dwv.view(c.Cout, Ct, 27)[:, C0:, :]
            for box in slab_boxes((c.D, c.H, c.W), plus, 2):
                tmp = _empty((c.Cout, C1, 27), dtype=_F32, device=dev)
                nat.call("u3d_conv3d_wgrad_box", dev.index, _stream(dev), ctypes.byref(s_up), _p(c.dz), _p(tmp), c.N, c.D, c.H, c.W,
                         c.Cout, _p(ws), ws.numel(), (ctypes.c_int * 6)(*box),
                         flops=54.0 * C1 * c.Cout * c.N * (box[3] - box[0]) * (box[4] - box[1]) * (box[5] - box[2]))
                slice_view += tmp
        else:
            nat.call("u3d_subpixel_conv_wgrad", dev.index, _stream(dev), _p(src.t1), _p(rec.affine.view(-1)[2 * C0:]), Ct * 2,
                     _p(c.dz), _p(dwv[C0 * 27:]), Ct, c.N, src.D1, src.H1, src.W1, C1, c.Cout, _p(ws), ws.numel(),
                     flops=128.0 * C1 * c.Cout * c.N * src.D1 * src.H1 * src.W1)
        a0 = rec.affine_lo if rec.affine_lo is not None else rec.affine[:, :C0].contiguous()
        s0 = VSrc(src.t0).struct(a0)
        nat.call("u3d_conv3d_wgrad_job", dev.index, _stream(dev), ctypes.byref(s0), _p(c.dz), _p(dwv), Ct, c.N, c.D, c.H, c.W,
                 c.Cout, _p(ws), ws.numel(), ctypes.byref(c.job) if c.job is not None else None,
                 flops=54.0 * C0 * c.Cout * c.N * c.D * c.H * c.W)
        c.job = None

    def _wgrad_fp32_side(self, c: "_BwdCall"):
        # small layer: neither kernel fills the chip on its own -> weight gradient on the side stream, data gradient
        # on the caller's stream; joined before anything consumes the flat gradient buffer
        cx, dev, src, rec = c.cx, c.cx.dev, c.src, c.rec
        s_aff = src.struct(rec.affine)
        need = nat.get_lib().u3d_wgrad_workspace_floats(c.N, c.D, c.H, c.W, src.C, c.Cout)
        side = cx.side_stream(need)
        side.wait_stream(torch.cuda.current_stream(dev))  # dz (and the flat buffer) are ready
        with torch.cuda.stream(side):
            nat.call("u3d_conv3d_wgrad", dev.index, _stream(dev), ctypes.byref(s_aff), _p(c.dz), _p(cx.gview(rec.idx_w)), c.N,
                     c.D, c.H, c.W, c.Cout, _p(cx.ws_side), cx.ws_side.numel(), flops=c.flops)
        c.dz.record_stream(side)  # dz is released on the main stream while the side stream may still read it
        cx.side_used = True

    def _wgrad_fp32(self, c: "_BwdCall"):
        cx, dev, src, rec, ws = c.cx, c.cx.dev, c.src, c.rec, c.cx.ws
        s_aff = src.struct(rec.affine)
        nat.call("u3d_conv3d_wgrad_job", dev.index, _stream(dev), ctypes.byref(s_aff), _p(c.dz), _p(cx.gview(rec.idx_w)), 0, c.N, c.D,
                 c.H, c.W, c.Cout, _p(ws), ws.numel(), ctypes.byref(c.job) if c.job is not None else None, flops=c.flops)
        c.job = None

    def _dgrad_subpixel(self, c: "_BwdCall"):
        # skip half at full resolution; upsampled half directly at LOW resolution (the children sum of the nearest
        # upsampling is folded into the 4x4x4-tap stride-2 gather).  dg = (dg_skip, dlow)
        cx, dev, src, rec, ws, pool = c.cx, c.cx.dev, c.src, c.rec, c.cx.ws, c.cx.pool
        Nn, Dd, Hh, Ww, Cout = c.N, c.D, c.H, c.W, c.Cout
        C0, C1 = rec.sub
        dg0 = _empty((Nn, Dd, Hh, Ww, C0), dtype=_F32, device=dev)
        dlow = _empty_like(src.t1)
        split0 = self._split_dgrad(C0, Cout)
        plus = src.plus
        gst0 = _take_reps(pool, Nn * C0 * 2, 1 if split0 else c.greps)
        gst1 = _take_reps(pool, Nn * C1 * 2, 1 if any(plus) else c.greps)  # (the windowed form of an n -> 2n + 1 level keeps one row)
        if split0:
            need = nat.get_lib().u3d_conv3d_bf16_workspace_floats(Nn, Dd, Hh, Ww, Cout, C0)
            kws = cx.ensure_ws(need) if need > 0 else None
            nat.call("u3d_conv3d_f32s", dev.index, _stream(dev), _p(c.dz), None, _p(self._packed_f32s(rec.conv_w, 1, dev, C0, 0)),
                     _p(dg0), Nn, Dd, Hh, Ww, Cout, C0, 0, None, _p(src.t0), _p(gst0), None, _p(kws), need,
                     flops=54.0 * C0 * Cout * Nn * Dd * Hh * Ww)
        else:
            s_dz = VSrc(c.dz).struct()
            s_x0 = VSrc(src.t0).struct()
            nat.call("u3d_conv3d_ex_reps", dev.index, _stream(dev), ctypes.byref(s_dz), _p(self._packed_sub(rec, 11, dev)), _p(dg0),
                     Nn, Dd, Hh, Ww, C0, 0, None, ctypes.byref(s_x0), _p(gst0), None, _p(ws), ws.numel(), _reps(gst0),
                     flops=54.0 * C0 * Cout * Nn * Dd * Hh * Ww)
        plus = src.plus
        if any(plus):
            # n -> 2n + 1: the share of the outputs d >= 2 from the windowed sub-pixel kernel; the slab's share as a full-resolution
            # gradient of the upsampled channels inside the slab's dilation (general kernel on dz masked to the slab), folded into the
            # first low-res cells of the shifted axes together with its part of the GroupNorm-backward sums
            win = (ctypes.c_int * 9)(Dd, Hh, Ww, *plus, *plus)
            nat.call("u3d_subpixel_conv_dgrad_win", dev.index, _stream(dev), _p(c.dz), _p(self._packed_sub(rec, 13, dev)), _p(src.t1),
                     _p(dlow), _p(gst1), Nn, src.D1, src.H1, src.W1, C1, Cout, win,
                     flops=128.0 * C1 * Cout * Nn * src.D1 * src.H1 * src.W1)
            dv = _empty((Nn, Dd, Hh, Ww, C1), dtype=_F32, device=dev)
            s_dz2 = VSrc(c.dz).struct()
            wpd1 = self._packed_sub(rec, 15, dev)
            mask = (ctypes.c_int * 3)(*(2 * e for e in plus))
            # (the two slab launches are one tile-time each on less than half of the chip's block slots — a 2-voxel slab in 4 x 8 x 8
            # tiles — but running them side by side on two streams was measured in round 6: each takes twice as long, 192 + 214 us
            # against 114 + 113; the waste is MFMA work on dead tile rows, not idle slots)
            for box in slab_boxes((Dd, Hh, Ww), plus, 3):
                nat.call("u3d_conv3d_box", dev.index, _stream(dev), ctypes.byref(s_dz2), _p(wpd1), _p(dv), Nn, Dd, Hh, Ww, C1,
                         (ctypes.c_int * 6)(*box), mask,
                         flops=54.0 * C1 * Cout * Nn * (box[3] - box[0]) * (box[4] - box[1]) * (box[5] - box[2]))
            lz, ly, lx = src.los
            nat.call("u3d_nearest_childsum_add", dev.index, _stream(dev), _p(dv), _p(src.t1), _p(dlow), _p(gst1), Nn, Dd, Hh, Ww,
                     src.D1, src.H1, src.W1, C1, _p(lz), _p(ly), _p(lx), *plus)
            del dv
        else:
            nat.call("u3d_subpixel_conv_dgrad_reps", dev.index, _stream(dev), _p(c.dz), _p(self._packed_sub(rec, 13, dev)), _p(src.t1),
                     _p(dlow), _p(gst1), Nn, src.D1, src.H1, src.W1, C1, Cout, _reps(gst1),
                     flops=128.0 * C1 * Cout * Nn * src.D1 * src.H1 * src.W1)
        return (dg0, dlow), (gst0, gst1)  # (_norm_bwd_finalize takes the two tables as they are)

    def _dgrad_f32s(self, c: "_BwdCall"):
        cx, dev, src, rec = c.cx, c.cx.dev, c.src, c.rec
        dg = _empty((c.N, c.D, c.H, c.W, src.C), dtype=_F32, device=dev)
        gst = cx.pool.take(c.N * src.C * 2)
        need = nat.get_lib().u3d_conv3d_bf16_workspace_floats(c.N, c.D, c.H, c.W, c.Cout, src.C)
        kws = cx.ensure_ws(need) if need > 0 else None
        nat.call("u3d_conv3d_f32s", dev.index, _stream(dev), _p(c.dz), None, _p(self._packed_f32s(rec.conv_w, 1, dev)), _p(dg),
                 c.N, c.D, c.H, c.W, c.Cout, src.C, 0, None, _p(src.t0), _p(gst), None, _p(kws), need, flops=c.flops)
        return dg, gst

    def _dgrad_bf16(self, c: "_BwdCall"):
        cx, dev, src, rec = c.cx, c.cx.dev, c.src, c.rec
        dg = _empty((c.N, c.D, c.H, c.W, src.C), dtype=c.dz.dtype if c.b16 else _F32, device=dev)
        gst = cx.pool.take(c.N * src.C * 2)
        need = nat.get_lib().u3d_conv3d_bf16_workspace_floats(c.N, c.D, c.H, c.W, c.Cout, src.C)
        kws = cx.ensure_ws(need) if need > 0 else None
        nat.call("u3d_conv3d_bf16_ex" + ("_b16" if c.b16 else ""), dev.index, _stream(dev), _p(c.dz), None,
                 _p(self._packed_bf16(rec.conv_w, 1, dev)), _p(dg), c.N, c.D, c.H, c.W, c.Cout, src.C, 0, None, _p(src.t0), _p(gst),
                 None, _p(kws), need, flops=c.flops)
        return dg, gst

    def _dgrad_fp32(self, c: "_BwdCall"):
        cx, dev, src, rec, ws = c.cx, c.cx.dev, c.src, c.rec, c.cx.ws
        wpd = self._packed(rec.conv_w, 1, dev)
        dg = _empty((c.N, c.D, c.H, c.W, src.C), dtype=_F32, device=dev)
        gst = _take_reps(cx.pool, c.N * src.C * 2, c.greps)
        s_dz = VSrc(c.dz).struct()
        s_x = src.struct()
        nat.call("u3d_conv3d_ex_reps", dev.index, _stream(dev), ctypes.byref(s_dz), _p(wpd), _p(dg), c.N, c.D, c.H, c.W, src.C, 0, None,
                 ctypes.byref(s_x), _p(gst), None, _p(ws), ws.numel(), _reps(gst), flops=c.flops)
        return dg, gst

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
        bf16 = src.t1 is None and rec.sub is None and not rec.small and self._bf16_layer(src.C, Cout)
        b16 = src.t0.dtype == torch.bfloat16  # bf16 activation storage
        assert not b16 or (bf16 and Cout % 64 == 0 and dz_.dtype == torch.bfloat16)
        call = _BwdCall(cx, rec, dz_, src, Nn, Dd, Hh, Ww, Cout, bf16, b16)
        # ---- data gradient (+ the GroupNorm-backward sums of the conv input), then weight gradient: one family decision each.  The
        # one-block reduction of those sums rides in the weight gradient's reduce launch where the family takes a job (round 6)
        # (replica rows for the sums only where the weight-gradient launch will take the reduction as a job: it reads them in that form)
        if (self.stat_reps > 1 and rec.pre_norm and rec.norm == "g" and _WGRAD_JOB and self._wgrad_family(call) in ("fp32", "subpixel")
                and nat.get_lib().u3d_conv3d_wgrad_job_supported(Nn, src.C, rec.G) == 1):
            call.greps = self.stat_reps
        dg, gst = getattr(self, self._DGRAD_KERNELS[self._dgrad_family(call)])(call)
        if self.debug is not None and rec.sub is None:
            self.debug[rec.name + ".dg"] = dg.clone()
        coef = None
        if rec.pre_norm:
            coef = _empty((Nn, 3, src.C), dtype=_F32, device=dev)
            call.job = self._norm_bwd_job(cx, rec, gst, Nn, src.C, float(Dd * Hh * Ww), coef)
        had_job = call.job is not None
        getattr(self, self._WGRAD_KERNELS[self._wgrad_family(call)])(call)
        if not rec.pre_norm:
            return dg, self._identity_coef(Nn, src.C, dev)  # no GroupNorm on the conv input: dx = dg
        if not had_job or call.job is not None:  # (no job, or a weight-gradient family without a reduce launch to carry it)
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
        if sub is not None:  # skip slice (fp32 kernels) + sub-pixel slice + the slab boxes of an n -> 2n + 1 level
            # (a sub-pixel level's odd axes are its n -> 2n + 1 axes)
            boxes = slab_boxes((D, H, W), (D % 2, H % 2, W % 2), 2)
            return max([lib.u3d_wgrad_workspace_floats(N, D, H, W, sub[0], Cout),
                        lib.u3d_subpixel_wgrad_workspace_floats(N, D // 2, H // 2, W // 2, sub[1], Cout),
                        lib.u3d_conv3d_workspace_floats(N, D, H, W, Cout, sub[0])] +
                       [lib.u3d_wgrad_workspace_floats(N, b[3] - b[0], b[4] - b[1], b[5] - b[2], sub[1], Cout) for b in boxes])
        if not virtual and self._bf16_layer(Cin, Cout):
            need = lib.u3d_conv3d_bf16_workspace_floats(N, D, H, W, Cout, Cin)  # data gradient: roles swapped
            wg = lib.u3d_wgrad_bf16_workspace_floats(N, D, H, W, Cin, Cout)
            return max(need, wg)
        return max(lib.u3d_wgrad_workspace_floats(N, D, H, W, Cin, Cout), lib.u3d_conv3d_workspace_floats(N, D, H, W, Cout, Cin))

    def _wgrad_workspace_floats(self, convs):
        """scratch floats the backward kernels of these recorded layers need (one shared buffer, sized once per backward)"""
        need = 0
        for r in convs:
            need = max(need, self._layer_ws_floats(r.src.N, r.src.D, r.src.H, r.src.W, r.src.C, r.y.shape[-1], r.sub, r.small,
                                                   r.src.t1 is not None))
        return int(need)
