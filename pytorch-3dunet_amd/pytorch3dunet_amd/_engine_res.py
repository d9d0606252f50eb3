"""The residual executors (ResidualUNet3D / ResidualUNetSE3D: reference model.py:193-278, buildingblocks.py:230-307,617-664,
se.py:18-114)."""
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
from ._engine_unet import UNet3DEngine

_CKPT_RERUN_LAST = os.environ.get("U3D_CKPT_RERUN_LAST", "0") == "1"  # A/B: recomputation re-runs a block's last convolution too (rounds 4-5)


class ResUNetEngine(UNet3DEngine):
    """Native executor of ResidualUNet3D (model.py:193-234): ResNetBlock encoders (max-pool down), decoders that upsample
    with ConvTranspose3d(k3,s2,p1) -> nearest resize -> sum with the skip, then a ResNetBlock; same head.

    The 3x3x3 convolutions (94 % of the FLOPs at BASELINE config 4) run on the same MFMA kernels as UNet3D, with the
    block's `out += residual; ReLU` fused into conv3's epilogue (u3d_conv3d_residual); GroupNorm statistics of the
    residual come out of the 1x1x1 conv's / the joining kernel's epilogue.  The 1x1x1 convolutions and the transposed
    convolution run on the FP32 vector units (csrc/u3d_res.hip)."""

    def __init__(self, model):
        super().__init__(model)
        self.stat_reps = 1  # (the residual executor's own consumers of the statistics tables read plain ones)
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
        # (recomputation under checkpointing: the block output is still alive in y_out — conv3 is NOT run again, only its record is
        # rebuilt; U3D_CKPT_RERUN_LAST=1 re-runs it as rounds 4-5 did (A/B; bit-identical).  An SE block's backward needs conv3's own
        # output, which the forward pass did not keep: re-run.)
        skip3 = (getattr(self, "_in_recompute", False) and se_mod is None and y_out is not None and not _CKPT_RERUN_LAST)
        y, y_st = self._single_conv_fwd(bm.conv3, name + ".c3", src3, (st2, Cout, 1.0, None, 0, 0.0), pool, tape,
                                        want_stats=se_mod is not None, residual=r, y_out=y_out if se_mod is None else None,
                                        act=(self.act, self.slope), record_only=skip3)
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
        nat.call("u3d_se_apply_fwd" + ("_b16" if self.act_bf16 else ""), dev.index, _stream(dev), _p(y), _p(st["gc"]), _p(ws), _p(bs), N, V, C, mode, _p(out),
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
        nat.call("u3d_se_bwd_reduce" + ("_b16" if self.act_bf16 else ""), dev.index, _stream(dev), _p(dout), _p(y), _p(se["gc"]), _p(se["a"]), _p(ws), N, V, C, mode,
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
        nat.call("u3d_se_bwd_apply" + ("_b16" if self.act_bf16 else ""), dev.index, _stream(dev), _p(dout), _p(y), _p(se["gc"]), _p(se["a"]), _p(ws), _p(dls), _p(ds),
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
            if tape is not None and self.checkpoint_encoders and (self.checkpoint_levels is None or i < self.checkpoint_levels):
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
                need = nat.get_lib().u3d_convtr3d_fwd_t8_workspace_floats(Nl, D1, H1, W1, Cl, Cs) if self.act_bf16 else 0
                if need > 0:  # small grid, many channels: the flat tile with a split channel reduction (csrc/u3d_bf16.hip)
                    kws = _empty(need, dtype=_F32, device=dev)
                    nat.call("u3d_convtr3d_fwd_t8_b16_ex", dev.index, _stream(dev), _p(cur), _p(self._packed_convtr_t8(ct.weight, 0, dev)),
                             _p(t), Nl, D1, H1, W1, Cl, Cs, _p(kws), need, flops=2.0 * 27 * Cl * Cs * Nl * D1 * H1 * W1)
                else:
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
                lib = nat.get_lib()
                need = max(lib.u3d_convtr3d_wgrad_t8_workspace_floats(Nl, D1, H1, W1, Cl, Cs),
                           lib.u3d_convtr3d_dgrad_t8_workspace_floats(Nl, D1, H1, W1, Cl, Cs))  # (both kernels: same stream, one after the other)
                wsb = cx.ensure_ws(need)
                nat.call("u3d_convtr3d_wgrad_t8" + sfx, dev.index, _stream(dev), _p(xl), _p(dt8),
                         _p(gview(self._pindex[id(up.weight)])), Nl, D1, H1, W1, Cl, Cs, _p(wsb), wsb.numel(),
                         flops=2.0 * 27 * Cl * Cs * Nl * D1 * H1 * W1)
                dxl = _empty_like(xl)
                nat.call("u3d_convtr3d_dgrad_t8" + sfx + "_ex", dev.index, _stream(dev), _p(dt8), _p(self._packed_convtr_t8(up.weight, 1, dev)),
                         _p(xl) if mk else None, _p(dxl), Nl, D1, H1, W1, Cl, Cs, _p(wsb), wsb.numel(),
                         flops=2.0 * 27 * Cl * Cs * Nl * D1 * H1 * W1)
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
