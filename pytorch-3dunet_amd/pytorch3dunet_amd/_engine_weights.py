"""Weight images of the native executor: the packed / pre-summed / bf16 / split-fp32 forms of every convolution weight that the
kernels read (include/u3d.h: u3d_pack_weights*), cached per parameter version and re-packed in ONE launch per step.  Mixin of
`engine.UNet3DEngine`; reference counterpart: none (ATen reorders weights inside its convolution algorithms, buildingblocks.py:56)."""
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


_PACK_BOTH = os.environ.get("U3D_PACK_BOTH", "1") != "0"  # A/B: 0 = the forward and data-gradient bf16 images of a weight as two reads of it
_PACK_ELEMENTWISE = os.environ.get("U3D_PACK_ELEMENTWISE", "0") == "1"  # A/B: every image through the thread-per-slot packer


class WeightImages:
    """mixin: needs self._pack_cache, self._salt, self._const, self.bf16, self.split, self.small_cin, self._sub_pairs"""

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

    def _t8_weights(self):
        """ConvTranspose3d weights whose forward / data gradient run in space-to-depth form (residual nets, summation joining)"""
        ws = getattr(self, "_t8w", None)
        if ws is None:
            ws = []
            concat = getattr(self, "dec_concat", None)
            if self.bf16 and concat is not None:
                for j, (ct, _) in enumerate(self.dec):
                    if isinstance(ct, torch.nn.ConvTranspose3d) and not concat[j] and self._convtr_t8(ct.in_channels, ct.out_channels):
                        ws.append(ct.weight)
            self._t8w = ws
        return ws

    def _conv_weights(self):
        """every 3x3x3 conv weight the MFMA kernels read through a packed image"""
        out = []
        for mod in self.model.modules():
            if isinstance(mod, torch.nn.Conv3d) and mod.kernel_size == (3, 3, 3):
                out.append(mod.weight)
        return out

    # pack modes: 0 forward, 1 data gradient (u3d_pack_weights).  Layers in self._sub (sub-pixel path) use instead: 10 / 11 =
    # forward / data-gradient image of the first C0 input channels, 12 / 13 = sub-pixel forward / data-gradient image of the
    # remaining C1 — and no mode-0 / mode-1 image; levels that upsample n -> 2n + 1 add 14 / 15 = the plain 27-tap images of
    # those C1 channels (slab launches).
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
            if mode in (14, 15):  # the plain 27-tap images of the upsampled channels: the slab launches of an n -> 2n + 1 level
                return w.data_ptr() + C0 * 27 * 4, C1, mode - 14, Cin, lib.u3d_packed_weight_floats(C1, Cout, mode - 14)
            return w.data_ptr() + C0 * 27 * 4, C1, 3, Cin, lib.u3d_subpixel_dgrad_packed_floats(Cout, C1)
        return w.data_ptr(), Cin, mode, 0, lib.u3d_packed_weight_floats(Cin, Cout, mode)

    def _repack_bf16_all(self, dev, modes, ws):
        """bf16 fragment images of every bf16 layer whose parameter changed: ONE launch at HBM rate (u3d_pack_weights_bf16_batch)
        instead of one strided-read launch per layer and mode (36 + 36 per config-4 step, 1.0 ms -> 0.25 ms)"""
        lib = nat.get_lib()
        stale = []  # (weight, C-ABI mode of the batch kernel, cache slot): 3x3x3 images 0 / 1 -> slots 20 / 21; T8 images 4 / 5 -> 30 / 31
        for w in ws:
            if not self._bf16_layer(w.shape[1], w.shape[0]) or id(w) in self._virtual_w or w.data_ptr() % 16 != 0:
                continue  # (the batch kernel reads 16 bytes per lane; an unaligned view is packed on demand by _packed_bf16)
            for mode in modes:
                hit = self._pack_cache.get((id(w), 20 + mode))
                if hit is None or hit[0] != self._ver(w):
                    stale.append((w, mode, 20 + mode))
        # the space-to-depth images of the transposed convolutions ride in the same launch (round 5; the per-weight kernel read 4
        # bytes per lane at a stride of 27 floats: 8 launches of ~33 us per config-4 step)
        for w in self._t8_weights():
            if w.data_ptr() % 16 != 0 or lib.u3d_pack_weights_bf16_blocks(w.shape[0], w.shape[1], 4) == 0:
                continue  # (Cs % 32 != 0: packed on demand by _packed_convtr_t8)
            for mode in modes:
                hit = self._pack_cache.get((id(w), 30 + mode))
                if hit is None or hit[0] != self._ver(w):
                    stale.append((w, 4 + mode, 30 + mode))
        if not stale:
            return
        # a 3x3x3 weight whose forward AND data-gradient image are stale (every training step) is read ONCE: mode 6 writes both images,
        # the data-gradient one right behind the forward one in one buffer (round 6; the two modes read 1.13 GB per config-4 step)
        if _PACK_BOTH:
            both = {id(w) for w, mode, _ in stale if mode == 0} & {id(w) for w, mode, _ in stale if mode == 1}
            merged = []
            for w, mode, slot in stale:
                if id(w) in both and mode in (0, 1) and lib.u3d_pack_weights_bf16_blocks(w.shape[1], w.shape[0], 6) > 0:
                    if mode == 0:
                        merged.append((w, 6, (20, 21)))
                else:
                    merged.append((w, mode, slot))
            stale = merged
        key = tuple((id(w), mode, w.data_ptr()) for w, mode, _ in stale)
        tab = getattr(self, "_pack_tables_bf16", None)
        if tab is None:
            tab = self._pack_tables_bf16 = {}
        ent = tab.get(key)
        if ent is None:
            descs = (nat.U3DPackDesc * len(stale))()
            bufs, first = [], 0
            for i, (w, mode, slot) in enumerate(stale):
                if mode in (4, 5):  # ConvTranspose3d weight (Cl, Cs, 3,3,3): desc.Cin = Cl, desc.Cout = Cs
                    Cin, Cout = w.shape[0], w.shape[1]
                    n = lib.u3d_convtr3d_t8_packed_elems(Cin, Cout, mode - 4)
                elif mode == 6:
                    Cout, Cin = w.shape[0], w.shape[1]
                    n0, n1 = lib.u3d_packed_weight_bf16_elems(Cin, Cout, 0), lib.u3d_packed_weight_bf16_elems(Cin, Cout, 1)
                    h0, h1 = self._pack_cache.get((id(w), 20)), self._pack_cache.get((id(w), 21))
                    if (h0 is not None and h1 is not None and h0[1].numel() == n0 and h1[1].numel() == n1 and h0[1].device == dev
                            and h1[1].data_ptr() == h0[1].data_ptr() + 2 * n0
                            and h0[1].untyped_storage().data_ptr() == h1[1].untyped_storage().data_ptr()):
                        pair = (h0[1], h1[1])  # (the two views of last step's buffer)
                    else:
                        whole = _empty(n0 + n1, dtype=torch.bfloat16, device=dev)
                        pair = (whole[:n0], whole[n0:])
                    bufs.append(pair)
                    descs[i].w, descs[i].packed, descs[i].first = w.data_ptr(), pair[0].data_ptr(), first
                    descs[i].Cout, descs[i].Cin, descs[i].mode, descs[i].cin_stride = Cout, Cin, 6, 0
                    first += lib.u3d_pack_weights_bf16_blocks(Cin, Cout, 6)
                    continue
                else:
                    Cout, Cin = w.shape[0], w.shape[1]
                    n = lib.u3d_packed_weight_bf16_elems(Cin, Cout, mode)
                hit = self._pack_cache.get((id(w), slot))
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
        for (w, mode, slot), buf in zip(stale, bufs):
            if mode == 6:
                self._pack_cache[(id(w), 20)] = (self._ver(w), buf[0])
                self._pack_cache[(id(w), 21)] = (self._ver(w), buf[1])
            else:
                self._pack_cache[(id(w), slot)] = (self._ver(w), buf)

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
                if id(w) in getattr(sub, "plus", ()):
                    wmodes += tuple(mm + 14 for mm in modes)
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
            # two descriptor tables: images whose runs are 16-byte aligned go through the LDS cell kernel (u3d_pack_weights_batch_cells,
            # `first` = first block), the rest through the element-wise kernel (`first` = first float)
            shapes = [self._pack_shape(w, mode) for w, mode in stale]
            blocks = [0 if _PACK_ELEMENTWISE else lib.u3d_pack_weights_cells_blocks(wptr, Cin, w.shape[0], cmode, cstride)
                      for (w, _), (wptr, Cin, cmode, cstride, _) in zip(stale, shapes)]
            bufs = []
            for (w, mode), (wptr, Cin, cmode, cstride, n) in zip(stale, shapes):
                hit = self._pack_cache.get((id(w), mode))
                bufs.append(hit[1] if hit is not None and hit[1].numel() == n and hit[1].device == dev else _empty(n, dtype=_F32, device=dev))
            tables = []
            for cells in (True, False):
                idx = [i for i, b in enumerate(blocks) if (b > 0) == cells]
                descs = (nat.U3DPackDesc * max(len(idx), 1))()
                first = 0
                for k, i in enumerate(idx):
                    (w, mode), (wptr, Cin, cmode, cstride, n) = stale[i], shapes[i]
                    descs[k].w, descs[k].packed, descs[k].first = wptr, bufs[i].data_ptr(), first
                    descs[k].Cout, descs[k].Cin, descs[k].mode, descs[k].cin_stride = w.shape[0], Cin, cmode, cstride
                    first += blocks[i] if cells else n
                host = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8)
                tables.append((host.to(dev), len(idx), first))
            ent = (tables, bufs)
            tab.clear()  # one live table per (set of stale weights): parameters are re-packed together every step
            tab[key] = ent
        tables, bufs = ent
        (tc, nc, bc), (te, ne, fe) = tables
        if nc:
            nat.call("u3d_pack_weights_batch_cells", dev.index, _stream(dev), _p(tc), nc, bc)
        if ne:
            nat.call("u3d_pack_weights_batch", dev.index, _stream(dev), _p(te), ne, fe)
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
