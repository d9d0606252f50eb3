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

import ctypes
from dataclasses import dataclass, field
from typing import List, Optional

import torch

from . import _native as nat
from ._native import U3DSrc

_F32 = torch.float32


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


@dataclass
class Tape:
    convs: List[ConvRec] = field(default_factory=list)
    pools: list = field(default_factory=list)  # (pooled, argmax, e_in) per encoder level > 0
    head_x: Optional[torch.Tensor] = None
    dims: tuple = ()
    x0: Optional[torch.Tensor] = None


class _StatPool:
    """one zero-filled double buffer per pass, handed out in slices (a single memset per forward/backward)"""

    def __init__(self, dev, doubles: int):
        self.buf = torch.zeros(max(doubles, 2), dtype=torch.float64, device=dev)
        self.off = 0

    def take(self, n: int) -> torch.Tensor:
        assert self.off + n <= self.buf.numel(), "stat pool exhausted"
        s = self.buf[self.off : self.off + n]
        self.off += n
        return s


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
        self.params = list(model.parameters())
        self._pindex = {id(p): i for i, p in enumerate(self.params)}
        # static layer table
        self.enc = []
        for enc in model.encoders:
            bm = enc.basic_module
            self.enc.append((enc.pooling is not None, bm.SingleConv1, bm.SingleConv2))
        self.dec = []
        for dec in model.decoders:
            bm = dec.basic_module
            self.dec.append((bm.SingleConv1, bm.SingleConv2))
        # split point of the flat gradient buffer: encoders first (module order), then decoders + head
        n_enc = sum(p.numel() for p in model.encoders.parameters())
        self.n_enc_params = n_enc
        self.n_params = sum(p.numel() for p in self.params)
        offs, o = [], 0
        for p in self.params:
            offs.append(o)
            o += p.numel()
        self.poffs = offs

    # -- helpers ------------------------------------------------------------------------------------
    def _packed(self, w: torch.Tensor, mode: int, dev) -> torch.Tensor:
        key = (id(w), mode)
        ver = (w._version, w.data_ptr())
        hit = self._pack_cache.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1]
        Cout, Cin = w.shape[0], w.shape[1]
        n = nat.get_lib().u3d_packed_weight_floats(Cin, Cout, mode)
        out = torch.empty(n, dtype=_F32, device=dev)
        nat.call("u3d_pack_weights", dev.index, _stream(dev), _p(w.detach()), Cout, Cin, mode, _p(out))
        self._pack_cache[key] = (ver, out)
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

    def _single_conv_fwd(self, sc, name, src: VSrc, st_in, pool: _StatPool, tape: Optional[Tape], want_stats=True):
        dev = src.t0.device
        gn, conv = sc.groupnorm, sc.conv
        N, D, H, W = src.N, src.D, src.H, src.W
        Ctot, Cout, G = src.C, conv.out_channels, gn.num_groups
        assert gn.num_channels == Ctot and conv.in_channels == Ctot
        st0, C0, sc0, st1, C1, sc1 = st_in
        affine = torch.empty((N, Ctot, 2), dtype=_F32, device=dev)
        mean_rstd = torch.empty((N, G, 2), dtype=_F32, device=dev)
        nat.call("u3d_gn_finalize", dev.index, _stream(dev), _p(st0), C0, sc0, _p(st1), C1, sc1, N, G,
                 float(D * H * W), _p(gn.weight.detach()), _p(gn.bias.detach()), float(gn.eps), _p(affine), _p(mean_rstd))
        y = torch.empty((N, D, H, W, Cout), dtype=_F32, device=dev)
        small = self.small_cin and src.t1 is None and Ctot <= 4 and Cout <= 32
        if small:
            # first layer of the network: K = 27*Cin is too small for the MFMA tiling (csrc/u3d_smallc.hip)
            ystats = None
            nat.call("u3d_conv3d_small_cin_fwd", dev.index, _stream(dev), _p(src.t0), _p(affine), _p(conv.weight.detach()),
                     _p(y), N, D, H, W, Ctot, Cout, 1, flops=54.0 * Ctot * Cout * N * D * H * W)
        else:
            wp = self._packed(conv.weight, 0, dev)
            ystats = pool.take(N * Cout * 2) if (want_stats and self.fused_stats) else None
            s = src.struct(affine)
            nat.call("u3d_conv3d", dev.index, _stream(dev), ctypes.byref(s), _p(wp), _p(y), N, D, H, W, Cout, 1, _p(ystats),
                     None, None, flops=54.0 * Ctot * Cout * N * D * H * W)
        if tape is not None:
            tape.convs.append(
                ConvRec(name, src, affine, mean_rstd, y, gn.weight, conv.weight, G, self._pindex[id(gn.weight)],
                        self._pindex[id(gn.bias)], self._pindex[id(conv.weight)], small)
            )
        return y, ystats

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
            x0 = torch.empty((N, D, H, W, Cin), dtype=_F32, device=dev)
            nat.call("u3d_ncdhw_to_ndhwc", dev.index, _stream(dev), _p(x), _p(x0), N, Cin, D * H * W)
        tape = Tape() if save else None
        if tape is not None:
            tape.x0 = x0
            tape.dims = (N, Cin, D, H, W)
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
                pooled = torch.empty((Np, Dp // 2, Hp // 2, Wp // 2, Cp), dtype=_F32, device=dev)
                argmax = torch.empty(pooled.shape, dtype=torch.uint8, device=dev)
                pst = pool.take(Np * Cp * 2)
                nat.call("u3d_maxpool2_fwd", dev.index, _stream(dev), _p(cur), Np, Dp, Hp, Wp, Cp, _p(pooled), _p(argmax),
                         _p(pst))
                if tape is not None:
                    tape.pools.append((pooled, argmax, cur))
                cur, cur_st = pooled, pst
            src = VSrc(cur)
            y1, s1 = self._single_conv_fwd(c1, f"enc{i}.c1", src, self._stats_of(src, cur_st, None, pool, dev), pool, tape)
            src2 = VSrc(y1)
            y2, s2 = self._single_conv_fwd(c2, f"enc{i}.c2", src2, self._stats_of(src2, s1, None, pool, dev), pool, tape)
            feats.append((y2, s2))
            cur, cur_st = y2, s2

        skips = feats[:-1][::-1]  # model.py:126-133
        for j, ((c1, c2), (sk, sk_st)) in enumerate(zip(self.dec, skips)):
            src = VSrc(sk, cur)  # skip channels first (buildingblocks.py:491)
            y1, s1 = self._single_conv_fwd(c1, f"dec{j}.c1", src, self._stats_of(src, sk_st, cur_st, pool, dev), pool, tape)
            src2 = VSrc(y1)
            y2, s2 = self._single_conv_fwd(c2, f"dec{j}.c2", src2, self._stats_of(src2, s1, None, pool, dev), pool, tape)
            cur, cur_st = y2, s2

        # head: 1x1x1 conv + bias + activation (model.py:141-147), NCDHW outputs
        fc = m.final_conv
        Co, Cf = fc.out_channels, fc.in_channels
        V = D * H * W
        logits = torch.empty((N, Co, D, H, W), dtype=_F32, device=dev)
        act = 0
        probs = None
        if m.final_activation is not None:
            act = 1 if isinstance(m.final_activation, torch.nn.Sigmoid) else 2
            probs = torch.empty_like(logits)
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
        flat = torch.empty(self.n_params, dtype=_F32, device=dev)

        def gview(idx):
            p = self.params[idx]
            return flat[self.poffs[idx] : self.poffs[idx] + p.numel()]

        # zeroed double scratch: head (dw,db) + 2 doubles per (n, channel) per conv layer
        fc = m.final_conv
        Co, Cf = fc.out_channels, fc.in_channels
        tot = Co * Cf + Co + sum(N * r.src.C * 2 for r in tape.convs)
        pool = _StatPool(dev, tot)
        # wgrad workspace: max over layers
        lib = nat.get_lib()
        ws_floats = 0
        for r in tape.convs:
            ws_floats = max(ws_floats, lib.u3d_wgrad_workspace_floats(r.src.N, r.src.D, r.src.H, r.src.W, r.src.C,
                                                                      r.y.shape[-1]))
        r0 = tape.convs[0]
        if r0.small:
            ws_floats = max(ws_floats, lib.u3d_small_cin_bwd_workspace_floats(r0.src.N, r0.src.D, r0.src.H, r0.src.W,
                                                                              r0.src.C, r0.y.shape[-1]))
        ws = torch.empty(ws_floats, dtype=_F32, device=dev)

        # ---- head backward: dz of the last decoder conv (ReLU mask fused)
        hacc = pool.take(Co * Cf + Co)
        dz = torch.empty_like(tape.head_x)
        nat.call("u3d_conv1x1_head_bwd", dev.index, _stream(dev), _p(dlogits), _p(tape.head_x), _p(fc.weight.detach()), N, V,
                 Cf, Co, 1, _p(dz), _p(hacc))
        iw, ib = self._pindex[id(fc.weight)], self._pindex[id(fc.bias)]
        assert self.poffs[ib] == self.poffs[iw] + Co * Cf
        nat.call("u3d_cvt_f64_f32", dev.index, _stream(dev), _p(hacc), _p(gview(iw)), Co * Cf + Co)

        n_levels = len(self.enc)
        n_dec = len(self.dec)
        skip_grad = {}  # encoder level -> gradient arriving through the skip connection (pre-mask)

        def conv_bwd(rec: ConvRec, dz_, need_dg=True):
            """wgrad + dgrad + GroupNorm-backward reductions of one SingleConv; returns (dg, coef)"""
            src = rec.src
            Nn, Dd, Hh, Ww = src.N, src.D, src.H, src.W
            Cout = rec.y.shape[-1]
            if self.debug is not None:
                self.debug[rec.name + ".dz"] = dz_.clone()
            if rec.small and not need_dg:
                # one pass gives dw and the GroupNorm-backward sums; no data gradient needed (csrc/u3d_smallc.hip)
                gst = pool.take(Nn * src.C * 2)
                nat.call("u3d_conv3d_small_cin_bwd", dev.index, _stream(dev), _p(src.t0), _p(rec.affine), _p(dz_),
                         _p(rec.conv_w.detach()), _p(gview(rec.idx_w)), _p(gst), Nn, Dd, Hh, Ww, src.C, Cout, _p(ws), ws.numel(),
                         flops=2 * 54.0 * src.C * Cout * Nn * Dd * Hh * Ww)
                coef = torch.empty((Nn, 3, src.C), dtype=_F32, device=dev)
                nat.call("u3d_gn_bwd_finalize", dev.index, _stream(dev), _p(gst), _p(rec.mean_rstd), _p(rec.gn_w.detach()), Nn,
                         src.C, rec.G, float(Dd * Hh * Ww), _p(gview(rec.idx_gw)), _p(gview(rec.idx_gb)), _p(coef))
                return None, coef
            s_aff = src.struct(rec.affine)
            flops = 54.0 * src.C * Cout * Nn * Dd * Hh * Ww
            nat.call("u3d_conv3d_wgrad", dev.index, _stream(dev), ctypes.byref(s_aff), _p(dz_), _p(gview(rec.idx_w)), Nn, Dd, Hh,
                     Ww, Cout, _p(ws), ws.numel(), flops=flops)
            wpd = self._packed(rec.conv_w, 1, dev)
            dg = torch.empty((Nn, Dd, Hh, Ww, src.C), dtype=_F32, device=dev)
            gst = pool.take(Nn * src.C * 2)
            s_dz = VSrc(dz_).struct()
            s_x = src.struct()
            nat.call("u3d_conv3d", dev.index, _stream(dev), ctypes.byref(s_dz), _p(wpd), _p(dg), Nn, Dd, Hh, Ww, src.C, 0, None,
                     ctypes.byref(s_x), _p(gst), flops=flops)
            if self.debug is not None:
                self.debug[rec.name + ".dg"] = dg.clone()
            coef = torch.empty((Nn, 3, src.C), dtype=_F32, device=dev)
            nat.call("u3d_gn_bwd_finalize", dev.index, _stream(dev), _p(gst), _p(rec.mean_rstd), _p(rec.gn_w.detach()), Nn, src.C,
                     rec.G, float(Dd * Hh * Ww), _p(gview(rec.idx_gw)), _p(gview(rec.idx_gb)), _p(coef))
            return dg, coef

        def plain_apply(dg, coef, x, relu_mask):
            out = torch.empty_like(x)
            Nn = x.shape[0]
            C = x.shape[-1]
            nat.call("u3d_gn_bwd_apply", dev.index, _stream(dev), _p(dg), C, 0, _p(x), C, _p(coef), C, x.numel() // (Nn * C), Nn,
                     relu_mask, _p(out))
            return out

        recs = tape.convs  # order: enc0.c1, enc0.c2, enc1.c1, ..., dec0.c1, dec0.c2, ...
        enc_recs = [(recs[2 * i], recs[2 * i + 1]) for i in range(n_levels)]
        dec_recs = [(recs[2 * n_levels + 2 * j], recs[2 * n_levels + 2 * j + 1]) for j in range(n_dec)]

        # ---- decoders, last to first
        for j in range(n_dec - 1, -1, -1):
            r1, r2 = dec_recs[j]
            dg2, coef2 = conv_bwd(r2, dz)
            dz1 = plain_apply(dg2, coef2, r2.src.t0, 1)  # r2.src.t0 is r1.y (post-ReLU)
            del dg2
            dg1, coef1 = conv_bwd(r1, dz1)
            src = r1.src
            C0, C1, Ct = src.C0, src.C1, src.C
            # skip half -> gradient of the encoder feature (mask applied later, merged with the pool path)
            lvl = n_levels - 2 - j
            sg = torch.empty_like(src.t0)
            nat.call("u3d_gn_bwd_apply", dev.index, _stream(dev), _p(dg1), Ct, 0, _p(src.t0), C0, _p(coef1), Ct,
                     src.D * src.H * src.W, src.N, 0, _p(sg))
            skip_grad[lvl] = sg
            # upsampled half -> low-res producer (previous decoder's conv2 or the deepest encoder), ReLU mask fused
            dzl = torch.empty_like(src.t1)
            lz, ly, lx = src.los
            nat.call("u3d_gn_bwd_apply_up", dev.index, _stream(dev), _p(dg1), Ct, C0, _p(src.t1), C1, _p(coef1), Ct, src.N, src.D,
                     src.H, src.W, src.D1, src.H1, src.W1, _p(lz), _p(ly), _p(lx), 1, _p(dzl))
            del dg1
            dz = dzl

        # decoder + head gradients are final: start their all-reduce now, overlapped with the encoder backward
        if self.grad_sync is not None:
            self.grad_sync.launch(flat[self.n_enc_params :])

        # ---- encoders, deepest to first
        dx0 = None
        for i in range(n_levels - 1, -1, -1):
            r1, r2 = enc_recs[i]
            dg2, coef2 = conv_bwd(r2, dz)
            dz1 = plain_apply(dg2, coef2, r2.src.t0, 1)
            del dg2
            dg1, coef1 = conv_bwd(r1, dz1, need_dg=(i > 0 or need_input_grad))
            if i > 0:
                pooled, argmax, e_in = tape.pools[i - 1]
                Ne, De, He, We, Ce = e_in.shape
                out = torch.empty_like(e_in)
                nat.call("u3d_maxpool2_bwd_merge", dev.index, _stream(dev), _p(dg1), _p(pooled), _p(argmax), _p(coef1),
                         _p(skip_grad.get(i - 1)), _p(e_in), Ne, De, He, We, Ce, 1, _p(out))
                dz = out
            elif need_input_grad:
                dx0 = plain_apply(dg1, coef1, tape.x0, 0)
            del dg1

        if self.grad_sync is not None:
            self.grad_sync.launch(flat[: self.n_enc_params])
            self.grad_sync.finish()

        dx = None
        if dx0 is not None:
            if Cin == 1:
                dx = dx0.view(N, 1, D, H, W)
            else:
                dx = torch.empty((N, Cin, D, H, W), dtype=_F32, device=dev)
                nat.call("u3d_ndhwc_to_ncdhw", dev.index, _stream(dev), _p(dx0), _p(dx), N, Cin, V)
        return flat, dx


class _UNet3DFunction(torch.autograd.Function):
    """The whole encoder-decoder as one autograd node (forward = engine.forward, backward = engine.backward)."""

    @staticmethod
    def forward(ctx, engine: UNet3DEngine, x: torch.Tensor, *params):
        # grad mode is always off inside Function.forward: needs_input_grad tells whether a backward can follow
        save = any(ctx.needs_input_grad)
        logits, probs, tape = engine.forward(x, save)
        ctx.engine = engine
        ctx.tape = tape
        ctx.has_probs = probs is not None
        ctx.x_requires_grad = x.requires_grad
        if probs is not None:
            ctx.save_for_backward(probs)
            return logits, probs
        return (logits,)

    @staticmethod
    def backward(ctx, *grads):
        engine, tape = ctx.engine, ctx.tape
        if tape is None:
            raise RuntimeError("u3d: backward called but forward ran without grad mode")
        dlogits = grads[0]
        if ctx.has_probs and len(grads) > 1 and grads[1] is not None:
            # gradient flowing through the probabilities (rare: the reference's trainer takes the loss on logits,
            # trainer.py:362-365): fold it into dlogits.  Tiny (N,Cout,D,H,W) tensors.
            (probs,) = ctx.saved_tensors
            gp = grads[1]
            if isinstance(engine.model.final_activation, torch.nn.Sigmoid):
                extra = gp * probs * (1 - probs)
            else:
                extra = probs * (gp - (gp * probs).sum(dim=1, keepdim=True))
            dlogits = extra if dlogits is None else dlogits + extra
        if dlogits is None:
            dlogits = torch.zeros_like(ctx.saved_tensors[0]) if ctx.has_probs else None
        flat, dx = engine.backward(tape, dlogits, ctx.x_requires_grad)
        ctx.tape = None
        out = [None, dx]
        for p, off in zip(engine.params, engine.poffs):
            out.append(flat[off : off + p.numel()].view(p.shape) if p.requires_grad else None)
        return tuple(out)


def run_model(engine: UNet3DEngine, x: torch.Tensor):
    """(probs_or_logits, logits) exactly like AbstractUNet._forward_logits (model.py:123-149)."""
    outs = _UNet3DFunction.apply(engine, x, *engine.params)
    if len(outs) == 2:
        logits, probs = outs
        return probs, logits
    return outs[0], outs[0]
