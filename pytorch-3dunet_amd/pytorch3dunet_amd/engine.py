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

from ._engine_base import *  # noqa: F401,F403
from ._engine_base import __all__ as _base_all
from ._engine_graph import GraphStep, _CaptureSplit, _GraphedUNet3DFunction, _graph_blocker, graph_step_for  # noqa: F401
from ._engine_res import ResUNetEngine  # noqa: F401
from ._engine_unet import UNet3DEngine  # noqa: F401




class _UNet3DFunction(torch.autograd.Function):
    """The whole encoder-decoder as one autograd node (forward = engine.forward, backward = engine.backward)."""

    @staticmethod
    def forward(ctx, engine: UNet3DEngine, grad_mode: bool, x: torch.Tensor, *params):
        # Grad mode is always off inside Function.forward, and `ctx.needs_input_grad` reports the inputs' requires_grad flags even
        # when the CALLER runs under torch.no_grad() (round 4: inference forwards therefore kept a tape, advanced the repack salt and
        # repacked every weight image each time — 6 ms per volume of BASELINE config 5).  The caller's grad mode is passed in.
        save = grad_mode and any(ctx.needs_input_grad)
        # an output the loss never touches (the reference's trainer takes the loss on the logits only, trainer.py:362-365) must reach
        # backward as None, not as a zero tensor: materialised, the unused `probs` cost a fill + four elementwise launches per step
        ctx.set_materialize_grads(False)
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
