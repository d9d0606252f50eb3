"""Static-shape step runner: the launch sequences of one training step captured in hipGraphs (hip_graph: true / U3D_GRAPH=1)."""
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
        # the warm-up step and the capture run WITHOUT the data-parallel hook: a capture is a per-rank event (a new input
        # shape, an LRU eviction, a re-attach on ONE rank), so nothing in it may issue a collective — the warm-up's gradients are
        # garbage anyway.  Only `backward()` below, which every rank runs once per step, talks to RCCL.
        self.sync = engine.grad_sync
        engine.grad_sync = None
        try:
            with torch.cuda.stream(side):
                # one eager step first: every lazily built constant (index maps, identity tables, the pack descriptor table, the
                # library's function attributes) must exist before capture — host-to-device copies are illegal inside it
                self.static_x.copy_(x)
                engine.begin_forward(True)
                logits, probs, tape = engine.forward(self.static_x, True)
                engine.backward(tape, torch.zeros_like(logits), need_dx)
                del logits, probs, tape
        finally:
            engine.grad_sync = self.sync
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
        ctx.set_materialize_grads(False)  # (as in _UNet3DFunction: an unused output's gradient arrives as None)
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
