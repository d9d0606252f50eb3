"""Data parallelism for the native path: one process per GPU, gradients all-reduced with RCCL over xGMI.

The reference's only parallelism is single-process nn.DataParallel (trainer.py:202-205, predict.py:63-66): params
broadcast + grads reduced to device 0 through Python threads every step.  Patches are independent samples and
GroupNorm statistics are per-sample (buildingblocks.py:75), so the path shards as pure data parallelism with ONE
exchange step per iteration: the gradient all-reduce.

MI355X design: the fused backward (engine.py) writes every parameter gradient into one flat fp32 buffer laid out
[encoders | decoders | head].  The decoder+head half (≈9.3 MB for UNet3D f_maps=32) is final before the encoder
backward starts, so its all-reduce is launched right there and runs on RCCL's stream WHILE the encoder backward
(≈1/3 of the step) computes; the encoder half (≈7 MB) follows at the end.  Two large buckets, not many small ones:
xGMI ring all-reduce is latency- not bandwidth-bound at these sizes (2·(7/8)·16.3 MB ≈ 0.19 ms at one 153 GB/s
link).
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as dist


class GradSync:
    """Asynchronous bucketed gradient averaging for `UNet3DEngine` (attach with `attach`)."""

    def __init__(self, process_group: Optional[dist.ProcessGroup] = None, force_single: Optional[bool] = None):
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        # a 1-rank group normally skips the collective; U3D_SYNC_SINGLE=1 (or force_single=True) issues it anyway so that
        # the engine -> RCCL hook placement can be executed and checked on a single-GPU box (tests/test_gpu_parallel.py)
        self.force_single = os.environ.get("U3D_SYNC_SINGLE", "0") == "1" if force_single is None else bool(force_single)
        self.launched = 0  # collectives issued (tests)
        self._pending: List = []
        backend = dist.get_backend(process_group)
        self._avg_op = dist.ReduceOp.AVG if backend == "nccl" else None  # gloo has no AVG

    def launch(self, bucket: torch.Tensor) -> None:
        """Start averaging `bucket` (a contiguous slice of the flat gradient buffer) across ranks."""
        if (self.world == 1 and not self.force_single) or bucket.numel() == 0:
            return
        self.launched += 1
        if self._avg_op is not None:
            work = dist.all_reduce(bucket, op=self._avg_op, group=self.group, async_op=True)
        else:
            work = dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._pending.append((work, bucket))

    def finish(self) -> None:
        """Make the compute stream wait for every launched bucket (no host sync with NCCL/RCCL)."""
        for work, bucket in self._pending:
            work.wait()
            if self._avg_op is None:
                bucket.mul_(1.0 / self.world)
        self._pending.clear()


class TreeSync(GradSync):
    """The same two-bucket exchange for a model that runs as the torch.nn MODULE TREE — CPU tensors (`device: cpu`, gloo) or a
    variant the native executor does not cover (2-D models, …): post-accumulate-grad hooks count the parameters of each bucket;
    when the last gradient of [decoders | head] has been accumulated its flat copy is all-reduced asynchronously while autograd
    is still inside the encoders, the encoder bucket follows, and the averaged values are written back into `.grad` before
    `loss.backward()` returns (autograd's end-of-pass callback).  (The native executor does not go through these hooks: it hands RCCL slices of its own flat
    gradient buffer, engine.py.)  Every parameter that requires grad must take part in the loss, like under torch DDP
    without `find_unused_parameters`."""

    def __init__(self, model: torch.nn.Module, process_group=None, force_single: Optional[bool] = None):
        super().__init__(process_group, force_single)
        params = [p for p in model.parameters() if p.requires_grad]
        enc = getattr(model, "encoders", None)
        enc_ids = {id(p) for p in enc.parameters()} if enc is not None else set()
        # bucket 0 = decoders + head (ready first in backward), bucket 1 = encoders
        self.buckets = [[p for p in params if id(p) not in enc_ids], [p for p in params if id(p) in enc_ids]]
        self._left = [len(b) for b in self.buckets]
        self._flat: List[Optional[torch.Tensor]] = [None, None]
        self._in_backward = False
        self._handles = []
        for bi, bucket in enumerate(self.buckets):
            for p in bucket:
                self._handles.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))
        # pass boundary: autograd skips its end-of-pass callbacks when `loss.backward()` RAISES after the first hook fired
        # (OOM, an anomaly check, a hook of the caller) — `_in_backward` would stay set, and the next pass would neither reset its
        # counters nor queue the callback.  The next forward of the model re-arms an aborted pass.
        self._handles.append(model.register_forward_pre_hook(self._on_forward))

    def _make_hook(self, bi: int):
        def hook(_param):
            self._ready(bi)

        return hook

    def _on_forward(self, _module, _args) -> None:
        if not self._in_backward:
            return
        # A forward that runs INSIDE an autograd pass is not a new step: torch.utils.checkpoint around the model, or a forward issued
        # from a backward hook, re-enter the model while this very backward is still exchanging — leave its state alone.
        try:
            in_graph_task = torch._C._current_graph_task_id() != -1
        except AttributeError:  # (very old torch: no way to tell; keep the re-arm, the common case)
            in_graph_task = False
        if in_graph_task:
            return
        # The previous backward never reached _end_of_backward (it raised after the first hook fired): drop its half-sent state.  The
        # step was invalid on THIS rank only — the other ranks may have launched more buckets than this one did, so after such an abort
        # the ranks' collective sequences are only guaranteed to pair up again if every rank aborts the step (or the group is
        # re-initialised); a mismatch shows as a hang / size error of the next collective, never as silently wrong gradients.
        self._in_backward = False
        for work, _bucket in self._pending:
            try:
                work.wait()
            except Exception as e:  # noqa: BLE001  (the collective of an aborted step may itself have failed: say so, then go on)
                import warnings

                warnings.warn(f"u3d TreeSync: a gradient all-reduce of the aborted backward pass failed while being drained: {e!r}")
        self._pending.clear()
        self._left = [len(b) for b in self.buckets]
        self._flat = [None, None]

    def _ready(self, bi: int) -> None:
        if self.world == 1 and not self.force_single:
            return
        if not self._in_backward:
            # first gradient of this backward pass: start from full counters, and have autograd call back when the pass is over
            self._in_backward = True
            self._left = [len(b) for b in self.buckets]
            self._flat = [None, None]
            torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)
        self._left[bi] -= 1
        if self._left[bi] == 0:
            flat = torch.cat([p.grad.reshape(-1) for p in self.buckets[bi]])
            self._flat[bi] = flat
            self.launch(flat)

    def _end_of_backward(self) -> None:
        """autograd's end-of-pass callback: every bucket must be complete — a parameter without a gradient in this pass (an unused
        branch, a loss over a subset of the outputs) would otherwise leave its bucket unsent HERE and fire it in the middle of the
        NEXT backward on partial gradients, pairing different buckets across ranks.  Like torch DDP without
        `find_unused_parameters`, that is an error, raised on the pass that caused it."""
        self._in_backward = False
        left, self._left = self._left, [len(b) for b in self.buckets]
        if any(n != 0 for n in left):
            self._pending.clear()  # (a launched bucket's result is dropped: the step is invalid on this rank)
            self._flat = [None, None]
            missing = [f"{n} of {len(b)} parameters of {name}" for n, b, name in zip(left, self.buckets, ("[decoders | head]", "[encoders]")) if n]
            raise RuntimeError("u3d TreeSync: this backward pass produced no gradient for " + " and ".join(missing) +
                               " — every parameter that requires grad must take part in the loss (the gradient exchange is "
                               "bucketed; torch DDP raises for unused parameters in the same situation)")
        self.finish()
        with torch.no_grad():
            for bucket, flat in zip(self.buckets, self._flat):
                o = 0
                for p in bucket:
                    n = p.grad.numel()
                    p.grad.copy_(flat[o:o + n].view_as(p.grad))
                    o += n
        self._flat = [None, None]

    def detach(self) -> None:
        for h in self._handles:
            h.remove()
        self._handles = []


def broadcast_parameters(model: torch.nn.Module, src: int = 0, process_group=None) -> None:
    """Identical initial weights on every rank (what DataParallel's replicate() does each step, once)."""
    with torch.no_grad():
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t, src=src, group=process_group)


def attach(model: torch.nn.Module, process_group=None, broadcast: bool = True, force_single: Optional[bool] = None) -> GradSync:
    """Enable data-parallel training: after this, `loss.backward()` leaves rank-averaged gradients in `.grad` exactly like torch
    DDP would, with the exchange overlapped as described.  A natively covered model on a HIP device gets the executor hook
    (flat buffer slices handed to RCCL from inside the fused backward); anything that runs as the torch.nn module tree — CPU
    tensors / gloo, uncovered variants — gets `TreeSync`'s post-accumulate-grad hooks.  Nothing is ever left unsynchronised
    silently: a natively attached model that later falls back to the module tree raises (unet3d/model.py)."""
    if not dist.is_initialized():
        raise RuntimeError("torch.distributed is not initialised")
    if broadcast:
        broadcast_parameters(model, 0, process_group)
    first = next(iter(model.parameters()), None)
    native = bool(getattr(model, "native_supported", False)) and first is not None and first.is_cuda
    old = getattr(model, "_u3d_grad_sync", None)
    if isinstance(old, TreeSync):
        old.detach()
    if native:
        sync = GradSync(process_group, force_single)
        model._get_engine().grad_sync = sync
    else:
        sync = TreeSync(model, process_group, force_single)
    object.__setattr__(model, "_u3d_grad_sync", sync)
    return sync


def cu_budget(slots: Optional[int] = None) -> int:
    """Leave `slots` block slots of the persistent convolution grids free (of 2 per CU; `u3d_set_tuning` key 12) so that RCCL's
    all-reduce kernels find CUs with room beside them: the persistent grids otherwise own every CU's LDS and most of its registers,
    and a kernel of another stream makes no progress until they end (profiles/r03_overlap_probe.txt: 3 % of the exchange hidden).
    Each slot costs 1/512 of the convolution throughput.  Key 12 acts on the fp32 persistent 3x3x3 convolution only — the bf16 kernels
    (compute_dtype: bf16) and all other launches are not persistent-grid kernels and ignore it.  `slots=None` reads U3D_RCCL_SLOTS (default 0: right for small models, whose
    whole exchange is ~1 % of a step).  Process-wide; results never depend on it.  (A CU-MASKED compute queue — the textbook way to
    reserve whole CUs — was measured and rejected: the same kernels run 40-75 % slower on a queue masked to 248 of 256 CUs,
    profiles/r04_cu_mask_layer_bench_reserve8.txt.)"""
    from . import _native as nat

    if slots is None:
        slots = int(os.environ.get("U3D_RCCL_SLOTS", "0"))
    nat.call("u3d_set_tuning", 12, int(slots))
    return int(slots)


def shard_batch(global_batch: int, rank: int, world: int) -> range:
    """Sample indices of this rank (contiguous shards; the reference scales the batch by the device count the same
    way for DataParallel, datasets/utils.py:399-403)."""
    per = global_batch // world
    if per * world != global_batch:
        raise ValueError(f"global batch {global_batch} not divisible by world size {world}")
    return range(rank * per, (rank + 1) * per)
