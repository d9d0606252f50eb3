"""Input hand-off: get the next batch into HBM while the current one computes.

The reference builds its DataLoaders without `pin_memory` (datasets/utils.py:408-422) and moves every batch with
`.to(device, non_blocking=True)` inside the step (`_split_and_move_to_device`, unet3d/trainer.py:81-90): from pageable
memory that copy is synchronous, so the H2D time of every patch is serialised with the compute of the step.

`DevicePrefetcher` wraps any iterable of (input, target) batches — the trainer only needs `for t in loader` and
`len(loader)` (trainer.py:231-237, :319-326) — and stays one batch ahead: batch i+1 is staged through a reusable pinned
buffer and copied on a dedicated HIP stream while batch i is being consumed; the consumer stream waits on the copy's
event only (no host synchronisation).  Nested tuples / lists are preserved exactly as `_split_and_move_to_device` does.
"""
from __future__ import annotations

from typing import Any, Iterable, Iterator

import torch


class DevicePrefetcher:
    """`for input, target in DevicePrefetcher(loader, "cuda")` — yields what the loader yields, already on `device`."""

    def __init__(self, loader: Iterable, device, depth: int = 1):
        self.loader = loader
        self.device = torch.device(device)
        self.depth = max(1, int(depth))
        self._stream = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None
        self._pinned: dict = {}  # (slot, position, shape, dtype) -> reusable pinned staging buffer

    def __len__(self):
        return len(self.loader)

    # -- one batch -> device (asynchronously on the copy stream) ------------------------------------------------------
    def _stage(self, x: Any, slot: int, pos: list):
        if isinstance(x, (tuple, list)):
            return tuple(self._stage(v, slot, pos) for v in x)
        if not torch.is_tensor(x):
            return x
        if self._stream is None:
            return x.to(self.device)
        pos[0] += 1
        if x.device.type == "cpu" and not x.is_pinned():
            key = (slot, pos[0], tuple(x.shape), x.dtype)
            buf = self._pinned.get(key)
            if buf is None:
                buf = torch.empty(x.shape, dtype=x.dtype, pin_memory=True)
                self._pinned[key] = buf
            buf.copy_(x)  # host memcpy into pinned memory; the H2D below is then truly asynchronous
            x = buf
        return x.to(self.device, non_blocking=True)

    @staticmethod
    def _record(x: Any, stream):
        if isinstance(x, (tuple, list)):
            for v in x:
                DevicePrefetcher._record(v, stream)
        elif torch.is_tensor(x) and x.is_cuda:
            x.record_stream(stream)  # the caching allocator must not recycle it while the consumer still reads it

    def __iter__(self) -> Iterator:
        it = iter(self.loader)
        if self._stream is None:
            for batch in it:
                yield self._stage(batch, 0, [0])
            return
        queue = []  # (device batch, copy-done event, slot)
        slot = 0
        nslots = self.depth + 1

        slot_events: dict = {}

        def push():
            nonlocal slot
            try:
                batch = next(it)
            except StopIteration:
                return False
            sl = slot % nslots
            prev = slot_events.get(sl)
            if prev is not None:
                prev.synchronize()  # the H2D that last read this slot's pinned buffers has finished (long ago, normally)
            with torch.cuda.stream(self._stream):
                dev_batch = self._stage(batch, sl, [0])
                ev = torch.cuda.Event()
                ev.record(self._stream)
            slot_events[sl] = ev
            queue.append((dev_batch, ev))
            slot += 1
            return True

        for _ in range(self.depth):
            if not push():
                break
        while queue:
            dev_batch, ev = queue.pop(0)
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            self._record(dev_batch, cur)
            push()
            yield dev_batch
