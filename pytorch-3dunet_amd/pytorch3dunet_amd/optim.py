"""The optimizer step of the training loop (reference trainer.py:246; create_optimizer, utils.py:246-316 builds torch.optim.Adam) as ONE
kernel launch over all parameters (csrc/u3d_optim.hip, u3d_adam_step): torch's multi-tensor Adam is 8 launches / 0.17 ms per step on
UNet3D f_maps=32's 44 parameters, 1 % of the 17 ms step, for 16 MB of parameters.

`FusedAdam` takes torch.optim.Adam's constructor arguments, keeps torch.optim.Adam's state layout (`step`, `exp_avg`, `exp_avg_sq` per
parameter: state_dicts are interchangeable, the reference's checkpoints load) and its update formula; parameters the kernel cannot take
(not fp32, not on a HIP device, sparse gradients) fall back to torch's own functional Adam.  `as_fused(optimizer)` converts an existing
torch.optim.Adam instance (e.g. the one the reference's create_optimizer built) in place of it."""
from __future__ import annotations

import ctypes

import torch

from . import _native as nat


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, maximize=False):
        if amsgrad or maximize:
            raise ValueError("u3d FusedAdam: amsgrad / maximize are not implemented (use torch.optim.Adam)")
        if not 0.0 <= lr or not 0.0 <= eps or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or not 0.0 <= weight_decay:
            raise ValueError("u3d FusedAdam: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False))
        self._tables: dict = {}

    def _init_state(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.tensor(0.0, dtype=torch.float32)  # (torch.optim.Adam: a CPU scalar tensor unless capturable / fused)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            native, other = {}, []
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self._init_state(p)
                g = p.grad
                ok = (p.is_cuda and p.dtype == torch.float32 and g.dtype == torch.float32 and not g.is_sparse and p.is_contiguous()
                      and g.is_contiguous() and g.device == p.device and st["exp_avg"].is_contiguous() and st["exp_avg_sq"].is_contiguous())
                if ok:
                    # one launch per (device, step count): parameters of a group normally share both
                    native.setdefault((p.device.index, int(st["step"].item()) if st["step"].numel() == 1 else 0), []).append((p, st))
                else:
                    other.append((p, st))
            b1, b2 = group["betas"]
            for (dev_index, step0), items in native.items():
                self._launch(gi, dev_index, step0 + 1, items, group, b1, b2)
            for p, st in other:  # torch's own update for what the kernel does not take (it advances `step` itself)
                torch.optim._functional.adam([p], [p.grad], [st["exp_avg"]], [st["exp_avg_sq"]], [], [st["step"]], amsgrad=False,
                                             beta1=b1, beta2=b2, lr=group["lr"], weight_decay=group["weight_decay"], eps=group["eps"],
                                             maximize=False)
        return loss

    def _launch(self, gi, dev_index, step, items, group, b1, b2):
        key = (gi, dev_index) + tuple((p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()) for p, st in items)
        ent = self._tables.get((gi, dev_index))
        if ent is None or ent[0] != key:
            # (gradients are views into the executor's flat buffer: the caching allocator hands out the same block step after step, so
            # the table is rebuilt once per change of the set of pointers, not per step)
            descs = (nat.U3DAdamDesc * len(items))()
            first = 0
            for i, (p, st) in enumerate(items):
                descs[i].p, descs[i].g = p.data_ptr(), p.grad.data_ptr()
                descs[i].m, descs[i].v = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                descs[i].first, descs[i].numel = first, p.numel()
                first += (p.numel() + 3) // 4 * 4
            dev = items[0][0].device
            table = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(dev)
            ent = (key, table, first)
            self._tables[(gi, dev_index)] = ent
        _, table, total = ent
        dev = items[0][0].device
        nat.call("u3d_adam_step", dev.index, torch.cuda.current_stream(dev).cuda_stream, table.data_ptr(), len(items), total,
                 float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]), int(step))
        for _, st in items:
            st["step"] += 1


def as_fused(optimizer: torch.optim.Optimizer) -> torch.optim.Optimizer:
    """A FusedAdam over the same parameter groups, hyper-parameters and state as `optimizer` when that is a plain torch.optim.Adam
    (amsgrad / maximize / capturable / differentiable off); anything else is returned unchanged."""
    if type(optimizer) is not torch.optim.Adam:
        return optimizer
    for g in optimizer.param_groups:
        if g.get("amsgrad") or g.get("maximize") or g.get("capturable") or g.get("differentiable"):
            return optimizer
    g0 = optimizer.param_groups[0]
    fused = FusedAdam([{k: v for k, v in g.items() if k in ("params", "lr", "betas", "eps", "weight_decay")} for g in optimizer.param_groups],
                      lr=g0["lr"], betas=g0["betas"], eps=g0["eps"], weight_decay=g0["weight_decay"])
    for p, st in optimizer.state.items():
        fused.state[p] = {"step": torch.as_tensor(float(st["step"])), "exp_avg": st["exp_avg"], "exp_avg_sq": st["exp_avg_sq"]}
    return fused
