"""The fused loss family of the training path.  `install_fused()` patches it into the reference's own
`pytorch3dunet.unet3d.losses` module, so the trainer's `get_loss_criterion` (losses.py:273-350) keeps its own option
handling, wrappers and every other loss — none of that host code is restated here.

Natively fused on an MI355X (csrc/u3d_loss.hip, C-ABI `u3d_bce_dice_fwd/_bwd`): `BCEDiceLoss` (losses.py:187-201),
`DiceLoss` with sigmoid normalisation (losses.py:119-127 on top of :84-116) and `nn.BCEWithLogitsLoss` without
`pos_weight`.  They are one family: loss = w_bce * mean(BCE-with-logits) + w_dice * (1 - mean_c dice_c).  The stock
path is ~15 ATen kernels, a permute+contiguous copy (`flatten`, losses.py:253-271) and six full-size autograd
temporaries; the fused path is two reads of (logits, target) and one write of dlogits, and the upstream scalar gradient
is consumed on the device (no host synchronisation on the step's critical path).

CPU tensors, other dtypes, softmax / no normalisation run the same formulas on torch operators (what the reference
does), so `device: cpu` configs behave identically.
"""
import ctypes
import functools
import importlib
import sys

import torch
from torch import nn
from torch.nn import functional as F


def flatten(tensor):
    """(N, C, *spatial) -> (C, N * prod(spatial)), channel axis first (losses.py:253-271)."""
    c = tensor.size(1)
    return tensor.transpose(0, 1).reshape(c, -1)


def compute_per_channel_dice(input, target, epsilon=1e-6, weight=None):
    """Per-channel Dice coefficient of already-normalised probabilities (losses.py:11-37; V-Net form with squared
    terms in the denominator)."""
    assert input.size() == target.size(), "'input' and 'target' must have the same shape"
    p = flatten(input)
    t = flatten(target).float()
    intersect = (p * t).sum(-1)
    if weight is not None:
        intersect = weight * intersect
    denominator = (p * p).sum(-1) + (t * t).sum(-1)
    return 2 * (intersect / denominator.clamp(min=epsilon))


def _native_ok(input, target):
    return (input.is_cuda and input.dtype == torch.float32 and target.dtype == torch.float32
            and input.shape == target.shape and input.dim() >= 3 and input.numel() > 0
            and input.shape[0] * input.shape[1] < 65536)


class _FusedBCEDice(torch.autograd.Function):
    """loss = w_bce * BCEWithLogits(mean) + w_dice * (1 - mean_c dice_c) through u3d_bce_dice_fwd/_bwd."""

    @staticmethod
    def forward(ctx, logits, target, weight, w_bce, w_dice, eps):
        from .. import _native as nat

        logits = logits.contiguous()
        target = target.contiguous()
        dev = logits.device
        n, c = logits.shape[0], logits.shape[1]
        v = logits.numel() // (n * c)
        sums = torch.empty(nat.get_lib().u3d_bce_dice_scratch_doubles(n, c, v), dtype=torch.float64, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        coef = torch.empty(2 * c + 1, dtype=torch.float32, device=dev)
        wt = None
        if weight is not None:
            wt = weight.to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
            assert wt.numel() == c, "DiceLoss weight must have one entry per channel"
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        nat.call("u3d_bce_dice_fwd", dev.index, stream, ctypes.c_void_p(logits.data_ptr()),
                 ctypes.c_void_p(target.data_ptr()), None if wt is None else ctypes.c_void_p(wt.data_ptr()), n, c, v,
                 float(w_bce), float(w_dice), float(eps), ctypes.c_void_p(sums.data_ptr()), ctypes.c_void_p(loss.data_ptr()),
                 ctypes.c_void_p(coef.data_ptr()))
        ctx.save_for_backward(logits, target, coef)
        ctx.dims = (n, c, v)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, grad_out):
        from .. import _native as nat

        logits, target, coef = ctx.saved_tensors
        n, c, v = ctx.dims
        dev = logits.device
        g = grad_out.to(dtype=torch.float32).reshape(1).contiguous()
        dlogits = torch.empty_like(logits)
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        nat.call("u3d_bce_dice_bwd", dev.index, stream, ctypes.c_void_p(logits.data_ptr()), ctypes.c_void_p(target.data_ptr()),
                 ctypes.c_void_p(coef.data_ptr()), ctypes.c_void_p(g.data_ptr()), n, c, v, ctypes.c_void_p(dlogits.data_ptr()))
        return dlogits, None, None, None, None, None


def fused_bce_dice(logits, target, w_bce=1.0, w_dice=1.0, weight=None, eps=1e-6):
    """Functional form: native on HIP fp32 tensors, torch operators otherwise (identical formulas)."""
    if _native_ok(logits, target):
        return _FusedBCEDice.apply(logits, target, weight, w_bce, w_dice, eps)
    out = 0.0
    if w_bce != 0:
        out = out + w_bce * F.binary_cross_entropy_with_logits(logits, target)
    if w_dice != 0:
        dice = compute_per_channel_dice(torch.sigmoid(logits), target, epsilon=eps, weight=weight)
        out = out + w_dice * (1.0 - torch.mean(dice))
    return out


class _AbstractDiceLoss(nn.Module):
    """Normalise the logits, take the per-channel Dice, return 1 - mean (losses.py:84-116)."""

    def __init__(self, weight=None, normalization="sigmoid"):
        super().__init__()
        self.register_buffer("weight", weight)
        assert normalization in ["sigmoid", "softmax", "none"]
        self.normalization_name = normalization
        if normalization == "sigmoid":
            self.normalization = nn.Sigmoid()
        elif normalization == "softmax":
            self.normalization = nn.Softmax(dim=1)
        else:
            self.normalization = lambda x: x

    def dice(self, input, target, weight):
        raise NotImplementedError

    def forward(self, input, target):
        per_channel_dice = self.dice(self.normalization(input), target, weight=self.weight)
        return 1.0 - torch.mean(per_channel_dice)


class DiceLoss(_AbstractDiceLoss):
    """Dice loss on logits (losses.py:119-127).  Sigmoid normalisation on an MI355X runs the fused kernels."""

    def __init__(self, weight=None, normalization="sigmoid"):
        super().__init__(weight, normalization)

    def dice(self, input, target, weight):
        return compute_per_channel_dice(input, target, weight=self.weight)

    def forward(self, input, target):
        if self.normalization_name == "sigmoid" and _native_ok(input, target):
            return fused_bce_dice(input, target, 0.0, 1.0, self.weight)
        return super().forward(input, target)


class BCEDiceLoss(nn.Module):
    """BCEWithLogitsLoss + alpha * DiceLoss (losses.py:187-201) — the loss of BASELINE config 2."""

    def __init__(self, alpha=1.0):
        super().__init__()
        self.alpha = alpha
        self.bce = nn.BCEWithLogitsLoss()
        self.dice = DiceLoss()

    def forward(self, input, target):
        if _native_ok(input, target):
            return fused_bce_dice(input, target, 1.0, self.alpha)
        return self.bce(input, target) + self.alpha * self.dice(input, target)


class BCEWithLogitsLoss(nn.BCEWithLogitsLoss):
    """nn.BCEWithLogitsLoss (losses.py:297-298); the plain mean-reduced, unweighted form is fused on an MI355X."""

    def forward(self, input, target):
        if self.weight is None and self.pos_weight is None and self.reduction == "mean" and _native_ok(input, target):
            return fused_bce_dice(input, target, 1.0, 0.0)
        return super().forward(input, target)


# ---------------------------------------------------------------------------------------------------------------
# Everything else of the reference's losses.py (option handling of `get_loss_criterion`, the masking / skip-last-channel
# wrappers, GeneralizedDice, weighted cross entropy / SmoothL1; losses.py:40-82,148-184,204-345) is host code that this
# repository does NOT restate: the three fused classes above are patched INTO the caller's own
# `pytorch3dunet.unet3d.losses` module, whose factory and wrappers keep running unchanged.
_FUSED = ("BCEDiceLoss", "DiceLoss")


def _upgrade_bce(module):
    """`_create_loss` builds `nn.BCEWithLogitsLoss(pos_weight=...)` from torch.nn directly (losses.py:312-313): give such
    instances the fused forward by switching their class to the subclass above (same state, no extra attributes)."""
    for m in module.modules():
        if type(m) is nn.BCEWithLogitsLoss:
            m.__class__ = BCEWithLogitsLoss
    return module


def install_fused(ref_losses):
    """Patch the fused loss family into the caller's `pytorch3dunet.unet3d.losses` module (idempotent): its own `_create_loss`
    (losses.py:310-345) looks `BCEDiceLoss` / `DiceLoss` up in its module globals at call time, and its `get_loss_criterion`
    is wrapped once so that plain `nn.BCEWithLogitsLoss` instances come back with the fused forward.  Must run before
    `pytorch3dunet.unet3d.trainer` is imported (trainer.py:16 binds `get_loss_criterion` by name)."""
    if ref_losses is sys.modules[__name__]:
        raise RuntimeError("install_fused() takes the REFERENCE's pytorch3dunet.unet3d.losses module, not this one")
    if getattr(ref_losses, "_u3d_fused", False):
        return ref_losses
    for name in _FUSED:
        setattr(ref_losses, name, globals()[name])
    inner = ref_losses.get_loss_criterion

    @functools.wraps(inner)
    def get_loss_criterion(config):
        return _upgrade_bce(inner(config))

    ref_losses.get_loss_criterion = get_loss_criterion
    ref_losses._u3d_fused = True
    return ref_losses


def get_loss_criterion(config):
    """The loss named by config['loss'], resolved by the CALLER's own `pytorch3dunet.unet3d.losses.get_loss_criterion`
    (losses.py:273-307: `ignore_index`, `skip_last_target`, `weight`, `pos_weight`, every loss name) with the fused family
    patched in.  The reference package must be importable — this library replaces one path of it, not the package."""
    try:
        ref = importlib.import_module("pytorch3dunet.unet3d.losses")
    except ImportError as e:
        raise ImportError("pytorch3dunet_amd.unet3d.losses.get_loss_criterion delegates to the reference's own "
                          "pytorch3dunet.unet3d.losses (wolny/pytorch-3dunet), which is not importable; construct "
                          "BCEDiceLoss / DiceLoss / BCEWithLogitsLoss of this module directly instead") from e
    if ref is sys.modules[__name__]:
        raise RuntimeError("pytorch3dunet.unet3d.losses is aliased to pytorch3dunet_amd.unet3d.losses (the pre-round-4 seam); "
                           "use pytorch3dunet_amd.launch.install_seam() / losses.install_fused(reference_module) instead")
    return install_fused(ref).get_loss_criterion(config)
