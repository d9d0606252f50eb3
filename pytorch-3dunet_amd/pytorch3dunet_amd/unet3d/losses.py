"""Loss functions of the training path, drop-in for the names the reference's trainer resolves through
`pytorch3dunet.unet3d.losses.get_loss_criterion` (losses.py:273-350).

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

import torch
from torch import nn
from torch.nn import L1Loss, MSELoss, SmoothL1Loss  # noqa: F401  (names the reference's module exposes, losses.py:4)
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
        sums = torch.empty(1 + 3 * c, dtype=torch.float64, device=dev)
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


class MaskingLossWrapper(nn.Module):
    """Zero the loss gradient where target == ignore_index (losses.py:40-64)."""

    def __init__(self, loss, ignore_index):
        super().__init__()
        assert ignore_index is not None, "ignore_index cannot be None"
        self.loss = loss
        self.ignore_index = ignore_index

    def forward(self, input, target):
        mask = (target != self.ignore_index).to(target.dtype)
        return self.loss(input * mask, target * mask)


class SkipLastTargetChannelWrapper(nn.Module):
    """Drop the last target channel before the loss (losses.py:67-94)."""

    def __init__(self, loss, squeeze_channel=False):
        super().__init__()
        self.loss = loss
        self.squeeze_channel = squeeze_channel

    def forward(self, input, target):
        assert target.size(1) > 1, "Target tensor has a singleton channel dimension, cannot remove channel"
        target = target[:, :-1, ...]
        if self.squeeze_channel:
            target = torch.squeeze(target, dim=1)
        return self.loss(input, target)


class GeneralizedDiceLoss(_AbstractDiceLoss):
    """Generalized Dice (arXiv:1707.03237), reference losses.py:148-184: every label's overlap and volume are weighted
    by 1 / (label volume)^2; a single-channel input is extended with its complement so that there are two labels.
    Plain torch reductions over (C, N*D*H*W) — outside the fused kernels (not on the measured path)."""

    def __init__(self, normalization="sigmoid", epsilon=1e-6):
        super().__init__(weight=None, normalization=normalization)
        self.epsilon = epsilon

    def dice(self, input, target, weight):
        assert input.size() == target.size(), "'input' and 'target' must have the same shape"
        p, t = flatten(input), flatten(target).float()
        if p.size(0) == 1:
            p, t = torch.cat((p, 1 - p), dim=0), torch.cat((t, 1 - t), dim=0)
        vol = t.sum(-1)
        w = (1 / (vol * vol).clamp(min=self.epsilon)).detach()
        overlap = ((p * t).sum(-1) * w).sum()
        total = ((p + t).sum(-1) * w).clamp(min=self.epsilon).sum()
        return 2 * overlap / total


class WeightedCrossEntropyLoss(nn.Module):
    """Cross entropy with per-class weights (1 - mean softmax mass) / (mean softmax mass) recomputed from every
    prediction, reference losses.py:204-227."""

    def __init__(self, ignore_index=-1):
        super().__init__()
        self.ignore_index = ignore_index

    @staticmethod
    def _class_weights(input):
        mass = flatten(F.softmax(input, dim=1))
        return ((1.0 - mass).sum(-1) / mass.sum(-1)).detach()

    def forward(self, input, target):
        return F.cross_entropy(input, target, weight=self._class_weights(input), ignore_index=self.ignore_index)


class WeightedSmoothL1Loss(nn.SmoothL1Loss):
    """SmoothL1 whose per-voxel terms are multiplied by `initial_weight` where the TARGET is below (or at/above) a
    threshold, then averaged — reference losses.py:230-250."""

    def __init__(self, threshold, initial_weight, apply_below_threshold=True):
        super().__init__(reduction="none")
        self.threshold = threshold
        self.apply_below_threshold = apply_below_threshold
        self.weight = initial_weight

    def forward(self, input, target):
        per_voxel = super().forward(input, target)
        picked = (target < self.threshold) if self.apply_below_threshold else (target >= self.threshold)
        return torch.where(picked, per_voxel * self.weight, per_voxel).mean()


def _create_loss(name, loss_config, weight, ignore_index, pos_weight):
    if name == "BCEWithLogitsLoss":
        return BCEWithLogitsLoss(pos_weight=pos_weight)
    if name == "BCEDiceLoss":
        return BCEDiceLoss(loss_config.get("alpha", 1.0))
    if name == "DiceLoss":
        return DiceLoss(weight=weight, normalization=loss_config.get("normalization", "sigmoid"))
    if name == "CrossEntropyLoss":
        return nn.CrossEntropyLoss(weight=weight, ignore_index=-100 if ignore_index is None else ignore_index)
    if name == "MSELoss":
        return nn.MSELoss()
    if name == "SmoothL1Loss":
        return nn.SmoothL1Loss()
    if name == "L1Loss":
        return nn.L1Loss()
    if name == "WeightedCrossEntropyLoss":
        return WeightedCrossEntropyLoss(ignore_index=-100 if ignore_index is None else ignore_index)
    if name == "GeneralizedDiceLoss":
        return GeneralizedDiceLoss(normalization=loss_config.get("normalization", "sigmoid"))
    if name == "WeightedSmoothL1Loss":
        return WeightedSmoothL1Loss(threshold=loss_config["threshold"], initial_weight=loss_config["initial_weight"],
                                    apply_below_threshold=loss_config.get("apply_below_threshold", True))
    raise RuntimeError(f"Unsupported loss function: '{name}'")


def get_loss_criterion(config):
    """Loss named by config['loss'] with the reference's option handling (losses.py:273-350): `ignore_index` wraps
    non-cross-entropy losses in MaskingLossWrapper, `skip_last_target` in SkipLastTargetChannelWrapper."""
    assert "loss" in config, "Could not find loss function configuration"
    device = config.get("device", None)
    assert device, "Device not specified in the config file and could not be inferred automatically"
    loss_config = dict(config["loss"])
    name = loss_config.pop("name")
    ignore_index = loss_config.pop("ignore_index", None)
    skip_last_target = loss_config.pop("skip_last_target", False)
    weight = loss_config.pop("weight", None)
    if weight is not None:
        weight = torch.tensor(weight).float()
    pos_weight = loss_config.pop("pos_weight", None)
    if pos_weight is not None:
        pos_weight = torch.tensor(pos_weight)
    loss = _create_loss(name, loss_config, weight, ignore_index, pos_weight)
    if not (ignore_index is None or name in ["CrossEntropyLoss", "WeightedCrossEntropyLoss"]):
        loss = MaskingLossWrapper(loss, ignore_index)
    if skip_last_target:
        loss = SkipLastTargetChannelWrapper(loss, loss_config.get("squeeze_channel", False))
    loss.to(device)
    return loss
