"""Drop-in model API of the reference's pytorch3dunet/unet3d/model.py (get_model :361, UNet3D :152,
ResidualUNet3D :193, ResidualUNetSE3D :237, UNet2D :281, ResidualUNet2D :321, is_model_2d :366) whose
forward/backward on an MI355X runs as hand-written gfx950 HIP kernels.

Contract kept:
  * constructor keyword arguments and defaults, extra keys swallowed by **kwargs (model.py:159-174),
  * forward(x, return_logits=False) -> probs | (probs, logits); regression models return (x, x) (model.py:103-149),
  * parameter names / shapes of state_dict() — checkpoints of the reference load with strict=True,
  * class lookup by name through get_model / get_class (utils.py:331-338), isinstance(model, UNet2D) for 2-D.

Dispatch: tensors on a gfx950 device -> native executor (pytorch3dunet_amd/engine.py), which raises if
libu3d_hip.so is missing (no silent fallback).  CPU tensors (`device: cpu`) run the torch.nn modules the tree is
made of.  ONE backend: a 3-D model variant the executor does not cover (layer orders outside engine.layer_spec's grammar,
conv kernels other than 3/pad 1, `pool_type: avg`) RAISES on a HIP device; U3D_ALLOW_TORCH_FALLBACK=1 opts into running
the same module tree through stock PyTorch-ROCm operators after a one-time warning (never counted as covered).  2-D
models are outside the 3-D path and keep that warning path by default; U3D_STRICT=1 makes them an error too.  Covered since round 2: every layer order with at most one
GroupNorm / BatchNorm, one non-linearity and a trailing dropout, every `upsample` value the reference itself can run
on a 3-D net, nn.DataParallel, activation checkpointing, and the opt-in compute modes `bf16` and `fp32_split`.
"""
import os
import warnings

import torch
from torch import nn

from .buildingblocks import DoubleConv, ResNetBlock, ResNetBlockSE, create_decoders, create_encoders
from .utils import get_class, number_of_features_per_level

_THIS_MODULE = __name__


def _mark_engine_stale(module, incompatible_keys):
    """load_state_dict post-hook: parameter OBJECTS may have been replaced (assign=True) -> re-walk them on the next forward"""
    object.__setattr__(module, "_engine_stale", True)


class AbstractUNet(nn.Module):
    """Encoder-decoder skeleton shared by all variants (reference model.py:7-149)."""

    def __init__(self, in_channels, out_channels, final_sigmoid, basic_module, f_maps=64, layer_order="gcr",
                 num_groups=8, num_levels=4, is_segmentation=True, conv_kernel_size=3, pool_kernel_size=2,
                 conv_padding=1, conv_upscale=2, upsample="default", dropout_prob=0.1, is3d=True, compute_dtype=None,
                 checkpoint_encoders=None, hip_graph=None, activation_dtype=None, checkpoint_levels=None):
        super().__init__()
        if isinstance(f_maps, int):
            f_maps = number_of_features_per_level(f_maps, num_levels=num_levels)
        assert isinstance(f_maps, (list, tuple))
        assert len(f_maps) > 1, "Required at least 2 levels in the U-Net"
        if "g" in layer_order:
            assert num_groups is not None, "num_groups must be specified if GroupNorm is used"

        self.encoders = create_encoders(in_channels, f_maps, basic_module, conv_kernel_size, conv_padding, conv_upscale,
                                        dropout_prob, layer_order, num_groups, pool_kernel_size, is3d)
        self.decoders = create_decoders(f_maps, basic_module, conv_kernel_size, conv_padding, layer_order, num_groups,
                                        upsample, dropout_prob, is3d)
        # 1x1(x1) convolution down to the number of labels
        self.final_conv = (nn.Conv3d if is3d else nn.Conv2d)(f_maps[0], out_channels, 1)
        if is_segmentation:
            self.final_activation = nn.Sigmoid() if final_sigmoid else nn.Softmax(dim=1)
        else:
            self.final_activation = None

        # ---- native-path eligibility (everything the gfx950 executor implements today)
        reasons = []
        if not is3d:
            reasons.append("2-D model")
        if basic_module not in (DoubleConv, ResNetBlock, ResNetBlockSE):
            reasons.append(f"basic_module {basic_module.__name__}")
        if basic_module is ResNetBlockSE and (any(f % 4 for f in f_maps) or max(f_maps) > 1024):
            reasons.append("SE gates need channel counts that are multiples of 4 and <= 1024")
        from ..engine import parse_order

        self.layer_order = layer_order
        if parse_order(layer_order) is None:
            reasons.append(f"layer_order '{layer_order}' (native: conv with at most one GroupNorm / BatchNorm before or after it, "
                           "one non-linearity, a trailing dropout)")
        elif basic_module in (ResNetBlock, ResNetBlockSE) and any(ch in layer_order for ch in "dD"):
            reasons.append(f"layer_order '{layer_order}': dropout inside residual blocks")
        if conv_kernel_size != 3 or conv_padding != 1:
            reasons.append("conv kernel/padding other than 3/1")
        if pool_kernel_size != 2:
            reasons.append("pool_kernel_size != 2")
        if basic_module is DoubleConv and upsample not in ("default", "nearest", "deconv", "trilinear", "area"):
            reasons.append(f"upsample '{upsample}'")
        if basic_module in (ResNetBlock, ResNetBlockSE) and upsample not in ("default", "deconv"):
            # (an EXPLICIT 'deconv' keeps concat joining and a 1x1x1 conv deep->shallow in the block, buildingblocks.py:441-468;
            # the interpolation modes do not even run in the reference for residual nets: tests/test_oracle.py)
            reasons.append(f"upsample '{upsample}' with residual blocks")
        if out_channels > 1024 or f_maps[0] > 256:
            reasons.append("head wider than 1024 outputs / 256 inputs")  # (> 16 outputs: the tiled wide-head kernels, round 5)
        # opt-in extras of the native executor (extra keys of the YAML's model section are swallowed by the reference's **kwargs):
        # bf16 MFMA operands with fp32 accumulation / master weights, and recomputation of the encoder blocks in backward
        if compute_dtype is None:
            compute_dtype = ("bf16" if os.environ.get("U3D_BF16", "0") == "1"
                             else "fp32_split" if os.environ.get("U3D_F32_SPLIT", "0") == "1" else "fp32")
        if str(compute_dtype).lower() not in ("fp32", "float32", "bf16", "bfloat16", "fp32_split"):
            raise ValueError(f"compute_dtype must be 'fp32', 'fp32_split' or 'bf16', got {compute_dtype!r}")
        self.compute_bf16 = str(compute_dtype).lower() in ("bf16", "bfloat16")
        # 'fp32_split': fp32-grade convolutions on the bf16 matrix pipe (three-way exact operand split, six partial products;
        # engine.UNet3DEngine.split) — same tensors, same tolerances as 'fp32'
        self.compute_split = str(compute_dtype).lower() == "fp32_split"
        # `checkpoint_encoders: true` (any truthy YAML value: true, 1) recomputes EVERY encoder block in backward.  `checkpoint_levels: k`
        # (its own key since round 6 — ADVICE r05: an integer `checkpoint_encoders: 1` had silently come to mean "one level") restricts
        # that to the k encoder levels of highest resolution (the first two levels of a 5-level net hold ~9/10 of the encoder tape; the
        # deeper ones cost recomputation for a few megabytes each).  An integer `checkpoint_encoders: k` with k >= 2 is still read as
        # `checkpoint_levels: k` (round-5 configurations).  U3D_CHECKPOINT=1 [+ U3D_CHECKPOINT_LEVELS=k] set the defaults of both keys.
        if checkpoint_encoders is None:
            checkpoint_encoders = os.environ.get("U3D_CHECKPOINT", "0") == "1"
            if checkpoint_encoders and checkpoint_levels is None and os.environ.get("U3D_CHECKPOINT_LEVELS", ""):
                checkpoint_levels = int(os.environ["U3D_CHECKPOINT_LEVELS"])
        if not isinstance(checkpoint_encoders, (bool, int)) or int(checkpoint_encoders) < 0:
            raise ValueError(f"checkpoint_encoders must be true / false (or a number of encoder levels >= 2), got {checkpoint_encoders!r}")
        if not isinstance(checkpoint_encoders, bool) and int(checkpoint_encoders) >= 2 and checkpoint_levels is None:
            checkpoint_levels = int(checkpoint_encoders)
        if checkpoint_levels is not None:
            if isinstance(checkpoint_levels, bool) or not isinstance(checkpoint_levels, int) or checkpoint_levels < 1:
                raise ValueError(f"checkpoint_levels must be a positive number of encoder levels, got {checkpoint_levels!r} "
                                 "(to switch checkpointing off set checkpoint_encoders: false)")
            if checkpoint_levels > len(f_maps):
                import warnings

                warnings.warn(f"u3d: checkpoint_levels={checkpoint_levels} exceeds the {len(f_maps)} encoder levels of this model: every level is recomputed")
                checkpoint_levels = len(f_maps)
        self.checkpoint_encoders = bool(checkpoint_encoders)
        self.checkpoint_levels = checkpoint_levels if self.checkpoint_encoders else None
        # `activation_dtype: bf16` / U3D_ACT_BF16=1 (with compute_dtype bf16, residual 'gcr' nets): activations and gradients between
        # kernels are stored as bf16 (engine.ResUNetEngine.act_bf16); anything else keeps fp32 storage
        if activation_dtype is None:
            activation_dtype = {"1": "bf16", "0": "fp32"}.get(os.environ.get("U3D_ACT_BF16", ""), "auto")
        if str(activation_dtype).lower() not in ("auto", "fp32", "float32", "bf16", "bfloat16"):
            raise ValueError(f"activation_dtype must be 'fp32', 'bf16' or 'auto', got {activation_dtype!r}")
        # 'auto' (default): bf16 storage whenever compute_dtype is bf16 and the model lies inside the `_b16` kernels' envelope
        # (decided by the executor: engine.ResUNetEngine._act_bf16_blocker), silently fp32 otherwise; 'bf16' warns when it cannot
        self.activation_dtype = {"float32": "fp32", "bfloat16": "bf16"}.get(str(activation_dtype).lower(), str(activation_dtype).lower())
        self.activation_bf16 = self.activation_dtype in ("bf16", "auto") and self.compute_bf16
        # `hip_graph: true` / U3D_GRAPH=1: training steps replay two captured hipGraphs per input shape (engine.GraphStep)
        if hip_graph is None:
            hip_graph = os.environ.get("U3D_GRAPH", "0") == "1"
        self.hip_graph = bool(hip_graph)
        self._native_blockers = reasons
        self._is3d = bool(is3d)
        self._residual = basic_module in (ResNetBlock, ResNetBlockSE)
        self._engine = None
        self._warned = False
        self._engine_stale = False
        # any load_state_dict on this module may have replaced parameter objects (assign=True): re-walk them on the next forward
        # (a module-level function, not a lambda: the hook dict is part of the module's pickled state — torch.save(model), mp.spawn)
        self.register_load_state_dict_post_hook(_mark_engine_stale)

    def __getstate__(self):
        """torch.save(model) / pickle / copy.deepcopy: the executor (ctypes handles, device scratch, a reference back to this
        module) is per-process state and is rebuilt by the next forward — the reference's models are plain picklable nn.Modules."""
        state = dict(self.__dict__)
        state["_engine"] = None
        state["_engine_stale"] = False
        return state

    # ------------------------------------------------------------------------------------------------
    @property
    def native_supported(self):
        return not self._native_blockers

    def _get_engine(self):
        """The executor bound to THIS module object and its current parameter tensors.  An nn.DataParallel replica
        (reference trainer.py:202-205, predict.py:63-66 wrap the model whenever more than one device is visible) is a shallow
        copy whose __dict__ still points at the original's executor: it gets its own (per replica, per forward — replicas
        are rebuilt by every DataParallel.forward), built from the replica's broadcast parameter copies, so nothing mutable
        is shared between replica threads and gradients flow back through the broadcast.  The executor is also rebuilt
        when parameters were replaced (load_state_dict(assign=True), parametrizations)."""
        from ..engine import ResUNetEngine, UNet3DEngine, module_params

        eng = self.__dict__.get("_engine")
        if eng is not None and eng.model is self:
            # cheap per-forward sentinel (this path is host-bound for small patches): the first and the last parameter OBJECT of the
            # module order; wholesale replacement (load_state_dict(assign=True), parametrizations) changes both.  A replaced MIDDLE
            # parameter (`module.weight = nn.Parameter(...)`, a partial load_state_dict(assign=True, strict=False), weight surgery)
            # is caught by the full identity walk, which runs when the sentinel moved, right after any load_state_dict on this
            # module (post-hook below) and otherwise every 16th forward — i.e. such an edit is seen at once through
            # load_state_dict and within 16 forwards through plain attribute assignment (call
            # `invalidate_native_caches()` after weight surgery to force it immediately)
            fc = self.final_conv
            last = fc.bias if fc.bias is not None else fc.weight
            owner = eng._first_param_owner  # (None on an nn.DataParallel replica: its parameters are plain attributes)
            eng._fwd_checks = getattr(eng, "_fwd_checks", 0) + 1
            stale = self.__dict__.get("_engine_stale", False) or (eng._fwd_checks & 15) == 0
            if (not stale and owner is not None and id(last) == eng._pids[-1]
                    and id(owner._parameters.get(eng._first_param_name)) == eng._pids[0]):
                return eng
            object.__setattr__(self, "_engine_stale", False)
            if eng._pids == [id(p) for p in module_params(self)]:
                return eng
        new = (ResUNetEngine if self._residual else UNet3DEngine)(self)
        if eng is not None and eng.model is self:
            new.grad_sync = eng.grad_sync  # same module, new parameter objects: keep the data-parallel hook
        object.__setattr__(self, "_engine", new)
        return new

    def invalidate_native_caches(self):
        """Drop the executor's packed weight images.  Needed only after writing parameters through `param.data` (which autograd's
        version counter does not see) while the model is in inference use; training forwards repack every step anyway."""
        eng = self.__dict__.get("_engine")
        object.__setattr__(self, "_engine_stale", True)  # also re-walk the parameter identities on the next forward
        if eng is not None:
            eng._salt += 1

    def forward(self, x, return_logits=False):
        """(N,C,D,H,W) -> probabilities, or (probabilities, logits) when return_logits (model.py:103-121)."""
        output, logits = self._forward_logits(x)
        if return_logits:
            return output, logits
        return output

    def _forward_logits(self, x):
        if x.is_cuda:
            if self.native_supported and x.dtype == torch.float32 and x.dim() == 5:
                from ..engine import StaleParameters, run_model

                try:
                    return run_model(self._get_engine(), x)
                except StaleParameters:
                    # a parameter OBJECT was replaced since the executor was built (`module.weight = nn.Parameter(...)`): the cheap
                    # per-forward sentinel of _get_engine only watches the first and the last one.  Rebuild and run again — nothing
                    # was handed to the caller yet
                    object.__setattr__(self, "_engine_stale", True)  # -> full identity walk -> new executor (keeps the grad_sync hook)
                    return run_model(self._get_engine(), x)
            why = ", ".join(self._native_blockers) or (
                f"input is {x.dim()}-D, the 3-D path takes (N,C,D,H,W)" if x.dim() != 5 else
                f"input dtype {x.dtype}: the native path takes float32 tensors (pass x.float(); reduced-precision arithmetic is the "
                "model key compute_dtype: bf16, not a half-precision input / model.half() / autocast)")
            eng = self.__dict__.get("_engine")
            if eng is not None and eng.grad_sync is not None and torch.is_grad_enabled():
                # parallel.attach hooked the gradient exchange into the native executor: the module tree would train unsynchronised
                raise RuntimeError(f"u3d: data-parallel model fell back to the module tree ({why}); its gradients would not be "
                                   "averaged across ranks — call pytorch3dunet_amd.parallel.attach again for this configuration")
            self._uncovered_on_hip(why)
        return self._forward_modules(x)

    def _uncovered_on_hip(self, why):
        """ONE backend by default: a 3-D model the executor does not cover is an ERROR on a HIP device.  The module tree on stock
        PyTorch-ROCm operators is an explicit opt-in (U3D_ALLOW_TORCH_FALLBACK=1) — it is not this library's product and is never
        counted as covered.  2-D models are outside the 3-D path altogether (north_star) and keep the warning; U3D_STRICT=1 makes
        every uncovered configuration an error."""
        allow = not self._is3d or os.environ.get("U3D_ALLOW_TORCH_FALLBACK", "0") == "1"
        if os.environ.get("U3D_STRICT", "0") == "1" or not allow:
            raise NotImplementedError(f"u3d: no native gfx950 path for this configuration ({why}); set "
                                      "U3D_ALLOW_TORCH_FALLBACK=1 to run the module tree on stock PyTorch-ROCm operators instead")
        if not self._warned:
            warnings.warn(f"u3d: {type(self).__name__} ({why}) is not covered by the native gfx950 executor; "
                          "running the module tree through stock PyTorch-ROCm operators", stacklevel=4)
            self._warned = True

    def _forward_modules(self, x):
        """Plain torch.nn execution of the module tree (CPU tensors; uncovered variants)."""
        features = []
        for encoder in self.encoders:
            x = encoder(x)
            features.insert(0, x)
        # the deepest encoder output is the decoder input, not a skip
        for decoder, skip in zip(self.decoders, features[1:]):
            x = decoder(skip, x)
        x = self.final_conv(x)
        if self.final_activation is not None:
            return self.final_activation(x), x
        return x, x


def _variant(name, basic_module, default_levels, is3d, doc):
    """The five public classes differ only in (basic_module, default num_levels, is3d): build them from one
    template so their signatures stay identical to the reference's (model.py:152-358)."""

    def __init__(self, in_channels, out_channels, final_sigmoid=True, f_maps=64, layer_order="gcr", num_groups=8,
                 num_levels=default_levels, is_segmentation=True, conv_padding=1, conv_upscale=2, upsample="default",
                 dropout_prob=0.1, **kwargs):
        AbstractUNet.__init__(self, in_channels=in_channels, out_channels=out_channels, final_sigmoid=final_sigmoid,
                              basic_module=basic_module, f_maps=f_maps, layer_order=layer_order, num_groups=num_groups,
                              num_levels=num_levels, is_segmentation=is_segmentation, conv_padding=conv_padding,
                              conv_upscale=conv_upscale, upsample=upsample, dropout_prob=dropout_prob, is3d=is3d,
                              compute_dtype=kwargs.get("compute_dtype"), checkpoint_encoders=kwargs.get("checkpoint_encoders"),
                              hip_graph=kwargs.get("hip_graph"), activation_dtype=kwargs.get("activation_dtype"),
                              checkpoint_levels=kwargs.get("checkpoint_levels"))

    return type(name, (AbstractUNet,), {"__init__": __init__, "__doc__": doc, "__module__": _THIS_MODULE})


UNet3D = _variant("UNet3D", DoubleConv, 4, True,
                  "3D U-Net (DoubleConv blocks, nearest-neighbour upsampling + concat) — reference model.py:152-190.")
ResidualUNet3D = _variant("ResidualUNet3D", ResNetBlock, 5, True,
                          "Residual 3D U-Net (ResNetBlock, transposed-conv upsampling + sum) — reference model.py:193-234.")
ResidualUNetSE3D = _variant("ResidualUNetSE3D", ResNetBlockSE, 5, True,
                            "Residual 3D U-Net with squeeze-and-excitation blocks — reference model.py:237-278.")
UNet2D = _variant("UNet2D", DoubleConv, 4, False, "2D U-Net — reference model.py:281-318.")
ResidualUNet2D = _variant("ResidualUNet2D", ResNetBlock, 5, False, "Residual 2D U-Net — reference model.py:321-358.")


def get_model(model_config):
    """Instantiate the class named by model_config['name'] with the whole dict as kwargs (model.py:361-363)."""
    model_class = get_class(model_config["name"], modules=[_THIS_MODULE])
    return model_class(**model_config)


def is_model_2d(model):
    """True for UNet2D (also when wrapped in nn.DataParallel) — model.py:366-369."""
    if isinstance(model, nn.DataParallel):
        model = model.module
    return isinstance(model, UNet2D)
