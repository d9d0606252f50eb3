"""TEST INFRASTRUCTURE ONLY — import the untouched reference (wolny/pytorch-3dunet 1.9.6) from /root/reference.

The reference's model code needs `skimage` / `h5py` only at import time (pytorch3dunet/unet3d/utils.py:10,
datasets/*.py); neither is installed here and there is no network, so empty stand-in modules are registered
before the import.  No numerics are involved.  /root/reference exists only in the build
container: callers must check `reference_available()` and never use this from `-m gpu` tests, smoke() or bench.py.
"""
import importlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("U3D_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "pytorch3dunet", "unet3d"))


def _stub(name, **attrs):
    if name in sys.modules:
        return sys.modules[name]
    mod = types.ModuleType(name)
    mod.__dict__.update(attrs)
    sys.modules[name] = mod
    return mod


def _missing(*_a, **_k):
    raise RuntimeError("stubbed third-party function called: not available in this container")


class _Permissive(types.ModuleType):
    """a stand-in module on which every attribute exists (a function that raises when CALLED): lets the reference's
    trainer / predictor / metrics modules import in a container without skimage, h5py, tensorboard, imageio"""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _missing


class _NullWriter:
    """torch.utils.tensorboard.SummaryWriter stand-in: accepts every call, records nothing"""

    def __init__(self, *a, **k):
        pass

    def __getattr__(self, name):
        return lambda *a, **k: None


def import_reference_runtime():
    """Import the reference's TRAINING / PREDICTION runtime (trainer.py, predictor.py, utils.py, metrics.py) with permissive
    stand-ins for the third-party packages this container lacks.  Returns the dict of stand-in modules installed (so a
    test can plug an in-memory `h5py.File`).  No numerics live in the stand-ins."""
    import_reference()
    made = {}
    for name in ("skimage", "skimage.color", "skimage.measure", "skimage.metrics", "skimage.exposure", "skimage.filters",
                 "skimage.segmentation", "h5py", "imageio", "tensorboard", "torch.utils.tensorboard"):
        mod = sys.modules.get(name)
        if mod is None or (not isinstance(mod, _Permissive) and getattr(mod, "__file__", None) is None):
            try:
                if mod is None:
                    importlib.import_module(name)
                    continue
            except Exception:
                pass
            new = _Permissive(name)
            if mod is not None:
                new.__dict__.update({k: v for k, v in mod.__dict__.items() if not k.startswith("__")})
            sys.modules[name] = new
            made[name] = new
    sys.modules["torch.utils.tensorboard"].SummaryWriter = _NullWriter
    return made


def import_reference():
    """Returns the reference's `pytorch3dunet.unet3d.model` module (get_model, UNet3D, ...)."""
    if not reference_available():
        raise RuntimeError(f"reference not found under {REFERENCE_ROOT}")
    for name, attrs in {
        "skimage": {},
        "skimage.color": {"label2rgb": _missing},
        "skimage.measure": {"label": _missing},
        "skimage.metrics": {"adapted_rand_error": _missing, "mean_squared_error": _missing,
                            "peak_signal_noise_ratio": _missing},
        "skimage.exposure": {},
        "skimage.filters": {"gaussian": _missing},
        "skimage.segmentation": {"find_boundaries": _missing},
        "h5py": {"File": _missing, "Dataset": type("Dataset", (), {})},
    }.items():
        try:
            importlib.import_module(name)
        except Exception:
            _stub(name, **attrs)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    return importlib.import_module("pytorch3dunet.unet3d.model")
