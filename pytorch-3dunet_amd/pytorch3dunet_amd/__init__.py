"""pytorch3dunet_amd — MI355X-native (gfx950) forward/backward path for the 3D U-Nets of wolny/pytorch-3dunet.

Public surface (mirrors pytorch3dunet.unet3d.model of the reference, model.py:361-369):
    from pytorch3dunet_amd.unet3d.model import get_model, UNet3D, ResidualUNet3D, ResidualUNetSE3D, UNet2D,
                                               ResidualUNet2D, is_model_2d
    pytorch3dunet_amd.install()   # make an installed reference (`pytorch3dunet`) use these classes
"""
from .version import __version__  # noqa: F401


def install():
    """Patch an importable reference package so `pytorch3dunet.unet3d.model.get_model` (and the class names
    looked up by `get_class`, utils.py:331-338) resolve to the MI355X-native classes.  train3dunet /
    predict3dunet and the YAML configs then run unchanged."""
    import importlib

    from .unet3d import model as native_model

    ref = importlib.import_module("pytorch3dunet.unet3d.model")
    for name in ("UNet3D", "ResidualUNet3D", "ResidualUNetSE3D", "UNet2D", "ResidualUNet2D", "get_model", "is_model_2d"):
        setattr(ref, name, getattr(native_model, name))
    return ref
