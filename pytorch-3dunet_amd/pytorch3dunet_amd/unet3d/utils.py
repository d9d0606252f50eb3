"""The two helpers of pytorch3dunet/unet3d/utils.py that the model constructor uses (utils.py:110-112, :331-338)."""
import importlib


def number_of_features_per_level(init_channel_number: int, num_levels: int) -> list:
    """[f, 2f, 4f, ...] — feature maps per encoder level (reference utils.py:110-112)."""
    return [init_channel_number << level for level in range(num_levels)]


def get_class(class_name: str, modules: list) -> type:
    """First attribute called `class_name` found in `modules` (reference utils.py:331-338; same RuntimeError)."""
    for module_name in modules:
        found = getattr(importlib.import_module(module_name), class_name, None)
        if found is not None:
            return found
    raise RuntimeError(f"Unsupported dataset class: {class_name}")
