"""Device-resident sliding-window inference: the patch loop of the reference's
`StandardPredictor.__call__` (pytorch3dunet/unet3d/predictor.py:112-214) with the volume, the patch gather, the halo
crop and the output volume all kept in HBM.

The reference loop, per batch: DataLoader workers slice a mirror-padded copy of the volume on the host
(datasets/hdf5.py:154-173, datasets/utils.py:518-546), `.to(device)`, forward, `remove_padding`
(datasets/utils.py:549-565), `.cpu().numpy()` — a device synchronisation and a D2H copy per batch — and
`prediction_array[index] = pred` on the host (predictor.py:169-196).  Here the whole volume makes ONE H2D trip, the
padded volume / patches / predictions live on the device (288 GB of HBM hold any volume the reference can hold in
host RAM), patches are gathered and written back with strided device copies, and the result makes ONE D2H trip.
Patch grid, halo handling, write order (later patches overwrite earlier ones), `save_segmentation` and
`prediction_channel` semantics are the reference's.
"""
from __future__ import annotations

from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F


def gen_indices(i: int, k: int, s: int) -> Iterator[int]:
    """Patch start positions along one axis (SliceBuilder._gen_indices, datasets/utils.py:277-282): a regular grid of
    stride s plus a final patch flush with the border when the grid does not end there."""
    assert i >= k, "Sample size has to be bigger than the patch size"
    j = 0
    for j in range(0, i - k + 1, s):
        yield j
    if j + k < i:
        yield i - k


def build_slices(volume_shape: Sequence[int], patch_shape: Sequence[int], stride_shape: Sequence[int]) -> List[Tuple[slice, slice, slice]]:
    """z-major list of (z, y, x) patch slices over the UNPADDED volume (SliceBuilder._build_slices, :237-274)."""
    i_z, i_y, i_x = volume_shape
    k_z, k_y, k_x = patch_shape
    s_z, s_y, s_x = stride_shape
    out = []
    for z in gen_indices(i_z, k_z, s_z):
        for y in gen_indices(i_y, k_y, s_y):
            for x in gen_indices(i_x, k_x, s_x):
                out.append((slice(z, z + k_z), slice(y, y + k_y), slice(x, x + k_x)))
    return out


def mirror_pad(volume: torch.Tensor, halo: Sequence[int]) -> torch.Tensor:
    """(C,Z,Y,X) -> reflect-padded by the halo on both sides of every spatial axis (np.pad(mode='reflect'),
    datasets/utils.py:518-546)."""
    if any(p < 0 for p in halo):
        raise ValueError("padding_shape must be non-negative")
    if all(p == 0 for p in halo):
        return volume
    hz, hy, hx = halo
    return F.pad(volume.unsqueeze(0), (hx, hx, hy, hy, hz, hz), mode="reflect").squeeze(0)


def remove_padding(m: torch.Tensor, halo: Optional[Sequence[int]]) -> torch.Tensor:
    """strip `halo` voxels from both ends of the trailing axes (datasets/utils.py:549-565)."""
    if halo is None:
        return m
    idx = (Ellipsis,) + tuple(slice(p, -p or None) for p in halo)
    return m[idx]


@torch.no_grad()
def predict_volume(model: torch.nn.Module, raw, patch_shape: Sequence[int], stride_shape: Sequence[int],
                   halo_shape: Sequence[int] = (0, 0, 0), batch_size: int = 1, device=None, mean: Optional[float] = None,
                   std: Optional[float] = None, eps: float = 1e-10, save_segmentation: bool = False,
                   prediction_channel: Optional[int] = None, return_tensor: bool = False):
    """Sliding-window prediction of a whole volume.

    raw: (Z,Y,X) or (C,Z,Y,X) numpy array / tensor.  `mean`/`std`: the Standardize transform of the test-phase raw
    transformer with pre-computed global statistics (augment/transforms.py:653-688), applied on the device.
    Returns what StandardPredictor stores: float32 (C_out,Z,Y,X) probabilities ((1,Z,Y,X) with prediction_channel),
    or a uint16 (Z,Y,X) segmentation when save_segmentation (predictor.py:119-128,171-196)."""
    from .unet3d.model import is_model_2d

    if is_model_2d(model):
        raise NotImplementedError("predict_volume covers the 3-D models; 2-D models go through the reference's predictor")
    if device is None:
        device = next(model.parameters()).device
    device = torch.device(device)
    vol = torch.as_tensor(np.ascontiguousarray(raw) if isinstance(raw, np.ndarray) else raw)
    if vol.dim() == 3:
        vol = vol.unsqueeze(0)  # ToTensor(expand_dims=True), transforms.py:816-820
    assert vol.dim() == 4, "raw must be (Z,Y,X) or (C,Z,Y,X)"
    if device.type == "cuda" and vol.device.type == "cpu":
        vol = vol.pin_memory()
    vol = vol.to(device=device, dtype=torch.float32, non_blocking=True)
    if mean is not None or std is not None:
        assert mean is not None and std is not None
        vol = (vol - float(mean)) / max(float(std), eps)
    volume_shape = tuple(vol.shape[1:])
    halo = tuple(int(h) for h in halo_shape)
    padded = mirror_pad(vol, halo)
    slices = build_slices(volume_shape, patch_shape, stride_shape)
    was_training = model.training
    model.eval()  # predictor.py:141-143
    out = None
    try:
        for b0 in range(0, len(slices), batch_size):
            chunk = slices[b0:b0 + batch_size]
            # padded index = [start, stop + 2*halo) in the padded volume (hdf5.py:16-20)
            batch = torch.stack([padded[:, sz.start:sz.stop + 2 * halo[0], sy.start:sy.stop + 2 * halo[1],
                                        sx.start:sx.stop + 2 * halo[2]] for (sz, sy, sx) in chunk])
            pred = model(batch)
            if sum(halo) > 0:
                pred = remove_padding(pred, halo)
            if out is None:
                n_out = pred.shape[1]
                if save_segmentation:
                    out = torch.zeros(volume_shape, dtype=torch.int32, device=device)
                else:
                    out = torch.zeros(((1 if prediction_channel is not None else n_out),) + volume_shape,
                                      dtype=torch.float32, device=device)
            for p, (sz, sy, sx) in zip(pred, chunk):
                if save_segmentation:
                    seg = (p[0] > 0.5) if p.shape[0] == 1 else torch.argmax(p, dim=0)
                    out[sz, sy, sx] = seg.to(torch.int32)
                elif prediction_channel is not None:
                    out[0:1, sz, sy, sx] = p[prediction_channel:prediction_channel + 1]
                else:
                    out[:, sz, sy, sx] = p
    finally:
        model.train(was_training)
    if return_tensor:
        return out
    res = out.cpu().numpy()
    return res.astype(np.uint16) if save_segmentation else res
