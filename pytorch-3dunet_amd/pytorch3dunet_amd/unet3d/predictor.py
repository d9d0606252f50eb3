"""Predictor classes with the reference's protocol (pytorch3dunet/unet3d/predictor.py:24-77 AbstractPredictor, :79-232
StandardPredictor, :235-283 LazyPredictor) whose patch loop keeps the predictions on the device.

`predict3dunet` looks the class up by name in the module `pytorch3dunet.unet3d.predictor` (predict.py:20-40:
`predictor_class(model, output_dir, out_channels, **predictor_config, device=config["device"])`) and calls it once per test
loader (predict.py:69-80); aliasing that module name to this one makes the unchanged CLI and YAML files use
these classes.  Same constructor arguments, same `__call__(test_loader)`, same output file (`<input stem>_predictions.h5` in
output_dir), same dataset name / dtype / shape conventions for `save_segmentation` and `prediction_channel`, same metrics.

What differs is where the bytes live.  The reference does, per batch, `prediction.cpu().numpy()` — a device synchronisation
plus a D2H copy — and places the patches into a host array (predictor.py:169-196).  Here
  * StandardPredictor allocates the output volume in HBM (288 GB per MI355X hold any volume the reference's host array can
    hold), crops halos and places patches with strided device copies in the reference's order (later patches overwrite earlier
    ones), and makes ONE D2H trip at the end; nothing in the loop synchronises, so the DataLoader, the H2D copies and the
    kernels of consecutive batches overlap;
  * LazyPredictor (volumes that do not fit in memory) streams each batch's predictions through two pinned host buffers on a copy
    stream and writes batch i into the H5 dataset while batch i+1 computes.
The test loader is consumed exactly as the reference consumes it: `(input, indices)` batches, `indices` a list of slice tuples
(datasets/utils.py:478-496), `dataset.volume_shape / halo_shape / file_path` (datasets/hdf5.py:70-113).
"""
from __future__ import annotations

import logging
import time
from pathlib import Path
from typing import Any, Optional

import numpy as np
import torch
from torch import nn

from ..predictor import remove_padding
from .model import is_model_2d

logger = logging.getLogger("UNetPredictor")

_DEVICES = ("cpu", "cuda", "mps")


def _device_str(device) -> str:
    # TorchDevice is a str-Enum in the reference (config.py:15-18); accept it, plain strings and torch.device
    value = getattr(device, "value", device)
    return torch.device(value).type if not isinstance(value, str) else value.split(":")[0]


class AbstractPredictor:
    """Constructor contract of predictor.py:38-61."""

    def __init__(self, model: nn.Module, output_dir: Optional[str], out_channels: int, device, output_dataset: str = "predictions",
                 save_segmentation: bool = False, prediction_channel: Optional[int] = None,
                 performance_metric: Optional[str] = None, gt_internal_path: Optional[str] = None, **kwargs):
        self.model = model
        self.output_dir = output_dir
        assert out_channels > 0, f"Invalid number of output channels: {out_channels}"
        self.out_channels = out_channels
        assert _device_str(device) in _DEVICES, f"Unsupported device: {device}"
        self.device = device
        self.output_dataset = output_dataset
        self.save_segmentation = save_segmentation
        self.prediction_channel = prediction_channel
        self.performance_metric = performance_metric
        self.gt_internal_path = gt_internal_path

    def __call__(self, test_loader) -> Any:
        raise NotImplementedError


def _torch_device(device) -> torch.device:
    value = getattr(device, "value", device)
    return value if isinstance(value, torch.device) else torch.device(str(value))


class StandardPredictor(AbstractPredictor):
    """Applies the model patch by patch and saves the assembled volume as an H5 file (predictor.py:79-232); the volume is
    assembled in device memory."""

    lazy = False

    def __call__(self, test_loader) -> Any:
        import h5py  # the output format is the reference's; imported here so that the module loads without h5py

        dataset = test_loader.dataset
        for attr in ("file_path", "volume_shape", "halo_shape"):
            assert hasattr(dataset, attr), f"test dataset lacks '{attr}' (an AbstractHDF5Dataset is expected, datasets/hdf5.py:39)"
        logger.info(f"Processing '{dataset.file_path}'...")
        start = time.perf_counter()
        volume_shape = tuple(dataset.volume_shape)
        if self.save_segmentation:
            prediction_shape = volume_shape                        # single channel segmentation map (:119-121)
        elif self.prediction_channel is not None:
            prediction_shape = (1,) + volume_shape                 # (:123-125)
        else:
            prediction_shape = (self.out_channels,) + volume_shape
        output_file = _get_output_file(dataset=dataset, output_dir=self.output_dir)
        logger.info(f"Saving predictions to: {output_file}")
        dev = _torch_device(self.device)
        with h5py.File(output_file, "w") as h5_output_file:
            prediction_array = self._allocate_prediction_array(prediction_shape, h5_output_file)
            patch_halo = tuple(dataset.halo_shape)
            logger.info(f"Using halo: {patch_halo}")
            self.model.eval()  # (:141-143)
            logger.info(f"Running inference on {len(test_loader)} batches")
            volume, streamed = None, self.lazy
            if not self.lazy:
                try:
                    volume = torch.zeros(prediction_shape, dtype=torch.int32 if self.save_segmentation else torch.float32, device=dev)
                except torch.OutOfMemoryError:
                    # the reference assembles the volume in HOST memory (predictor.py:130-133): a volume that fits there but not
                    # beside the model's activations in HBM must still predict — stream the patches through the pinned slabs
                    logger.warning("output volume does not fit in free device memory: staging the patches through pinned host buffers")
                    torch.cuda.empty_cache()
                    streamed = True
            stager = _HostStager(dev) if streamed else None
            with torch.no_grad():
                for input, indices in test_loader:
                    input = input.to(dev, non_blocking=True)
                    if is_model_2d(self.model):
                        prediction = torch.unsqueeze(self.model(torch.squeeze(input, dim=-3)), dim=-3)  # (:154-160)
                    else:
                        prediction = self.model(input)
                    if sum(patch_halo) > 0:
                        prediction = remove_padding(prediction, patch_halo)  # (:166-167)
                    assert len(prediction) == len(indices), "batch of predictions and of patch indices differ in length"
                    placed = [self._patch_and_index(pred, index) for pred, index in zip(prediction, indices)]
                    if streamed:
                        stager.push(placed, prediction_array)
                    else:
                        for index, pred in placed:  # strided device copies; later patches overwrite earlier ones
                            volume[index] = pred
            if streamed:
                stager.drain(prediction_array)
            else:
                prediction_array[...] = volume.cpu().numpy().astype(prediction_array.dtype, copy=False)  # the ONE D2H trip
            logger.info(f"Finished inference in {time.perf_counter() - start:.2f} seconds")
            self._create_prediction_dataset(h5_output_file, prediction_array)
            if self.performance_metric is not None:
                assert self.gt_internal_path is not None
                gt = _load_dataset(dataset, self.gt_internal_path)
                prediction_array = prediction_array[...]
                assert self.performance_metric in ["dice", "mean_iou"], (
                    f"Unsupported performance metric: {self.performance_metric}, only dice and mean_iou are supported")
                if self.performance_metric == "dice":
                    return dice_score(prediction_array, gt)
                return mean_iou(prediction_array, gt, n_classes=self.out_channels)

    def _patch_and_index(self, pred: torch.Tensor, index):
        """one sample's (destination index, values) with the reference's rules (predictor.py:171-193), on the device"""
        if self.save_segmentation:
            seg = (pred[0] > 0.5) if pred.shape[0] == 1 else torch.argmax(pred, dim=0)
            return tuple(index), seg.to(torch.int32)
        if self.prediction_channel is None:
            return (slice(0, self.out_channels),) + tuple(index), pred
        return (slice(0, 1),) + tuple(index), pred[self.prediction_channel:self.prediction_channel + 1]

    def _create_prediction_dataset(self, h5_output_file, prediction_array):
        h5_output_file.create_dataset(self.output_dataset, data=prediction_array, compression="gzip")

    def _allocate_prediction_array(self, output_shape, output_file):
        return np.zeros(output_shape, dtype="uint16" if self.save_segmentation else "float32")


class LazyPredictor(StandardPredictor):
    """Writes the predicted patches straight into the H5 dataset (predictor.py:235-283) — for volumes that fit neither in host
    nor in device memory.  The per-batch D2H copy runs on a copy stream into one of two pinned buffers while the next batch
    computes."""

    lazy = True

    def _allocate_prediction_array(self, output_shape, output_file):
        dtype = "uint16" if self.save_segmentation else "float32"
        return output_file.create_dataset(self.output_dataset, shape=output_shape, dtype=dtype, chunks=True, compression="gzip")

    def _create_prediction_dataset(self, h5_output_file, prediction_array):
        pass  # already in the file


class _HostStager:
    """two-slot pinned staging of per-batch predictions: D2H of batch i on a side stream, H5 write of batch i-1 on the host.
    The two slabs are allocated once (grown if a later batch is larger) and used alternately — pinned allocations are
    expensive host calls, one per sample per batch would cost more than the copies."""

    def __init__(self, dev: torch.device):
        self.dev = dev
        self.cuda = dev.type == "cuda"
        self.stream = torch.cuda.Stream(dev) if self.cuda else None
        self.pending = []  # [(event, [(index, host view)])], at most two
        self.slabs = [None, None]  # pinned byte buffers
        self.turn = 0

    def _slab(self, nbytes: int) -> torch.Tensor:
        buf = self.slabs[self.turn]
        if buf is None or buf.numel() < nbytes:
            buf = self.slabs[self.turn] = torch.empty(max(nbytes, 1), dtype=torch.uint8, pin_memory=True)
        return buf

    def push(self, placed, dest):
        if not self.cuda:
            for index, pred in placed:
                dest[index] = pred.numpy().astype(dest.dtype, copy=False)
            return
        # the slab of this turn was last used two pushes ago; its batch has been flushed (at most one batch stays pending)
        sizes = [(pred.numel() * pred.element_size() + 63) // 64 * 64 for _, pred in placed]
        slab = self._slab(sum(sizes))
        self.stream.wait_stream(torch.cuda.current_stream(self.dev))
        items, off = [], 0
        with torch.cuda.stream(self.stream):
            for (index, pred), size in zip(placed, sizes):
                host = slab[off:off + pred.numel() * pred.element_size()].view(pred.dtype).view(pred.shape)
                off += size
                host.copy_(pred, non_blocking=True)
                pred.record_stream(self.stream)
                items.append((index, host))
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.pending.append((ev, items))
        self.turn ^= 1
        while len(self.pending) > 1:  # write the previous batch while this one is in flight
            self._flush_one(dest)

    def _flush_one(self, dest):
        ev, items = self.pending.pop(0)
        ev.synchronize()
        for index, host in items:
            dest[index] = host.numpy().astype(dest.dtype, copy=False)

    def drain(self, dest):
        while self.pending:
            self._flush_one(dest)


def _get_output_file(dataset, suffix: str = "_predictions", output_dir=None) -> Path:
    """<output_dir or the input's directory>/<input stem>_predictions.h5 (predictor.py:333-356)"""
    file_path = Path(dataset.file_path)
    out_dir = file_path.parent if output_dir is None else Path(output_dir)
    return out_dir / (file_path.stem + suffix + ".h5")


def _load_dataset(dataset, internal_path: str) -> np.ndarray:
    import h5py

    with h5py.File(dataset.file_path, "r") as f:
        return f[internal_path][...]


def mean_iou(pred: np.ndarray, gt: np.ndarray, n_classes: int, avg: bool = False):
    """per-class intersection over union, background (class 0) skipped (predictor.py:365-393)"""
    pred, gt = pred.astype("uint16"), gt.astype("uint16")
    assert pred.shape == gt.shape, f"Predictions and ground truth have different shapes: {pred.shape} != {gt.shape}"
    per_class = [np.logical_and(gt == c, pred == c).sum() / np.logical_or(gt == c, pred == c).sum() for c in range(1, n_classes)]
    return np.mean(per_class) if avg else per_class


def dice_score(pred: np.ndarray, gt: np.ndarray, avg: bool = False):
    """per-channel Dice of binarised volumes (predictor.py:396-416)"""
    pred, gt = pred.astype("uint16"), gt.astype("uint16")
    assert pred.shape == gt.shape, f"Predictions and ground truth have different shapes: {pred.shape} != {gt.shape}"
    assert len(pred) == len(gt)
    per_class = [2 * np.logical_and(c_gt, c_pred).sum() / (c_gt.sum() + c_pred.sum()) for c_pred, c_gt in zip(pred, gt)]
    return np.mean(per_class) if avg else per_class
