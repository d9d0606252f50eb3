"""TEST INFRASTRUCTURE — run the reference's UNMODIFIED StandardPredictor / LazyPredictor
(/root/reference/pytorch3dunet/unet3d/predictor.py:79-283) and this repository's drop-in classes
(pytorch3dunet_amd/unet3d/predictor.py) on the same model and the same test loader, and the host-side restatement of the loop
(oracle/predictor_oracle.py) on the same volume: all three must produce the same H5 contents bit for bit (CPU, one model).
That pins the oracle's FULL loop to the reference and the drop-in classes to both.  Fresh interpreter:

    python tests/drive_reference_predictor.py

Stand-ins (no numerics): in-memory h5py (tests/fake_h5py.py), permissive skimage; the dataset is a subclass of the reference's
AbstractHDF5Dataset built without its HDF5-reading constructor (the predictor asserts isinstance, predictor.py:113) but served
by the reference's own __getitem__ (datasets/hdf5.py:154-173), mirror_pad, SliceBuilder and default_prediction_collate.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "pytorch-3dunet_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def main():
    import torch

    import fake_h5py
    fake_h5py.install()
    from ref_import import import_reference_runtime

    import_reference_runtime()
    fake_h5py.install()  # (the permissive stand-in may have replaced it)
    import pytorch3dunet.unet3d.predictor as RP
    from pytorch3dunet.datasets.hdf5 import AbstractHDF5Dataset
    from pytorch3dunet.datasets.utils import SliceBuilder, default_prediction_collate, mirror_pad
    from pytorch3dunet.unet3d.config import TorchDevice

    import predictor_oracle as PO
    import pytorch3dunet_amd.unet3d.predictor as MP
    from pytorch3dunet_amd.unet3d.model import UNet3D

    class MemDataset(AbstractHDF5Dataset):
        """test-phase dataset over an in-memory volume; everything but the constructor is the reference's code"""

        def __init__(self, raw, patch, stride, halo, mean, std, file_path):
            self.phase = "test"
            self.file_path = file_path
            self.halo_shape = tuple(halo)
            self.volume_shape = raw.shape if raw.ndim == 3 else raw.shape[1:]
            self.raw_padded = mirror_pad(raw, self.halo_shape)  # hdf5.py:277
            self.raw_slices = SliceBuilder(raw, None, patch, stride, skip_shape_check=True).raw_slices
            self.label = None
            self.patch_count = len(self.raw_slices)
            self.mean, self.std = mean, std

        def raw_transform(self, m):  # Standardize + ToTensor(expand_dims) of the test transformer (transforms.py:653-688,801-826)
            m = (m - self.mean) / np.clip(self.std, a_min=1e-10, a_max=None)
            if m.ndim == 3:
                m = np.expand_dims(m, axis=0)
            return torch.from_numpy(m.astype(np.float32))

        def get_raw_padded_patch(self, idx):
            return self.raw_padded[idx]

        def get_raw_patch(self, idx):
            raise NotImplementedError

        def get_label_patch(self, idx):
            raise NotImplementedError

        def is_lazy(self):
            return False

    torch.manual_seed(0)
    rng = np.random.RandomState(0)
    results = {}
    for case, (cin, cout, shape, patch, stride, halo, kw) in {
        "probs": (1, 2, (20, 33, 29), (8, 16, 16), (6, 12, 12), (2, 4, 4), {}),
        "channel": (1, 3, (16, 24, 24), (8, 16, 16), (8, 8, 8), (0, 0, 0), {"prediction_channel": 1}),
        "segm_multi": (2, 3, (16, 24, 24), (8, 16, 16), (8, 16, 16), (2, 2, 2), {"save_segmentation": True}),
        "segm_single": (1, 1, (16, 24, 24), (8, 16, 16), (8, 16, 16), (2, 2, 2), {"save_segmentation": True}),
    }.items():
        model = UNet3D(cin, cout, f_maps=[4, 8], num_groups=2, final_sigmoid=(cout == 1)).eval()
        raw = rng.randn(*((cin,) + shape if cin > 1 else shape)).astype(np.float32) * 3 + 1
        mean, std = float(raw.mean()), float(raw.std())
        ds = MemDataset(raw, patch, stride, halo, mean, std, f"/mem/{case}.h5")
        loader = torch.utils.data.DataLoader(ds, batch_size=3, num_workers=0, collate_fn=default_prediction_collate)
        outs = {}
        for tag, cls in (("ref_standard", RP.StandardPredictor), ("ref_lazy", RP.LazyPredictor),
                         ("our_standard", MP.StandardPredictor), ("our_lazy", MP.LazyPredictor)):
            pred = cls(model, f"/mem/{tag}", cout, TorchDevice.CPU, output_dataset="predictions", **kw)
            pred(loader)
            outs[tag] = fake_h5py.STORE[f"/mem/{tag}/{case}_predictions.h5"]["predictions"].copy()
        outs["oracle"] = PO.standard_predict(model, raw, patch, stride, halo, batch_size=3, mean=mean, std=std, **kw)
        ref = outs["ref_standard"]
        results[case] = {k: bool(v.shape == ref.shape and v.dtype == ref.dtype and np.array_equal(v, ref)) for k, v in outs.items()}
        results[case]["shape"] = list(ref.shape)
        results[case]["dtype"] = str(ref.dtype)
    # metric branch (predictor.py:204-217): dice against a stored ground truth
    model = UNet3D(1, 2, f_maps=[4, 8], num_groups=2, final_sigmoid=True).eval()
    raw = rng.randn(16, 24, 24).astype(np.float32)
    ds = MemDataset(raw, (8, 16, 16), (8, 8, 8), (0, 0, 0), 0.0, 1.0, "/mem/gt.h5")
    fake_h5py.STORE["/mem/gt.h5"] = {"label": (rng.rand(2, 16, 24, 24) > 0.5).astype(np.uint16)}
    loader = torch.utils.data.DataLoader(ds, batch_size=2, collate_fn=default_prediction_collate)
    a = RP.StandardPredictor(model, "/mem/m_ref", 2, TorchDevice.CPU, performance_metric="dice", gt_internal_path="label")(loader)
    b = MP.StandardPredictor(model, "/mem/m_our", 2, TorchDevice.CPU, performance_metric="dice", gt_internal_path="label")(loader)
    results["dice_equal"] = bool(np.allclose(a, b, rtol=0, atol=0))
    # the class lookup of predict.get_predictor (predict.py:20-40) resolves by name in whatever module carries that name
    results["names"] = all(hasattr(MP, n) for n in ("AbstractPredictor", "StandardPredictor", "LazyPredictor", "mean_iou", "dice_score"))
    print("RESULT " + json.dumps(results), flush=True)


if __name__ == "__main__":
    main()
