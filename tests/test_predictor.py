"""Sliding-window inference: pytorch3dunet_amd.predictor against the host-side restatement of the
reference's StandardPredictor loop (oracle/predictor_oracle.py), the patch grid / padding helpers against the LIVE
reference when /root/reference is present, and — on the GPU — the native forward path inside the loop."""
import numpy as np
import pytest
import torch

import predictor_oracle as porc
from pytorch3dunet_amd import predictor as P
from pytorch3dunet_amd.unet3d.model import UNet3D, get_model
from ref_import import import_reference, reference_available


@pytest.mark.parametrize("i,k,s", [(10, 4, 2), (10, 4, 3), (7, 7, 1), (100, 32, 20), (65, 64, 64), (9, 3, 9)])
def test_gen_indices_restatements_agree(i, k, s):
    assert list(P.gen_indices(i, k, s)) == list(porc.gen_indices(i, k, s))
    idx = list(P.gen_indices(i, k, s))
    assert idx[0] == 0 and idx[-1] == i - k and all(b > a for a, b in zip(idx, idx[1:]))


@pytest.mark.skipif(not reference_available(), reason="needs /root/reference (build container only)")
def test_grid_and_padding_match_live_reference():
    import importlib

    import_reference()
    ru = importlib.import_module("pytorch3dunet.datasets.utils")
    rng = np.random.default_rng(0)
    for shape, patch, stride in [((20, 70, 90), (8, 64, 64), (4, 32, 40)), ((16, 64, 64), (16, 64, 64), (8, 8, 8)),
                                 ((33, 100, 81), (10, 64, 70), (7, 30, 11))]:
        vol = rng.standard_normal(shape).astype(np.float32)
        ref = ru.SliceBuilder._build_slices(vol, patch, stride)
        assert [tuple(s) for s in ref] == P.build_slices(shape, patch, stride)
    vol = rng.standard_normal((2, 9, 11, 13)).astype(np.float32)
    for halo in [(0, 0, 0), (2, 3, 4), (1, 0, 5)]:
        ref = ru.mirror_pad(vol, halo)
        got = P.mirror_pad(torch.from_numpy(vol), halo).numpy()
        assert np.array_equal(ref, got)
        assert np.array_equal(ru.remove_padding(ref, halo), P.remove_padding(torch.from_numpy(ref), halo).numpy())
    with pytest.raises(ValueError):
        P.mirror_pad(torch.zeros(1, 4, 4, 4), (1, -1, 0))


def _tiny_model(name="UNet3D", cin=1, cout=1, **kw):
    torch.manual_seed(7)
    cfg = dict(name=name, in_channels=cin, out_channels=cout, f_maps=[8, 16], num_groups=4, **kw)
    return get_model(cfg).eval()


@pytest.mark.parametrize("cin,cout,halo,kw", [(1, 1, (2, 4, 4), {}), (2, 3, (0, 0, 0), dict(final_sigmoid=False)),
                                              (1, 2, (1, 2, 3), dict(final_sigmoid=False))])
def test_predict_volume_cpu_matches_reference_loop(cin, cout, halo, kw):
    model = _tiny_model(cin=cin, cout=cout, **kw)
    rng = np.random.default_rng(1)
    shape = (14, 30, 27)
    raw = rng.standard_normal(shape if cin == 1 else (cin,) + shape).astype(np.float32) * 3 + 1
    patch, stride = (8, 16, 16), (5, 9, 11)
    args = dict(patch_shape=patch, stride_shape=stride, halo_shape=halo, batch_size=3, mean=1.0, std=3.0)
    got = P.predict_volume(model, raw, device="cpu", **args)
    ref = porc.standard_predict(model, raw, patch, stride, halo, 3, mean=1.0, std=3.0)
    assert got.shape == ref.shape == (cout,) + shape
    assert np.abs(got - ref).max() < 1e-6
    seg = P.predict_volume(model, raw, device="cpu", save_segmentation=True, **args)
    seg_ref = porc.standard_predict(model, raw, patch, stride, halo, 3, mean=1.0, std=3.0, save_segmentation=True)
    assert seg.dtype == np.uint16 and np.array_equal(seg, seg_ref)
    ch = P.predict_volume(model, raw, device="cpu", prediction_channel=cout - 1, **args)
    assert ch.shape == (1,) + shape and np.abs(ch[0] - ref[cout - 1]).max() < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("name,cin,cout,halo", [("UNet3D", 1, 1, (4, 8, 8)), ("ResidualUNetSE3D", 3, 2, (2, 4, 4)),
                                                ("ResidualUNet3D", 1, 1, (0, 0, 0))])
def test_predict_volume_native_matches_reference_loop(name, cin, cout, halo):
    """the native forward inside the device-resident loop vs the reference loop running the same weights on the CPU"""
    from pytorch3dunet_amd import _native as nat

    model = _tiny_model(name, cin, cout, **(dict(final_sigmoid=False) if cout > 1 else {}))
    rng = np.random.default_rng(2)
    shape = (20, 45, 52)
    raw = rng.standard_normal(shape if cin == 1 else (cin,) + shape).astype(np.float32)
    patch, stride = (8, 16, 24), (6, 12, 17)
    ref = porc.standard_predict(model, raw, patch, stride, halo, 2)
    dev = torch.device("cuda", 0)
    gmodel = get_model(dict(name=name, in_channels=cin, out_channels=cout, f_maps=[8, 16], num_groups=4,
                            **(dict(final_sigmoid=False) if cout > 1 else {})))
    gmodel.load_state_dict(model.state_dict())
    gmodel = gmodel.to(dev)
    n0 = nat.launch_count
    got = P.predict_volume(gmodel, raw, patch, stride, halo, batch_size=2)
    assert nat.launch_count > n0, "native path did not run"
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() < 1e-3 * max(1.0, np.abs(ref).max())
    seg = P.predict_volume(gmodel, raw, patch, stride, halo, batch_size=2, save_segmentation=True)
    seg_ref = porc.standard_predict(model, raw, patch, stride, halo, 2, save_segmentation=True)
    assert (seg != seg_ref).mean() < 1e-4  # only probabilities within round-off of the decision threshold may differ


# ---- the predictor CLASSES (reference protocol: predictor.py:24-283, looked up by name in predict.py:20-40) -------------
class _MemTestDataset(torch.utils.data.Dataset):
    """what StandardPredictor needs from an AbstractHDF5Dataset in the test phase (datasets/hdf5.py:70-113,154-173): file_path,
    volume_shape, halo_shape and (standardised padded patch, unpadded spatial index) items"""

    def __init__(self, raw, patch, stride, halo, file_path):
        self.file_path = file_path
        self.halo_shape = tuple(halo)
        vol = raw if raw.ndim == 4 else raw[None]
        self.volume_shape = vol.shape[1:]
        self.mean, self.std = float(raw.mean()), float(raw.std())
        self.padded = np.pad(vol, [(0, 0)] + [(p, p) for p in halo], mode="reflect") if any(halo) else vol
        self.slices = porc.build_slices(self.volume_shape, patch, stride)

    def __len__(self):
        return len(self.slices)

    def __getitem__(self, i):
        sz, sy, sx = self.slices[i]
        h = self.halo_shape
        m = self.padded[:, sz.start:sz.stop + 2 * h[0], sy.start:sy.stop + 2 * h[1], sx.start:sx.stop + 2 * h[2]]
        m = (m - self.mean) / np.clip(self.std, a_min=1e-10, a_max=None)
        return torch.from_numpy(m.astype(np.float32)), (sz, sy, sx)


def _collate(batch):  # default_prediction_collate (datasets/utils.py:478-496): stack the tensors, keep the slice tuples as a list
    return torch.stack([b[0] for b in batch], 0), [b[1] for b in batch]


def _run_predictor_classes(device, model_name, cin, cout, halo, kw, atol):
    import fake_h5py
    from pytorch3dunet_amd.unet3d import predictor as MP

    fake_h5py.install()
    torch.manual_seed(0)
    rng = np.random.RandomState(1)
    model = get_model(dict(name=model_name, in_channels=cin, out_channels=cout, f_maps=[8, 16], num_groups=4,
                           final_sigmoid=(cout == 1))).eval()
    shape, patch, stride = (20, 33, 29), (8, 16, 16), (6, 12, 12)
    raw = (rng.randn(*((cin,) + shape if cin > 1 else shape)) * 2 + 0.5).astype(np.float32)
    ds = _MemTestDataset(raw, patch, stride, halo, "/mem/in/vol_a.h5")
    loader = torch.utils.data.DataLoader(ds, batch_size=3, collate_fn=_collate)
    expect = porc.standard_predict(model, raw, patch, stride, halo, batch_size=3, mean=ds.mean, std=ds.std, **kw)
    model = model.to(device)
    for cls in (MP.StandardPredictor, MP.LazyPredictor):
        out_dir = f"/mem/out_{cls.__name__}"
        res = cls(model, out_dir, cout, device, output_dataset="predictions", **kw)(loader)
        assert res is None
        got = fake_h5py.STORE[f"{out_dir}/vol_a_predictions.h5"]["predictions"]
        assert got.shape == expect.shape and got.dtype == expect.dtype
        if kw.get("save_segmentation"):
            assert (got != expect).mean() <= (0.0 if atol == 0 else 1e-3)  # a class flips only where two probabilities tie
        else:
            assert np.abs(got - expect).max() <= atol
    assert not model.training


@pytest.mark.parametrize("cin,cout,halo,kw", [(1, 2, (2, 4, 4), {}), (2, 3, (0, 0, 0), {"prediction_channel": 2}),
                                              (1, 1, (2, 2, 2), {"save_segmentation": True}),
                                              (2, 3, (1, 2, 3), {"save_segmentation": True})])
def test_predictor_classes_cpu_equal_reference_loop(cin, cout, halo, kw):
    _run_predictor_classes("cpu", "UNet3D", cin, cout, halo, kw, atol=0)


@pytest.mark.gpu
@pytest.mark.parametrize("name,cin,cout,halo,kw", [("UNet3D", 1, 2, (2, 4, 4), {}), ("ResidualUNetSE3D", 3, 3, (0, 0, 0), {"prediction_channel": 1}),
                                                   ("ResidualUNet3D", 1, 1, (2, 2, 2), {"save_segmentation": True})])
def test_predictor_classes_native_forward_on_device(name, cin, cout, halo, kw):
    """the same classes with the model on the MI355X: native forward inside the loop, volume assembled in HBM (Standard) or
    streamed through pinned buffers (Lazy); against the host loop with the torch.nn module tree"""
    from pytorch3dunet_amd import _native as nat

    n0 = nat.launch_count
    _run_predictor_classes("cuda", name, cin, cout, halo, kw, atol=2e-5)
    assert nat.launch_count > n0


@pytest.mark.skipif(not reference_available(), reason="needs /root/reference (build container only)")
@pytest.mark.timeout(300)
def test_drop_in_predictors_equal_the_unmodified_reference_predictors():
    """tests/drive_reference_predictor.py: reference StandardPredictor / LazyPredictor == ours == the oracle's loop, bit for bit
    (probabilities, prediction_channel, multi- and single-channel segmentation, the dice metric branch)"""
    import json
    import os
    import subprocess
    import sys

    from conftest import ROOT

    proc = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "drive_reference_predictor.py")], capture_output=True,
                          text=True, timeout=280, env=dict(os.environ, OMP_NUM_THREADS="4"))
    assert proc.returncode == 0, proc.stderr[-3000:]
    r = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    for case in ("probs", "channel", "segm_multi", "segm_single"):
        assert all(r[case][k] for k in ("ref_standard", "ref_lazy", "our_standard", "our_lazy", "oracle")), (case, r[case])
    assert r["dice_equal"] and r["names"]
