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
