"""Sliding-window inference throughput of pytorch3dunet_amd.predictor.predict_volume (BASELINE config 5 style:
ResidualUNetSE3D, multi-channel (3,96,192,192) volume) — whole-volume wall time including the H2D / D2H trips.

    python tools/predict_bench.py [--name ResidualUNetSE3D --f-maps 64 --levels 5 --in-channels 3 --volume 96,192,192
                                   --patch 48,96,96 --stride 48,96,96 --halo 8,16,16 --batch 2 --reps 3]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-3dunet_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pytorch3dunet_amd import _native as nat  # noqa: E402
from pytorch3dunet_amd.predictor import build_slices, predict_volume  # noqa: E402
from pytorch3dunet_amd.unet3d.model import get_model  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--name", default="ResidualUNetSE3D")
    ap.add_argument("--f-maps", type=int, default=64)
    ap.add_argument("--levels", type=int, default=5)
    ap.add_argument("--in-channels", type=int, default=3)
    ap.add_argument("--out-channels", type=int, default=1)
    ap.add_argument("--volume", default="96,192,192")
    ap.add_argument("--patch", default="48,96,96")
    ap.add_argument("--stride", default="48,96,96")
    ap.add_argument("--halo", default="8,16,16")
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    t3 = lambda s: tuple(int(v) for v in s.split(","))  # noqa: E731
    vol, patch, stride, halo = t3(args.volume), t3(args.patch), t3(args.stride), t3(args.halo)
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = get_model(dict(name=args.name, in_channels=args.in_channels, out_channels=args.out_channels, f_maps=args.f_maps,
                           num_levels=args.levels, layer_order="gcr", num_groups=8, final_sigmoid=True)).to(dev).eval()
    assert model.native_supported, model._native_blockers
    raw = np.random.default_rng(0).standard_normal((args.in_channels,) + vol).astype(np.float32)
    n_patches = len(build_slices(vol, patch, stride))
    predict_volume(model, raw, patch, stride, halo, batch_size=args.batch)  # warm-up
    torch.cuda.synchronize()
    prof = nat.EventProfiler()
    nat.profiler = prof
    times = []
    for _ in range(args.reps):
        t0 = time.perf_counter()
        out = predict_volume(model, raw, patch, stride, halo, batch_size=args.batch)
        times.append(time.perf_counter() - t0)
    nat.profiler = None
    times.sort()
    med = times[len(times) // 2]
    summ = prof.summary()
    fams = {k: round(v["ms"] / args.reps, 3) for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])[:8]}
    # roofline of the dominant MFMA family (HIP events around its C-ABI calls; FLOPs as declared by the executor = algorithmic FLOPs of
    # the launches): fp32 MFMA peak 157.3 TFLOP/s, dense bf16 2500 (MI355X_MICROARCH.md)
    mf = {k: v for k, v in summ.items() if v.get("flops")}
    roof = None
    if mf:
        top = max(mf, key=lambda k: mf[k]["ms"])
        peak = 2500.0 if "bf16" in top or "_t8" in top else 157.3
        tf = mf[top]["flops"] / (mf[top]["ms"] * 1e-3) / 1e12
        tot_fl = sum(v["flops"] for v in mf.values())
        roof = {"bound": "mfma", "kernel": top, "achieved": round(tf, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 3),
                "launches_per_volume": mf[top]["calls"] // args.reps, "ms_per_volume": round(mf[top]["ms"] / args.reps, 3),
                "all_mfma_tflops_over_the_whole_volume_time": round(tot_fl / args.reps / med / 1e12, 1)}
    print(json.dumps({"model": args.name, "f_maps": args.f_maps, "levels": args.levels, "volume": [args.in_channels, *vol],
                      "patch": patch, "stride": stride, "halo": halo, "model_input": [p + 2 * h for p, h in zip(patch, halo)],
                      "patches": n_patches, "batch": args.batch, "seconds_per_volume": round(med, 4),
                      "Mvoxels_per_s": round(vol[0] * vol[1] * vol[2] / med / 1e6, 2), "patches_per_s": round(n_patches / med, 2),
                      "out_shape": list(out.shape), "compute": "bf16" if os.environ.get("U3D_BF16") == "1" else ("fp32_split" if os.environ.get("U3D_F32_SPLIT") == "1" else "fp32"),
                      "roofline": roof, "kernel_ms_per_volume_top": fams}))


if __name__ == "__main__":
    main()
