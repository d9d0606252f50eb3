"""Time forward+backward of any model configuration on the native path, with the per-entry-point HIP-event breakdown
bench.py uses for its roofline leg.

    python tools/model_bench.py --name ResidualUNet3D --f-maps 64 --levels 5 --patch 80,160,160 --batch 1 --steps 5
    python tools/model_bench.py --bf16 --checkpoint      # BASELINE config 4: bf16 MFMA operands, encoder blocks recomputed

With --bf16 the bf16 kernel families get a roofline against BOTH bounds: the dense bf16 MFMA peak (2.5 PFLOP/s) and HBM
(8 TB/s) with their algorithmic bytes (fp32 input + output once, + the forward input re-read by the data gradient; the two
operands once for the weight gradient).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-3dunet_amd"))
import torch  # noqa: E402

from pytorch3dunet_amd import _native as nat  # noqa: E402
from pytorch3dunet_amd.unet3d.losses import BCEDiceLoss  # noqa: E402
from pytorch3dunet_amd.unet3d.model import get_model  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--name", default="ResidualUNet3D")
    ap.add_argument("--f-maps", type=int, default=64)
    ap.add_argument("--levels", type=int, default=5)
    ap.add_argument("--in-channels", type=int, default=1)
    ap.add_argument("--out-channels", type=int, default=1)
    ap.add_argument("--patch", default="80,160,160")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--forward-only", action="store_true")
    ap.add_argument("--bf16", action="store_true", help="compute_dtype='bf16' (fp32 master weights / activations / statistics)")
    ap.add_argument("--split", action="store_true", help="compute_dtype='fp32_split' (fp32-grade convolutions on the bf16 pipe)")
    ap.add_argument("--checkpoint", action="store_true", help="checkpoint_encoders=True")
    ap.add_argument("--checkpoint-levels", type=int, default=0, help="checkpoint_encoders=k: only the k highest-resolution encoder levels")
    ap.add_argument("--no-events", action="store_true", help="bare timing only (for runs under rocprofv3)")
    ap.add_argument("--act-bf16", action="store_true", help="activation_dtype='bf16' (with --bf16: activations and gradients stored as bf16)")
    ap.add_argument("--graph", action="store_true", help="hip_graph=True")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = get_model(dict(name=args.name, in_channels=args.in_channels, out_channels=args.out_channels, f_maps=args.f_maps,
                           num_levels=args.levels, layer_order="gcr", num_groups=8, final_sigmoid=True,
                           compute_dtype="bf16" if args.bf16 else ("fp32_split" if args.split else "fp32"), checkpoint_encoders=bool(args.checkpoint or args.checkpoint_levels > 0), checkpoint_levels=(args.checkpoint_levels if args.checkpoint_levels > 0 else None),
                           activation_dtype="bf16" if args.act_bf16 else "fp32", hip_graph=args.graph)).to(dev)
    assert model.native_supported, model._native_blockers
    D, H, W = (int(v) for v in args.patch.split(","))
    x = torch.randn(args.batch, args.in_channels, D, H, W, device=dev)
    target = (torch.rand(args.batch, args.out_channels, D, H, W, device=dev) > 0.5).float()
    crit = BCEDiceLoss()

    def step():
        if args.forward_only:
            with torch.no_grad():
                return model(x)
        model.zero_grad(set_to_none=True)
        _, logits = model(x, return_logits=True)
        loss = crit(logits, target)
        loss.backward()
        return loss

    model.train(not args.forward_only)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    # bare step time and peak memory first (the event profiler adds ~2 us of device time per call)
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt_bare = (time.perf_counter() - t0) / args.steps
    peak = torch.cuda.max_memory_allocated()
    if args.no_events:
        print(json.dumps({"model": args.name, "activation_dtype": "bf16" if args.act_bf16 else "fp32", "checkpoint_encoders": (args.checkpoint_levels if args.checkpoint_levels > 0 else args.checkpoint),
                          "ms_per_step_bare": round(dt_bare * 1e3, 2), "peak_mem_gb": round(peak / 2**30, 3)}))
        return
    prof = nat.EventProfiler()
    nat.profiler = prof
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    nat.profiler = None
    summ = prof.summary()
    fams = {k: {"ms_per_step": round(v["ms"] / args.steps, 3),
                "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["flops"] and v["ms"] > 0 else None}
            for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])}
    PEAK = {"u3d_conv3d_bf16_ex": 2500.0, "u3d_conv3d_wgrad_bf16": 2500.0, "u3d_conv3d_wgrad_bf16_job": 2500.0, "u3d_conv3d_bf16_ex_b16": 2500.0,
            "u3d_conv3d_wgrad_bf16_b16_job": 2500.0}
    for k, v in summ.items():
        if k in PEAK and v["flops"] and v["ms"] > 0:
            tf = v["flops"] / (v["ms"] * 1e-3) / 1e12
            # algorithmic HBM bytes of these launches: 54*Cin*Cout FLOP per voxel, square layers (Cin == Cout == C) move
            # 4*C (+4*C) bytes in and 4*C out per voxel -> bytes = flops * 8 / (54 * C); reported for C = f_maps (level 0, the
            # most bandwidth-hungry level)
            fams[k]["frac_of_bf16_mfma_peak"] = round(tf / PEAK[k], 3)
            fams[k]["hbm_gbps_algorithmic_level0"] = round(tf * 1e12 * 8.0 / (54.0 * args.f_maps) / 1e9, 1)
    print(json.dumps({"model": args.name, "compute": "bf16" if args.bf16 else ("fp32_split" if args.split else "fp32"), "checkpoint_encoders": (args.checkpoint_levels if args.checkpoint_levels > 0 else args.checkpoint), "activation_dtype": "bf16" if args.act_bf16 else "fp32",
                      "ms_per_step_bare": round(dt_bare * 1e3, 2), "patches_per_s_bare": round(args.batch / dt_bare, 3), "f_maps": args.f_maps, "levels": args.levels, "patch": [D, H, W], "batch": args.batch,
                      "mode": "fwd" if args.forward_only else "fwd+bwd", "ms_per_step": round(dt * 1e3, 2),
                      "patches_per_s": round(args.batch / dt, 3), "peak_mem_gb": round(peak / 2**30, 3),
                      "families": fams}))


if __name__ == "__main__":
    main()
