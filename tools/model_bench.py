"""Time forward+backward of any model configuration on the native path, with the per-entry-point HIP-event breakdown
bench.py uses for its roofline leg.

    python tools/model_bench.py --name ResidualUNet3D --f-maps 64 --levels 5 --patch 80,160,160 --batch 1 --steps 5
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
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = get_model(dict(name=args.name, in_channels=args.in_channels, out_channels=args.out_channels, f_maps=args.f_maps,
                           num_levels=args.levels, layer_order="gcr", num_groups=8, final_sigmoid=True)).to(dev)
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
    print(json.dumps({"model": args.name, "f_maps": args.f_maps, "levels": args.levels, "patch": [D, H, W], "batch": args.batch,
                      "mode": "fwd" if args.forward_only else "fwd+bwd", "ms_per_step": round(dt * 1e3, 2),
                      "patches_per_s": round(args.batch / dt, 3), "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2**30, 2),
                      "families": fams}))


if __name__ == "__main__":
    main()
