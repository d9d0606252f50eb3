"""Where the peak of a training step's device memory sits: torch.cuda.memory_allocated() sampled at every C-ABI call of one
forward + backward (BASELINE config 4 by default), printed as the running maximum with the entry point that set it.

    python tools/mem_trace.py [--checkpoint] [--bf16] [--name ResidualUNet3D --f-maps 64 --patch 80,160,160]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-3dunet_amd"))
import torch  # noqa: E402

from pytorch3dunet_amd import _native as nat  # noqa: E402
from pytorch3dunet_amd.unet3d.losses import BCEDiceLoss  # noqa: E402
from pytorch3dunet_amd.unet3d.model import get_model  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--name", default="ResidualUNet3D")
ap.add_argument("--f-maps", type=int, default=64)
ap.add_argument("--levels", type=int, default=5)
ap.add_argument("--patch", default="80,160,160")
ap.add_argument("--bf16", action="store_true")
ap.add_argument("--checkpoint", action="store_true")
ap.add_argument("--act-bf16", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = get_model(dict(name=args.name, in_channels=1, out_channels=1, f_maps=args.f_maps, num_levels=args.levels, layer_order="gcr",
                       num_groups=8, final_sigmoid=True, compute_dtype="bf16" if args.bf16 else "fp32",
                       checkpoint_encoders=args.checkpoint, activation_dtype="bf16" if args.act_bf16 else "fp32")).to(dev).train()
D, H, W = (int(v) for v in args.patch.split(","))
x = torch.randn(1, 1, D, H, W, device=dev)
t = (torch.rand(1, 1, D, H, W, device=dev) > 0.5).float()
crit = BCEDiceLoss()


def step():
    model.zero_grad(set_to_none=True)
    _, logits = model(x, return_logits=True)
    crit(logits, t).backward()


step()
torch.cuda.synchronize()
trace = []
orig = nat.call


def traced(name, *a, **k):
    trace.append((name, torch.cuda.memory_allocated()))
    return orig(name, *a, **k)


nat.call = traced
import pytorch3dunet_amd.engine as E  # noqa: E402

E.nat.call = traced
base = torch.cuda.memory_allocated()
torch.cuda.reset_peak_memory_stats()
step()
torch.cuda.synchronize()
print(f"resident before the step {base / 2**30:.3f} GiB, peak {torch.cuda.max_memory_allocated() / 2**30:.3f} GiB, {len(trace)} calls")
run = 0
for i, (name, m) in enumerate(trace):
    if m > run:
        run = m
        print(f"  call {i:4d} {name:32s} {m / 2**30:.3f} GiB  (new maximum)")
peak_i = max(range(len(trace)), key=lambda i: trace[i][1])
print(f"live device tensors at the peak (call {peak_i} {trace[peak_i][0]}), one line per storage >= 32 MiB:")
import gc  # noqa: E402

trace2 = []


def dumping(name, *a, **k):
    trace2.append(name)
    if len(trace2) - 1 == peak_i:
        seen = {}
        for o in gc.get_objects():
            try:
                if torch.is_tensor(o) and o.is_cuda:
                    st = o.untyped_storage()
                    if st.nbytes() >= 32 << 20 and st.data_ptr() not in seen:
                        who = []
                        for r in gc.get_referrers(o):
                            if isinstance(r, dict):
                                owners = [type(x).__name__ for x in gc.get_referrers(r) if hasattr(x, "__dict__") and x.__dict__ is r]
                                keys = [k for k, v in r.items() if v is o]
                                who.append(f"{'/'.join(owners) or 'dict'}.{','.join(map(str, keys))}")
                            elif isinstance(r, (list, tuple)):
                                who.append(type(r).__name__)
                            else:
                                who.append(type(r).__name__)
                        seen[st.data_ptr()] = (st.nbytes(), tuple(o.shape), str(o.dtype) + "  <- " + "; ".join(sorted(set(who))[:6]))
            except Exception:
                pass
        for nb, shp, dt in sorted(seen.values(), reverse=True):
            print(f"    {nb / 2**20:9.1f} MiB  {shp} {dt}")
        print(f"    sum {sum(v[0] for v in seen.values()) / 2**30:.3f} GiB of {torch.cuda.memory_allocated() / 2**30:.3f} GiB allocated")
    return orig(name, *a, **k)


nat.call = dumping
E.nat.call = dumping
step()
torch.cuda.synchronize()
print("every 10th call:")
for i, (name, m) in enumerate(trace):
    if i % 10 == 0:
        print(f"  call {i:4d} {name:32s} {m / 2**30:.3f} GiB")
