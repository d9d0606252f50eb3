"""GPU diagnostic: per-layer deviation of (a) our HIP path and (b) the fp32 CPU oracle from an fp64 CPU oracle.
Prints max-abs error / max-abs truth for every SingleConv's output y, gradient wrt conv output (dz), gradient wrt
GroupNorm output (dg), weight gradient, GroupNorm gamma/beta gradients."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("oracle", "pytorch-3dunet_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import torch.nn.functional as F
import unet3d_oracle as orc
from pytorch3dunet_amd.unet3d.model import UNet3D


def trace(sd, x, target, G, dtype):
    sd = {k: v.to(dtype).clone().requires_grad_(True) for k, v in sd.items()}
    x = x.to(dtype); target = target.to(dtype)
    acts = {}
    def sc(x, pfx, name):
        g = F.group_norm(x, orc.groups_for(x.shape[1], G), sd[pfx + ".groupnorm.weight"], sd[pfx + ".groupnorm.bias"], 1e-5)
        g.retain_grad()
        z = F.conv3d(g, sd[pfx + ".conv.weight"], None, padding=1)
        z.retain_grad()
        y = F.relu(z)
        acts[name] = (g, z, y)
        return y
    n_enc, n_dec = orc._count(sd, "encoders"), orc._count(sd, "decoders")
    feats = []
    for i in range(n_enc):
        if i > 0:
            x = F.max_pool3d(x, 2)
        x = sc(x, f"encoders.{i}.basic_module.SingleConv1", f"enc{i}.c1")
        x = sc(x, f"encoders.{i}.basic_module.SingleConv2", f"enc{i}.c2")
        feats.insert(0, x)
    feats = feats[1:]
    for j in range(n_dec):
        x = torch.cat((feats[j], F.interpolate(x, size=feats[j].shape[2:], mode="nearest")), 1)
        x = sc(x, f"decoders.{j}.basic_module.SingleConv1", f"dec{j}.c1")
        x = sc(x, f"decoders.{j}.basic_module.SingleConv2", f"dec{j}.c2")
    logits = F.conv3d(x, sd["final_conv.weight"], sd["final_conv.bias"])
    loss = orc.bce_dice_loss(logits, target)
    loss.backward()
    out = {}
    for name, (g, z, y) in acts.items():
        out[name + ".y"] = y.detach()
        out[name + ".dz"] = z.grad
        out[name + ".dg"] = g.grad
    grads = {k: v.grad for k, v in sd.items()}
    return out, grads, logits.detach()


def rel(a, b):
    d = b.abs().max().item()
    return (a.double() - b.double()).abs().max().item() / (d if d > 0 else 1)


def main():
    f_maps = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    shape = tuple(int(v) for v in sys.argv[2].split("x")) if len(sys.argv) > 2 else (1, 1, 16, 32, 32)
    torch.manual_seed(0)
    model = UNet3D(1, 1, f_maps=f_maps, num_groups=8)
    x = torch.randn(shape)
    target = (torch.rand(shape) > 0.5).float()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    t64, g64, l64 = trace(sd, x, target, 8, torch.float64)
    t32, g32, l32 = trace(sd, x, target, 8, torch.float32)
    dev = torch.device("cuda", 0)
    model = model.to(dev).train()
    eng = model._get_engine()
    eng.debug = {}
    for fused in (True, False):
        eng.fused_stats = fused
        eng.debug = {}
        model.zero_grad()
        probs, logits = model(x.to(dev), return_logits=True)
        tape_ref = None
        loss = orc.bce_dice_loss(logits, target.to(dev))
        # grab forward activations from the autograd node's tape before backward frees it
        node = logits.grad_fn
        loss.backward()
        torch.cuda.synchronize()
        print(f"==== fused_stats={fused}  f_maps={f_maps} shape={shape}: logits ours-vs-f64 {rel(logits.detach().cpu(), l64):.2e}, f32-vs-f64 {rel(l32, l64):.2e}")
        print(f"{'tensor':24s} {'ours-vs-f64':>12s} {'ref32-vs-f64':>12s} {'ours-vs-ref32':>13s}")
        ncdhw = lambda t: t.permute(0, 4, 1, 2, 3).contiguous().cpu()
        for k in t64:
            if k.endswith(".y"):
                continue
            if k in eng.debug:
                o = ncdhw(eng.debug[k])
                print(f"{k:24s} {rel(o, t64[k]):12.2e} {rel(t32[k], t64[k]):12.2e} {rel(o, t32[k]):13.2e}")
        for k, p in model.named_parameters():
            o = p.grad.detach().cpu()
            flag = "" if orc.grad_within_tolerance(o, g32[k], (g32[k].double() - g64[k]).abs().max().item()) else "  <-- FAIL"
            print(f"{k:58s} {rel(o, g64[k]):10.2e} {rel(g32[k], g64[k]):10.2e} {rel(o, g32[k]):10.2e}{flag}")


if __name__ == "__main__":
    main()
