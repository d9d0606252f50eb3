"""Why golden g11 (ResidualUNetSE3D, config 5's channel ladder) misses the per-parameter flip band in `compute_dtype: fp32_split`
while it passes on the default fp32-MFMA path ("which parameter, which reduction").

For both compute modes, on the fixture's own weights / input / target:
  1. the sampled per-parameter ratio |ours - ref32| / max(1e-3 * absmax, ref_err) of tests/test_gpu_model.py, worst parameters by name;
  2. the ReLU masks / pool arg-maxes the step took against the fp32 oracle's OWN decisions, layer by layer (count, |pre-activation| of
     the flipped elements relative to the layer's range);
  3. the float64 oracle with THIS run's decisions imposed (oracle.forward_backward_decided): per-parameter relative error — what is
     left when the discrete decisions are taken out of the comparison, i.e. the arithmetic error of the kernels themselves.

    python tools/diag_split_golden.py [--name g11_resunetse3d_in3_ladder] [--cpu-only]   ->  one JSON line per mode
(--cpu-only: time the oracle part without a GPU, with the oracle's own decisions)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("pytorch-3dunet_amd", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--name", default="g11_resunetse3d_in3_ladder")
    ap.add_argument("--cpu-only", action="store_true")
    args = ap.parse_args()
    import unet3d_oracle as orc
    from conftest import Golden

    g = Golden(args.name)
    x, target = g.inputs()
    sd = {k: v.detach().clone() for k, v in g.build_model().state_dict().items()}
    G, fs, seg = g.cfg.get("num_groups", 8), g.cfg.get("final_sigmoid", True), g.cfg.get("is_segmentation", True)
    t0 = time.time()
    trace, _ = orc.forward_decisions(sd, x, G, fs, seg)
    t_dec = time.time() - t0
    if args.cpu_only:
        masks = [z > 0 for z in trace["pre"]]
        import torch.nn.functional as F

        ams = []
        for h in trace["pool"]:
            _, idx = F.max_pool3d(h, 2, return_indices=True)
            Hh, Ww = h.shape[3:]
            ams.append((((idx // (Hh * Ww)) % 2) * 4 + (((idx // Ww) % Hh) % 2) * 2 + (idx % Ww) % 2).to(torch.uint8))
        t0 = time.time()
        orc.forward_backward_decided(sd, x, target, masks, ams, G, fs, seg, g.loss_name)
        print(json.dumps({"forward_decisions_s": round(t_dec, 1), "forward_backward_decided_s": round(time.time() - t0, 1)}))
        return
    import test_gpu_model as tm
    from pytorch3dunet_amd.unet3d.model import get_model

    for mode in ("fp32", "fp32_split"):
        model = get_model(dict(g.cfg, compute_dtype=mode))
        model.load_state_dict(sd)
        dec = {}
        probs, logits, loss, grads = tm._run_native(model, x, target, g.loss_name, dec)
        s = g.sample
        # 1. the golden's sampled per-parameter ratio
        ratios = []
        for k, rs in g.group("grad_s/").items():
            am, re = float(g.z["grad_absmax/" + k]), float(g.z["ref_err/" + k])
            err = (grads[k].flatten()[::s].double() - rs.double()).abs().max().item()
            ratios.append((err / max(tm.REL * am, re, 1e-30), k, err, am, re))
        ratios.sort(reverse=True)
        # 2. decisions against the fp32 oracle's own
        layers, flips = [], 0
        for name, ours, z in zip(dec["names"], dec["masks"], trace["pre"]):
            diff = ours != (z > 0)
            n = int(diff.sum())
            flips += n
            if n:
                layers.append({"layer": name, "flips": n, "of": z.numel(), "max_flipped_preact_rel": float(z[diff].abs().max() / z.abs().max())})
        import torch.nn.functional as F

        pool_flips = 0
        for ours, h in zip(dec["argmax"], trace["pool"]):
            _, idx = F.max_pool3d(h, 2, return_indices=True)
            Hh, Ww = h.shape[3:]
            theirs = ((idx // (Hh * Ww)) % 2) * 4 + (((idx // Ww) % Hh) % 2) * 2 + (idx % Ww) % 2
            pool_flips += int((ours.long() != theirs).sum())
        # 3. float64 with OUR decisions imposed
        _, _, g64 = orc.forward_backward_decided(sd, x, target, dec["masks"], dec["argmax"], G, fs, seg, g.loss_name)
        dc = sorted(((orc.rel_err(v.double(), g64[k]), k) for k, v in grads.items()), reverse=True)
        print(json.dumps({
            "golden": args.name, "compute_dtype": mode, "loss": loss, "loss_ref": g.loss,
            "worst_sampled_ratio": [{"ratio": round(r, 2), "param": k, "err": e, "absmax": a, "ref_err": re} for r, k, e, a, re in ratios[:5]],
            "params_over_factor_4": sum(1 for r in ratios if r[2] > max(tm.REL * r[3], tm.GRAD_FLIP_FACTOR * r[4])),
            "relu_flips_vs_fp32_oracle": flips, "relu_decisions": sum(z.numel() for z in trace["pre"]), "pool_flips": pool_flips,
            "flipped_layers": layers,
            "decision_consistent_fp64_worst": [{"rel_err": e, "param": k} for e, k in dc[:5]],
        }), flush=True)


if __name__ == "__main__":
    main()
