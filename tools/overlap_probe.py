"""Single-GPU probe of the gradient-exchange overlap (VERDICT r02 "weak" 10 / task 7): do kernels of ANOTHER HIP stream make
progress beside the persistent convolution kernels of the encoder backward, or only at kernel boundaries / after them?

RCCL itself cannot be exercised with one rank (a 1-rank all-reduce in place launches nothing), so the exchange is replaced by a
STAND-IN on the same hook (`engine.grad_sync.launch(bucket)` right after the decoder backward, `finish()` at the end): `passes`
streaming passes over the bucket on a side stream, sized to the time a ring all-reduce of that bucket takes on one 153 GB/s xGMI
link (2 * 7/8 * bytes / 153e9).  Reported per model: step time without exchange, with the stand-in on the side stream
(overlapped), with the stand-in on the compute stream (serialised), and the stand-in's standalone duration.

    python tools/overlap_probe.py            # BASELINE configs 2 and 4
    python tools/overlap_probe.py --reserve 8 [--only 4]

--reserve k (round 4): the step runs on a stream CU-masked to all but k CUs (`parallel.reserve_cus`: the library sizes its
persistent grids for 256 - k) and the stand-in runs on a stream confined to exactly those k CUs — RCCL's kernels are link-bound and
need few CUs, so the stand-in is re-sized on ITS CUs: as many passes as take the ring all-reduce's time there.  Reported in
addition: the step time without any exchange at 256 CUs, i.e. what the budget itself costs.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-3dunet_amd"))
import torch  # noqa: E402

from pytorch3dunet_amd.unet3d.losses import BCEDiceLoss  # noqa: E402
from pytorch3dunet_amd.unet3d.model import get_model  # noqa: E402

dev = torch.device("cuda", 0)


class StandIn:
    def __init__(self, side: bool, gbps: float = 153.0, side_stream=None):
        self.side = (side_stream or torch.cuda.Stream(dev)) if side else None
        self.gbps = gbps
        self.pending = False
        self.ms_target = 0.0
        self.pass_tbps = 4.0  # in-place pass rate on the stream the stand-in runs on (calibrate() measures it on a CU-masked stream)

    def calibrate(self, bucket):
        """measured rate of one in-place pass over `bucket` on the stand-in's own stream (a few reserved CUs stream far below 4 TB/s)"""
        s = self.side or torch.cuda.current_stream(dev)
        with torch.cuda.stream(s):
            for _ in range(2):
                bucket.mul_(1.0)
            s.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                bucket.mul_(1.0)
            s.synchronize()
            dt = (time.perf_counter() - t0) / 4
        self.pass_tbps = 2 * bucket.numel() * 4 / dt / 1e12
        return self.pass_tbps

    def passes_for(self, bucket):
        # one in-place pass moves 2 x bytes at pass_tbps; the exchange takes 2 * 7/8 * bytes / link rate
        t_x = 2 * 7 / 8 * bucket.numel() * 4 / (self.gbps * 1e9)
        t_pass = 2 * bucket.numel() * 4 / (self.pass_tbps * 1e12)
        return max(1, int(round(t_x / t_pass)))

    def launch(self, bucket):
        if bucket.numel() == 0:
            return
        n = self.passes_for(bucket)
        if self.side is None:
            for _ in range(n):
                bucket.mul_(1.0)
            return
        self.side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self.side):
            for _ in range(n):
                bucket.mul_(1.0)
        self.pending = True

    def finish(self):
        if self.pending:
            torch.cuda.current_stream(dev).wait_stream(self.side)
            self.pending = False


def run(name, cfg, shape, steps=8, reserve=0):
    from pytorch3dunet_amd import _native as nat
    from pytorch3dunet_amd import parallel

    torch.manual_seed(0)
    model = get_model(cfg).to(dev).train()
    x = torch.randn(shape, device=dev)
    t = (torch.rand(shape, device=dev) > 0.5).float()
    crit = BCEDiceLoss()
    eng = model._get_engine()

    def step():
        model.zero_grad(set_to_none=True)
        _, logits = model(x, return_logits=True)
        crit(logits, t).backward()

    def timed(mode, side_stream=None, tbps=None):
        eng.grad_sync = None if mode == "none" else StandIn(side=(mode == "side"), side_stream=side_stream)
        if tbps is not None and eng.grad_sync is not None:
            eng.grad_sync.pass_tbps = tbps
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / steps

    out = {}
    full = timed("none")  # all CUs, no exchange: the reference point for what a budget costs
    comp = res = None
    tbps = None
    default_stream = torch.cuda.current_stream(dev)
    if reserve > 0:
        comp, res = parallel.reserve_cus(dev, reserve)  # becomes the current stream; tuning key 12 = reserve
        probe = StandIn(side=True, side_stream=res)
        n_enc = eng.n_enc_params
        tbps = probe.calibrate(torch.zeros(eng.n_params - n_enc, device=dev))
    for mode in ("none", "side", "serial"):
        out[mode] = timed(mode, side_stream=res, tbps=tbps if mode == "side" else None)
    eng.grad_sync = None
    n_enc = eng.n_enc_params
    flat = torch.zeros(eng.n_params, device=dev)
    s = StandIn(side=res is not None, side_stream=res)
    if tbps is not None:
        s.pass_tbps = tbps
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        s.launch(flat[n_enc:])
        s.launch(flat[:n_enc])
        s.finish()
    torch.cuda.synchronize()
    alone = 1e3 * (time.perf_counter() - t0) / 5
    if reserve > 0:
        # with a budget the stand-in's cost when NOT hidden is its standalone duration on its own CUs
        hidden = 1.0 - (out["side"] - out["none"]) / max(alone, 1e-9)
        print(f"{name} [reserve {reserve} CUs]: step {full:.2f} ms on all CUs without exchange -> {out['none']:.2f} ms on {256 - reserve} CUs "
              f"(the budget costs {100 * (out['none'] / full - 1):.1f} %); with the stand-in on the {reserve} reserved CUs {out['side']:.2f} ms; "
              f"stand-in alone on those CUs {alone:.2f} ms at {tbps:.2f} TB/s per pass -> {100 * hidden:.0f} % of it hidden; "
              f"net vs unhidden exchange on all CUs ({full + alone:.2f} ms): {out['side']:.2f} ms", flush=True)
        torch.cuda.set_stream(default_stream)
        nat.call("u3d_set_tuning", 12, 0)
        torch.cuda.synchronize()
        for st in (comp, res):
            if st is not None:
                nat.call("u3d_stream_destroy", dev.index, st.cuda_stream)
        return
    hidden = (out["serial"] - out["side"]) / max(out["serial"] - out["none"], 1e-9)
    print(f"{name}: params {eng.n_params / 1e6:.1f} M (decoder+head bucket {(eng.n_params - n_enc) * 4 / 1e6:.0f} MB, encoder bucket "
          f"{n_enc * 4 / 1e6:.0f} MB); step {out['none']:.2f} ms without exchange, {out['side']:.2f} ms with the stand-in on a side stream, "
          f"{out['serial']:.2f} ms serialised; stand-in alone {alone:.2f} ms -> {100 * hidden:.0f} % of its cost hidden", flush=True)


if __name__ == "__main__":
    import argparse

    ap = argparse.ArgumentParser()
    ap.add_argument("--reserve", type=int, nargs="*", default=[0])
    ap.add_argument("--only", type=int, default=0, help="2 or 4: just that BASELINE config")
    a = ap.parse_args()
    for r in a.reserve:
        if a.only in (0, 2):
            run("config 2 (UNet3D f_maps=32, 2x1x64x128x128, fp32)",
                dict(name="UNet3D", in_channels=1, out_channels=1, f_maps=32, num_groups=8, final_sigmoid=True), (2, 1, 64, 128, 128),
                reserve=r)
        if a.only in (0, 4):
            run("config 4 (ResidualUNet3D f_maps=64, 1x80x160x160, bf16)",
                dict(name="ResidualUNet3D", in_channels=1, out_channels=1, f_maps=64, num_groups=8, final_sigmoid=True,
                     compute_dtype="bf16"), (1, 1, 80, 160, 160), reserve=r)
