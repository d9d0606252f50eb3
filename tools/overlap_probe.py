"""Single-GPU probe of the gradient-exchange overlap (VERDICT r02 "weak" 10 / task 7): do kernels of ANOTHER HIP
stream make progress beside the backward, or only after it?

RCCL itself cannot be exercised with one rank (a 1-rank all-reduce in place launches nothing), so the exchange is replaced by a
STAND-IN on the same hooks (`engine.grad_sync.launch(bucket)` wherever the engine hands a final gradient bucket to the exchange,
`finish()` at the end), on a side stream.  Two stand-ins:

  --standin torch   (round 3) chip-wide streaming passes (`bucket.mul_(1)`, thousands of workgroups at HBM rate), as many as take the
                    time of a ring all-reduce of the bucket on one 153 GB/s xGMI link.  Pessimistic: it competes for HBM and for
                    every CU, which a link-bound collective does not.
  --standin link    (round 4, default) `u3d_debug_stream_pass`: 16 workgroups — the shape of RCCL's ring kernels, one per channel —
                    streaming the bucket once in place and PACED against the wall clock so that the launch lasts the ring
                    all-reduce's time (2 * 7/8 * bytes / 153 GB/s) however much bandwidth it gets: like the collective it needs few CUs
                    and little HBM bandwidth and does not finish early.

--slots k: the persistent convolution grids leave k block slots free (parallel.cu_budget, tuning key 12).  Reported per model: step
time without exchange (and without / with the budget), with the stand-in on the side stream, with the stand-in serialised on the
compute stream, the stand-in's standalone duration, and the share of it that was hidden.

    python tools/overlap_probe.py [--only 2|4] [--slots 0 16 32] [--standin link|torch]
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
    """engine.grad_sync stand-in: every launch(bucket) issues a kernel sequence that lasts as long as the ring all-reduce of that
    bucket on one xGMI link; finish() makes the compute stream wait for them"""

    BLOCKS = 16  # workgroups of the link-rate stand-in (RCCL: one per channel)

    def __init__(self, side: bool, kind: str = "link", gbps: float = 153.0):
        self.side = torch.cuda.Stream(dev) if side else None
        self.kind, self.gbps = kind, gbps
        self.pending = False
        self.launched = 0
        self.bytes_per_ms = None  # link stand-in: measured in-place streaming rate of BLOCKS workgroups on the idle GPU

    def calibrate(self):
        return None  # (the link stand-in paces itself against the wall clock: nothing to calibrate)

    def _issue(self, bucket, stream):
        from pytorch3dunet_amd import _native as nat

        t_x = 2 * 7 / 8 * bucket.numel() * 4 / (self.gbps * 1e9)  # ring all-reduce of this bucket on one xGMI link, seconds
        if self.kind == "torch":
            n = max(1, int(round(t_x / (2 * bucket.numel() * 4 / 4.0e12))))
            for _ in range(n):
                bucket.mul_(1.0)
            return
        off = (-(bucket.data_ptr() // 4)) % 4  # 16-byte alignment of the slice
        n4 = (bucket.numel() - off) // 4 * 4
        view = bucket[off : off + n4]
        nat.call("u3d_debug_stream_pass", dev.index, stream.cuda_stream, view.data_ptr(), view.numel(), self.BLOCKS, 1, t_x)

    def launch(self, bucket):
        if bucket.numel() < 16:
            return
        self.launched += 1
        if self.side is None:
            self._issue(bucket, torch.cuda.current_stream(dev))
            return
        self.side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self.side):
            self._issue(bucket, self.side)
        self.pending = True

    def finish(self):
        if self.pending:
            torch.cuda.current_stream(dev).wait_stream(self.side)
            self.pending = False


def run(name, cfg, shape, steps=8, slots=0, kind="link"):
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

    def timed(mode):
        sync = None if mode == "none" else StandIn(side=(mode == "side"), kind=kind)
        eng.grad_sync = sync
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / steps, (sync.launched // (steps + 3) if sync else 0)

    parallel.cu_budget(0)
    full, _ = timed("none")
    parallel.cu_budget(slots)
    out = {m: timed(m) for m in ("none", "side", "serial")}
    eng.grad_sync = None
    parallel.cu_budget(0)
    n_enc = eng.n_enc_params
    alone = out["serial"][0] - out["none"][0]  # the stand-in's own duration = what serialising it adds
    hidden = (out["serial"][0] - out["side"][0]) / max(alone, 1e-9)
    print(f"{name} [{kind} stand-in, {slots} free slots]: params {eng.n_params / 1e6:.1f} M (decoder+head {(eng.n_params - n_enc) * 4 / 1e6:.0f} MB, "
          f"encoders {n_enc * 4 / 1e6:.0f} MB in {out['side'][1] - 1} level buckets); step {full:.2f} ms without exchange"
          + (f" ({out['none'][0]:.2f} ms with the budget: {100 * (out['none'][0] / full - 1):+.1f} %)" if slots else "")
          + f", {out['side'][0]:.2f} ms with the stand-in on a side stream, {out['serial'][0]:.2f} ms serialised (stand-in {alone:.2f} ms) -> "
          f"{100 * hidden:.0f} % of the exchange hidden; net cost of the exchange {out['side'][0] - full:.2f} ms = "
          f"{100 * (out['side'][0] / full - 1):.1f} % of the step", flush=True)


if __name__ == "__main__":
    import argparse

    ap = argparse.ArgumentParser()
    ap.add_argument("--slots", type=int, nargs="*", default=[0])
    ap.add_argument("--only", type=int, default=0, help="2 or 4: just that BASELINE config")
    ap.add_argument("--standin", default="link", choices=["link", "torch"])
    a = ap.parse_args()
    for k in a.slots:
        if a.only in (0, 2):
            run("config 2 (UNet3D f_maps=32, 2x1x64x128x128, fp32)",
                dict(name="UNet3D", in_channels=1, out_channels=1, f_maps=32, num_groups=8, final_sigmoid=True), (2, 1, 64, 128, 128),
                slots=k, kind=a.standin)
        if a.only in (0, 4):
            run("config 4 (ResidualUNet3D f_maps=64, 1x80x160x160, bf16)",
                dict(name="ResidualUNet3D", in_channels=1, out_channels=1, f_maps=64, num_groups=8, final_sigmoid=True,
                     compute_dtype="bf16"), (1, 1, 80, 160, 160), slots=k, kind=a.standin)
