"""How much of a training step is host (Python + ctypes enqueue) time: enqueue K steps without synchronising, stamp the host clock
when the last call returned, then synchronise.  host_ms_per_step close to step_ms = the GPU waits for the host."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "pytorch-3dunet_amd"))
from pytorch3dunet_amd.unet3d.losses import BCEDiceLoss  # noqa: E402
from pytorch3dunet_amd.unet3d.model import UNet3D  # noqa: E402

dev = torch.device("cuda", 0)
for mode, graph in (("fp32", False), ("fp32", True), ("fp32_split", False), ("fp32_split", True)):
    for f_maps, shape in ((32, (2, 1, 64, 128, 128)), (16, (1, 1, 32, 64, 64))):
        torch.manual_seed(0)
        model = UNet3D(in_channels=1, out_channels=1, f_maps=f_maps, num_groups=8, compute_dtype=mode, hip_graph=graph).to(dev).train()
        opt = torch.optim.Adam(model.parameters(), lr=2e-4)
        x = torch.randn(shape, device=dev)
        t = (torch.rand(shape, device=dev) > 0.5).float()
        crit = BCEDiceLoss()

        def step():
            p, l = model(x, return_logits=True)
            loss = crit(l, t)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        K = 10
        t0 = time.perf_counter()
        for _ in range(K):
            step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        # the model alone (forward + loss + backward, no optimizer): what the step runner itself leaves on the host
        def model_only():
            p, l = model(x, return_logits=True)
            model.zero_grad(set_to_none=True)
            crit(l, t).backward()

        for _ in range(2):
            model_only()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        for _ in range(K):
            model_only()
        t4 = time.perf_counter()
        torch.cuda.synchronize()
        print(f"{mode}{' hip_graph' if graph else ''} f_maps={f_maps} {shape}: model + loss only: host enqueue {1e3 * (t4 - t3) / K:.2f} ms/step", flush=True)
        print(f"{mode}{' hip_graph' if graph else ''} f_maps={f_maps} {shape}: host enqueue {1e3 * (t1 - t0) / K:.2f} ms/step, step {1e3 * (t2 - t0) / K:.2f} ms", flush=True)
