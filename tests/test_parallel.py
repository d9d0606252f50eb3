"""CPU: the N>1 gradient-averaging path (pytorch3dunet_amd/parallel.py) with world_size 2 over gloo."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT  # noqa: F401  (sets sys.path)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "pytorch-3dunet_amd"))
    from pytorch3dunet_amd import parallel
    from pytorch3dunet_amd.unet3d.model import UNet3D

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(rank)  # different init per rank on purpose
        model = UNet3D(1, 1, f_maps=[4, 8], num_groups=2)
        parallel.broadcast_parameters(model)
        flat0 = torch.cat([p.detach().flatten() for p in model.parameters()])
        gathered = [torch.empty_like(flat0) for _ in range(world)]
        dist.all_gather(gathered, flat0)
        assert all(torch.equal(g, gathered[0]) for g in gathered), "broadcast did not equalise parameters"

        # bucketed async averaging of a flat gradient buffer split like the engine does: [enc | dec+head]
        n = flat0.numel()
        n_enc = sum(p.numel() for p in model.encoders.parameters())
        torch.manual_seed(100 + rank)
        flat = torch.randn(n)
        mine = flat.clone()
        sync = parallel.GradSync()
        sync.launch(flat[n_enc:])
        sync.launch(flat[:n_enc])
        sync.finish()
        all_local = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(all_local, mine)
        expect = torch.stack(all_local).mean(0)
        assert torch.allclose(flat, expect, atol=1e-6), "bucketed all-reduce != mean over ranks"
        assert list(parallel.shard_batch(8, rank, world)) == list(range(rank * 4, rank * 4 + 4))

        # same answer as one process seeing the whole batch: grads of a per-sample-mean loss average over shards
        xs = torch.randn(2, 1, 8, 8, 8, generator=torch.Generator().manual_seed(5))
        shard = xs[rank:rank + 1]
        model.zero_grad()
        model(shard).mean().backward()
        flat_g = torch.cat([p.grad.flatten() for p in model.parameters()])
        sync.launch(flat_g)
        sync.finish()
        model.zero_grad()
        (0.5 * (model(xs[0:1]).mean() + model(xs[1:2]).mean())).backward()
        whole = torch.cat([p.grad.flatten() for p in model.parameters()])
        assert torch.allclose(flat_g, whole, atol=1e-6, rtol=1e-4)

        # parallel.attach on a CPU model: the module tree syncs through post-accumulate-grad hooks, loss.backward() alone
        # leaves the rank-averaged gradients in .grad (no hand-written launch) — twice, the second time accumulating
        sync2 = parallel.attach(model, broadcast=False)
        assert type(sync2).__name__ == "TreeSync" and model._u3d_grad_sync is sync2
        model.zero_grad()
        model(shard).mean().backward()
        hooked = torch.cat([p.grad.flatten() for p in model.parameters()])
        assert torch.allclose(hooked, whole, atol=1e-6, rtol=1e-4) and sync2.launched == 2
        model(shard).mean().backward()  # gradient accumulation: avg(avg(g) + g_local) = 2 avg(g)
        twice = torch.cat([p.grad.flatten() for p in model.parameters()])
        assert torch.allclose(twice, 2 * whole, atol=2e-6, rtol=1e-4) and sync2.launched == 4
        # a pass that leaves a bucket incomplete (here: only two head parameters get a gradient) raises at the END OF
        # THAT PASS, launches nothing, and does not leak its counters into the next pass
        model.zero_grad()
        try:
            (model.final_conv.weight.sum() + model.final_conv.bias.sum()).backward()
            raise AssertionError("an incomplete bucket must raise")
        except RuntimeError as e:
            assert "produced no gradient" in str(e), e
        assert sync2.launched == 4
        model.zero_grad()
        model(shard).mean().backward()
        again = torch.cat([p.grad.flatten() for p in model.parameters()])
        assert torch.allclose(again, whole, atol=1e-6, rtol=1e-4) and sync2.launched == 6
        # a backward that RAISES after the first hook fired (autograd then skips its end-of-pass callbacks) must not leave
        # the pass flag set — the next forward re-arms, and the next backward is a complete, correctly paired exchange again
        w_first = next(model.encoders.parameters())

        def boom(_g):
            raise ValueError("boom")

        h = w_first.register_hook(boom)
        model.zero_grad()
        try:
            model(shard).mean().backward()
            raise AssertionError("the hook must abort this pass")
        except ValueError:
            pass
        h.remove()
        assert sync2._in_backward and sync2.launched == 7  # [decoders | head] went out, the encoder bucket never completed
        model.zero_grad()
        model(shard).mean().backward()
        healed = torch.cat([p.grad.flatten() for p in model.parameters()])
        assert torch.allclose(healed, whole, atol=1e-6, rtol=1e-4) and sync2.launched == 9 and not sync2._in_backward
        # a forward that runs INSIDE the backward pass (torch.utils.checkpoint around the model, a forward issued from a
        # hook) is not a new step — the exchange state of the running pass must survive it
        def reenter(_g):
            with torch.no_grad():
                model(shard)

        h = w_first.register_hook(reenter)  # an encoder weight: fires after [decoders | head] went out
        model.zero_grad()
        model(shard).mean().backward()
        h.remove()
        re = torch.cat([p.grad.flatten() for p in model.parameters()])
        assert torch.allclose(re, whole, atol=1e-6, rtol=1e-4) and sync2.launched == 11 and not sync2._in_backward
        out.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        out.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_gradsync_world2_gloo():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_attach_requires_native_model_and_process_group():
    from pytorch3dunet_amd import parallel
    from pytorch3dunet_amd.unet3d.model import UNet3D

    with pytest.raises(RuntimeError):
        parallel.attach(UNet3D(1, 1, f_maps=[4, 8], num_groups=2))
    with pytest.raises(ValueError):
        parallel.shard_batch(5, 0, 2)
