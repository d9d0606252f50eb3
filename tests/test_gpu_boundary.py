"""-m gpu: the drop-in boundary under the reference's callers' usage patterns — the training loop of
UNetTrainer (trainer.py:214-300, 301-349, 351-368, 381-403) restated on the GPU box (the unmodified trainer itself is driven
in the build container by tests/test_reference_trainer.py), nn.DataParallel as trainer.py:202-205 / predict.py:63-66 wrap
it, retain_graph / second backward, parameter placement errors, and the 1-rank RCCL gradient path."""
import copy
import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT, diag

pytestmark = pytest.mark.gpu

CFG = dict(in_channels=1, out_channels=1, f_maps=[8, 16, 32], num_groups=4, final_sigmoid=True)


def _batches(n, seed, shape=(2, 1, 8, 16, 16)):
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        x = torch.randn(shape, generator=g)
        out.append((x, (x > 0.3).float()))
    return out


def _train_steps(model, opt, crit, batches, dev):
    losses = []
    model.train()
    for x, t in batches:
        out, logits = model(x.to(dev), return_logits=True)  # trainer.py:362
        loss = crit(logits, t.to(dev))                        # :365, always on the logits
        losses.append(loss.item())                            # :241
        opt.zero_grad()                                       # :244
        loss.backward()
        opt.step()
    return losses


def test_training_loop_validate_checkpoint_resume_identical_next_step(tmp_path):
    """train 3 iterations (Adam over the flat-buffer gradient views) -> validate under no_grad/eval -> save the
    reference-format checkpoint (utils.py:17-34) -> resume in fresh objects -> the next training step gives the IDENTICAL
    loss and parameters as continuing in the original objects; and the whole trajectory follows the stock torch.nn module
    tree (our CPU branch = the reference's semantics) within fp32 round-off."""
    from pytorch3dunet_amd import _native as nat
    from pytorch3dunet_amd.unet3d.losses import BCEDiceLoss
    from pytorch3dunet_amd.unet3d.model import UNet3D

    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    cpu_model = UNet3D(**CFG)
    model = copy.deepcopy(cpu_model).to(dev)
    crit = BCEDiceLoss()
    mk_opt = lambda m: torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-5)  # noqa: E731  utils.py:307-309
    opt, cpu_opt = mk_opt(model), mk_opt(cpu_model)
    init = [p.detach().clone() for p in cpu_model.parameters()]
    train, val = _batches(5, 1), _batches(2, 2)

    n0 = nat.launch_count
    losses = _train_steps(model, opt, crit, train[:3], dev)
    assert nat.launch_count > n0
    cpu_losses = _train_steps(cpu_model, cpu_opt, crit, train[:3], torch.device("cpu"))
    assert max(abs(a - b) for a, b in zip(losses, cpu_losses)) < 2e-4, (losses, cpu_losses)
    # parameters after 3 Adam steps: Adam normalises every coordinate's step to ~lr, so coordinates whose gradient is pure
    # round-off noise (e.g. the first GroupNorm's gamma: analytically 0) move by +-lr in an implementation-dependent
    # direction; compare the update as a whole (relative L2 of the difference of the two updates)
    num = den = 0.0
    for (k, p), (_, q), p0 in zip(model.named_parameters(), cpu_model.named_parameters(), init):
        assert p.grad is not None and p.grad.shape == p.shape, k
        num += (p.detach().cpu() - q.detach()).double().pow(2).sum().item()
        den += (q.detach() - p0).double().pow(2).sum().item()
    upd_rel = (num / den) ** 0.5
    diag(test="adam_trajectory", losses=losses, cpu_losses=cpu_losses, update_rel_l2=upd_rel)
    assert upd_rel < 0.1, upd_rel

    # validation (trainer.py:301-349): eval mode + no_grad, loss on logits, probabilities for the metric
    model.eval()
    with torch.no_grad():
        vls = []
        for x, t in val:
            out, logits = model(x.to(dev), return_logits=True)
            assert not logits.requires_grad and float(out.min()) >= 0 and float(out.max()) <= 1
            vls.append(crit(logits, t.to(dev)).item())
    model.train()

    # checkpoint in the reference's layout (utils.py:17-34, trainer.py:381-403)
    path = os.path.join(tmp_path, "last_checkpoint.pytorch")
    torch.save({"num_epochs": 1, "num_iterations": 3, "model_state_dict": model.state_dict(), "best_eval_score": 0.5,
                "optimizer_state_dict": opt.state_dict()}, path)
    state = torch.load(path, map_location="cpu")  # utils.py:59
    model2 = UNet3D(**CFG).to(dev)
    opt2 = mk_opt(model2)
    model2.load_state_dict(state["model_state_dict"])  # strict
    opt2.load_state_dict(state["optimizer_state_dict"])

    nxt = _train_steps(model, opt, crit, train[3:4], dev)
    nxt2 = _train_steps(model2, opt2, crit, train[3:4], dev)
    assert nxt == nxt2, (nxt, nxt2)
    for (k, p), (_, q) in zip(model.named_parameters(), model2.named_parameters()):
        assert torch.equal(p, q), k
    # validation gives the same numbers again in the resumed model before that step? (forward is a pure function of state)
    model2.eval()
    with torch.no_grad():
        _, lg_a = model(val[0][0].to(dev), return_logits=True)
        _, lg_b = model2(val[0][0].to(dev), return_logits=True)
    assert torch.equal(lg_a, lg_b)


def test_second_backward_with_retain_graph_and_error_without():
    """The reference's modules support loss.backward(retain_graph=True) followed by another backward; without it autograd
    raises its own 'backward through the graph a second time'.  The activation tape is owned by autograd
    (ctx.save_for_backward), so both behaviours are the stock ones."""
    from pytorch3dunet_amd.unet3d.model import UNet3D

    dev = torch.device("cuda", 0)
    torch.manual_seed(3)
    model = UNet3D(**CFG).to(dev).train()
    x = torch.randn(1, 1, 8, 16, 16, device=dev)
    _, logits = model(x, return_logits=True)
    loss = (logits * logits).mean()
    loss.backward(retain_graph=True)
    g1 = [p.grad.clone() for p in model.parameters()]
    model.zero_grad()
    loss.backward()  # second pass over the retained graph: identical gradients
    for a, p in zip(g1, model.parameters()):
        assert torch.equal(a, p.grad)
    with pytest.raises(RuntimeError, match="second time|already been freed"):
        loss.backward()
    # two forwards alive at once, at different input sizes, backward in the opposite order (per-call state, no engine globals)
    xa, xb = torch.randn(1, 1, 8, 16, 16, device=dev), torch.randn(1, 1, 12, 20, 20, device=dev)
    model.zero_grad()
    _, la = model(xa, return_logits=True)
    _, lb = model(xb, return_logits=True)
    (la * la).mean().backward()
    ga = [p.grad.clone() for p in model.parameters()]
    model.zero_grad()
    (lb * lb).mean().backward()
    model.zero_grad()
    _, la2 = model(xa, return_logits=True)
    (la2 * la2).mean().backward()
    for a, p in zip(ga, model.parameters()):
        assert torch.equal(a, p.grad)


def test_dataparallel_wrap_trains_and_predicts():
    """trainer.py:202-205 / predict.py:63-66 wrap the model in nn.DataParallel whenever more than one device is visible.
    Replicas are shallow copies whose parameters are broadcast NON-LEAF tensors kept as plain attributes: every replica
    gets its own executor (nothing mutable shared between the replica threads) and gradients flow back through the
    broadcast.  On this one-GPU box the two replicas both live on device 0 (device_ids=[0, 0]) — same code path: scatter,
    replicate, one Python thread per replica calling into the C-ABI concurrently, gather."""
    from pytorch3dunet_amd import _native as nat
    from pytorch3dunet_amd.unet3d.model import UNet3D, is_model_2d

    dev = torch.device("cuda", 0)
    torch.manual_seed(4)
    model = UNet3D(**CFG).to(dev).train()
    ref = copy.deepcopy(model)
    dp = torch.nn.DataParallel(model, device_ids=[0, 0])
    assert not is_model_2d(dp)
    x = torch.randn(4, 1, 8, 16, 16, device=dev)
    t = (x > 0.3).float()
    n0 = nat.launch_count
    try:
        out, logits = dp(x, return_logits=True)
    except RuntimeError as e:  # pragma: no cover - torch refusing duplicate device ids would be an environment limit
        if "device" in str(e) and "duplicate" in str(e).lower():
            pytest.skip(f"this torch build refuses device_ids=[0, 0]: {e}")
        raise
    assert nat.launch_count > n0 and logits.shape == x.shape and logits.requires_grad
    loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, t)
    loss.backward()
    # single-model result on the whole batch: the same mean loss -> the same gradients (samples are independent: GroupNorm)
    out_r, logits_r = ref(x, return_logits=True)
    torch.nn.functional.binary_cross_entropy_with_logits(logits_r, t).backward()
    assert (logits - logits_r).abs().max().item() <= 1e-5 * logits_r.abs().max().item()
    for (k, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        assert p.grad is not None, k
        scale = max(q.grad.abs().max().item(), 1e-12)
        assert (p.grad - q.grad).abs().max().item() <= 2e-4 * scale + 1e-9, k
    # the original model is untouched by the replicas and still runs on its own executor
    eng = model._get_engine()
    assert eng.model is model
    dp.eval()
    with torch.no_grad():
        y = dp(x)
        y1 = model(x)
    assert torch.allclose(y, y1, atol=1e-6)


def test_parameter_placement_and_dtype_errors_are_loud():
    from pytorch3dunet_amd.unet3d.model import UNet3D

    dev = torch.device("cuda", 0)
    model = UNet3D(**CFG)  # parameters left on the CPU, CUDA input: stock modules raise a device-mismatch RuntimeError too
    with pytest.raises(RuntimeError, match="u3d: parameter"):
        model(torch.randn(1, 1, 8, 16, 16, device=dev))
    model = model.to(dev).half()
    with pytest.raises(RuntimeError, match="u3d: parameter"):
        model(torch.randn(1, 1, 8, 16, 16, device=dev))
    model = model.float()
    assert model(torch.randn(1, 1, 8, 16, 16, device=dev)).shape == (1, 1, 8, 16, 16)
    # parameters replaced wholesale (load_state_dict(assign=True)): the executor follows the new tensors
    sd = {k: v.detach().clone() * 0.5 for k, v in model.state_dict().items()}
    x = torch.randn(1, 1, 8, 16, 16, device=dev)
    y0 = model(x)
    sd_copy = {k: v.clone() for k, v in sd.items()}  # assign=True makes the model's Parameters WRAP the tensors of `sd`
    model.load_state_dict(sd, assign=True)
    y1 = model(x)
    fresh = UNet3D(**CFG).to(dev)
    fresh.load_state_dict(sd_copy)
    assert torch.equal(y1, fresh(x)) and not torch.equal(y0, y1)
    # versioned in-place updates (what optimizers and EMA swaps under no_grad do) invalidate the packed weight images
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(2.0)
    fresh2 = UNet3D(**CFG).to(dev)
    fresh2.load_state_dict({k: v * 2.0 for k, v in sd_copy.items()})
    assert torch.equal(model(x), fresh2(x))


def _spawn(args, env_extra, timeout=600):
    env = dict(os.environ, **env_extra)
    return subprocess.run(args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


@pytest.mark.timeout(900)
def test_one_rank_rccl_gradient_path_equals_unsynced_run():
    """The engine -> RCCL hook placement (decoder+head bucket launched before the encoder backward, encoder bucket after,
    engine.py backward) executed on a real device with a 1-rank `nccl` group: averaging over one rank is the identity, so the
    gradients must equal the unsynced run bit for bit, twice in a row; both buckets are really issued."""
    code = r'''
import os, sys, json
sys.path.insert(0, os.path.join(os.getcwd(), "pytorch-3dunet_amd"))
import torch, torch.distributed as dist
from pytorch3dunet_amd import parallel
from pytorch3dunet_amd.unet3d.model import UNet3D, ResidualUNet3D
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
res = {}
for name, cls in (("unet", UNet3D), ("res", ResidualUNet3D)):
    torch.manual_seed(0)
    model = cls(1, 1, f_maps=[8, 16, 32], num_groups=4).to(dev).train()
    x = torch.randn(2, 1, 8, 16, 16, device=dev)
    def grads():
        model.zero_grad()
        _, lg = model(x, return_logits=True)
        (lg * lg).mean().backward()
        torch.cuda.synchronize()
        return torch.cat([p.grad.flatten() for p in model.parameters()]).clone()
    g_plain = grads()
    sync = parallel.attach(model, force_single=True)
    g1 = grads(); n1 = sync.launched
    g2 = grads()
    res[name] = {"equal_plain": bool(torch.equal(g_plain, g1)), "equal_rerun": bool(torch.equal(g1, g2)),
                 "launched_per_backward": n1, "backend": dist.get_backend(), "world": dist.get_world_size()}
dist.destroy_process_group()
print("RESULT " + json.dumps(res))
'''
    proc = _spawn([sys.executable, "-c", code], {"HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert proc.returncode == 0, proc.stderr[-3000:]
    r = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    for name in ("unet", "res"):
        assert r[name] == {"equal_plain": True, "equal_rerun": True, "launched_per_backward": 2, "backend": "nccl", "world": 1}, r


@pytest.mark.timeout(900)
@pytest.mark.parametrize("min_bucket_mb,per_step", [("1", 2), ("0", 4)])
def test_hip_graph_under_a_one_rank_rccl_group_is_bitwise_the_eager_synced_run(min_bucket_mb, per_step):
    """`hip_graph` used to refuse to run with the data-parallel exchange attached (every rank paid the eager
    host enqueue).  Now the backward is captured as two graphs cut where engine.backward hands the [decoders | head] bucket to
    RCCL; the all-reduces are launched eagerly between / after the replays (trainer.py:202-205's loop, one process per GPU).  On
    a 1-rank `nccl` group: three SGD steps with graphs == three eager steps with the same hooks, bit for bit (losses, gradients,
    parameters), for UNet3D and ResidualUNet3D.  The encoder gradients are exchanged level by level, deepest first, in buckets
    of at least U3D_MIN_BUCKET_MB: these small nets make ONE encoder bucket at the default 1 MiB (2 collectives per backward) and
    one per level at 0 (1 + 3 collectives: the backward is captured as a chain of four graphs)."""
    code = r'''
import os, sys, json, copy
sys.path.insert(0, os.path.join(os.getcwd(), "pytorch-3dunet_amd"))
import torch, torch.distributed as dist
from pytorch3dunet_amd import parallel
from pytorch3dunet_amd.unet3d.model import UNet3D, ResidualUNet3D
from pytorch3dunet_amd.unet3d.losses import BCEDiceLoss
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29547")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
res = {}
g = torch.Generator().manual_seed(3)
xs = [torch.randn(2, 1, 8, 16, 16, generator=g).to(dev) for _ in range(3)]
for name, cls in (("unet", UNet3D), ("res", ResidualUNet3D)):
    torch.manual_seed(0)
    base = cls(1, 1, f_maps=[8, 16, 32], num_groups=4)
    out = {}
    for mode in ("eager", "graph"):
        model = copy.deepcopy(base).to(dev).train()
        if mode == "graph":
            model.hip_graph = True
            model._get_engine().hip_graph = True
        sync = parallel.attach(model, force_single=True)
        opt = torch.optim.SGD(model.parameters(), lr=1e-2, momentum=0.9)
        crit = BCEDiceLoss()
        losses, grads = [], []
        for x in xs:
            _, lg = model(x, return_logits=True)
            loss = crit(lg, (x > 0.3).float())
            opt.zero_grad(); loss.backward()
            grads.append(torch.cat([p.grad.flatten() for p in model.parameters()]).clone())
            losses.append(loss.detach().clone()); opt.step()
        torch.cuda.synchronize()
        eng = model._get_engine()
        out[mode] = (losses, grads, [p.detach().clone() for p in model.parameters()], sync.launched, len(eng._graph_steps), eng._graph_off_reason)
    e, gr = out["eager"], out["graph"]
    res[name] = {"loss": all(torch.equal(a, b) for a, b in zip(e[0], gr[0])), "grads": all(torch.equal(a, b) for a, b in zip(e[1], gr[1])),
                 "params": all(torch.equal(a, b) for a, b in zip(e[2], gr[2])), "launched_eager": e[3], "launched_graph": gr[3],
                 "captured": gr[4], "off_reason": gr[5]}
dist.destroy_process_group()
print("RESULT " + json.dumps(res))
'''
    proc = _spawn([sys.executable, "-c", code], {"HSA_ENABLE_IPC_MODE_LEGACY": "0", "U3D_MIN_BUCKET_MB": min_bucket_mb})
    assert proc.returncode == 0, proc.stderr[-3000:]
    r = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    for name in ("unet", "res"):
        # per_step collectives x 3 steps in BOTH modes: GraphStep's warm-up step and its capture issue no collective (a
        # capture is a per-rank event — a new shape or an eviction on one rank must not unpair the ranks' collective sequences)
        assert r[name] == {"loss": True, "grads": True, "params": True, "launched_eager": 3 * per_step, "launched_graph": 3 * per_step,
                           "captured": 1, "off_reason": None}, r


@pytest.mark.timeout(900)
def test_bench_under_torch_distributed_run_one_rank():
    """bench.py's N>1 branch (process group, attach, barrier, max-over-ranks) through the driver's own launcher with one rank"""
    proc = _spawn([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                   "--master-port", "29543", "bench.py", "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                  {"HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert proc.returncode == 0, proc.stderr[-3000:]
    line = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")][-1]
    r = json.loads(line)
    # [decoders | head], then the encoder levels deepest first in buckets of >= 1 MiB: enc3 (5.3 MB), enc2 (1.3 MB), enc1 + enc0
    assert r["ranks_seen"] == {"world_size": 1, "backend": "nccl", "allreduce_per_step": 4}, r["ranks_seen"]
    assert r["n_gpus"] == 1 and r["value"] > 0 and "roofline" in r


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` without an external launcher re-execs itself under torch.distributed.run (forced at N = 1 here:
    the box has one GPU); the one JSON line comes from rank 0 of an RCCL group"""
    proc = _spawn([sys.executable, "bench.py", "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                  {"HSA_ENABLE_IPC_MODE_LEGACY": "0", "U3D_BENCH_SELF_LAUNCH": "1"})
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, proc.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["ranks_seen"] == {"world_size": 1, "backend": "nccl", "allreduce_per_step": 4}, r["ranks_seen"]
    assert r["n_gpus"] == 1 and r["value"] > 0 and "roofline" in r


def test_weight_edits_through_param_data_are_seen():
    """packed weight images are cached on autograd's version counter, which `param.data` writes bypass: a training forward
    repacks regardless; in inference `invalidate_native_caches()` (or U3D_ALWAYS_REPACK=1) is the documented hook"""
    from pytorch3dunet_amd.unet3d.model import UNet3D

    DEV = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = UNet3D(in_channels=1, out_channels=1, f_maps=[16, 32], num_groups=8).to(DEV)
    x = torch.randn(1, 1, 8, 16, 16, device=DEV)
    target = (torch.rand(1, 1, 8, 16, 16, device=DEV) > 0.5).float()
    w = model.encoders[1].basic_module.SingleConv1.conv.weight
    model.train()
    _, l0 = model(x, return_logits=True)
    w.data.add_(0.1 * torch.randn_like(w))  # invisible to w._version (a pure rescaling would vanish in the next GroupNorm)
    _, l1 = model(x, return_logits=True)
    assert (l1 - l0).abs().max().item() > 1e-4  # the training forward saw the new values
    torch.nn.functional.binary_cross_entropy_with_logits(l1, target).backward()
    ref = UNet3D(in_channels=1, out_channels=1, f_maps=[16, 32], num_groups=8).to(DEV).train()
    ref.load_state_dict(model.state_dict())
    _, lr = ref(x, return_logits=True)
    assert torch.equal(lr, l1)
    # inference: cached by design, explicit invalidation
    model.eval()
    with torch.no_grad():
        e0 = model(x)
        w.data.add_(0.1 * torch.randn_like(w))
        model.invalidate_native_caches()
        e1 = model(x)
    assert (e1 - e0).abs().max().item() > 1e-5


def test_replaced_middle_parameter_objects_are_seen():
    """the cached executor used to be re-validated by the first and last parameter object only.  A partial
    `load_state_dict(assign=True, strict=False)` (a backbone checkpoint without the first conv and the head) is now seen on the next
    forward (load_state_dict post-hook -> full identity walk); a plain `module.weight = nn.Parameter(...)` after
    `invalidate_native_caches()` likewise; without it a conv / norm parameter is missed by the executor's index at once (rebuild +
    re-run), anything else within 16 forwards."""
    from pytorch3dunet_amd.unet3d.model import UNet3D

    DEV = torch.device("cuda", 0)
    torch.manual_seed(0)
    cfg = dict(in_channels=1, out_channels=1, f_maps=[16, 32], num_groups=8)
    model = UNet3D(**cfg).to(DEV).eval()
    x = torch.randn(1, 1, 8, 16, 16, device=DEV)

    def fresh_output():
        ref = UNet3D(**cfg).to(DEV).eval()
        ref.load_state_dict(model.state_dict())
        with torch.no_grad():
            return ref(x)

    with torch.no_grad():
        y0 = model(x)
        eng0 = model._get_engine()
        # 1. partial load with assign=True: only the middle layers' tensors are replaced
        part = {k: (v + 0.05 * torch.randn_like(v)) for k, v in model.state_dict().items()
                if not k.startswith("final_conv") and "encoders.0.basic_module.SingleConv1" not in k}
        model.load_state_dict(part, strict=False, assign=True)
        y1 = model(x)
        assert model._get_engine() is not eng0 and (y1 - y0).abs().max().item() > 1e-5 and torch.equal(y1, fresh_output())
        # 2. attribute assignment + the documented hook
        eng1 = model._get_engine()
        conv = model.decoders[0].basic_module.SingleConv2.conv
        conv.weight = torch.nn.Parameter(conv.weight.detach() * 1.5 + 0.01)
        model.invalidate_native_caches()
        y2 = model(x)
        assert model._get_engine() is not eng1 and (y2 - y1).abs().max().item() > 1e-5 and torch.equal(y2, fresh_output())
        # 3. attribute assignment alone.  Inference reads the module's tensors live (the weight image cache is keyed by object and
        #    version), so the output is right at once; the periodic identity walk replaces the executor within 16 forwards
        eng2 = model._get_engine()
        conv.weight = torch.nn.Parameter(conv.weight.detach() * 0.5)
        want = fresh_output()
        assert torch.equal(model(x), want)
        outs = [model(x) for _ in range(17)]
        assert torch.equal(outs[-1], want) and model._get_engine() is not eng2
    # 4. ... and a TRAINING forward indexes its parameters (tape records): the replaced object is missed at once (StaleParameters ->
    #    the executor is rebuilt and the forward re-run), so the gradient lands in the NEW parameter
    model.train()
    eng3 = model._get_engine()
    conv.weight = torch.nn.Parameter(conv.weight.detach() * 1.25)
    _, logits = model(x, return_logits=True)
    assert model._get_engine() is not eng3
    logits.square().mean().backward()
    assert conv.weight.grad is not None and conv.weight.grad.abs().sum().item() > 0


def test_forward_under_no_grad_is_a_real_inference_forward():
    """`ctx.needs_input_grad` inside an autograd.Function reports the parameters' requires_grad flags even when the caller is under
    torch.no_grad(): until round 4 every inference forward (predictor.py:144-163 runs the model under no_grad) therefore kept an
    activation tape, advanced the repack salt and repacked every weight image — 6 ms per volume of BASELINE config 5
    (profiles/r04_cfg5_*).  Now the caller's grad mode decides: no pack launch after the first no_grad forward, no tape (peak memory
    well below a forward that can be followed by a backward), identical outputs."""
    from pytorch3dunet_amd import _native as nat
    from pytorch3dunet_amd.unet3d.model import ResidualUNetSE3D

    DEV = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = ResidualUNetSE3D(in_channels=3, out_channels=1, f_maps=[16, 32, 64], num_groups=8).to(DEV).eval()
    assert all(p.requires_grad for p in model.parameters())
    x = torch.randn(2, 3, 16, 32, 32, device=DEV)
    with torch.no_grad():
        y0 = model(x)
        prof = nat.EventProfiler()
        nat.profiler = prof
        try:
            y1 = model(x)
            torch.cuda.synchronize()
        finally:
            nat.profiler = None
    assert torch.equal(y0, y1)
    packs = [k for k in prof.summary() if "pack" in k]
    assert not packs, packs
    del y0, y1

    def peak(fn):
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        out = fn()
        torch.cuda.synchronize()
        return torch.cuda.max_memory_allocated() - base, out

    def infer():
        with torch.no_grad():
            return model(x)

    m_inf, y_inf = peak(infer)
    m_grad, y_grad = peak(lambda: model(x))
    assert y_grad.grad_fn is not None and y_inf.grad_fn is None and torch.equal(y_inf, y_grad.detach())
    assert m_inf < 0.6 * m_grad, (m_inf, m_grad)
