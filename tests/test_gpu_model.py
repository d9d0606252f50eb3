"""-m gpu: the whole hot path (model forward + backward through the fused executor) against the golden vectors
of the live reference, against the CPU oracle on extra configurations, and — at BASELINE.json's full sizes — through
size-independent properties."""
import pytest
import torch
import torch.nn.functional as F

from conftest import Golden, GOLDEN_NAMES, diag, loss_by_name

pytestmark = pytest.mark.gpu

REL = 1e-3  # BASELINE.json north_star: within 1e-3 rel fp32 of the reference CPU path
GRAD_FLIP_FACTOR = 4.0  # see test_model_matches_reference_golden (sampled fixtures)


def _make(cfg):
    """cfg without 'name' -> UNet3D; with name -> get_model (e.g. ResidualUNet3D)"""
    from pytorch3dunet_amd.unet3d.model import UNet3D, get_model

    return get_model(dict(cfg)) if "name" in cfg else UNet3D(**cfg)


def _run_native(model, x, target, loss_name, decisions=None):
    """one native step; `decisions` (a dict) receives the ReLU masks / pool arg-maxes the step took (NCDHW, CPU)"""
    from pytorch3dunet_amd import _native as nat

    dev = torch.device("cuda", 0)
    model = model.to(dev).train()
    before = nat.launch_count
    eng = model._get_engine() if decisions is not None else None
    if eng is not None:
        eng.debug = {}
    probs, logits = model(x.to(dev), return_logits=True)
    if eng is not None:
        tape = eng.debug["tape"]
        ncdhw = lambda t: t.permute(0, 4, 1, 2, 3).contiguous().cpu()  # noqa: E731
        decisions["masks"] = [ncdhw(r.y > 0) for r in tape.convs]
        decisions["argmax"] = [ncdhw(am) for (_, am, _) in tape.pools]
        decisions["names"] = [r.name for r in tape.convs]
    loss = loss_by_name(loss_name, probs, logits, target.to(dev))
    model.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    if eng is not None:
        eng.debug = None
    assert nat.launch_count > before, "native HIP path did not run"
    grads = {k: p.grad.detach().cpu() for k, p in model.named_parameters()}
    return probs.detach().cpu(), logits.detach().cpu(), loss.item(), grads


def _flip_audit(g, sd, x, target, decisions, grads):
    """A `full` golden whose direct per-parameter gate fails: the only legitimate cause is a DISCRETE decision taken differently at a
    pre-activation within fp32 round-off of zero (one ReLU flip is an O(1e-3) event in these tiny nets; the reference's own fixtures
    for them happen to contain none).  Audit exactly that: (1) every ReLU mask of ours that differs from the fp32 oracle's own sits at
    |pre-activation| <= DECISION_TOL of its layer's range, at most 4 per layer, no arg-max differs; (2) with OUR decisions imposed, the
    float64 oracle reproduces every gradient to 1e-4 (first GroupNorm gamma 1e-3) — nothing but those flips separates the two."""
    import unet3d_oracle as orc

    cfg = g.cfg
    G, fs, seg = cfg.get("num_groups", 8), cfg.get("final_sigmoid", True), cfg.get("is_segmentation", True)
    trace, _ = orc.forward_decisions(sd, x, G, fs, seg)
    flips, worst = 0, 0.0
    assert len(decisions["masks"]) == len(trace["pre"])
    for name, ours, z in zip(decisions["names"], decisions["masks"], trace["pre"]):
        diff = ours != (z > 0)
        n = int(diff.sum())
        if n:
            rel = (z[diff].abs().max() / z.abs().max()).item()
            assert rel <= DECISION_TOL and n <= 4, (name, n, rel)
            worst = max(worst, rel)
        flips += n
    for ours, h in zip(decisions["argmax"], trace["pool"]):
        _, idx = F.max_pool3d(h, 2, return_indices=True)
        Hh, Ww = h.shape[3:]
        theirs = ((idx // (Hh * Ww)) % 2) * 4 + (((idx // Ww) % Hh) % 2) * 2 + (idx % Ww) % 2
        assert int((ours.long() != theirs).sum()) == 0
    assert flips > 0, "the direct gate failed without a single decision flip: an arithmetic error"
    _, _, g64 = orc.forward_backward_decided(sd, x, target, decisions["masks"], decisions["argmax"], G, fs, seg, g.loss_name)
    first_gamma = next(k for k in grads if k.endswith("groupnorm.weight"))
    for k, v in grads.items():
        e = orc.rel_err(v.double(), g64[k])
        assert e < (1e-3 if k == first_gamma else 1e-4), (k, e)
    return flips, worst


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_model_matches_reference_golden(name):
    import unet3d_oracle as orc

    g = Golden(name)
    model = g.build_model()
    x, target = g.inputs()
    decisions = {} if g.full else None
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    probs, logits, loss, grads = _run_native(model, x, target, g.loss_name, decisions)
    assert abs(loss - g.loss) <= REL * max(1.0, abs(g.loss))
    if g.full:
        assert orc.rel_err(logits, g.tensor("logits")) < REL
        assert orc.rel_err(probs, g.tensor("probs")) < REL
        worst, failing = 0.0, []
        for k, rg in g.group("grad/").items():
            e = orc.rel_err(grads[k], rg)
            worst = max(worst, e)
            if not orc.grad_within_tolerance(grads[k], rg, float(g.z["ref_err/" + k]), REL):
                failing.append((k, e))
        print(f"{name}: logits rel {orc.rel_err(logits, g.tensor('logits')):.2e}, worst grad rel {worst:.2e}")
        if failing:
            # (round 5: g3's last layer — 8 output channels on a 12 x 20 x 17 volume — moved from the generic kernel to the 16-column
            # persistent variant, whose MFMA shape sums in another order: ONE ReLU mask at a pre-activation of 1e-7 of the layer's range
            # flips, and with it every gradient upstream by 1e-3 ... 5e-3, tools/diag_golden.py)
            flips, worst_pre = _flip_audit(g, sd, x, target, decisions, grads)
            print(f"{name}: {len(failing)} parameters outside the direct gate through {flips} ReLU flip(s) at |pre-activation| <= "
                  f"{worst_pre:.1e} of the layer's range; decision-consistent float64 gate passed")
            assert worst < 2e-2, failing[:3]  # (a handful of round-off flips cannot move a gradient further than this)
    else:
        # sampled fixture of a BASELINE-size configuration (g4 = config 1, g9 = config 2 at full size, g10 / g11 = the channel
        # ladders of configs 4 / 5).  Logits: 1e-3 of the reference's range.  Gradients, per parameter:
        #   |ours - ref32| <= max(1e-3 * max|ref32|, GRAD_FLIP_FACTOR * ref_err)
        # ref_err = max|ref32 - ref64| of the reference itself on this very step: its fp32 path takes ReLU / arg-max
        # decisions at pre-activations within round-off of 0 differently from exact arithmetic, each flip an O(1) local
        # event.  Ours takes an independent set of such decisions, so |ours - ref32| is the difference of two such error
        # processes.  Measured worst ratio err / max(1e-3*absmax, ref_err) over all parameters (profiles/r02_parity_diag.jsonl):
        # g4 1.37, g9 (config 2 at full size) 3.67, g10 2.45, g11 1.00 -> GRAD_FLIP_FACTOR = 4 (round 1 used 10).  The global relative L2 distance from the float64 gradient must not
        # exceed 2x the reference's own; the decision-consistent test below is the tight (1e-4) gradient gate.
        s = g.sample
        assert (logits.flatten()[::97] - g.tensor("logits_s")).abs().max().item() < REL * float(g.z["logits_absmax"])
        has64 = any(k.startswith("grad64_s/") for k in g.z.files)
        g64 = g.group("grad64_s/") if has64 else {}
        worst, num32, num64, den, bad = (0.0, ""), 0.0, 0.0, 0.0, []
        for k, rs in g.group("grad_s/").items():
            am = float(g.z["grad_absmax/" + k])
            re = float(g.z["ref_err/" + k])  # the reference's own fp32-vs-fp64 deviation for this parameter
            ours = grads[k].flatten()[::s].double()
            err = (ours - rs.double()).abs().max().item()
            ratio = err / max(REL * am, re, 1e-30)
            if ratio > worst[0]:
                worst = (ratio, k)
            if err > max(REL * am, GRAD_FLIP_FACTOR * re):
                bad.append((k, err, am, re))
            n_ref = float(g.z["grad_norm/" + k])
            if abs(grads[k].norm().item() - n_ref) > max(REL * n_ref, GRAD_FLIP_FACTOR * re * grads[k].numel() ** 0.5) + 1e-12:
                bad.append((k, "norm", grads[k].norm().item(), n_ref))
            if has64:
                num64 += (ours - g64[k]).pow(2).sum().item()
                num32 += (rs.double() - g64[k]).pow(2).sum().item()
                den += g64[k].pow(2).sum().item()
        rec = {"test": "golden_big", "name": name, "worst_ratio_to_max(1e-3*absmax,ref_err)": worst[0], "worst_param": worst[1]}
        if has64:
            ours_l2, ref_l2 = (num64 / den) ** 0.5, (num32 / den) ** 0.5
            rec.update(ours_vs_fp64_rel_l2=ours_l2, ref32_vs_fp64_rel_l2=ref_l2, ref32_vs_fp64_rel_l2_all=float(g.z["ref_grad_rel_l2"]))
        diag(**rec)
        print(rec)
        assert not bad, bad[:5]
        if has64:
            assert ours_l2 <= max(REL, 2.0 * ref_l2), (ours_l2, ref_l2)


@pytest.mark.parametrize("cfg,shape,loss_name", [
    (dict(in_channels=1, out_channels=1, f_maps=32, num_groups=8), (2, 1, 16, 32, 32), "bce_dice"),  # cfg2 model, small patch, batch 2
    (dict(in_channels=1, out_channels=1, f_maps=16, num_groups=8), (1, 1, 33, 65, 65), "bce_dice"),  # the reference's odd test shape
    (dict(in_channels=3, out_channels=2, f_maps=[16, 32, 64, 128], num_groups=4, final_sigmoid=False), (1, 3, 16, 24, 40), "probs_sum"),
    # 12x20x24 -> 6x10x12 -> 3x5x6 -> 1x2x3: the two upper decoder levels are exact 2x (sub-pixel kernels), the deepest one is
    # not (virtual-concat kernel with index maps) — both paths in one network, ragged tiles everywhere
    (dict(in_channels=2, out_channels=1, f_maps=8, num_groups=4), (2, 2, 12, 20, 24), "bce_dice"),
    # conv_upscale=1: the FIRST conv of an encoder DoubleConv widens the channels (buildingblocks.py:200-227)
    (dict(in_channels=1, out_channels=1, f_maps=16, num_levels=3, num_groups=8, conv_upscale=1), (1, 1, 16, 32, 32), "bce_dice"),
    # residual variant: aligned sizes (persistent conv kernels) and the reference's odd test shape
    (dict(name="ResidualUNet3D", in_channels=1, out_channels=1, f_maps=16, num_levels=4, num_groups=8), (1, 1, 16, 32, 32), "bce_dice"),
    (dict(name="ResidualUNet3D", in_channels=1, out_channels=2, f_maps=[16, 32, 64], num_groups=8, final_sigmoid=False), (1, 1, 17, 33, 35), "probs_sum"),
])
def test_model_matches_cpu_oracle(cfg, shape, loss_name):
    import unet3d_oracle as orc

    torch.manual_seed(1234)
    model = _make(cfg)
    with torch.no_grad():
        for k, p in model.named_parameters():
            if "groupnorm" in k:
                p.add_(0.2 * torch.randn_like(p))
    x = torch.randn(shape)
    target = (torch.rand((shape[0], cfg["out_channels"]) + shape[2:]) > 0.5).float()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    p_ref, l_ref, loss_ref, g_ref, ref_err = orc.forward_backward_with_truth(sd, x, target, cfg["num_groups"],
                                                                              cfg.get("final_sigmoid", True), True, loss_name)
    probs, logits, loss, grads = _run_native(model, x, target, loss_name)
    assert orc.rel_err(logits, l_ref) < REL
    assert orc.rel_err(probs, p_ref) < REL
    assert abs(loss - loss_ref.item()) < REL * max(1.0, abs(loss_ref.item()))
    # gradients: a ReLU net's gradient is discontinuous in its pre-activations, and isolated mask / arg-max flips
    # make even the reference's own fp32 path 0.3-3 % (max-norm) away from float64 (tools/gpu_layer_diag.py).  Here
    # we only require that our distance from the reference's fp32 gradient (global relative L2) is no more than a
    # few times the reference's own distance from exact; the TIGHT gradient gate is the decision-consistent test.
    keys = list(g_ref)
    ours = torch.cat([grads[k].flatten().double() for k in keys])
    ref = torch.cat([g_ref[k].flatten().double() for k in keys])
    e_ours = ((ours - ref).norm() / ref.norm()).item()
    e_ref = (sum(ref_err[k] ** 2 * g_ref[k].numel() for k in keys) ** 0.5) / ref.norm().item()  # upper bound of ref's own L2 error
    print(f"global grad rel-L2 ours-vs-ref32 {e_ours:.2e}; reference fp32-vs-fp64 bound {e_ref:.2e}")
    if e_ours > max(REL, 5 * e_ref):
        # The discrepancy must be fully explained by discrete decisions (ReLU masks / pool arg-maxes taken at
        # pre-activations within round-off of 0): with OUR decisions imposed on the float64 oracle the gradients must
        # agree tightly, and the imposed decisions must move the exact gradient by about as much as we differ.
        dev = torch.device("cuda", 0)
        eng = model._get_engine()
        eng.debug = {}
        pr, lg = model(x.to(dev), return_logits=True)
        tape = eng.debug["tape"]
        eng.debug = None
        ncdhw = lambda t: t.permute(0, 4, 1, 2, 3).contiguous().cpu()  # noqa: E731
        masks = [ncdhw(r.y > 0) for r in tape.convs]
        argmax = [ncdhw(am) for (_, am, _) in tape.pools]
        _, _, g_dec = orc.forward_backward_decided(sd, x, target, masks, argmax, cfg["num_groups"],
                                                   cfg.get("final_sigmoid", True), True, loss_name)
        dec = torch.cat([g_dec[k].flatten().double() for k in keys])
        e_dec = ((ours - dec).norm() / dec.norm()).item()
        e_flip = ((dec - ref).norm() / ref.norm()).item()
        print(f"decision-consistent: ours-vs-decided {e_dec:.2e}; decided-vs-ref32 {e_flip:.2e}")
        assert e_dec < 1e-4, (e_ours, e_dec, e_flip)
        assert e_ours <= 1.5 * e_flip + REL, (e_ours, e_dec, e_flip)


@pytest.mark.parametrize("cfg,shape,loss_name", [
    (dict(in_channels=1, out_channels=1, f_maps=16, num_groups=8), (1, 1, 16, 32, 32), "bce_dice"),
    (dict(in_channels=1, out_channels=1, f_maps=16, num_groups=8), (1, 1, 32, 64, 64), "bce_dice"),  # BASELINE config 1's shape (= g4)
    (dict(in_channels=1, out_channels=1, f_maps=32, num_groups=8), (2, 1, 16, 32, 32), "bce_dice"),
    (dict(in_channels=1, out_channels=1, f_maps=16, num_groups=8), (1, 1, 33, 65, 65), "bce_dice"),
    (dict(in_channels=2, out_channels=3, f_maps=[8, 16, 32], num_groups=4, final_sigmoid=False), (2, 2, 9, 13, 11), "probs_sum"),
    (dict(name="ResidualUNet3D", in_channels=1, out_channels=1, f_maps=16, num_levels=4, num_groups=8), (1, 1, 16, 32, 32), "bce_dice"),
    (dict(name="ResidualUNet3D", in_channels=2, out_channels=3, f_maps=[8, 16, 24], num_groups=4, final_sigmoid=False), (2, 2, 9, 13, 11), "probs_sum"),
    (dict(name="ResidualUNet3D", in_channels=1, out_channels=1, f_maps=32, num_levels=3, num_groups=8), (1, 1, 8, 32, 40), "bce_dice"),
    (dict(name="ResidualUNet3D", in_channels=1, out_channels=2, f_maps=[16, 32, 64], num_groups=8, final_sigmoid=False), (1, 1, 17, 33, 35), "probs_sum"),
    (dict(name="ResidualUNetSE3D", in_channels=1, out_channels=1, f_maps=16, num_levels=3, num_groups=8), (1, 1, 16, 32, 32), "bce_dice"),
    (dict(name="ResidualUNetSE3D", in_channels=3, out_channels=2, f_maps=[8, 24, 40], num_groups=4, final_sigmoid=False), (2, 3, 9, 13, 11), "probs_sum"),
    # BASELINE config 2 EXACTLY (UNet3D f_maps=32, per-GPU batch 2x1x64x128x128, BCEDiceLoss): the tight gate at full size — slow,
    # the float64 oracle is a minute or two of host time
    pytest.param(dict(in_channels=1, out_channels=1, f_maps=32, num_groups=8), (2, 1, 64, 128, 128), "bce_dice",
                 marks=pytest.mark.timeout(1800), id="config2-full-size"),
    # the reference's SHIPPED training patch (resources/3DUnet_confocal_boundary/train_config.yml:94): 80 -> 40 -> 20 -> 10 planes,
    # 170 -> 85 -> 42 -> 21 rows: ragged tiles at every level, two n -> 2n + 1 decoder levels, one plain 2x level
    pytest.param(dict(in_channels=1, out_channels=1, f_maps=32, num_groups=8), (1, 1, 80, 170, 170), "bce_dice",
                 marks=pytest.mark.timeout(1800), id="shipped-80x170x170"),
])
def test_gradients_match_decision_consistent_fp64_oracle(cfg, shape, loss_name):
    """The tight gradient check: the float64 oracle with OUR ReLU masks and max-pool arg-maxes imposed
    (oracle.forward_backward_decided) — no flip noise left, so the logits and every parameter gradient must agree to 1e-4 (the cancellation-dominated gamma of
    the first GroupNorm: 1e-3)."""
    import unet3d_oracle as orc

    dev = torch.device("cuda", 0)
    torch.manual_seed(4321)
    model = _make(cfg)
    with torch.no_grad():
        for k, p in model.named_parameters():
            if "groupnorm" in k:
                p.add_(0.2 * torch.randn_like(p))
    x = torch.randn(shape)
    target = (torch.rand((shape[0], cfg["out_channels"]) + shape[2:]) > 0.5).float()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(dev).train()
    eng = model._get_engine()
    eng.debug = {}
    probs, logits = model(x.to(dev), return_logits=True)
    tape = eng.debug["tape"]
    ncdhw = lambda t: t.permute(0, 4, 1, 2, 3).contiguous().cpu()
    masks = [ncdhw(r.y > 0) for r in tape.convs]
    argmax = [ncdhw(am) for (_, am, _) in tape.pools]
    loss_by_name(loss_name, probs, logits, target.to(dev)).backward()
    torch.cuda.synchronize()
    eng.debug = None
    l64, _, g64 = orc.forward_backward_decided(sd, x, target, masks, argmax, cfg["num_groups"], cfg.get("final_sigmoid", True),
                                               True, loss_name)
    e_logits = orc.rel_err(logits.detach().cpu().double(), l64)
    # The first GroupNorm's gamma is the one cancellation-dominated gradient of these nets: the loss is (almost) invariant to the scale
    # of the first conv's input because the NEXT GroupNorm renormalises it, so d/dgamma is a sum of large terms that cancel to ~0 and
    # carries the round-off of every term (tests/golden/make_golden.py measures the same for the reference's own fp32-vs-fp64 step).
    # Measured (profiles/r04_parity_diag.jsonl, test "decided_fp64_gate"): that parameter 4.4e-5 ... 1.0e-4, every other 6e-6 ... 8e-6.
    FIRST_GAMMA = next(k for k, _ in model.named_parameters() if k.endswith("groupnorm.weight"))  # (SingleConv1 / ResNetBlock.conv2)
    worst, first = ("", 0.0), 0.0
    for k, p in model.named_parameters():
        e = orc.rel_err(p.grad.detach().cpu().double(), g64[k])
        if k == FIRST_GAMMA:
            first = e
        elif e > worst[1]:
            worst = (k, e)
    print(f"decision-consistent fp64 oracle: worst gradient rel err {worst[1]:.2e} ({worst[0]}); first GroupNorm gamma {first:.2e}")
    from conftest import diag

    diag(test="decided_fp64_gate", cfg=str(cfg), shape=list(shape), logits_rel=e_logits, worst_grad_rel=worst[1], worst_param=worst[0],
         first_gamma_rel=first)
    # the docstring's 1e-4 — round 3 still asserted the north_star's 1e-3 here for everything ("What's weak" 3)
    assert e_logits < 1e-4 and worst[1] < 1e-4 and first < 1e-3, (e_logits, worst, first)


# |pre-activation| (relative to the layer's largest pre-activation) below which OUR ReLU mask may differ from the fp32
# oracle's, and the same for the gap between the two candidates of a max-pool window: 64 * eps_fp32 = 7.6e-6.  Both
# implementations carry a forward error of ~1e-6 of a layer's range; a mask can only flip where the pre-activation is
# smaller than that error.  Measured (profiles/r02_parity_diag.jsonl): 0-5 flips among 0.1-10 M decisions per network, the
# largest flipped pre-activation 1.9e-6 of its layer's range, no arg-max flip at all -> 4x margin on the level, and at most
# 1e-5 of a layer's decisions (or 4) may differ.
DECISION_TOL = 64 * 1.1920929e-07
MAX_FLIP_FRACTION = 1e-5


@pytest.mark.parametrize("cfg,shape", [
    (dict(in_channels=1, out_channels=1, f_maps=16, num_groups=8), (1, 1, 32, 64, 64)),   # BASELINE config 1's shape
    (dict(in_channels=1, out_channels=1, f_maps=32, num_groups=8), (2, 1, 16, 32, 32)),   # config 2's model, small patch
    (dict(in_channels=1, out_channels=1, f_maps=16, num_groups=8), (1, 1, 33, 65, 65)),   # odd sizes: virtual-concat kernels
    (dict(in_channels=2, out_channels=3, f_maps=[8, 16, 32], num_groups=4, final_sigmoid=False), (2, 2, 9, 13, 11)),
    (dict(name="ResidualUNet3D", in_channels=1, out_channels=1, f_maps=16, num_levels=4, num_groups=8), (1, 1, 16, 32, 32)),
    (dict(name="ResidualUNetSE3D", in_channels=3, out_channels=2, f_maps=[8, 24, 40], num_groups=4, final_sigmoid=False), (2, 3, 9, 13, 11)),
])
def test_discrete_decisions_agree_with_fp32_oracle(cfg, shape):
    """Every ReLU mask and every max-pool arg-max the native path took (they decide which gradient the backward computes)
    against the fp32 oracle's OWN decisions, layer by layer: they may differ only where the oracle's pre-activation (or the
    gap between the two window candidates) is at fp32 round-off level, and only for a bounded number of elements.  A wrong
    mask / arg-max kernel fails here directly (the decision-consistent gradient test imposes OUR decisions on the oracle and
    would reproduce such an error on both sides)."""
    import unet3d_oracle as orc

    dev = torch.device("cuda", 0)
    torch.manual_seed(777)
    model = _make(cfg)
    with torch.no_grad():
        for k, p in model.named_parameters():
            if "groupnorm" in k:
                p.add_(0.2 * torch.randn_like(p))
    x = torch.randn(shape)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    trace, l_ref = orc.forward_decisions(sd, x, cfg["num_groups"], cfg.get("final_sigmoid", True))
    model = model.to(dev).train()
    eng = model._get_engine()
    eng.debug = {}
    _, logits = model(x.to(dev), return_logits=True)
    tape = eng.debug["tape"]
    eng.debug = None
    assert orc.rel_err(logits.detach().cpu(), l_ref) < REL
    ncdhw = lambda t: t.permute(0, 4, 1, 2, 3).contiguous().cpu()  # noqa: E731
    assert len(tape.convs) == len(trace["pre"]) and len(tape.pools) == len(trace["pool"])
    worst_rel, flips, total = 0.0, 0, 0
    for rec, z in zip(tape.convs, trace["pre"]):
        ours = ncdhw(rec.y > 0)
        assert ours.shape == z.shape
        diff = ours != (z > 0)
        n = int(diff.sum())
        flips += n
        total += z.numel()
        if n:
            rel = (z[diff].abs().max() / z.abs().max()).item()
            worst_rel = max(worst_rel, rel)
            assert rel <= DECISION_TOL, (rec.name, n, rel)
        assert n <= max(4, MAX_FLIP_FRACTION * z.numel()), (rec.name, n, z.numel())
    pool_flips, pool_total, worst_gap = 0, 0, 0.0
    for (pooled, argmax, _), h in zip(tape.pools, trace["pool"]):
        win = orc.pool_windows(h)                       # (N,C,d,h,w,8), k = dz*4 + dy*2 + dx
        _, idx = F.max_pool3d(h, 2, return_indices=True)  # ATen's own choice among ties (flat index into D*H*W)
        Hh, Ww = h.shape[3:]
        theirs = ((idx // (Hh * Ww)) % 2) * 4 + (((idx // Ww) % Hh) % 2) * 2 + (idx % Ww) % 2
        ours = ncdhw(argmax).long()
        assert ours.shape == theirs.shape
        diff = ours != theirs
        n = int(diff.sum())
        pool_flips += n
        pool_total += theirs.numel()
        if n:
            gap = (win.gather(-1, theirs.unsqueeze(-1)) - win.gather(-1, ours.unsqueeze(-1))).squeeze(-1)[diff]
            rel = (gap.abs().max() / h.abs().max()).item()
            worst_gap = max(worst_gap, rel)
            assert rel <= DECISION_TOL, (n, rel)
        assert n <= max(4, MAX_FLIP_FRACTION * theirs.numel()), (n, theirs.numel())
    rec = {"test": "decisions", "cfg": str(cfg), "shape": list(shape), "relu_flips": flips, "relu_total": total,
           "worst_flipped_preact_rel": worst_rel, "pool_flips": pool_flips, "pool_total": pool_total, "worst_pool_gap_rel": worst_gap}
    diag(**rec)
    print(rec)


def test_wide_multi_class_head_24_outputs_softmax():
    """model.py:88-101 allows any out_channels; heads wider than 16 outputs used to raise on a HIP device.  UNet3D
    with 24 classes and Softmax: logits / probabilities / every gradient against the CPU oracle, like test_model_matches_cpu_oracle"""
    import unet3d_oracle as orc

    cfg = dict(in_channels=2, out_channels=24, f_maps=[16, 32], num_groups=4, final_sigmoid=False)
    shape = (2, 2, 8, 16, 16)
    torch.manual_seed(99)
    model = _make(cfg)
    assert model.native_supported
    with torch.no_grad():
        for k, p in model.named_parameters():
            if "groupnorm" in k:
                p.add_(0.2 * torch.randn_like(p))
    x = torch.randn(shape)
    target = torch.rand((shape[0], 24) + shape[2:])
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    p_ref, l_ref, loss_ref, g_ref, ref_err = orc.forward_backward_with_truth(sd, x, target, 4, False, True, "probs_sum")
    probs, logits, loss, grads = _run_native(model, x, target, "probs_sum")
    assert orc.rel_err(logits, l_ref) < 1e-4 and orc.rel_err(probs, p_ref) < 1e-4
    assert abs(loss - loss_ref.item()) < REL * max(1.0, abs(loss_ref.item()))
    assert orc.rel_err(grads["final_conv.weight"], g_ref["final_conv.weight"]) < 1e-4
    assert orc.rel_err(grads["final_conv.bias"], g_ref["final_conv.bias"]) < 1e-4
    keys = list(g_ref)
    ours = torch.cat([grads[k].flatten().double() for k in keys])
    ref = torch.cat([g_ref[k].flatten().double() for k in keys])
    e_ref = (sum(ref_err[k] ** 2 * g_ref[k].numel() for k in keys) ** 0.5) / ref.norm().item()
    assert ((ours - ref).norm() / ref.norm()).item() <= max(3e-3, 4 * e_ref)


def test_inference_no_grad_and_eval_matches_train_forward():
    from pytorch3dunet_amd.unet3d.model import UNet3D

    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = UNet3D(1, 1, f_maps=16).to(dev)
    x = torch.rand(1, 1, 33, 65, 65, device=dev)  # tests/test_models.py:17-24 of the reference
    model.eval()
    with torch.no_grad():
        y = model(x)
    assert y.shape == x.shape and torch.all(0 <= y) and torch.all(y <= 1)
    model.train()
    y2, logits = model(x, return_logits=True)
    assert torch.allclose(y, y2.detach(), atol=1e-6)
    assert torch.allclose(torch.sigmoid(logits.detach()), y, atol=1e-6)


def test_input_gradient():
    import unet3d_oracle as orc
    from pytorch3dunet_amd.unet3d.model import UNet3D

    dev = torch.device("cuda", 0)
    torch.manual_seed(2)
    model = UNet3D(2, 1, f_maps=[8, 16], num_groups=2)
    x = torch.randn(1, 2, 8, 8, 8)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    xl = x.clone().requires_grad_(True)
    _, lr = orc.unet3d_forward(sd, xl, 2)
    (lr * lr).mean().backward()
    model = model.to(dev)
    xd = x.to(dev).requires_grad_(True)
    _, lg = model(xd, return_logits=True)
    (lg * lg).mean().backward()
    assert orc.rel_err(xd.grad.cpu(), xl.grad) < REL


def test_full_size_cfg2_properties():
    """BASELINE.json config 2 at full size (f_maps=32, 2x1x64x128x128): runs, finite, probabilities in [0,1],
    run-to-run reproducible; the dominant conv layer shape (96->32 @ 64x128x128) agrees with the naive device
    kernel; the conv is linear in its input (size-independent properties, no CPU oracle at this size)."""
    import gpu_utils as U
    from pytorch3dunet_amd.engine import VSrc
    from pytorch3dunet_amd.unet3d.model import UNet3D

    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = UNet3D(1, 1, f_maps=32).to(dev).train()
    x = torch.randn(2, 1, 64, 128, 128, device=dev)
    target = (torch.rand_like(x) > 0.5).float()
    outs = []
    for _ in range(2):
        probs, logits = model(x, return_logits=True)
        loss = loss_by_name("bce_dice", probs, logits, target)
        model.zero_grad()
        loss.backward()
        outs.append((logits.detach().clone(), torch.cat([p.grad.flatten() for p in model.parameters()]).clone()))
        assert torch.isfinite(logits).all() and torch.all(0 <= probs) and torch.all(probs <= 1)
    assert torch.isfinite(outs[0][1]).all()
    assert U.relerr(outs[1][0], outs[0][0]) < 1e-6 and U.relerr(outs[1][1], outs[0][1]) < 1e-5
    del outs
    # dominant layer shape vs naive kernel
    torch.manual_seed(1)
    a = torch.randn(1, 64, 128, 128, 96, device=dev)
    w = torch.randn(32, 96, 3, 3, 3) / (27 * 96) ** 0.5
    src = VSrc(a)
    y = U.conv3d(src, w, 32, relu=0)
    yn = U.conv3d_naive(src, w, 32)
    assert U.relerr(y, yn) < 2e-5
    y2 = U.conv3d(VSrc(a * 2.0), w, 32, relu=0)
    assert U.relerr(y2, 2.0 * y) < 1e-6
    dz = torch.randn(1, 64, 128, 128, 32, device=dev)
    dw = U.wgrad(src, dz, 32)
    # <dw, w> == <dz, conv(a, w)>  (adjoint identity ties wgrad to the forward kernel at full size)
    lhs = (dw.double().cpu() * w.double()).sum().item()
    rhs = (dz.double() * y.double()).sum().item()
    assert abs(lhs - rhs) < 1e-4 * max(abs(lhs), abs(rhs), 1.0)
    # <dgrad(dz), a> == <dz, conv(a, w)>
    dg = U.conv3d(VSrc(dz), w, 96, relu=0, mode=1)
    lhs2 = (dg.double() * a.double()).sum().item()
    assert abs(lhs2 - rhs) < 1e-4 * max(abs(lhs2), abs(rhs), 1.0)


def test_uncovered_3d_variant_raises_by_default_and_runs_only_on_opt_in(monkeypatch):
    """ONE backend by default: a 3-D model outside the executor's envelope raises on a HIP device; the module
    tree on stock PyTorch-ROCm operators is the explicit opt-in U3D_ALLOW_TORCH_FALLBACK=1 (one warning, no native launch);
    U3D_STRICT=1 wins over the opt-in."""
    from pytorch3dunet_amd import _native as nat
    from pytorch3dunet_amd.unet3d.model import ResidualUNet3D

    dev = torch.device("cuda", 0)
    # dropout inside residual blocks: a configuration the reference runs (buildingblocks.py:230-288) and the executor does not cover
    model = ResidualUNet3D(1, 1, f_maps=16, num_levels=3, layer_order="gcrd").to(dev).eval()
    assert not model.native_supported
    x = torch.rand(1, 1, 8, 16, 16, device=dev)
    monkeypatch.delenv("U3D_ALLOW_TORCH_FALLBACK", raising=False)
    monkeypatch.delenv("U3D_STRICT", raising=False)
    with pytest.raises(NotImplementedError, match="U3D_ALLOW_TORCH_FALLBACK"):
        model(x)
    monkeypatch.setenv("U3D_ALLOW_TORCH_FALLBACK", "1")
    n0 = nat.launch_count
    with pytest.warns(UserWarning, match="stock PyTorch-ROCm operators"):
        y = model(x)
    assert nat.launch_count == n0 and y.shape == x.shape and bool(((y >= 0) & (y <= 1)).all())
    monkeypatch.setenv("U3D_STRICT", "1")
    with pytest.raises(NotImplementedError):
        model(x)


def test_2d_models_keep_the_warning_path(monkeypatch):
    """2-D variants (reference model.py:281-358) are outside the 3-D path: module tree + one warning by default, error under
    U3D_STRICT=1"""
    from pytorch3dunet_amd.unet3d.model import UNet2D

    dev = torch.device("cuda", 0)
    monkeypatch.delenv("U3D_ALLOW_TORCH_FALLBACK", raising=False)
    monkeypatch.delenv("U3D_STRICT", raising=False)
    model = UNet2D(1, 1, f_maps=8, num_levels=2, num_groups=4).to(dev).eval()
    x = torch.rand(1, 1, 16, 16, device=dev)  # (N, C, H, W): the trainer squeezes z for 2-D models (trainer.py:352-360)
    with pytest.warns(UserWarning, match="2-D model"):
        y = model(x)
    assert y.shape == x.shape
    monkeypatch.setenv("U3D_STRICT", "1")
    with pytest.raises(NotImplementedError):
        model(x)


@pytest.mark.parametrize("name,levels,shape", [("UNet3D", 4, (1, 1, 8, 8, 8)), ("UNet3D", 3, (3, 1, 4, 12, 20)),
                                               ("ResidualUNet3D", 3, (1, 1, 4, 4, 4)), ("ResidualUNetSE3D", 3, (2, 1, 6, 10, 14))])
def test_edge_shapes_minimum_size_and_odd_batch(name, levels, shape):
    """smallest legal inputs (one voxel at the deepest level), batch 3, sizes that are not multiples of the pooling
    factor: forward + every gradient against the CPU oracle"""
    import unet3d_oracle as orc

    torch.manual_seed(11)
    model = _make(dict(name=name, in_channels=1, out_channels=1, f_maps=8, num_levels=levels, num_groups=4))
    x = torch.randn(shape)
    target = (torch.rand(shape) > 0.5).float()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    p_ref, l_ref, loss_ref, g_ref, ref_err = orc.forward_backward_with_truth(sd, x, target, 4, True, True, "bce_dice")
    probs, logits, loss, grads = _run_native(model, x, target, "bce_dice")
    assert orc.rel_err(logits, l_ref) < REL and orc.rel_err(probs, p_ref) < REL
    keys = list(g_ref)
    ours = torch.cat([grads[k].flatten().double() for k in keys])
    ref = torch.cat([g_ref[k].flatten().double() for k in keys])
    assert ((ours - ref).norm() / ref.norm()).item() < 2e-2  # tiny tensors: a single ReLU flip is a percent-level event
    assert torch.isfinite(ours).all()


def test_non_contiguous_and_strided_inputs():
    """the boundary accepts what nn.Module.forward accepts: a permuted / sliced (non-contiguous) input tensor"""
    from pytorch3dunet_amd.unet3d.model import UNet3D

    dev = torch.device("cuda", 0)
    torch.manual_seed(5)
    model = UNet3D(2, 1, f_maps=[8, 16], num_groups=4).to(dev).eval()
    base = torch.randn(1, 8, 16, 16, 2, device=dev)
    xnc = base.permute(0, 4, 1, 2, 3)  # (N,C,D,H,W) view of an NDHWC buffer
    assert not xnc.is_contiguous()
    with torch.no_grad():
        y1 = model(xnc)
        y2 = model(xnc.contiguous())
        big = torch.randn(1, 2, 10, 20, 20, device=dev)
        y3 = model(big[:, :, 1:9, 2:18, 2:18])
        y4 = model(big[:, :, 1:9, 2:18, 2:18].contiguous())
    assert torch.equal(y1, y2) and torch.equal(y3, y4)


def test_eval_no_grad_on_residual_variants_saves_no_tape():
    from pytorch3dunet_amd.unet3d.model import ResidualUNetSE3D

    dev = torch.device("cuda", 0)
    torch.manual_seed(6)
    model = ResidualUNetSE3D(1, 2, f_maps=[8, 16], num_groups=4, final_sigmoid=False).to(dev).eval()
    x = torch.randn(2, 1, 8, 16, 16, device=dev)
    with torch.no_grad():
        y = model(x)
    assert y.shape == (2, 2, 8, 16, 16) and torch.allclose(y.sum(dim=1), torch.ones_like(y[:, 0]), atol=1e-5)
    assert not y.requires_grad
    model.train()
    y2, logits = model(x, return_logits=True)
    assert torch.allclose(y2.detach(), y, atol=1e-6) and logits.requires_grad


@pytest.mark.parametrize("shape", [(2, 1, 16, 24, 32), (1, 1, 12, 22, 18)], ids=["aligned", "odd"])
def test_statistics_replica_rows_do_not_change_the_model(shape):
    """The DoubleConv executor keeps every f64 statistics table as 8 replica rows (round 6: the blocks of a persistent kernel flush a
    sample's sums at the same time, and same-address f64 atomics serialize).  With one row (`stat_reps = 1`, the plain entry points'
    layout) the same model gives the same loss, logits and gradients up to the order of a few f64 additions; the replica-aware entry
    points really are the ones that run."""
    from pytorch3dunet_amd import _native as nat

    cfg = dict(in_channels=1, out_channels=2, f_maps=16, layer_order="gcr", num_groups=4, final_sigmoid=True, num_levels=3)
    torch.manual_seed(9)
    model = _make(cfg)
    x = torch.randn(*shape)
    target = (torch.rand(shape[0], 2, *shape[2:]) > 0.5).float()
    engine = model.to(torch.device("cuda", 0))._get_engine()
    keep = engine.stat_reps
    res = {}
    try:
        for reps in (8, 1):
            engine.stat_reps = reps
            prof = nat.EventProfiler()
            nat.profiler = prof
            try:
                res[reps] = _run_native(model, x, target, "bce_dice")
            finally:
                nat.profiler = None
            names = set(prof.summary())
            assert "u3d_conv3d_ex_reps" in names and ("u3d_gn_finalize_reps" in names) == (reps > 1)
    finally:
        engine.stat_reps = keep
    (p8, l8, loss8, g8), (p1, l1, loss1, g1) = res[8], res[1]
    assert abs(loss8 - loss1) < 1e-6 * max(1.0, abs(loss1))
    assert (l8 - l1).abs().max().item() < 1e-5 * l1.abs().max().item()
    gmax = max(v.abs().max().item() for v in g1.values())
    for k in g1:
        scale = max(g1[k].abs().max().item(), 1e-3 * gmax)
        assert (g8[k] - g1[k]).abs().max().item() < 2e-4 * scale, k


def test_subpixel_decoder_path_agrees_with_virtual_concat_path():
    """The decoder first convs run the upsampled half as sub-pixel convolutions over the low-res tensor (8/27 of the
    multiply-adds, csrc/u3d_subpix.hip) whenever the upsampling is an exact 2x; with the path switched off the same layers
    run the one-kernel virtual-concat convolution.  Same loss, same parameter gradients (fp32 association only), and the
    path really is taken / not taken."""
    from pytorch3dunet_amd import _native as nat

    cfg = dict(in_channels=1, out_channels=2, f_maps=16, layer_order="gcr", num_groups=4, final_sigmoid=True, num_levels=3)
    torch.manual_seed(5)
    model = _make(cfg)
    x = torch.randn(2, 1, 16, 24, 32)
    target = (torch.rand(2, 2, 16, 24, 32) > 0.5).float()
    engine = model.to(torch.device("cuda", 0))._get_engine()
    res = {}
    for on in (True, False):
        engine.subpixel = on
        prof = nat.EventProfiler()
        nat.profiler = prof
        try:
            res[on] = _run_native(model, x, target, "bce_dice")
        finally:
            nat.profiler = None
        names = set(prof.summary())
        assert ("u3d_subpixel_conv_fwd" in names) == on and ("u3d_subpixel_conv_wgrad" in names) == on
    engine.subpixel = True
    (p1, l1, loss1, g1), (p0, l0, loss0, g0) = res[True], res[False]
    assert abs(loss1 - loss0) < 1e-5 * max(1.0, abs(loss0))
    assert (l1 - l0).abs().max().item() < 1e-4 * l0.abs().max().item()
    gmax = max(v.abs().max().item() for v in g0.values())
    for k in g0:
        # relative to the tensor's own scale, with a floor for gradients that are pure cancellation noise (the first GroupNorm's
        # gamma over a single normalised channel is ~1e-8)
        scale = max(g0[k].abs().max().item(), 1e-3 * gmax)
        assert (g1[k] - g0[k]).abs().max().item() < 2e-4 * scale, k


@pytest.mark.parametrize("name", ["UNet3D", "ResidualUNet3D", "ResidualUNetSE3D"])
def test_reference_own_model_tests_run_native_in_strict_mode(name, monkeypatch):
    """The 3-D cases of the reference's tests/test_models.py:17-69, constructor for constructor — `X(1, 1, f_maps=16,
    final_sigmoid=True)`, eval mode, `torch.rand(1, 1, 33, 65, 65)`, outputs in [0, 1] — on the native path with U3D_STRICT=1 (no
    stock-operator fallback), and equal to the module tree on the CPU for the same weights"""
    from pytorch3dunet_amd import _native as nat
    from pytorch3dunet_amd.unet3d import model as M

    monkeypatch.setenv("U3D_STRICT", "1")
    torch.manual_seed(0)
    model = getattr(M, name)(1, 1, f_maps=16, final_sigmoid=True).eval()
    x = torch.rand(1, 1, 33, 65, 65)
    with torch.no_grad():
        y_cpu = model(x)
    dev = torch.device("cuda", 0)
    model = model.to(dev)
    n0 = nat.launch_count
    with torch.no_grad():
        y = model(x.to(dev))
    assert nat.launch_count > n0
    assert torch.all(0 <= y) and torch.all(y <= 1)
    assert (y.cpu() - y_cpu).abs().max().item() < 1e-4
    # the residual variants are called WITHOUT no_grad in the reference's tests: the autograd graph must build, too
    y2 = model(x.to(dev))
    assert y2.requires_grad and torch.equal(y2.detach(), y)
