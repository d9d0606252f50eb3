"""Developer aid: one golden (tests/golden) under the ragged persistent kernels (tuning key 3 = 0) and under the round-4 gate (key 3 = 2:\ngeneric kernels), per-parameter gradient distance to the reference fixture.  python tools/diag_golden.py g3_unet3d_regression"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("tests", "pytorch-3dunet_amd", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, d))
import torch
from conftest import Golden
from pytorch3dunet_amd import _native as nat
import unet3d_oracle as orc
from test_gpu_model import _run_native
name = sys.argv[1] if len(sys.argv) > 1 else "g3_unet3d_regression"
g = Golden(name)
res = {}
for gate in (2, 0):
    nat.call("u3d_set_tuning", 3, gate)
    model = g.build_model()
    x, target = g.inputs()
    probs, logits, loss, grads = _run_native(model, x, target, g.loss_name)
    res[gate] = (logits, grads)
    print("gate", gate, "logits rel", orc.rel_err(logits, g.tensor("logits")))
    for k, rg in g.group("grad/").items():
        print("   %-70s %.3e" % (k, orc.rel_err(grads[k], rg)))
nat.call("u3d_set_tuning", 3, 0)

# ---- per-layer: the engine's debug taps (dz, dg of every SingleConv) under both gates
taps = {}
for gate in (2, 0):
    nat.call("u3d_set_tuning", 3, gate)
    model = g.build_model().to("cuda").train()
    eng = model._get_engine()
    eng.debug = {}
    x, target = g.inputs()
    from test_gpu_model import loss_by_name
    probs, logits = model(x.cuda(), return_logits=True)
    loss_by_name(g.loss_name, probs, logits, target.cuda()).backward()
    torch.cuda.synchronize()
    taps[gate] = {k: v.detach().float().cpu() for k, v in eng.debug.items() if torch.is_tensor(v)}
    taps[gate].update({"grad/" + k: p.grad.detach().cpu() for k, p in model.named_parameters()})
nat.call("u3d_set_tuning", 3, 0)
print("---- ragged (gate 0) vs generic (gate 2), max|a-b| / max|b| per debug tap, in recording order")
for k in taps[2]:
    if k in taps[0] and taps[0][k].shape == taps[2][k].shape:
        a, b = taps[0][k], taps[2][k]
        e = orc.rel_err(a, b)
        bad = (a - b).abs() > 1e-4 * b.abs().max()
        print("   %-60s %.3e   shape %s  bad elems %d  first bad %s" % (k, e, tuple(a.shape), int(bad.sum()), bad.nonzero()[:3].tolist() if bad.any() else ""))
