"""Per-wave timelines of u3d_conv3d from the instrumented kernel twin (u3d_set_profile_buffer): where do the cycles of
a block go (prologue / k-loops / inter-chunk gaps / epilogue) and how busy is each SIMD's MFMA pipe.

    python tools/wave_timeline.py [layer ...]        layers: names of tools/layer_bench.py, suffix :dgrad for dgrad
"""
import ctypes
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("pytorch-3dunet_amd", "tests", "tools"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch  # noqa: E402

from pytorch3dunet_amd import _native as nat  # noqa: E402
from pytorch3dunet_amd.engine import VSrc, _p, _stream  # noqa: E402
import gpu_utils as U  # noqa: E402
from layer_bench import LAYERS  # noqa: E402

dev = U.DEV
N, D0, H0, W0 = 2, 64, 128, 128
for _kv in os.environ.get("U3D_TUNE", "").split(","):  # e.g. U3D_TUNE=6:1 = one block per CU
    if ":" in _kv:
        nat.call("u3d_set_tuning", int(_kv.split(":")[0]), int(_kv.split(":")[1]))


def run(name, dgrad, forced_nt=0, dims=None, abl=0):
    (_, C0, C1, Cout, lvl) = next(x for x in LAYERS if x[0] == name)
    global N
    D, H, W = D0 >> lvl, H0 >> lvl, W0 >> lvl
    N = 2
    if dims:
        N, D, H, W = dims
    Cin = C0 + C1
    t0 = torch.randn(N, D, H, W, C0, device=dev)
    t1 = torch.randn(N, D // 2, H // 2, W // 2, C1, device=dev) if C1 else None
    src = VSrc(t0, t1)
    aff = torch.randn(N, Cin, 2, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, 3, device=dev) / (27 * Cin) ** 0.5
    nat.call("u3d_set_tuning", 0, forced_nt)
    if not dgrad:
        wp = U.pack(w, 0)
        y = torch.empty((N, D, H, W, Cout), device=dev)
        st = torch.zeros((N, Cout, 2), dtype=torch.float64, device=dev)
        s = src.struct(aff)
        fn = lambda: nat.call("u3d_conv3d", 0, _stream(dev), ctypes.byref(s), _p(wp), _p(y), N, D, H, W, Cout, 1, _p(st), None, None)  # noqa: E731
        kin, kout = Cin, Cout
    else:
        dz = torch.randn(N, D, H, W, Cout, device=dev)
        wpd = U.pack(w, 1)
        dg = torch.empty((N, D, H, W, Cin), device=dev)
        gst = torch.zeros((N, Cin, 2), dtype=torch.float64, device=dev)
        s_dz = VSrc(dz).struct()
        s_x = src.struct()
        fn = lambda: nat.call("u3d_conv3d", 0, _stream(dev), ctypes.byref(s_dz), _p(wpd), _p(dg), N, D, H, W, Cin, 0, None,  # noqa: E731
                              ctypes.byref(s_x), _p(gst))
        kin, kout = Cout, Cin
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    nrec = 4 * 65536
    buf = torch.zeros((nrec, 24), dtype=torch.int64, device=dev)
    nat.call("u3d_set_profile_buffer", _p(buf), buf.numel() * 8)
    nat.call("u3d_set_tuning", 2, abl)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    nat.call("u3d_set_profile_buffer", None, 0)
    nat.call("u3d_set_tuning", 2, 0)
    nat.call("u3d_set_tuning", 0, 0)
    ms = e0.elapsed_time(e1)
    r = buf.cpu()
    r = r[r[:, 3] != 0]
    nw = r.shape[0]
    nch = int(r[0, 7])
    entry, staged, epi, exit_ = r[:, 3], r[:, 8], r[:, 5], r[:, 6]
    span_all = (exit_.max() - entry.min()).item()
    flops = 54.0 * kin * kout * N * D * H * W
    print(f"== {name}{':dgrad' if dgrad else ''}{' abl=%d' % abl if abl else ''} {kin}->{kout} @{D}x{H}x{W}: {ms:.3f} ms, {flops / ms / 1e9:.1f} TF; {nw} waves, "
          f"{nch} chunks; counter span {span_all} ticks = {span_all / ms / 1e6:.3f} GHz-equivalent")
    # MFMA cycles per wave: total MFMAs (incl. padded lanes) = nchunks*54*8*NT ; infer NT from grid
    ntot = (kout + 31) // 32
    ntiles = N * ((D + 3) // 4) * ((H + 7) // 8) * ((W + 7) // 8)
    if ncb_hint := 0:
        pass
    nblk = nw // 4
    done = r[:, 4].double().clamp(min=1)  # tiles walked by each wave (persistent kernel), 1 for the generic kernel
    items = int(done.sum().item()) // 4
    ncb = max(1, items // ntiles)
    NT = ntot // ncb
    mfma_cyc = nch * 54 * 8 * NT * 64
    life = (exit_ - entry).double()
    pro = (staged - entry).double()
    ep = (exit_ - epi).double()
    kl, gaps = [], []
    if (r[:, 22] != 0).all() and (r[:, 23] != 0).all():
        e_dur = (r[:, 22] - r[:, 5]).double()
        e_gap = (r[:, 23] - r[:, 22]).double()
        print(f"   epilogue of the stamped tile: {e_dur.mean():.0f} ticks (min {e_dur.min():.0f} max {e_dur.max():.0f}); from its end to the "
              f"next tile's first k-loop: {e_gap.mean():.0f}")
    for c in range(min(nch, 7)):
        kl.append((r[:, 9 + 2 * c] - r[:, 8 + 2 * c]).double())
        if c > 0:
            gaps.append((r[:, 8 + 2 * c] - r[:, 7 + 2 * c]).double())
    # gap between chunks = (start of next k-loop) - (end of this) is folded into next chunk's duration here; report
    # chunk durations instead
    print(f"   NT={NT} blocks={nblk} tiles/block {done.mean():.1f}  ideal MFMA cycles/wave/tile {mfma_cyc}  | mean lifetime {life.mean():.0f} "
          f"(prologue {pro.mean():.0f} = {100 * pro.mean() / life.mean():.1f}%, epilogue {ep.mean():.0f} = "
          f"{100 * ep.mean() / life.mean():.1f}%)")
    print("   k-loop durations (mean ticks): " + " ".join(f"{k.mean():.0f}" for k in kl) +
          f"   [ideal if alone on the SIMD: {54 * 8 * NT * 64}]; restage gaps: " + " ".join(f"{k.mean():.0f}" for k in gaps))
    # per-SIMD utilisation
    key = (r[:, 2] & 0xF) * 65536 + ((r[:, 1] >> 4) & 0xFFF)
    groups = defaultdict(list)
    for i, k in enumerate(key.tolist()):
        groups[k].append(i)
    utils, conc = [], []
    for k, idx in groups.items():
        idx_t = torch.tensor(idx)
        s0, s1 = entry[idx_t].min().item(), exit_[idx_t].max().item()
        utils.append(done[idx_t].sum().item() * mfma_cyc / (s1 - s0))
        conc.append(life[idx_t].sum().item() / (s1 - s0))
    # which blocks share a SIMD?  (the dispatcher's placement decides who must be staggered against whom)
    pairs = defaultdict(int)
    for k, idx in list(groups.items()):
        blks = sorted(set(int(r[i, 0]) for i in idx))
        if len(blks) == 2:
            pairs[blks[1] - blks[0]] += 1
    top = sorted(pairs.items(), key=lambda kv: -kv[1])[:6]
    print(f"   co-resident block-id differences (count of SIMDs): {top}; first-k-loop start spread per SIMD pair (ticks): "
          f"{torch.tensor([abs(int(staged[i[0]]) - int(staged[i[-1]])) for i in groups.values() if len(i) >= 2]).double().mean():.0f}")
    ut = torch.tensor(utils)
    print(f"   SIMDs seen {len(groups)}; waves/SIMD {nw / len(groups):.1f}; MFMA-pipe utilisation per SIMD (ticks basis): "
          f"mean {ut.mean():.3f} min {ut.min():.3f} max {ut.max():.3f}; mean concurrent waves/SIMD {sum(conc) / len(conc):.2f}")
    # kernel-level: first entry to last exit per XCD
    for x in sorted(set((r[:, 2] & 0xF).tolist())):
        m = (r[:, 2] & 0xF) == x
        print(f"   xcc {x}: waves {int(m.sum())} span {(exit_[m].max() - entry[m].min()).item()} first-entry spread "
              f"{(entry[m].max() - entry[m].min()).item()}", end=";")
    print()


if __name__ == "__main__":
    layers = sys.argv[1:] or ["dec2.c1", "dec2.c1:dgrad", "dec2.c2", "enc0.c2", "enc0.c2:dgrad", "dec1.c1"]
    for spec in layers:
        parts = spec.split(":")
        name = parts[0]
        dgrad = "dgrad" in parts[1:]
        nt, dims, abl = 0, None, 0
        for q in parts[1:]:
            if q.startswith("nt"):
                nt = int(q[2:])
            if q.startswith("dims"):
                dims = tuple(int(v) for v in q[4:].split("x"))
            if q.startswith("abl"):
                abl = int(q[3:])
        run(name, dgrad, nt, dims, abl)
