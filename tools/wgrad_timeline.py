"""Wave timeline of the bf16-storage weight gradient (conv3d_wgrad_b16v2_kernel), from a TRACE build of the library:

    bash tools/ab_libs.sh u3d_bf16.hip trace "-DU3D_WG_TRACE"          # + any -DU3D_WG_ABLATE=.. to combine
    U3D_LIB_PATH=$PWD/pytorch-3dunet_amd/pytorch3dunet_amd/lib/libu3d_hip_trace.so python tools/wgrad_timeline.py [--level 0]

The trace build stamps s_memtime (core clock) at 14 points of tiles 4 .. 15 of every wave of the middle block and dumps them into
that block's workspace region instead of its partial sums (results are wrong; timing only).  Stamps: 0 loop top, 1 after the barrier, then
per part p = 0..3: 2+3p after the loads were issued, 3+3p after the part's 16 steps of fragment reads + MFMAs, 4+3p after the
LDS stores of the batch written in that part.  Every stamp drains the wave's LDS queue (s_memtime is a scalar-memory read that the
compiler waits for with lgkmcnt(0)): the fragment ring restarts 4 times per tile, so read the numbers as an upper bound.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pytorch-3dunet_amd"))
from pytorch3dunet_amd import _native as nat  # noqa: E402
from pytorch3dunet_amd.engine import _p, _stream  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--level", type=int, default=0)
    ap.add_argument("--fmaps", type=int, default=64)
    ap.add_argument("--patch", default="80,160,160")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = nat.get_lib()
    D, H, W = (int(v) >> args.level for v in args.patch.split(","))
    C = args.fmaps << args.level
    N = 1
    x = torch.randn(N, D, H, W, C, device=dev).to(torch.bfloat16)
    dz = torch.randn(N, D, H, W, C, device=dev).to(torch.bfloat16)
    aff = torch.randn(N, C, 2, device=dev)
    dw = torch.empty(C, C, 3, 3, 3, device=dev)
    nw = lib.u3d_wgrad_bf16_workspace_floats(N, D, H, W, C, C)
    ws = torch.zeros(nw, device=dev)
    for _ in range(3):
        nat.call("u3d_conv3d_wgrad_bf16_b16", 0, _stream(dev), _p(x), _p(aff), _p(dz), _p(dw), N, D, H, W, C, C, _p(ws), nw)
    torch.cuda.synchronize()
    blocks = nw // (27 * 2048)  # S * P; the trace build stamps the block with (XCD-remapped) index blocks / 2
    off = (blocks // 2) * 27 * 2048
    st = ws[off: off + 8 * 12 * 14].view(torch.int32).cpu().view(8, 12, 14).long() & 0xFFFFFFFF
    if int(st.max()) == 0:
        print("no stamps: not a -DU3D_WG_TRACE build of the library (U3D_LIB_PATH)")
        return
    names = ["barrier"] + [f"{k}{p}" for p in range(4) for k in ("ld", "mfma", "st")]
    print(f"level {args.level}: {C}->{C} @{D}x{H}x{W}; cycles between consecutive stamps, median over tiles 4..15 of the middle block")
    print("wave " + " ".join(f"{n:>7s}" for n in names) + "   tile")
    for w in range(8):
        d = (st[w, :, 1:] - st[w, :, :-1]) & 0xFFFFFFFF
        med = d.median(dim=0).values
        tile = ((st[w, 1:, 0] - st[w, :-1, 0]) & 0xFFFFFFFF).median()
        print(f"{w:4d} " + " ".join(f"{int(v):7d}" for v in med) + f" {int(tile):6d}")
    t0 = st[:, :, 0]
    print("loop-top stamp of tile 8, relative to wave 0:", [int(v) for v in (t0[:, 4] - t0[0, 4])])


if __name__ == "__main__":
    main()
