#!/usr/bin/env python
"""Where a bench step's wall time goes between kernels: from a `rocprofv3 --kernel-trace --output-format csv` trace compute, per
step (a step starts at a launch of --marker, default the one-per-step weight packer, "pack_weights_"), the wall time first start -> next step's
first start, the sum of kernel durations, the idle time between consecutive kernels and which kernels the idle time follows.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 3 --warmup 3
    python tools/gap_analysis.py gpurun_out/trace [--marker pack_weights_batch_kernel] > profiles/rNN_gap_analysis.md
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(?:<[^(]{0,60}>)?)", name)
    s = m.group(1) if m else name[:60]
    s = s.replace("at::native::", "")
    return s[:70]


def main():
    d = sys.argv[1]
    marker = "pack_weights_"
    if "--marker" in sys.argv:
        marker = sys.argv[sys.argv.index("--marker") + 1]
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if marker in r[2]]
    if "--list" in sys.argv and len(starts) >= 2:
        # every launch of the last complete step, in order, with its duration (the trace's durations include the dispatch gap:
        # consecutive kernels of one stream start where the previous one ended)
        a, b = starts[-2], starts[-1]
        print(f"# launches of one bench step in order ({b - a} kernels, {(rows[b][0] - rows[a][0]) / 1e6:.3f} ms)\n")
        for i in range(a, b):
            s, e, n = rows[i]
            print(f"{i - a:3d} {(e - s) / 1000:8.1f} us  {short(n)}")
        return
    if len(starts) < 3:
        print(f"only {len(starts)} marker launches ({marker}) among {len(rows)} kernels")
        return
    print(f"# idle time between kernels per bench step ({len(rows)} kernel records, {len(starts)} steps by marker `{marker}`)\n")
    print("| step | launches | wall ms | kernel ms | idle ms | overlap ms |")
    print("|---|---|---|---|---|---|")
    gap_after = defaultdict(lambda: [0, 0.0])
    gap_hist = defaultdict(int)
    last_steps = list(zip(starts[:-1], starts[1:]))[-3:]
    for si, (a, b) in enumerate(last_steps):
        seg = rows[a:b + 1]  # up to and including the next step's first kernel
        wall = (seg[-1][0] - seg[0][0]) / 1e6
        ksum = sum(e - s for s, e, _ in seg[:-1]) / 1e6
        idle = 0.0
        over = 0.0
        cur_end = seg[0][1]
        for j in range(1, len(seg)):
            g = seg[j][0] - cur_end
            if g > 0:
                idle += g / 1e6
                k = short(seg[j - 1][2])
                gap_after[k][0] += 1
                gap_after[k][1] += g / 1e6
                gap_hist[min(int(g / 1000), 20)] += 1
            else:
                over += -g / 1e6 if seg[j][1] > cur_end else (seg[j][1] - seg[j][0]) / 1e6
            cur_end = max(cur_end, seg[j][1])
        print(f"| {si} | {len(seg) - 1} | {wall:.3f} | {ksum:.3f} | {idle:.3f} | {over:.3f} |")
    n = len(last_steps)
    print("\n## idle time by the kernel it FOLLOWS (per step, average over the steps above)\n")
    print("| kernel before the gap | gaps/step | idle ms/step | avg gap us |")
    print("|---|---|---|---|")
    for k, (c, ms) in sorted(gap_after.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"| `{k}` | {c / n:.1f} | {ms / n:.4f} | {1000 * ms / c:.2f} |")
    print("\n## gap histogram (us -> count per step)\n")
    print(", ".join(f"{k}{'+' if k == 20 else ''}us: {v / n:.1f}" for k, v in sorted(gap_hist.items())))


if __name__ == "__main__":
    main()
