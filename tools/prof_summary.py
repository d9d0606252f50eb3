#!/usr/bin/env python
"""Turn rocprofv3 output (rocpd sqlite .db or CSVs) into small committed summaries under profiles/.

  python tools/prof_summary.py stats  <results.db>  <steps_in_run>  > profiles/rNN_kernel_stats.md
  python tools/prof_summary.py pmc    <dir with *counter_collection.csv>     > profiles/rNN_pmc.md
"""
import csv
import glob
import os
import sqlite3
import sys
from collections import defaultdict


def stats(db_path, steps):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    # durations are in ns in rocpd 'top_kernels'? detect: bench step ~30 ms
    tot = sum(r[2] for r in rows)
    unit = 1e-3  # values are microseconds when total is ~1e5 for a few 30 ms steps; convert to ms
    if tot > 1e8:
        unit = 1e-6  # nanoseconds
    print(f"# rocprofv3 --kernel-trace --stats summary ({os.path.basename(db_path)}, {steps} bench steps incl. warm-up)\n")
    print("| kernel | calls | total ms | avg us | ms/step | % |")
    print("|---|---|---|---|---|---|")
    for name, calls, total, avg, pct in rows[:30]:
        print(f"| `{name[:90]}` | {calls} | {total * unit:.3f} | {avg * unit * 1e3:.1f} | {total * unit / steps:.3f} | {pct:.2f} |")
    print(f"\ntotal kernel time {tot * unit:.2f} ms = {tot * unit / steps:.2f} ms/step")


def pmc(d):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    agg = defaultdict(lambda: defaultdict(float))
    ndisp = defaultdict(set)
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name") or row.get("kernel_name")
                c = row.get("Counter_Name") or row.get("counter_name")
                v = float(row.get("Counter_Value") or row.get("counter_value") or 0)
                agg[k][c] += v
                ndisp[k].add((f, row.get("Dispatch_Id") or row.get("dispatch_id")))
    print(f"# rocprofv3 --pmc summary ({d}); counter sums over all dispatches of a kernel\n")
    for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", agg[k].get("GRBM_GUI_ACTIVE", 0))):
        if not any(s in k for s in ("conv3d", "wgrad", "head_", "chan_stats", "gn_", "maxpool")):
            continue
        print(f"## `{k[:100]}`  ({len(ndisp[k])} dispatch records)")
        for c, v in sorted(agg[k].items()):
            print(f"- {c}: {v:.6g}")
        print()


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], int(sys.argv[3]))
    else:
        pmc(sys.argv[2])
