#!/usr/bin/env python
"""Turn rocprofv3 output (rocpd sqlite .db or CSVs) into small committed summaries under profiles/.

  python tools/prof_summary.py stats  <results.db>  <steps_in_run>  > profiles/rNN_kernel_stats.md
  python tools/prof_summary.py pmc    <dir with *counter_collection.csv>     > profiles/rNN_pmc.md
  python tools/prof_summary.py traffic <FETCH_SIZE dir> <WRITE_SIZE dir> <steps> > profiles/rNN_pmc_traffic.json
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
        if not any(s in k for s in ("conv3d", "wgrad", "subpixel", "head_", "chan_stats", "gn_", "maxpool")):
            continue
        print(f"## `{k[:100]}`  ({len(ndisp[k])} dispatch records)")
        for c, v in sorted(agg[k].items()):
            print(f"- {c}: {v:.6g}")
        print()


FAMILIES = {  # C-ABI entry point -> substrings of the kernels it launches
    "u3d_conv3d": ("conv3d_mfma_reg_kernel", "conv3d_mfma_kernel", "splitk_reduce_kernel"),
    "u3d_conv3d_wgrad": ("conv3d_wgrad_kernel", "wgrad_reduce_kernel"),
    "u3d_subpixel_conv_fwd": ("subpixel_fwd_kernel",),
}


def traffic(fetch_dir, write_dir, steps):
    """HBM traffic per entry-point family from two separate --pmc passes (FETCH_SIZE, WRITE_SIZE; both in KiB).
    MI355X_MICROARCH.md §HBM: on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, i.e. reports 1/2 of a wide coalesced
    read -> doubled here (confirmed on gn_bwd_apply_kernel, a pure 16 B/lane stream: raw FETCH == WRITE although it
    reads two tensors and writes one).  WRITE_SIZE is taken as is (matches the known byte counts of the streaming
    kernels)."""
    import json

    def load(d, ctr):
        out = defaultdict(lambda: [0, 0.0])
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    if row["Counter_Name"] != ctr:
                        continue
                    out[row["Kernel_Name"]][0] += 1
                    out[row["Kernel_Name"]][1] += float(row["Counter_Value"])
        return out

    fe, wr = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    res = {"steps": steps, "unit": "bytes", "fetch_correction": 2.0, "families": {}, "kernels": {}}
    for k in sorted(fe, key=lambda k: -fe[k][1]):
        fb, wb = fe[k][1] * 1024 * 2.0, wr.get(k, [0, 0.0])[1] * 1024
        if fb + wb > 1e7:
            res["kernels"][k[:80]] = {"launches": fe[k][0], "fetch_bytes": fb, "write_bytes": wb}
    for fam, subs in FAMILIES.items():
        ks = [k for k in fe if any(s in k for s in subs)]
        main = [k for k in ks if "reduce" not in k]
        fb = sum(fe[k][1] for k in ks) * 1024 * 2.0
        wb = sum(wr.get(k, [0, 0.0])[1] for k in ks) * 1024
        n = sum(fe[k][0] for k in main)
        res["families"][fam] = {"launches": n, "launches_per_step": n / steps, "fetch_bytes": fb, "write_bytes": wb,
                                "bytes_per_launch": (fb + wb) / max(n, 1), "bytes_per_step": (fb + wb) / steps}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], int(sys.argv[3]))
    elif sys.argv[1] == "traffic":
        traffic(sys.argv[2], sys.argv[3], int(sys.argv[4]))
    else:
        pmc(sys.argv[2])
