#!/usr/bin/env python
"""Turn rocprofv3 output (rocpd sqlite .db or CSVs) into small committed summaries under profiles/.

  python tools/prof_summary.py stats  <results.db>  <steps_in_run>  > profiles/rNN_kernel_stats.md
  python tools/prof_summary.py pmc    <dir with *counter_collection.csv>     > profiles/rNN_pmc.md
  python tools/prof_summary.py traffic <FETCH_SIZE dir> <WRITE_SIZE dir> <steps> > profiles/rNN_pmc_traffic.json
  python tools/prof_summary.py tables <results.db> <steps_in_run> <SQ counter dir> <bench.json.log> > profiles/rNN_tables.md
      -> the two tables DESIGN.md quotes, GENERATED: per-family roofline fraction from the rocprofv3 kernel durations (family
         FLOPs per step taken from the bench line of the same build) and per-kernel MFMA pipe utilisation from the SQ counters
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
        if not any(s in k for s in ("conv3d", "wgrad", "subpixel", "head_", "chan_stats", "gn_", "maxpool", "gconv", "nearest_", "pack_")):
            continue
        print(f"## `{k[:100]}`  ({len(ndisp[k])} dispatch records)")
        for c, v in sorted(agg[k].items()):
            print(f"- {c}: {v:.6g}")
        print()


FAMILIES = {  # C-ABI entry point -> substrings of the kernels it launches
    "u3d_conv3d": ("conv3d_mfma_reg_kernel", "conv3d_mfma_kernel", "splitk_reduce_kernel"),
    "u3d_conv3d_wgrad": ("conv3d_wgrad_kernel", "wgrad_reduce_kernel"),
    "u3d_subpixel_conv_fwd": ("subpixel_fwd_kernel",),
    "u3d_conv3d_bf16_ex": ("conv3d_bf16_kernel", "splitk_bf16_reduce_kernel"),
    "u3d_conv3d_wgrad_bf16": ("conv3d_wgrad_bf16_kernelILi3", "conv3d_wgrad_b16v2_kernel", "wgrad_bf16_reduce"),
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


ROOFLINE_FAMILIES = {  # bench.py family (entry points merged when they launch the same kernels) -> kernel-name substrings
    "u3d_conv3d": (("u3d_conv3d",), ("conv3d_mfma_reg_kernel", "conv3d_mfma_kernel", "splitk_reduce_kernel")),
    "u3d_conv3d_wgrad (+_strided)": (("u3d_conv3d_wgrad", "u3d_conv3d_wgrad_strided"), ("conv3d_wgrad_kernel", "wgrad_reduce_kernel")),
    "u3d_subpixel_conv_fwd": (("u3d_subpixel_conv_fwd",), ("subpixel_fwd_kernel", "sum_partials_kernel")),
    "u3d_subpixel_conv_dgrad": (("u3d_subpixel_conv_dgrad",), ("subpixel_dgrad_kernel",)),
    "u3d_subpixel_conv_wgrad": (("u3d_subpixel_conv_wgrad",), ("subpixel_wgrad_kernel", "subpixel_wgrad_reduce_kernel")),
}
PEAK_F32_MFMA = 157.3


def tables(db_path, steps, sq_dir, bench_log):
    import json

    bench = json.loads([ln for ln in open(bench_log).read().splitlines() if ln.startswith("{")][-1])
    fam_bench = bench["roofline"]["families"]
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, total_calls, total_duration from top_kernels").fetchall()
    tot = sum(r[2] for r in rows)
    unit = 1e-6 if tot > 1e8 else 1e-3  # -> ms
    print(f"# Generated by tools/prof_summary.py tables — do not edit ({os.path.basename(db_path)}; bench line: "
          f"{bench['value']} {bench['unit']}, {bench['ms_per_step']} ms/step)\n")
    print("## Per-family fp32-MFMA roofline (FLOPs per step from the bench line's HIP-event table, time from rocprofv3 kernel durations)\n")
    print("| family | GFLOP/step | rocprofv3 ms/step | TFLOP/s | frac of 157.3 | HIP-event ms/step | HIP-event frac |")
    print("|---|---|---|---|---|---|---|")
    all_gf = all_ms = 0.0
    for fam, (entries, subs) in ROOFLINE_FAMILIES.items():
        gf = sum(fam_bench[e]["tflops"] * fam_bench[e]["ms_per_step"] for e in entries if e in fam_bench and fam_bench[e]["tflops"])
        ev_ms = sum(fam_bench[e]["ms_per_step"] for e in entries if e in fam_bench)
        ms = sum(r[2] for r in rows if any(sub in r[0] for sub in subs)) * unit / steps
        if gf <= 0 or ms <= 0:
            continue
        all_gf += gf
        all_ms += ms
        print(f"| `{fam}` | {gf:.1f} | {ms:.3f} | {gf / ms:.1f} | {gf / ms / PEAK_F32_MFMA:.3f} | {ev_ms:.3f} | {gf / ev_ms / PEAK_F32_MFMA:.3f} |")
    print(f"| all MFMA families | {all_gf:.1f} | {all_ms:.3f} | {all_gf / all_ms:.1f} | {all_gf / all_ms / PEAK_F32_MFMA:.3f} | | |")
    print(f"\nwhole step (rocprofv3): {tot * unit / steps:.2f} ms of kernels; executed {all_gf / (tot * unit / steps):.1f} TFLOP/s = "
          f"{all_gf / (tot * unit / steps) / PEAK_F32_MFMA:.3f} of the fp32-MFMA peak\n")
    # MFMA pipe utilisation from the SQ counters
    files = glob.glob(os.path.join(sq_dir, "**", "*counter_collection.csv"), recursive=True)
    agg = defaultdict(lambda: defaultdict(float))
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                agg[row.get("Kernel_Name") or row.get("kernel_name")][row.get("Counter_Name") or row.get("counter_name")] += float(
                    row.get("Counter_Value") or row.get("counter_value") or 0)
    print("## MFMA pipe utilisation (SQ counters; busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128): per SIMD and active cycle)\n")
    print("| kernel | MFMA pipe busy | VALU : MFMA instructions | SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES |")
    print("|---|---|---|---|")
    for k in sorted(agg, key=lambda k: -agg[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0)):
        a = agg[k]
        if a.get("SQ_INSTS_MFMA", 0) <= 0 or a.get("GRBM_GUI_ACTIVE", 0) <= 0:
            continue
        print(f"| `{k[:80]}` | {100 * a['SQ_VALU_MFMA_BUSY_CYCLES'] / (a['GRBM_GUI_ACTIVE'] * 128):.1f} % | "
              f"{a['SQ_INSTS_VALU'] / a['SQ_INSTS_MFMA']:.2f} | {a.get('SQ_WAIT_INST_ANY', 0) / max(a.get('SQ_WAVE_CYCLES', 1), 1):.2f} |")


if __name__ == "__main__":
    if sys.argv[1] == "tables":
        tables(sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5])
    elif sys.argv[1] == "stats":
        stats(sys.argv[2], int(sys.argv[3]))
    elif sys.argv[1] == "traffic":
        traffic(sys.argv[2], sys.argv[3], int(sys.argv[4]))
    else:
        pmc(sys.argv[2])
