#!/usr/bin/env python
"""Per bench STEP (a step starts at a launch of the one-per-step weight packer) the wall time and the kernel time of the MFMA families, from a
`rocprofv3 --kernel-trace --output-format csv` trace — shows which steps of a run are steady state (the first steps after an idle phase run
on ramping clocks) and what the dominant family's time is THERE (tools/prof_summary.py averages over every step of the trace).

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 20 --warmup 5
    python tools/step_family_times.py gpurun_out/trace > profiles/rNN_step_family_times.txt
"""
import csv
import glob
import os
import sys

FAMILIES = [  # (label, substrings of the kernel name, FLOPs per step in G for the fraction of the fp32 peak; bench workload)
    ("u3d_conv3d", ("conv3d_mfma_reg_kernel", "conv3d_mfma_kernel", "splitk_reduce_kernel"), 1079.9),
    ("wgrad", ("conv3d_wgrad_kernel", "wgrad_reduce_kernel"), 539.9),
    ("subpixel", ("subpixel_",), 360.6),
]


def main():
    d = sys.argv[1]
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if "pack_weights_" in r[2]]
    print(f"# {len(rows)} kernel records, {len(starts)} steps (marker: the weight packer); fractions of the 157.3 TFLOP/s fp32-MFMA peak")
    print("# step  launches  wall ms  kernels ms | " + " | ".join(f"{lab} ms (frac)" for lab, _, _ in FAMILIES) + " | other ms")
    for si, (a, b) in enumerate(zip(starts[:-1], starts[1:])):
        seg = rows[a:b]
        wall = (rows[b][0] - seg[0][0]) / 1e6
        ksum = sum(e - s for s, e, _ in seg) / 1e6
        fam = []
        rest = ksum
        for lab, pats, gf in FAMILIES:
            ms = sum(e - s for s, e, n in seg if any(p in n for p in pats)) / 1e6
            rest -= ms
            fam.append(f"{ms:7.3f} ({gf / ms / 157.3:.3f})" if ms > 0 else "   -   ")
        print(f"{si:5d} {len(seg):8d} {wall:8.3f} {ksum:10.3f} | " + " | ".join(fam) + f" | {rest:6.3f}")


if __name__ == "__main__":
    main()
