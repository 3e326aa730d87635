#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (one directory per pass) for the render kernel:
per-launch averages of every counter."""
import csv, glob, sys, collections
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
kern = sys.argv[2] if len(sys.argv) > 2 else "render_kernel<1, false>"  # the timed kernel, not its instrumented variant
acc = collections.defaultdict(list)
for f in sorted(glob.glob(root + "/*/*/*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        if kern in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:32s} launches={len(v):3d} avg={sum(v)/len(v):.6g}")
