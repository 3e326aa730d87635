#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (one directory per pass) for the render kernel:
per-launch averages of every counter.  python tools/pmc_summary.py DIR [kernel name part] [launches to skip]
(the first frame of a configuration runs in image order, before any tile order exists: skip it)."""
import csv, glob, sys, collections
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
kern = sys.argv[2] if len(sys.argv) > 2 else "render_kernel<1, false>"  # the timed kernel, not its instrumented variant
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
acc = collections.defaultdict(list)
for f in sorted(glob.glob(root + "/*/*/*counter_collection.csv")):
    per = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if kern in row["Kernel_Name"]:
            per[row["Counter_Name"]].append((int(row.get("Dispatch_Id", 0) or 0), float(row["Counter_Value"])))
    for name, rows in per.items():
        rows.sort()
        acc[name].extend(v for _, v in rows[skip:])
for k, v in sorted(acc.items()):
    print(f"{k:32s} launches={len(v):3d} avg={sum(v)/len(v):.6g}")
