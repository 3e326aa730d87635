#!/usr/bin/env python3
"""Fold PMC summaries (tools/pmc_summary.py output) into profiles/pmc_traffic.json, the per-launch
figures bench.py attaches to its roofline object.  python tools/pmc_to_json.py key=summary.txt ...
(key = workload:pipeline, e.g. c3:pooled)."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
data = json.load(open(path)) if os.path.exists(path) else {}
for arg in sys.argv[1:]:
    key, f = arg.split("=", 1)
    txt = open(f).read()
    avg = {m.group(1): float(m.group(2)) for m in re.finditer(r"^(\w+)\s+launches=\s*\d+\s+avg=([0-9.e+-]+)", txt, re.M)}
    der = {m.group(1).strip(): float(m.group(2)) for m in re.finditer(r"^# ([A-Za-z0-9 /\-]+?)\s{2,}([0-9.]+)", txt, re.M)}
    rec = {"source": os.path.relpath(f, ROOT),
           "note": "rocprofv3 --pmc, one pass per counter group, per launch of the timed kernel; FETCH_SIZE / WRITE_SIZE in KB as "
                   "reported (FETCH_SIZE uncorrected: these are 16-B per-lane gathers, for which the gfx950 x2 rule of wide streams "
                   "is not calibrated)"}
    for k in ("FETCH_SIZE", "WRITE_SIZE"):
        if k in avg:
            rec[k + "_KB"] = avg[k]
    names = {"VALU-active wave time / SIMD time": "valu_active_frac", "VALU lane utilisation": "lane_utilisation", "L2 hit rate": "l2_hit",
             "VALU issue slots used": "valu_issue_frac", "L1 / texture-address path busy": "ta_busy",
             "wave cycles waiting on memory/LDS": "wait_frac"}
    for long, short in names.items():
        if long in der:
            rec[short] = der[long]
    if "SQ_INSTS_VALU_FMA_F32" in avg:
        rec["fp32_wave_insts"] = {k: avg["SQ_INSTS_VALU_" + k.upper() + "_F32"] for k in ("add", "mul", "fma", "trans")}
    if "SQ_INSTS_VALU" in avg:
        rec["valu_wave_insts"] = avg["SQ_INSTS_VALU"]
    data[key] = rec
json.dump(data, open(path, "w"), indent=1)
print("wrote", path, sorted(data))
