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

# ---- derived figures (MI355X: 256 CUs x 4 SIMDs, 8 XCDs; SQ cycle counters tick once per 4 clocks) ----
m = {k: sum(v) / len(v) for k, v in acc.items()}
def have(*names):
    return all(n in m for n in names)
print("# derived")
if have("GRBM_GUI_ACTIVE"):
    clocks = m["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
    print(f"# kernel clocks per launch            {clocks:.4g}  (= {clocks / 2.4e6:.2f} ms at 2.4 GHz)")
    simd_quads = 1024.0 * clocks / 4.0
    if have("SQ_ACTIVE_INST_VALU"):
        print(f"# VALU-active wave time / SIMD time   {m['SQ_ACTIVE_INST_VALU'] / simd_quads:.3f}  (instructions of different waves overlap in the pipeline: ~1 = saturated)")
    if have("SQ_ACTIVE_INST_SCA"):
        print(f"# scalar-active wave time / SIMD time {m['SQ_ACTIVE_INST_SCA'] / simd_quads:.3f}")
    if have("SQ_WAVE_CYCLES"):
        print(f"# waves resident per SIMD             {m['SQ_WAVE_CYCLES'] / simd_quads:.2f}")
    if have("TA_BUSY_avr"):
        print(f"# L1 / texture-address path busy      {m['TA_BUSY_avr'] / clocks:.3f}")
    if have("TCC_REQ_sum"):
        print(f"# L2 requests                         {m['TCC_REQ_sum'] * 128.0 / (clocks / 2.4e9) / 1e9:.0f} GB/s at 128 B each (upper bound: 16-B gathers fill little of a line)")
    if have("FETCH_SIZE", "WRITE_SIZE"):
        print(f"# HBM-side traffic                    {(m['FETCH_SIZE'] + m['WRITE_SIZE']) * 1024.0 / (clocks / 2.4e9) / 1e9:.0f} GB/s of 8000")
if have("SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU"):
    print(f"# VALU lane utilisation               {m['SQ_THREAD_CYCLES_VALU'] / (64.0 * m['SQ_ACTIVE_INST_VALU']):.3f}")
if have("SQ_WAIT_ANY", "SQ_WAVE_CYCLES"):
    print(f"# wave cycles waiting on memory/LDS   {m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']:.3f}")
if have("SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"):
    print(f"# wave cycles waiting to issue        {m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES']:.3f}")
if have("TCC_HIT_sum", "TCC_REQ_sum"):
    print(f"# L2 hit rate                         {m['TCC_HIT_sum'] / m['TCC_REQ_sum']:.3f}")
if have("TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum"):
    print(f"# L1 hit rate                         {1.0 - m['TCP_TCC_READ_REQ_sum'] / m['TCP_TOTAL_CACHE_ACCESSES_sum']:.3f}")
if have("SQ_INSTS_VALU", "GRBM_GUI_ACTIVE"):
    # every vector instruction occupies its SIMD's issue port for 4 clocks (a 64-wide wave on 16 lanes)
    print(f"# VALU issue slots used               {m['SQ_INSTS_VALU'] * 4.0 / (1024.0 * m['GRBM_GUI_ACTIVE'] / 8.0):.3f}")
if have("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32"):
    a, mu, f, t = (m[k] for k in ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32"))
    print(f"# FP32 wave instructions              add {a:.4g} mul {mu:.4g} fma {f:.4g} transcendental {t:.4g}  (x 64 lanes, fma = 2 flop: {(a + mu + 2 * f + t) * 64 / 1e9:.1f} Gflop issued per launch)")
