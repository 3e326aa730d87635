#!/usr/bin/env python3
"""Where a frame launched alone spends its end: when every pixel started and ended, and what it cost.  Needs the library
built with the diagnostic compiled in, which stores those three numbers where the primary-hit ids go:
    make -C raytracer_amd/csrc EXTRA_HIPFLAGS=-DRT_TAIL_DIAG=1 OBJDIR=/tmp/rt_diag LIB=/tmp/rt_diag/lib.so
    RT_LIB_PATH=/tmp/rt_diag/lib.so python tools/tail_diag.py c5
(the fourth frame of the workload on the pooled kernel; times in ms from the first pixel's start)."""
import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
from raytracer_amd import capi
from raytracer_amd.renderer import Context, Scene, make_params
from raytracer_amd.workloads import WORKLOADS, workload_camera
name = sys.argv[1]
preset, FW, FH, spp, mb, _ = WORKLOADS[name]
s = Scene(preset); ctx = Context(0); ctx.upload_scene(s); ctx.resize(FW, FH)
cam = workload_camera(name, FW, FH)
for f in range(4):
    ctx.render_frame(cam, s.push_constants(0.0, f, f == 0), make_params(spp, mb, capi.RT_ACCEL_BVH, 0, 1, 0, capi.RT_FLAG_POOLED | capi.RT_FLAG_PRIMARY_IDS))
    ctx.sync()
print("kernel ms", ctx.stats().kernel_ms, "pipeline", ctx.stats().pipeline)
ids, _ = ctx.read_debug()
v = ids.astype(np.uint32).ravel()
start, end, cost = (v >> 20) & 0xfff, (v >> 8) & 0xfff, v & 0xff
t0 = int(np.min(start))  # (no wrap handling: a frame is far shorter than the counter's range; rerun if it wraps)
st = (start.astype(np.int64) - t0) * 0.04096; en = (end.astype(np.int64) - t0) * 0.04096
print("frame span ms", en.max(), " pixels", v.size)
edges = np.arange(0, en.max() + 0.5, 0.5)
h_end, _ = np.histogram(en, edges); h_st, _ = np.histogram(st, edges)
print("bucket(ms) started ended  mean-cost-of-ended  mean-start-of-ended  mean-life-ms")
for i in range(len(edges) - 1):
    m = (en >= edges[i]) & (en < edges[i + 1])
    if m.any():
        print(f"{edges[i]:5.1f} {h_st[i]:8d} {h_end[i]:8d}   {cost[m].mean():6.1f}   {st[m].mean():6.2f}   {(en[m]-st[m]).mean():6.2f}")
last = en >= en.max() - 1.5
print("pixels ending in the last 1.5 ms:", int(last.sum()), "mean cost", cost[last].mean(), "mean start", st[last].mean(), "start quantiles", np.quantile(st[last], [0.1, 0.5, 0.9]), "life quantiles", np.quantile((en - st)[last], [0.1, 0.5, 0.9]))
for c in (60, 100, 140):
    m = cost >= c
    print(f"pixels with cost >= {c}: {int(m.sum())}, start quantiles {np.quantile(st[m], [0.1,0.5,0.9]) if m.any() else None}, life quantiles {np.quantile((en-st)[m], [0.1,0.5,0.9]) if m.any() else None}, end max {en[m].max() if m.any() else None}")
print("overall: mean life", (en - st).mean(), "mean cost", cost.mean(), " life per cost unit (us):", 1e3 * (en - st).sum() / max(1, cost.sum()))
H, W = FH, FW
c2 = cost.reshape(H, W)
hb, wb = (H + 7) // 8, (W + 7) // 8
pad = np.zeros((hb * 8, wb * 8), dtype=c2.dtype); pad[:H, :W] = c2
bmax = pad.reshape(hb, 8, wb, 8).max(axis=(1, 3))
life = (en - st).reshape(H, W)
for thr in (60, 100, 140):
    nb = int((bmax >= thr).sum())
    print(f"blocks of 8x8 whose dearest pixel costs >= {thr}: {nb} of {hb * wb} ({100.0 * nb / (hb * wb):.2f} %), holding {int((c2 >= thr).sum())} such pixels; "
          f"rays (cost units) of all pixels in those blocks: {100.0 * pad.reshape(hb, 8, wb, 8).sum(axis=(1, 3))[bmax >= thr].sum() / max(1, c2.sum()):.2f} % of the frame's")
print("cost histogram (pixels):", np.histogram(cost, [0, 10, 20, 40, 60, 100, 140, 256])[0])
