#!/usr/bin/env python3
"""Render N frames of a workload (for profilers).  python tools/run_frames.py c3 4 [single|pooled|auto] [stats]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import capi
from raytracer_amd.renderer import Context, Scene, make_params
from raytracer_amd.workloads import WORKLOADS, workload_camera
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
mode = sys.argv[3] if len(sys.argv) > 3 else "single"
flags = {"pooled": capi.RT_FLAG_POOLED, "single": capi.RT_FLAG_SINGLE_KERNEL, "auto": 0}[mode]
if len(sys.argv) > 4 and sys.argv[4] == "stats":
    flags |= capi.RT_FLAG_TRAVERSAL_STATS
preset, FW, FH, spp, mb, _ = WORKLOADS[name]
s = Scene(preset); ctx = Context(0); ctx.upload_scene(s); ctx.resize(FW, FH)
cam = workload_camera(name, FW, FH)
strip = tuple(int(v) for v in os.environ.get("RT_STRIP", "0,1,0").split(","))  # rows per strip, ranks, this rank
for f in range(n):
    ctx.render_frame(cam, s.push_constants(0.0, f, f == 0), make_params(spp, mb, capi.RT_ACCEL_BVH, strip[0], strip[1], strip[2], flags))
    ctx.sync()
    st = ctx.stats()
    print(f"frame {f}: {st.kernel_ms:.3f} ms rays {st.rays} nodes/ray {st.nodes_visited/max(1,st.rays):.2f} tests/ray {st.prims_tested/max(1,st.rays):.2f} width {st.bvh_width} "
          f"occupancy {st.lane_iterations/(64*max(1,st.wave_iterations)):.3f}", flush=True)
    if mode == "pooled" and (flags & capi.RT_FLAG_TRAVERSAL_STATS):
        names = ["first-hit", "later-hit", "light-term", "sample-end/pixel-end/miss"]
        dp = list(st.cycles_shade_parts)[4]
        print(f"  pushes at depth >=4: {dp >> 40}, >=6: {(dp >> 20) & 0xfffff}, >=8: {dp & 0xfffff} of {st.nodes_visited} node visits")
        tot = max(1, st.cycles_raygen + st.cycles_search + st.cycles_shade)
        jc = list(st.pool_job_cycles)
        print(f"  wave cycles: search {st.cycles_search / tot:.3f}, other {st.cycles_raygen / tot:.3f}, shading " + ", ".join(
            f"{n} {c / tot:.3f} ({c / max(1, v >> 32):.0f}/item)" for n, c, v in zip(names, jc, list(st.cycles_shade_parts)[:4])))
        print(f"  search steps: lanes with a ray {st.busy_lane_iterations / (64 * max(1, st.wave_iterations)):.3f}, of those taking the step "
              f"{st.lane_iterations / max(1, st.busy_lane_iterations):.3f}; bursts of steps {st.cycles_steps / tot:.3f} of all wave cycles "
              f"({st.cycles_steps / max(1, st.wave_iterations):.0f} cycles per wave step), exchange of rays {(st.cycles_search - st.cycles_steps) / tot:.3f}")
        print(f"  search queue: at most {st.search_queue_peak} rays waiting of {st.search_queue_size} it holds")
        print("  job fill: " + ", ".join(f"{n} {(v >> 32) / max(1, v & 0xffffffff):.1f}/64 x {v & 0xffffffff}" for n, v in zip(names, list(st.cycles_shade_parts)[:4])), flush=True)
