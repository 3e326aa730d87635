#!/usr/bin/env python3
"""Instrumented frame: where the render kernel's wave cycles go and how full the waves are
in the traversal loop.  python tools/phase_stats.py [c2|c3|c5]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import capi
from raytracer_amd.renderer import Context, Scene, make_params
from raytracer_amd.workloads import WORKLOADS, workload_camera
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
preset, FW, FH, spp, mb, _ = WORKLOADS[name]
s = Scene(preset); ctx = Context(0); ctx.upload_scene(s); ctx.resize(FW, FH)
cam = workload_camera(name, FW, FH)
for flags in (capi.RT_FLAG_SINGLE_KERNEL, capi.RT_FLAG_SINGLE_KERNEL, capi.RT_FLAG_SINGLE_KERNEL | capi.RT_FLAG_TRAVERSAL_STATS):
    ctx.render_frame(cam, s.push_constants(0.0, 1, True), make_params(spp, mb, capi.RT_ACCEL_BVH, flags=flags))
    st = ctx.stats()
    print(f"flags={flags} kernel_ms={st.kernel_ms:.3f} rays={st.rays} Mrays/s={st.rays/st.kernel_ms/1e3:.1f}")
tot = st.cycles_raygen + st.cycles_search + st.cycles_shade
print(f"nodes/ray={st.nodes_visited/st.rays:.2f} tests/ray={st.prims_tested/st.rays:.2f}")
print(f"wave cycles: raygen {st.cycles_raygen/tot:.3f} search {st.cycles_search/tot:.3f} shade {st.cycles_shade/tot:.3f}")
sp = list(st.cycles_shade_parts)
print("continuation split (hit+light pick, direct BSDF, sampling, indirect BSDF, end): " + " ".join(f"{x/max(1,sum(sp)):.3f}" for x in sp))
print(f"traversal lane occupancy: {st.lane_iterations/(64*max(1,st.wave_iterations)):.3f} (lane iters {st.lane_iterations}, wave iters {st.wave_iterations})")
