import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import capi
from raytracer_amd.renderer import Context, Scene, make_params
from raytracer_amd.workloads import WORKLOADS, workload_camera
F = int(sys.argv[1])
wl = sys.argv[2] if len(sys.argv) > 2 else "c3"
preset, FW, FH, spp, mb, _ = WORKLOADS[wl]
s = Scene(preset); cam = workload_camera(wl, FW, FH)
ctx = Context(0); ctx.upload_scene(s); ctx.resize(FW, FH)
strip = tuple(int(v) for v in os.environ.get("RT_STRIP", "0,1,0").split(","))  # rows per strip, ranks, this rank
for l in range(4):
    flags = capi.RT_FLAG_POOLED | (capi.RT_FLAG_TRAVERSAL_STATS if l == 3 else 0)
    params = make_params(spp, mb, capi.RT_ACCEL_BVH, strip[0], strip[1], strip[2], flags)
    pcs = [s.push_constants(0.0, l * F + k, l == 0 and k == 0) for k in range(F)]
    ctx.render_frames(cam, pcs, params); ctx.sync()
st = ctx.stats()
tot = max(1, st.cycles_raygen + st.cycles_search + st.cycles_shade)
jc = list(st.pool_job_cycles); parts = list(st.cycles_shade_parts)
print(f"{wl} F={F}: kernel {st.kernel_ms:.3f} ms = {st.kernel_ms/F:.3f} per frame, rays {st.rays}; search {st.cycles_search/tot:.3f} (steps {st.cycles_steps/tot:.3f}) other {st.cycles_raygen/tot:.3f} shading {st.cycles_shade/tot:.3f}: "
      + " ".join(f"{c/tot:.3f}" for c in jc) + f"; lanes with ray {st.busy_lane_iterations/(64*max(1,st.wave_iterations)):.3f}, taking the step {st.lane_iterations/max(1,st.busy_lane_iterations):.3f}; jobs " + " ".join(str(v & 0xffffffff) for v in parts[:4]))
