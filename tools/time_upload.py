import time, sys, os
sys.path.insert(0, os.getcwd())
from raytracer_amd import capi
from raytracer_amd.renderer import Context, Scene
from raytracer_amd.workloads import WORKLOADS
for name in ("c3", "c5"):
    preset = WORKLOADS[name][0]
    t0 = time.perf_counter(); s = Scene(preset); t1 = time.perf_counter()
    ctx = Context(0); t2 = time.perf_counter()
    ctx.upload_scene(s); t3 = time.perf_counter()
    print(name, f"scene build {t1-t0:.2f} s, context {t2-t1:.2f} s, upload (BVH build + copies) {t3-t2:.2f} s", flush=True)
    ctx.close()
