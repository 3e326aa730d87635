#!/usr/bin/env python3
"""Edge-tie pixels of the tree search against the device's flat search (bit-identical to the oracle's flat loops):
python tools/edge_ties.py [workload] [width height spp bounces].  Used to choose the builder's box padding
(RT_BVH_PAD overrides it, a tuning aid): the padding must keep every hit the primitive test accepts inside its leaf box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from helpers import DEBUG_FLAGS, bit_equal_pixels, gpu_frames
from raytracer_amd import capi
from raytracer_amd.renderer import Context, Scene
from raytracer_amd.workloads import WORKLOADS, workload_camera
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
preset = WORKLOADS[name][0]
W, H, spp, mb = (int(v) for v in sys.argv[2:6]) if len(sys.argv) > 5 else (480, 270, 2, 4)
s = Scene(preset); ctx = Context(0)
cam = workload_camera(name, W, H)
f = gpu_frames(ctx, s, W, H, spp, mb, camera=cam, accel=capi.RT_ACCEL_FLAT)
b = gpu_frames(ctx, s, W, H, spp, mb, camera=cam, accel=capi.RT_ACCEL_BVH, flags=DEBUG_FLAGS | capi.RT_FLAG_POOLED, upload=False)
same = bit_equal_pixels(b["accum"], f["accum"]) & (b["path_hash"] == f["path_hash"])
print(f"pad {os.environ.get('RT_BVH_PAD', 'default')}: {name} {W}x{H} {spp} spp {mb} bounces: {int((~same).sum())} edge-tie pixels of {same.size}, rays {f['total_rays']}", flush=True)
