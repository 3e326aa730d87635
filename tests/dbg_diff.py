import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from helpers import *
from raytracer_amd import capi
from raytracer_amd.renderer import Context, Scene
ctx = Context(0)
s = Scene(capi.RT_SCENE_CORNELL_STAR)
W,H=96,54
for spp,mb in ((1,0),(1,1),(2,6)):
    o = oracle_frames(s, W,H, spp=spp, max_bounces=mb)
    g = gpu_frames(ctx, s, W,H, spp=spp, max_bounces=mb, accel=capi.RT_ACCEL_FLAT)
    same = bit_equal_pixels(g['accum'], o['accum'])
    print('spp',spp,'mb',mb,'diff pixels', (~same).sum(), 'of', same.size, 'hash eq', (g['path_hash']==o['path_hash']).all(), 'rays', g['total_rays'], o['total_rays'])
    ys,xs = np.nonzero(~same)
    for y,x in list(zip(ys,xs))[:8]:
        print('  ', y,x, g['accum'][y,x], o['accum'][y,x], g['accum'][y,x].view(np.uint32)-o['accum'][y,x].view(np.uint32), 'id', o['primary_ids'][y,x])
    d = np.abs(g['accum']-o['accum'])
    print('  max abs diff', d.max())
