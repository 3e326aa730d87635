#!/usr/bin/env python3
"""A scripted interactive session, headless: the camera controller replays a list of input
events (keys held, mouse motion, scroll) at a fixed frame rate, every frame is rendered on the
GPU with the controller's camera block and push constants, and snapshots are written as PNG.
    python tools/fly_through.py out_dir [workload] [width height]
Script (seconds are the session clock): rest and converge, walk forward, look left, zoom, rest."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import capi
from raytracer_amd.renderer import Context, Controller, Scene, make_params
from raytracer_amd.workloads import WORKLOADS
from render_png import write_png

FPS = 30.0
SCRIPT = [  # (duration s, keys held, mouse dx px/frame, mouse dy px/frame, scroll per frame)
    (0.5, 0, 0.0, 0.0, 0.0),
    (0.5, capi.RT_KEY_W | capi.RT_KEY_LEFT_SHIFT, 0.0, 0.0, 0.0),
    (0.5, 0, -12.0, 2.0, 0.0),
    (0.3, 0, 0.0, 0.0, 1.0),
    (0.7, 0, 0.0, 0.0, 0.0),
]


def main():
    out_dir = sys.argv[1]
    name = sys.argv[2] if len(sys.argv) > 2 else "c3"
    w, h = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (960, 540)
    os.makedirs(out_dir, exist_ok=True)
    preset, _, _, spp, mb, _ = WORKLOADS[name]
    scene = Scene(preset)
    ctx = Context(0); ctx.upload_scene(scene); ctx.resize(w, h)
    ctl = Controller()
    clock, frame, mx, my = 0.0, 0, 0.0, 0.0
    ctl.mouse_move(mx, my)
    for seg, (duration, keys, dx, dy, scroll) in enumerate(SCRIPT):
        for _ in range(int(round(duration * FPS))):
            clock += 1.0 / FPS
            ctl.begin_frame(clock, keys)
            if dx or dy:
                mx += dx; my += dy
                ctl.mouse_move(mx, my)
            if scroll:
                ctl.scroll(scroll)
            cam, pc = ctl.frame_inputs(scene, w, h)
            ctx.render_frame(cam, pc, make_params(spp, mb))
            ctl.end_frame(clock)
            frame += 1
        _, cnt = ctx.read_accum()
        img = ctx.read_bgra8()
        path = os.path.join(out_dir, f"segment{seg}_frame{frame:04d}.png")
        write_png(path, img[:, :, [2, 1, 0]].copy())
        st = ctl.state
        print(f"{path}: accumulated frames {int(cnt.max())}, pos ({st.pos[0]:.2f},{st.pos[1]:.2f},{st.pos[2]:.2f}) "
              f"yaw {st.yaw:.1f} pitch {st.pitch:.1f} fov {st.fov:.0f}, last kernel {ctx.stats().kernel_ms:.2f} ms", flush=True)


if __name__ == "__main__":
    main()
