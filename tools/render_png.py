#!/usr/bin/env python3
"""Render a workload on the GPU for some frames (progressive accumulation) and save a PNG.
    python tools/render_png.py c3 out.png [frames] [width height]"""
import os, struct, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import capi
from raytracer_amd.renderer import Context, Scene, make_params
from raytracer_amd.workloads import WORKLOADS, workload_camera


def write_png(path, rgb):
    h, w, _ = rgb.shape
    raw = b"".join(b"\x00" + rgb[y].tobytes() for y in range(h))
    def chunk(t, d):
        c = struct.pack(">I", len(d)) + t + d
        return c + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                           chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def main():
    name, out = sys.argv[1], sys.argv[2]
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    w, h = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (960, 540)
    preset, _, _, spp, mb, _ = WORKLOADS[name]
    s = Scene(preset); ctx = Context(0); ctx.upload_scene(s); ctx.resize(w, h)
    cam = workload_camera(name, w, h)
    for f in range(frames):
        ctx.render_frame(cam, s.push_constants(0.0, f, f == 0), make_params(spp, mb))
    img = ctx.read_bgra8()
    write_png(out, img[:, :, [2, 1, 0]].copy())
    print("wrote", out, "kernel_ms(last)", ctx.stats().kernel_ms)


if __name__ == "__main__":
    main()
