#!/usr/bin/env python3
"""Benchmark of the path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One step = --frames-per-launch frames (32 unless told otherwise; a frame = one execution of the reference
shader's main() per pixel) of the workload BASELINE.json's metric is quoted on: Cornell box + teapot.glb,
1920x1080, 16 samples per pixel, 4 bounces, still camera, frameCount advancing from frame to frame as in the
reference's frame loop.  The frames of a step are handed to the library in one rt_render_frames call, which
fuses them into one launch; `value` is rays per second over all of it, `ms_per_frame` the step's time divided
by its frames, and `ms_single_frame_launch` what one frame costs launched alone (--frames-per-launch 1 makes
that the step).  At N > 1 (launched by torch.distributed.run, one rank per GPU) each frame is split into row
strips, the scene is replicated, and the accumulated frame is completed on every rank by one RCCL all-gather
per step; the total work is fixed, so scaling is
"strong".  Rank 0 prints one JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from raytracer_amd import capi, parallel  # noqa: E402
from raytracer_amd.renderer import Context, Scene, default_camera, make_params  # noqa: E402
from raytracer_amd.workloads import WORKLOADS, workload_camera  # noqa: E402

FRAMES_PER_BATCH = 16  # sample sharding: frames per progressive batch (c4: 16 x 4 spp = 64 spp)
STRIP_ROWS = 9  # 1080 / 9 = 120 strips: divisible by 1, 2, 4 and 8 ranks

# MI355X vector FP32 peak (spec): 256 CU x 4 SIMD x 32 lanes per clock x 2 flop (FMA) x 2.4 GHz = 157.3 TFLOP/s.  A SIMD
# issues a wave's v_add / v_mul / v_fma_f32 over 2 clocks - plain instructions, no packing needed: tools/ubench/valu_rate
# measures 2.2-2.4 clocks per wave instruction at 4-8 waves per SIMD, 144 TFLOP/s (profiles/r02_valu_rate.txt) - and
# everything else the search is made of (min / max 4.7 clocks, integer and address arithmetic 4.3, a compare + select
# pair 8.3, reciprocal 8.5) over 4 clocks or more.  78.6 TFLOP/s is what the peak would be at 4 clocks per instruction.
PEAK_VALU_TFLOPS = 157.3
PEAK_VALU_TFLOPS_4CLK = 78.6
PEAK_HBM_GBPS = 8000.0


def lavapipe_probe() -> str:
    """The baseline the metric names is the reference shader under Mesa lavapipe on the host cores: probe for what that
    would take (a Vulkan loader, lavapipe's ICD manifest, a GLSL compiler) and say what is missing instead of failing."""
    import ctypes.util
    import glob
    import shutil
    missing = []
    if not ctypes.util.find_library("vulkan") and not glob.glob("/usr/lib/*/libvulkan.so*") and not glob.glob("/usr/lib64/libvulkan.so*"):
        missing.append("no libvulkan")
    if not [m for d in ("/usr/share/vulkan/icd.d", "/etc/vulkan/icd.d", "/usr/local/share/vulkan/icd.d") for m in glob.glob(d + "/*lvp*")]:
        missing.append("no lavapipe ICD")
    if not (shutil.which("glslc") or shutil.which("glslangValidator")):
        missing.append("no GLSL compiler")
    if missing:
        return "unavailable (" + ", ".join(missing) + " on this box)"
    return "present but unused: this repository has no Vulkan harness for the reference shader"


def gpu_same_config(device: int) -> dict:
    """The CPU baseline's configuration (C1, same scene, camera, seeds) on the GPU, beside it: the flat search (the
    reference's algorithm, what the CPU figure runs) and the tree search, kernel time of a frame launched alone,
    median of five after three settling frames."""
    preset, w, h, spp, mb, _ = WORKLOADS["c1"]
    scene = Scene(preset)
    ctx = Context(device)
    out = {}
    try:
        ctx.upload_scene(scene)
        ctx.resize(w, h)
        cam = workload_camera("c1", w, h)
        for name, accel in (("flat", capi.RT_ACCEL_FLAT), ("bvh", capi.RT_ACCEL_BVH)):
            ms, rays = [], 0
            for f in range(8):
                ctx.render_frame(cam, scene.push_constants(0.0, 0, True), make_params(spp, mb, accel))
                st = ctx.stats()
                rays = st.rays
                if f >= 3:
                    ms.append(st.kernel_ms)
            t = sorted(ms)[len(ms) // 2]
            out[name] = {"ms_per_frame": t, "Mrays_per_s": rays / (t * 1e-3) / 1e6, "rays_per_frame": rays}
    finally:
        ctx.close()
    return out


def cpu_baseline() -> dict:
    """The CPU oracle (flat search = the reference's algorithm: every ray against every primitive) on
    BASELINE.json's CPU-runnable configuration C1 in full: Cornell box + teapot.glb, 640x360, 1 spp,
    primary + 1 bounce, canonical camera, time 0, frame 0 (a few seconds on the GPU box's 16-core share;
    median of three frames)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob

    preset, w, h, spp, mb, desc = WORKLOADS["c1"]
    scene = Scene(preset)
    threads = max(1, min(16, os.cpu_count() or 1))
    cam = workload_camera("c1", w, h)
    osc = ob.OracleScene(scene.arrays())
    pc = scene.push_constants(0.0, 0, True)
    times, rays = [], 0
    for _ in range(3):
        t0 = time.perf_counter()
        out = ob.render(osc, cam, pc, w, h, spp, mb, threads=threads, want_debug=False)
        times.append(time.perf_counter() - t0)
        rays = out["rays"]
    dt = sorted(times)[1]
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port", "ms_per_frame": dt * 1e3,
            "sample": f"c1: {desc}; flat search (every ray x every primitive, as the reference shader), "
                      f"{rays} rays per frame, median of 3 frames {dt * 1e3:.0f} ms"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--accel", default="bvh", choices=["bvh", "flat"])
    ap.add_argument("--partition", default=None, choices=["strips", "samples"],
                    help="how N GPUs share the work: row strips of every frame (default) or whole frames "
                         "(sample sharding, default for c4)")
    ap.add_argument("--pipeline", default="auto", choices=["auto", "single", "pooled"],
                    help="schedule of the per-path program (same results): the library's own choice, the single "
                         "persistent kernel, or the pooled kernel (LDS queues)")
    ap.add_argument("--frames-per-launch", type=int, default=32,
                    help="row-strip / single-GPU runs: frames of the still camera handed to the library per step (rt_render_frames "
                         "fuses up to 32 into one launch; with several ranks the strips are exchanged once per step); "
                         "1 = one rt_render_frame call (and one exchange) per frame")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true",
                    help="skip the untimed check of the timed frames (rank 0 renders the same frames again, one per call, on the "
                         "single kernel and compares colours, counts and BGRA8 bytes bit for bit)")
    ap.add_argument("--force-collective", action="store_true",
                    help="run the strip all-gather even with one rank (exercises the RCCL/stream path on one GPU)")
    ap.add_argument("--rehearse", action="store_true",
                    help="N ranks on however many GPUs there are (rank r renders on GPU r mod count; RCCL wants one GPU per "
                         "rank, so the ranks talk over gloo through host memory), and rank 0 checks the completed frame bit for "
                         "bit against the same frames rendered whole on its GPU: the N-rank path of this script on a one-GPU box; "
                         "timings mean nothing")
    args = ap.parse_args()
    # stdout carries ONE line, the result; whatever the libraries print there meanwhile (RCCL announces its version
    # on stdout) goes to stderr: file descriptor 1 is pointed at 2 until the line is written
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback path)")
    if args.rehearse:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    collective = world > 1 or args.force_collective
    if collective:
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        if args.rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    small = torch.device("cpu") if args.rehearse else dev  # where the few scalars exchanged at the end live

    preset, W, H, spp, max_bounces, desc = WORKLOADS[args.workload]
    accel = capi.RT_ACCEL_BVH if args.accel == "bvh" else capi.RT_ACCEL_FLAT
    scene = Scene(preset)
    ctx = Context(local_rank)
    ctx.upload_scene(scene)
    # framebuffer in torch tensors so that RCCL reads the library's output in place
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    cnt = torch.zeros((H, W), dtype=torch.int32, device=dev)
    bgra = torch.zeros((H, W), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.bind_framebuffer(W, H, acc.data_ptr(), cnt.data_ptr(), bgra.data_ptr())
    cam = workload_camera(args.workload, W, H)
    partition = args.partition or ("samples" if args.workload == "c4" else "strips")
    fpl = max(1, args.frames_per_launch)
    frames_per_step = FRAMES_PER_BATCH if partition == "samples" else fpl
    strip = (STRIP_ROWS, world, rank) if (collective and partition == "strips") else (0, 1, 0)
    pipe_flags = {"auto": 0, "single": capi.RT_FLAG_SINGLE_KERNEL, "pooled": capi.RT_FLAG_POOLED}[args.pipeline]
    params = make_params(spp, max_bounces, accel, strip[0], strip[1], strip[2], pipe_flags)
    rt_stream = torch.cuda.ExternalStream(ctx.stream, device=dev)
    main_stream = torch.cuda.current_stream(dev)
    exchange = None
    if collective and partition == "strips":
        # one all-gather per frame: colours + counts of own rows packed by the library (20 B per pixel), the
        # other ranks' rows unpacked and their BGRA8 bytes rebuilt by it
        exchange = parallel.PackedStripExchange(acc, cnt, bgra, STRIP_ROWS, world, rank, ctx=ctx, include_self=args.force_collective,
                                                via_host=args.rehearse)
    packed, gathered = torch.cuda.Event(), torch.cuda.Event()
    pending = [False]  # a gathered frame still waits to be unpacked

    def step(frame: int, ev=None) -> None:
        if partition == "samples":
            # one step = one progressive batch: FRAMES_PER_BATCH frames of `spp` samples, frames dealt
            # round-robin to the ranks, each rank accumulating privately (reset on its first frame)
            own = parallel.frames_of(FRAMES_PER_BATCH, world, rank)
            if ev is not None:
                ev[0].record(rt_stream)
            # (one call: the library fuses a rank's frames into one launch where that pays)
            ctx.render_frames(cam, [scene.push_constants(0.0, frame * FRAMES_PER_BATCH + f, i == 0) for i, f in enumerate(own)], params)
            if ev is not None:
                ev[1].record(rt_stream)
            if collective:
                main_stream.wait_stream(rt_stream)
                parallel.reduce_samples(acc, cnt, via_host=args.rehearse)
                rt_stream.wait_stream(main_stream)
            ctx.resolve()
            return
        pc = scene.push_constants(0.0, frame * fpl, frame == 0)
        if collective and args.force_collective:
            # one rank unpacking its OWN rows too: the rows gathered after step N - 1 must land before kernel N adds to
            # them (with several ranks a rank never unpacks rows it renders, and the unpack may follow the kernel)
            flush()
        if ev is not None:
            ev[0].record(rt_stream)
        if fpl > 1:
            ctx.render_frames(cam, [scene.push_constants(0.0, frame * fpl + k, frame == 0 and k == 0) for k in range(fpl)], params)
        else:
            ctx.render_frame(cam, pc, params)
        if ev is not None:
            ev[1].record(rt_stream)
        if collective:
            # Render stream: kernel(N), unpack(N-1), pack(N); collective stream: gather(N) after pack(N).
            # The all-gather of frame N thus runs beside kernel N+1 (which writes own rows only; the
            # gather reads the packed copy), and its rows are unpacked once that kernel is through.
            flush()
            exchange.pack()
            packed.record(rt_stream)
            main_stream.wait_event(packed)
            exchange.gather()
            gathered.record(main_stream)
            pending[0] = True

    def flush() -> None:
        if pending[0]:
            rt_stream.wait_event(gathered)
            exchange.unpack()
            pending[0] = False

    def fence() -> None:
        if collective and partition == "strips":
            flush()
        ctx.sync()
        torch.cuda.synchronize()
        if collective:
            dist.barrier()
            torch.cuda.synchronize()

    # The library times its two kernels against each other on a configuration's first frames and
    # orders the tiles by the previous frame's costs: settle that before the W warm-up steps, so
    # that the timed frames are steady-state frames whatever W is.
    SETTLE = 6
    for f in range(SETTLE):
        step(f)
    for f in range(args.warmup):
        step(SETTLE + f)
    fence()
    rays0 = ctx.stats().total_rays
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(SETTLE + args.warmup + k, events[k])
    fence()
    t1 = time.perf_counter()
    rays = ctx.stats().total_rays - rays0
    own_frames = len(parallel.frames_of(FRAMES_PER_BATCH, world, rank)) if partition == "samples" else fpl
    kernel_ms = [a.elapsed_time(b) / max(1, own_frames) for a, b in events]  # per frame (a launch may hold several)

    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=small)
    total_rays = torch.tensor([float(rays)], dtype=torch.float64, device=small)
    kern = torch.tensor([sum(kernel_ms) / max(1, len(kernel_ms))], dtype=torch.float64, device=small)
    if collective:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
        dist.all_reduce(total_rays, op=dist.ReduceOp.SUM)
        dist.all_reduce(kern, op=dist.ReduceOp.MAX)
    elapsed_s = float(elapsed.item())
    job_rays = float(total_rays.item())
    avg_kernel_ms = float(kern.item())

    # ---- untimed: are the frames just timed the right frames?  Rank 0 renders the same frames again, one rt_render_frame
    # call each, whole, on the single kernel (another schedule of the per-path program, frame by frame instead of
    # fused, unsplit instead of in strips) in a context of its own, and compares accumulated colours, sample counts and
    # BGRA8 bytes bit for bit with what the timed path left in the framebuffer (strips: the frame all ranks completed
    # together; samples: the last batch, summed over ranks - in another order than frame order, so within rounding there).
    verified, verify_note = None, "skipped (--no-verify)"
    frames_done = (SETTLE + args.warmup + args.steps) * frames_per_step
    if not args.no_verify:
        mine = (acc.clone(), cnt.clone(), bgra.clone())
        if rank == 0:
            whole = Context(local_rank)
            whole.upload_scene(scene)
            whole.resize(W, H)
            whole_params = make_params(spp, max_bounces, accel, 0, 1, 0, capi.RT_FLAG_SINGLE_KERNEL)
            if partition == "samples":  # every step resets: what remains is the last batch
                first = (SETTLE + args.warmup + args.steps - 1) * FRAMES_PER_BATCH
                todo = [(first + f, f == 0) for f in range(FRAMES_PER_BATCH)]
            else:
                todo = [(f, f == 0) for f in range(frames_done)]
            for f, reset in todo:
                whole.render_frame(cam, scene.push_constants(0.0, f, reset), whole_params)
            wa, wc = whole.read_accum()
            wb = whole.read_bgra8()
            whole.close()
            ga, gc, gb = mine[0].cpu().numpy(), mine[1].cpu().numpy(), mine[2].cpu().numpy().view("uint8").reshape(H, W, 4)
            counts_ok = bool((gc == wc).all()) and bool((gc == len(todo)).all())
            if partition == "samples" and world > 1:
                import numpy as np
                verified = counts_ok and bool(np.allclose(ga, wa, rtol=1e-5, atol=1e-6))
                verify_note = f"last batch of {len(todo)} frames summed over {world} ranks against the same frames accumulated in order on the single kernel: counts equal, colours within 1e-5 (the sum's order differs)"
            else:
                verified = counts_ok and bool((ga.view("uint32") == wa.view("uint32")).all()) and bool((gb == wb).all())
                verify_note = (f"all {len(todo)} frames of the run ({SETTLE} settling + {args.warmup} warm-up + {args.steps} timed steps) rendered again one rt_render_frame "
                               "call each, whole, on the single kernel: accumulated colours, counts and BGRA8 bytes bit-identical"
                               if verified else "MISMATCH between the timed path's framebuffer and the same frames on the single kernel")
            print(f"verify: {verify_note}: {verified}", file=sys.stderr)

    # traversal work per ray, from one extra (untimed) instrumented frame on this rank's rows
    pipeline = {capi.RT_PIPELINE_SINGLE: "single", capi.RT_PIPELINE_POOLED: "pooled"}[ctx.stats().pipeline]  # what the timed frames ran on
    launch_frames = ctx.stats().frames  # frames the library put into the last timed launch
    # what ONE frame costs when it is launched alone (the latency of a frame, where the steps above measure throughput):
    # kernel time of single rt_render_frame calls on this rank's rows, median of five
    # (on the kernel the timed launches ran on: left to itself the library would time its two kernels against each other
    # on these first single-frame calls of the configuration)
    forced = capi.RT_FLAG_POOLED if pipeline == "pooled" else capi.RT_FLAG_SINGLE_KERNEL
    alone = make_params(spp, max_bounces, accel, strip[0], strip[1], strip[2], forced)
    single_ms = []
    for k in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(rt_stream)
        ctx.render_frame(cam, scene.push_constants(0.0, 100000 + k, False), alone)
        b.record(rt_stream)
        ctx.sync()
        single_ms.append(a.elapsed_time(b))
    single_frame_ms = sorted(single_ms)[len(single_ms) // 2]
    flags = capi.RT_FLAG_TRAVERSAL_STATS | forced
    pc = scene.push_constants(0.0, SETTLE + args.warmup, False)
    ctx.render_frame(cam, pc, make_params(spp, max_bounces, accel, strip[0], strip[1], strip[2], flags))
    st = ctx.stats()
    nodes_per_ray = st.nodes_visited / max(1, st.rays)
    tests_per_ray = st.prims_tested / max(1, st.rays)

    if rank == 0:
        v = scene.view()
        n_prims = st.total_prims
        if partition == "samples":  # a launch is a whole frame on one rank
            pixels_per_launch = W * H
            rays_per_launch = job_rays / args.steps / FRAMES_PER_BATCH
        else:
            pixels_per_launch = W * H / world
            rays_per_launch = job_rays / args.steps / world / fpl  # per frame of a launch of fpl frames
        if accel == capi.RT_ACCEL_BVH:
            # per ray: one box test (24 flop) per child of a visited node - 2 children in 64 B, or 4 in 128 B
            # (the pooled kernel's four-wide tree) -, 51 flop per triangle test; 48 B per tested primitive
            # record; + 48 B framebuffer per pixel
            width = st.bvh_width if st.bvh_width in (2, 4) else 2
            node_bytes = st.bvh_node_bytes if st.bvh_node_bytes > 0 else 32 * width  # 64 binary, 128 or 80 (binary16 boxes) four-wide
            flops_per_ray = 24.0 * width * nodes_per_ray + 51.0 * tests_per_ray
            bytes_per_ray = float(node_bytes) * nodes_per_ray + 48.0 * tests_per_ray
        else:
            flops_per_ray = 30.0 * v.total_spheres + 51.0 * (n_prims - v.total_spheres)
            bytes_per_ray = 48.0 * n_prims
        launch_flops = flops_per_ray * rays_per_launch
        launch_bytes = bytes_per_ray * rays_per_launch + 48.0 * pixels_per_launch
        achieved_tflops = launch_flops / (avg_kernel_ms * 1e-3) / 1e12
        # HBM-side bytes per launch from the committed rocprofv3 PMC passes (tools/pmc_passes.sh) of
        # this workload on one GPU; bench.py cannot run the profiler on itself.
        traffic, traffic_note, issue, executed = None, None, {}, {}
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(f"{args.workload}:{pipeline}")
            if rec and world == 1 and args.accel == "bvh":
                traffic = (rec["FETCH_SIZE_KB"] + rec["WRITE_SIZE_KB"]) * 1024.0
                traffic_note = rec["note"] + " [" + rec["source"] + "]"
                # from the same passes: share of the SIMDs' vector issue slots used (an instruction holds its SIMD's port
                # for 4 clocks), share of the SIMDs' time with a vector instruction in flight, share of lanes active in
                # those instructions, L2 hit rate, L1 / texture-address path busy, wave cycles waiting on memory
                issue = {k: rec[k] for k in ("valu_issue_frac", "valu_active_frac", "lane_utilisation", "l2_hit", "ta_busy", "wait_frac") if k in rec}
                fp = rec.get("fp32_wave_insts")
                if fp:
                    # FP32 arithmetic the kernel ISSUED (counted by the hardware per wave instruction, x 64 lanes, fma = 2),
                    # and what is left of it after the nearest-hit search's algorithmic flops: shading (BSDF evaluation and
                    # sampling, light sampling, hit records, ray generation, gamma), plus lanes masked off in either part
                    issued = (fp["add"] + fp["mul"] + 2.0 * fp["fma"] + fp["trans"]) * 64.0
                    executed = {"issued_fp32_flops_per_launch": issued,
                                "issued_fp32_flops_per_ray": issued / max(1.0, rays_per_launch),
                                "shading_and_masked_flops_per_ray": issued / max(1.0, rays_per_launch) - flops_per_ray,
                                "transcendental_wave_insts_per_launch": fp["trans"]}
        except (OSError, ValueError, KeyError):
            pass
        result = {
            "metric": "Mrays/s",
            "value": job_rays / elapsed_s / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed_s / args.steps * 1e3,
            "ms_per_frame": elapsed_s / args.steps / frames_per_step * 1e3,
            "ms_single_frame_launch": single_frame_ms,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {desc}", "search": args.accel,
                       "partition": ("single GPU" if not collective else
                                     f"row strips of {STRIP_ROWS} rows, one all-gather of packed colours + counts per step of {fpl} frame(s)" if partition == "strips" else
                                     f"frames of a {FRAMES_PER_BATCH}-frame batch dealt to ranks, all-reduce of accumulators"),
                       "pipeline": pipeline, "frames_per_step": frames_per_step, "frames_per_launch": launch_frames,
                       "rays_per_step": job_rays / args.steps, "ray_definition": "1 ray = 1 nearest-hit search"},
            "roofline": {
                "bound": "valu",
                "kernel": {"single": "render_kernel", "pooled": "pool_kernel"}[pipeline],
                "achieved": achieved_tflops,
                "peak": PEAK_VALU_TFLOPS,
                "peak_note": "vector FP32 peak of the guide and the spec sheet (plain v_fma_f32 at 2 clocks per wave instruction: measured "
                             "144 TFLOP/s, profiles/r02_valu_rate.txt); frac_4clk prices the same number against "
                             f"{PEAK_VALU_TFLOPS_4CLK} TFLOP/s, the rate at 4 clocks per instruction, which is what compares, selects, "
                             "min / max and integer arithmetic cost",
                "unit": "TFLOP/s",
                "frac": achieved_tflops / PEAK_VALU_TFLOPS,
                "frac_4clk": achieved_tflops / PEAK_VALU_TFLOPS_4CLK,
                "traffic": traffic,
                "traffic_note": traffic_note,
                "avg_kernel_ms": avg_kernel_ms,
                "per": (f"frame: the timed launch carries {launch_frames} frames, avg_kernel_ms is its measured duration / {launch_frames}; traffic, "
                        "issued flops and the instruction counters come from PMC passes of one frame per launch") if launch_frames > 1 else "launch = frame",
                "flops_per_ray": flops_per_ray,
                "bytes_per_ray": bytes_per_ray,
                "nodes_per_ray": nodes_per_ray,
                "children_per_node": st.bvh_width,
                "node_bytes": st.bvh_node_bytes,
                "prim_tests_per_ray": tests_per_ray,
                "algorithmic_GBps": launch_bytes / (avg_kernel_ms * 1e-3) / 1e9,
                "hbm_peak_GBps": PEAK_HBM_GBPS,
                **issue,
                **executed,
            },
        }
        result["verified"] = verified
        result["verified_note"] = verify_note
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline()
            result["cpu_baseline"]["gpu_same_config"] = gpu_same_config(local_rank)
            result["cpu_baseline"]["lavapipe"] = lavapipe_probe()
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(result) + "\n").encode())

    ctx.close()
    if collective:
        dist.destroy_process_group()
    if rank == 0 and verified is False:
        raise SystemExit("bench.py: the timed frames differ from the same frames rendered one by one on the single kernel")


if __name__ == "__main__":
    main()
