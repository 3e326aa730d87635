"""Multi-GPU partitions, rehearsed with two gloo ranks on the CPU: the oracle renders each
rank's share, raytracer_amd.parallel combines the shares exactly as the GPU path does."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
W, H, SPP, MB = 40, 30, 1, 3
STRIP = 4


def _worker(rank, world, port, mode, outdir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_binding as ob
    from raytracer_amd import capi, parallel
    from raytracer_amd.renderer import Scene, default_camera

    s = Scene(capi.RT_SCENE_CORNELL_STAR)
    cam = default_camera(W, H)
    osc = ob.OracleScene(s.arrays())
    if mode in ("strips", "packed"):
        out = ob.render(osc, cam, s.push_constants(0.0, 0, True), W, H, SPP, MB, threads=2, strip=(STRIP, world, rank))
        acc = torch.from_numpy(out["accum"])
        cnt = torch.from_numpy(out["counts"])
        bgra = torch.from_numpy(out["bgra"].view(np.int32).reshape(H, W).copy())
        if mode == "strips":
            parallel.gather_strips([acc, cnt, bgra], H, STRIP, world, rank)
        else:  # one all-gather of packed colours + counts; the BGRA8 bytes of the other rank's rows are rebuilt
            ex = parallel.PackedStripExchange(acc, cnt, bgra, STRIP, world, rank)
            ex.pack()
            ex.gather()
            ex.unpack()
        np.savez(os.path.join(outdir, f"{mode}_{rank}.npz"), accum=acc.numpy(), counts=cnt.numpy(), bgra=bgra.numpy())
    else:
        acc_np = np.zeros((H, W, 4), np.float32)
        cnt_np = np.zeros((H, W), np.int32)
        for f in parallel.frames_of(4, world, rank):
            ob.render(osc, cam, s.push_constants(0.0, f, False), W, H, SPP, MB, threads=2, accum=acc_np, counts=cnt_np)
        acc, cnt = torch.from_numpy(acc_np), torch.from_numpy(cnt_np)
        parallel.reduce_samples(acc, cnt)
        np.savez(os.path.join(outdir, f"samples_{rank}.npz"), accum=acc.numpy(), counts=cnt.numpy())
    dist.destroy_process_group()


def _run(mode, tmp_path, port):
    mp.spawn(_worker, args=(2, port, mode, str(tmp_path)), nprocs=2, join=True)


@pytest.mark.parametrize("mode,port", [("strips", 29611), ("packed", 29613)])
def test_row_strips_two_ranks_equal_one_rank(tmp_path, mode, port):
    sys.path.insert(0, HERE)
    from helpers import bit_equal_pixels, oracle_frames
    from raytracer_amd import capi
    from raytracer_amd.renderer import Scene
    _run(mode, tmp_path, port)
    whole = oracle_frames(Scene(capi.RT_SCENE_CORNELL_STAR), W, H, SPP, MB, threads=2)
    for r in range(2):
        got = np.load(tmp_path / f"{mode}_{r}.npz")
        assert bit_equal_pixels(got["accum"], whole["accum"]).all()  # identical bits: global pixel coordinates seed the RNG
        assert np.array_equal(got["counts"], whole["counts"])
        assert np.array_equal(got["bgra"].view(np.uint8).reshape(H, W, 4), whole["bgra"])


def test_sample_sharding_two_ranks_equals_sequential_accumulation(tmp_path):
    sys.path.insert(0, HERE)
    from helpers import oracle_frames
    from raytracer_amd import capi
    from raytracer_amd.renderer import Scene
    _run("samples", tmp_path, 29612)
    seq = oracle_frames(Scene(capi.RT_SCENE_CORNELL_STAR), W, H, SPP, MB, frames=4, threads=2)
    for r in range(2):
        got = np.load(tmp_path / f"samples_{r}.npz")
        assert (got["counts"] == 4).all()
        # the sum order differs from frame-sequential accumulation: equal to float32 rounding
        assert np.allclose(got["accum"], seq["accum"], rtol=0, atol=4e-7 * 4)


def test_strip_ownership_covers_every_row_once():
    from raytracer_amd import parallel
    for height, strip, world in ((1080, 9, 8), (1080, 9, 4), (37, 4, 3), (5, 8, 2)):
        rows = torch.cat([parallel.strip_rows_of(height, strip, world, r) for r in range(world)])
        assert sorted(rows.tolist()) == list(range(height))
    assert parallel.frames_of(5, 2, 0) == [0, 2, 4] and parallel.frames_of(5, 2, 1) == [1, 3]
