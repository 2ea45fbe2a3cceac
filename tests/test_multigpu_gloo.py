"""CPU: the N>1 path's sharding + reduce, world_size 2 over gloo.  The per-rank renderer is the
oracle standing in for the GPU library (same frame schedule / sum-accumulation contract); what is
under test is pathtracer_amd.multigpu -- the schedule, the reduce(sum) and the final mean -- i.e.
the code bench.py runs over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, frames_per_rank, out_path):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle as O
    import pathtracer_amd as pa
    from pathtracer_amd import multigpu
    sc = pa.scenes.cornell_box()
    osc = O.OracleScene(sc.vertices, sc.indices, sc.materials, sc.mat_indices)
    W, H = 32, 24
    view, proj = O.reference_camera(W, H)
    first, stride = multigpu.frame_schedule(rank, world)
    img, _ = osc.render(view, proj, W, H, frames=frames_per_rank, spp_per_frame=4, max_depth=6, seed=11,
                        first_frame=first, frame_stride=stride, accum_sum=True, nthreads=2)
    t = torch.from_numpy(img)
    multigpu.reduce_sum_image(t, dist, dst=0)
    if rank == 0:
        np.save(out_path, multigpu.finalize_mean(t.numpy(), world * frames_per_rank))
    dist.destroy_process_group()


def test_frame_schedule_partitions_frames():
    from pathtracer_amd import multigpu
    for world in (1, 2, 4, 8):
        seen = sorted(f for r in range(world) for f in multigpu.global_frames(r, world, 5))
        assert seen == list(range(5 * world))
    with pytest.raises(ValueError):
        multigpu.frame_schedule(2, 2)


def test_two_rank_reduce_equals_single_process_render(tmp_path, O, ocornell):
    world, fpr = 2, 3
    out = str(tmp_path / "img.npy")
    mp.spawn(_worker, args=(world, _free_port(), fpr, out), nprocs=world, join=True)
    got = np.load(out)
    W, H = 32, 24
    view, proj = O.reference_camera(W, H)
    ref, _ = ocornell.render(view, proj, W, H, frames=world * fpr, spp_per_frame=4, max_depth=6, seed=11, accum_sum=True)
    ref[..., :3] /= float(world * fpr)
    # same frames, different fp32 summation order: (f0+f2+f4)+(f1+f3+f5) vs f0+...+f5
    assert np.allclose(got, ref, rtol=2e-6, atol=1e-6)
    assert np.all(got[..., 3] == 1.0)


def _tile_worker(rank, world, port, frames, out_path):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle as O
    import pathtracer_amd as pa
    from pathtracer_amd import multigpu
    sc = pa.scenes.cornell_box()
    osc = O.OracleScene(sc.vertices, sc.indices, sc.materials, sc.mat_indices)
    W, H = 33, 20
    view, proj = O.reference_camera(W, H)
    x0, y0, w, h = multigpu.tile_schedule(rank, world, W, H)
    ys, xs = np.meshgrid(np.arange(y0, y0 + h), np.arange(x0, x0 + w), indexing="ij")
    pixels = (ys * W + xs).reshape(-1).astype(np.uint32)
    img = np.zeros((H, W, 4), dtype=np.float32)  # what pt_set_tile + PT_ACCUM_SUM leaves: zero outside the strip
    for f in range(frames):
        rgb = osc.render_pixels(view, proj, W, H, pixels, gframe=f, spp_per_frame=3, max_depth=5, seed=4)
        img.reshape(-1, 4)[pixels, :3] += rgb
    t = torch.from_numpy(img)
    multigpu.reduce_sum_image(t, dist, dst=0)
    if rank == 0:
        np.save(out_path, multigpu.finalize_mean(t.numpy(), frames))
    dist.destroy_process_group()


def test_interleaved_columns_partition_the_image():
    from pathtracer_amd import multigpu
    for world, W, B in ((1, 7, 32), (2, 33, 4), (8, 1920, 32), (8, 1921, 32), (3, 150, 7), (5, 40, 16)):
        cols = sorted(x for r in range(world) for x in multigpu.interleaved_columns(r, world, W, B))
        assert cols == list(range(W))
        counts = [len(multigpu.interleaved_columns(r, world, W, B)) for r in range(world)]
        assert max(counts) - min(counts) <= B
    with pytest.raises(ValueError):
        multigpu.interleaved_columns(2, 2, 10)


def test_tile_schedule_partitions_columns():
    from pathtracer_amd import multigpu
    for world, W in ((1, 7), (2, 33), (8, 1920), (8, 1921), (3, 3)):
        cols = []
        for r in range(world):
            x0, y0, w, h = multigpu.tile_schedule(r, world, W, 10)
            assert (y0, h) == (0, 10) and w >= 1
            cols += list(range(x0, x0 + w))
        assert cols == list(range(W))
    with pytest.raises(ValueError):
        multigpu.tile_schedule(0, 5, 4, 4)


def test_two_rank_tile_split_equals_single_process_render(tmp_path, O, ocornell):
    """Tiles: every pixel is rendered by exactly one rank with full-image seeds, so the reduced image is
    bit-identical to the one-process render (x + 0 = x), whatever the world size."""
    world, frames = 2, 3
    out = str(tmp_path / "tiles.npy")
    mp.spawn(_tile_worker, args=(world, _free_port(), frames, out), nprocs=world, join=True)
    got = np.load(out)
    W, H = 33, 20
    view, proj = O.reference_camera(W, H)
    ref, _ = ocornell.render(view, proj, W, H, frames=frames, spp_per_frame=3, max_depth=5, seed=4, accum_sum=True)
    ref[..., :3] /= float(frames)
    ref[..., 3] = 1.0
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


# ---------------------------------------------------------------- strong scaling: a fixed total of frames ----
def _strong_worker(rank, world, port, total_frames, out_path):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle as O
    import pathtracer_amd as pa
    from pathtracer_amd import multigpu
    sc = pa.scenes.cornell_box()
    osc = O.OracleScene(sc.vertices, sc.indices, sc.materials, sc.mat_indices)
    W, H = 32, 24
    view, proj = O.reference_camera(W, H)
    first, stride = multigpu.frame_schedule(rank, world)
    n = multigpu.strong_frame_count(rank, world, total_frames)  # what bench.py --scaling strong renders on this rank
    img, _ = osc.render(view, proj, W, H, frames=n, spp_per_frame=4, max_depth=6, seed=11, first_frame=first,
                        frame_stride=stride, accum_sum=True, nthreads=2)
    t = torch.from_numpy(img)
    multigpu.reduce_sum_image(t, dist, dst=0)
    if rank == 0:
        np.save(out_path, multigpu.finalize_mean(t.numpy(), total_frames))
    dist.destroy_process_group()


def test_strong_frame_count_partitions_a_fixed_total():
    from pathtracer_amd import multigpu
    for world in (1, 2, 3, 4, 8):
        for total in (0, 1, 5, 7, 8, 64, 1024):
            counts = [multigpu.strong_frame_count(r, world, total) for r in range(world)]
            assert sum(counts) == total and max(counts) - min(counts) <= 1
            frames = sorted(f for r in range(world) for f in multigpu.global_frames(r, world, counts[r]))
            assert frames == list(range(total))
    with pytest.raises(ValueError):
        multigpu.strong_frame_count(3, 3, 10)


def test_two_rank_strong_split_of_an_odd_total_equals_single_process_render(tmp_path, O, ocornell):
    """5 frames over 2 ranks (3 + 2): the reduced image is the 5-frame render up to fp32 summation order."""
    world, total = 2, 5
    out = str(tmp_path / "strong.npy")
    mp.spawn(_strong_worker, args=(world, _free_port(), total, out), nprocs=world, join=True)
    got = np.load(out)
    W, H = 32, 24
    view, proj = O.reference_camera(W, H)
    ref, _ = ocornell.render(view, proj, W, H, frames=total, spp_per_frame=4, max_depth=6, seed=11, accum_sum=True)
    ref[..., :3] /= float(total)
    assert np.allclose(got, ref, rtol=2e-6, atol=1e-6)
    assert np.all(got[..., 3] == 1.0)
