"""GPU: the three execution pipelines of pt_render_frame (persistent path kernel, one fused kernel per
bounce, split extend/shade) are three implementations of one specification: each must reproduce the
oracle's image and ray count bit for bit."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, bits

pytestmark = pytest.mark.gpu
KAT = np.load(os.path.join(GOLDEN, "oracle_kat.npz"))
PIPELINES = {"fused": 0, "split": 1, "persistent": 2}


@pytest.mark.parametrize("pipeline", list(PIPELINES))
def test_golden_small_image_every_pipeline(rt, pa, cornell, pipeline):
    W, H = 64, 48
    view, proj = pa.reference_camera(W, H)
    rt.setPipeline(PIPELINES[pipeline])
    rt.buildBlas(cornell)
    rt.resize(W, H)
    rt.setCamera(view, proj)
    rt.setParams(spp_per_frame=4, max_depth=8, seed=0)
    rt.raytrace(3)
    assert np.array_equal(bits(rt.readImage()), bits(KAT["cornell_image_64x48_f3_s4_d8"]))
    assert rt.stats().rays_total == int(KAT["cornell_image_rays"][0])
    rt.setParams(spp_per_frame=10, max_depth=20, seed=7)  # the reference's own constants
    rt.resetFrame()
    rt.raytrace(2)
    assert np.array_equal(bits(rt.readImage()), bits(KAT["cornell_image_64x48_f2_s10_d20_seed7"]))


@pytest.mark.parametrize("pipeline", list(PIPELINES))
@pytest.mark.parametrize("W,H,spp,depth", [(1, 1, 1, 1), (7, 5, 3, 2), (257, 3, 2, 8), (63, 65, 5, 20), (130, 2, 17, 3),
                                            (640, 360, 16, 8)])
def test_ragged_extents_every_pipeline(rt, pa, cornell, ocornell, pipeline, W, H, spp, depth):
    """Partial pixel groups (W*H not a multiple of 64), more samples than a claim holds, depth 1 and 20."""
    view, proj = pa.reference_camera(W, H)
    rt.setPipeline(PIPELINES[pipeline])
    rt.buildBlas(cornell)
    rt.resize(W, H)
    rt.setCamera(view, proj)
    rt.setParams(spp_per_frame=spp, max_depth=depth, seed=9)
    frames = 3 if W * H < 100000 else 2
    rt.raytrace(frames)
    want, rays = ocornell.render(view, proj, W, H, frames=frames, spp_per_frame=spp, max_depth=depth, seed=9)
    assert rt.stats().rays_total == rays
    assert np.array_equal(bits(rt.readImage()), bits(want))


@pytest.mark.parametrize("pipeline", list(PIPELINES))
def test_emission_that_does_not_terminate(rt, pa, O, cornell, pipeline):
    """rchit:56 terminates a path only when Ke.x+Ke.y+Ke.z > 0, while rgen:78 adds emission whenever a
    component is non-zero: with Ke = (2, -2, 0) on the walls a path collects emission at several bounces."""
    mats = cornell.materials.copy()
    for m in range(len(mats)):
        if not np.any(mats[m, 3:6]):
            mats[m, 3:6] = (2.0, -2.0, 0.0) if m % 2 else (0.0, 0.5, -0.5)
    sc = pa.Scene(cornell.vertices, cornell.indices, mats, cornell.mat_indices)
    osc = O.OracleScene(sc.vertices, sc.indices, sc.materials, sc.mat_indices)
    W, H = 96, 64
    view, proj = pa.reference_camera(W, H)
    rt.setPipeline(PIPELINES[pipeline])
    rt.buildBlas(sc)
    rt.resize(W, H)
    rt.setCamera(view, proj)
    rt.setParams(spp_per_frame=5, max_depth=6, seed=3)
    rt.raytrace(2)
    want, rays = osc.render(view, proj, W, H, frames=2, spp_per_frame=5, max_depth=6, seed=3)
    assert rt.stats().rays_total == rays
    assert np.array_equal(bits(rt.readImage()), bits(want))
    assert (want[..., 1] < 0).any()  # negative emission really was collected


@pytest.mark.parametrize("pipeline", list(PIPELINES))
def test_mesh_scene_every_pipeline(rt, pa, O, pipeline):
    """A scene with a deeper tree than Cornell (4.5k triangles, open to the sky: most paths end in a miss)."""
    sc = pa.scenes.grid_mesh(48, 47)
    osc = O.OracleScene(sc.vertices, sc.indices, sc.materials, sc.mat_indices)
    W, H = 160, 90
    view, proj = pa.reference_camera(W, H)
    rt.setPipeline(PIPELINES[pipeline])
    rt.buildBlas(sc)
    rt.resize(W, H)
    rt.setCamera(view, proj)
    rt.setParams(spp_per_frame=3, max_depth=5, seed=1)
    rt.raytrace(2)
    want, rays = osc.render(view, proj, W, H, frames=2, spp_per_frame=3, max_depth=5, seed=1)
    assert rt.stats().rays_total == rays
    assert np.array_equal(bits(rt.readImage()), bits(want))


def test_pipelines_agree_at_full_size(rt, pa, cornell):
    """BASELINE.json configs[1] extent (1920x1080, 16 spp, depth 8): too slow for the oracle, so the three
    pipelines check each other -- whole image and ray count, bit for bit, frames 0 and 1."""
    W, H = 1920, 1080
    view, proj = pa.reference_camera(W, H)
    rt.buildBlas(cornell)
    rt.resize(W, H)
    rt.setCamera(view, proj)
    rt.setParams(spp_per_frame=16, max_depth=8, seed=0)
    out = {}
    for name, mode in PIPELINES.items():
        rt.setPipeline(mode)
        rt.resetFrame()
        rt.resetStats()
        rt.raytrace(2)
        out[name] = (rt.readImage().copy(), rt.stats().rays_total)
    for name in ("split", "persistent"):
        assert out[name][1] == out["fused"][1]
        assert np.array_equal(bits(out[name][0]), bits(out["fused"][0])), name


@pytest.mark.parametrize("pipeline", list(PIPELINES))
def test_tiles_reproduce_the_full_image(rt, pa, cornell, ocornell, pipeline):
    """pt_set_tile: four ragged tiles rendered one after the other into zeroed images sum to the full image
    bit for bit (x + 0 = x), and each tile alone equals the oracle restricted to it; rays add up."""
    from pathtracer_amd.api import PT_ACCUM_SUM
    W, H = 150, 67
    view, proj = pa.reference_camera(W, H)
    rt.setPipeline(PIPELINES[pipeline])
    rt.buildBlas(cornell)
    rt.resize(W, H)
    rt.setCamera(view, proj)
    rt.setParams(spp_per_frame=3, max_depth=6, seed=2)
    rt.setAccumMode(PT_ACCUM_SUM)
    want, rays = ocornell.render(view, proj, W, H, frames=2, spp_per_frame=3, max_depth=6, seed=2, accum_sum=True)
    total = np.zeros_like(want)
    nrays = 0
    for (x0, y0, w, h) in [(0, 0, 71, 30), (71, 0, 79, 30), (0, 30, 1, 37), (1, 30, 149, 37)]:
        rt.resize(W, H)  # zeroes the image, clears the tile
        rt.setTile(x0, y0, w, h)
        rt.resetStats()
        rt.raytrace(2)
        img = rt.readImage()
        inside = np.zeros((H, W), dtype=bool)
        inside[y0:y0 + h, x0:x0 + w] = True
        assert np.array_equal(bits(img[inside]), bits(want[inside]))
        assert np.all(img[~inside] == 0.0)
        total += img
        nrays += rt.stats().rays_total
    assert np.array_equal(bits(total[..., :3]), bits(want[..., :3])) and nrays == rays
    with pytest.raises(pa.PathTracerError):
        rt.setTile(100, 0, 51, 10)
    rt.setTile()  # back to the whole image
    rt.resetFrame()
    rt.raytrace(2)
    assert np.array_equal(bits(rt.readImage()), bits(want))


@pytest.mark.parametrize("pipeline", sorted(PIPELINES))
def test_interleaved_tiles_reproduce_the_full_image(rt, pa, cornell, ocornell, pipeline, tmp_path):
    """pt_set_tile_interleaved: blocks of 7 columns dealt round-robin to 3 ranks (150 columns: the last block is cut short and
    the ranks get 52 / 49 / 49 columns): every rank's image equals the oracle on its columns and is zero elsewhere, the three
    sum to the full image bit for bit, rays add up; a checkpoint taken on an interleaved tile resumes on it."""
    from pathtracer_amd import multigpu
    from pathtracer_amd.api import PT_ACCUM_SUM
    W, H, G, B = 150, 67, 3, 7
    view, proj = pa.reference_camera(W, H)
    rt.setPipeline(PIPELINES[pipeline])
    rt.buildBlas(cornell)
    rt.resize(W, H)
    rt.setCamera(view, proj)
    rt.setParams(spp_per_frame=3, max_depth=6, seed=2)
    rt.setAccumMode(PT_ACCUM_SUM)
    want, rays = ocornell.render(view, proj, W, H, frames=3, spp_per_frame=3, max_depth=6, seed=2, accum_sum=True)
    total = np.zeros_like(want)
    nrays = 0
    seen = []
    for r in range(G):
        cols = multigpu.interleaved_columns(r, G, W, B)
        seen += cols
        rt.resize(W, H)  # zeroes the image, clears the tile
        rt.setTileInterleaved(r, G, B)
        rt.resetStats()
        rt.raytrace(2)
        if r == 1:
            ck = str(tmp_path / "tile.ckpt")
            rt.saveCheckpoint(ck)
            rt.resize(W, H)
            rt.loadCheckpoint(ck)
        rt.raytrace(1)
        img = rt.readImage()
        inside = np.zeros((H, W), dtype=bool)
        inside[:, cols] = True
        assert np.array_equal(bits(img[inside]), bits(want[inside]))
        assert np.all(img[~inside] == 0.0)
        assert rt.stats().paths_last_frame == len(cols) * H * 3
        total += img
        nrays += rt.stats().rays_total
    assert sorted(seen) == list(range(W))
    assert np.array_equal(bits(total[..., :3]), bits(want[..., :3])) and nrays == rays
    for bad in ((3, 3, 7), (0, 0, 7), (0, 3, 0), (2, 3, 75)):  # rank outside the world; no world; no block; a rank without a column
        with pytest.raises(pa.PathTracerError):
            rt.setTileInterleaved(*bad)


def test_many_frames_per_launch_equal_frame_by_frame(rt, pa, cornell, ocornell):
    """pt_render_frames(n) puts several frames into one launch of the persistent kernel (PT_FRAMES_PER_LAUNCH: 8, flat scenes
    like this one 16; a call's last launch takes up to twice that rather than leave a short one behind): 19 frames in one
    call (one launch of 19) and 53 frames (16 + 16 + 21) == as many single-frame calls == the oracle, bit for bit; the
    max_frames cut-off may fall inside a launch."""
    W, H = 50, 21
    view, proj = pa.reference_camera(W, H)
    rt.buildBlas(cornell)
    rt.resize(W, H)
    rt.setCamera(view, proj)
    rt.setParams(spp_per_frame=2, max_depth=6, seed=5, max_frames=0)
    rt.raytrace(19)
    assert rt.frame == 18
    a, rays_a = rt.readImage(), rt.stats().rays_total
    want, rays = ocornell.render(view, proj, W, H, frames=19, spp_per_frame=2, max_depth=6, seed=5)
    assert rays_a == rays and np.array_equal(bits(a), bits(want))
    rt.resetFrame()
    rt.resetStats()
    for _ in range(19):
        rt.raytrace(1)
    assert rt.stats().rays_total == rays and np.array_equal(bits(rt.readImage()), bits(want))
    rt.resetFrame()
    rt.resetStats()
    rt.raytrace(53)
    want53, rays53 = ocornell.render(view, proj, W, H, frames=53, spp_per_frame=2, max_depth=6, seed=5)
    assert rt.frame == 52 and rt.stats().rays_total == rays53 and np.array_equal(bits(rt.readImage()), bits(want53))
    rt.setParams(spp_per_frame=2, max_depth=6, seed=5, max_frames=11)  # frames 0..10 render, the rest are no-ops
    rt.resetFrame()
    rt.raytrace(19)
    assert rt.frame == 18
    want11, _ = ocornell.render(view, proj, W, H, frames=11, spp_per_frame=2, max_depth=6, seed=5)
    assert np.array_equal(bits(rt.readImage()), bits(want11))


def test_every_node_source_of_the_path_kernel(rt, pa, O):
    """k_paths reads its nodes from decoded LDS records (<= 8 wide nodes), from a packed LDS copy (<= 16) or from memory
    (pt_wave_traverse.h): small meshes of growing size land in each of the three, with half-empty nodes among them;
    image and ray count against the oracle bit for bit, node visits and triangle tests equal to the fused pipeline's."""
    from pathtracer_amd.api import PT_PROFILE_TRAVERSAL
    W, H = 96, 64
    view = pa.look_at((0.0, 1.1, 2.6), (0.0, 0.1, 0.0), (0, 1, 0))
    proj = pa.perspective(np.radians(45.0), W / H, 0.1, 100.0)
    proj[5] *= -1
    seen = set()
    for nx, nz, leaf in [(1, 1, 0), (2, 2, 0), (3, 3, 0), (4, 3, 0), (4, 4, 0), (5, 5, 0), (6, 5, 1), (7, 7, 0), (9, 8, 0)]:
        sc = pa.scenes.grid_mesh(nx, nz, size=2.0, amplitude=0.3)
        osc = O.OracleScene(sc.vertices, sc.indices, sc.materials, sc.mat_indices)
        want, rays = osc.render(view, proj, W, H, frames=2, spp_per_frame=4, max_depth=6, seed=3)
        # with the entry node (default: the tree top is tested while a ray is staged) and without (every ray starts at
        # the root: then the persistent kernel takes exactly the fused kernel's steps, with nodes from LDS or memory)
        for entry_flag in (0, pa.api.PT_ACCEL_NO_ENTRY_NODE):
            rt.uploadScene(sc)
            rt.buildBlas(None, pa.api.PT_ACCEL_BVH4 | (leaf << 8) | entry_flag)
            nodes = rt.stats().bvh_nodes
            seen.add("planes" if nodes <= 8 else ("packed" if nodes <= 16 else "global"))
            rt.resize(W, H)
            rt.setCamera(view, proj)
            rt.setParams(spp_per_frame=4, max_depth=6, seed=3)
            trav = {}
            for name in ("persistent", "fused"):
                rt.setPipeline(PIPELINES[name])
                for prof in (0, PT_PROFILE_TRAVERSAL):  # the counting kernels are instantiations of their own
                    rt.setProfiling(prof)
                    rt.resetFrame()
                    rt.resetStats()
                    rt.raytrace(2)
                    st = rt.stats()
                    assert st.rays_total == rays, (nx, nz, name)
                    assert np.array_equal(bits(rt.readImage()), bits(want)), (nx, nz, nodes, name, prof, entry_flag)
                trav[name] = (st.trav_nodes, st.trav_tris, st.trav_hits)
            if entry_flag:
                assert trav["persistent"] == trav["fused"], (nx, nz, nodes, trav)
            else:
                assert trav["persistent"][2] == trav["fused"][2], (nx, nz, nodes, trav)  # the same hits either way
            rt.setProfiling(0)
    assert seen == {"planes", "packed", "global"}, seen
