"""GPU: edge cases and error behaviour of the C ABI (empty and ragged inputs, limits, call order)."""
import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu


def test_empty_scene_renders_black(rt, pa):
    sc = pa.Scene(np.zeros((0, 11), np.float32), np.zeros(0, np.uint32), np.array([[.5, .5, .5, 0, 0, 0]], np.float32),
                  np.zeros(0, np.uint32))
    rt.buildBlas(sc)
    W, H = 33, 17
    view, proj = pa.reference_camera(W, H)
    rt.resize(W, H)
    rt.setCamera(view, proj)
    rt.setParams(spp_per_frame=3, max_depth=4)
    rt.raytrace(2)
    img = rt.readImage()
    assert np.all(img[..., :3] == 0.0) and np.all(img[..., 3] == 1.0)
    assert rt.stats().rays_total == 2 * W * H * 3  # every primary ray is traced once and misses
    assert np.all(rt.debugPrimaryHits()["prim"] == 0xFFFFFFFF)


@pytest.mark.parametrize("W,H,spp,depth", [(1, 1, 1, 1), (7, 5, 3, 2), (257, 3, 2, 8), (63, 65, 5, 20), (130, 2, 17, 3),
                                           (40, 30, 2, 64),   # (64: the largest depth; it shares a byte with a flag)
                                           (9, 7, 70, 6)])    # (more than 64 samples: k_accumulate_items fetches the mask in 64s)
def test_ragged_extents_bit_exact(rt, pa, cornell, ocornell, W, H, spp, depth):
    view, proj = pa.reference_camera(W, H)
    rt.buildBlas(cornell)
    rt.resize(W, H)
    rt.setCamera(view, proj)
    rt.setParams(spp_per_frame=spp, max_depth=depth, seed=9)
    rt.raytrace(3)
    want, rays = ocornell.render(view, proj, W, H, frames=3, spp_per_frame=spp, max_depth=depth, seed=9)
    assert rt.stats().rays_total == rays
    assert np.array_equal(bits(rt.readImage()), bits(want))


def test_max_frames_stops_accumulation(rt, pa, cornell):
    """VulkanRaytracing::raytrace is a no-op once frame >= maxRTFrames, but the counter keeps running
    (VulkanRaytracing.cpp:877-886, 904-908)."""
    W, H = 32, 32
    view, proj = pa.reference_camera(W, H)
    rt.buildBlas(cornell)
    rt.resize(W, H)
    rt.setCamera(view, proj)
    rt.setParams(spp_per_frame=2, max_depth=4, max_frames=3)
    rt.raytrace(3)
    a = rt.readImage()
    rays = rt.stats().rays_total
    rt.raytrace(4)
    assert rt.frame == 6
    assert np.array_equal(bits(rt.readImage()), bits(a)) and rt.stats().rays_total == rays


def test_resize_and_reupload_reset_state(rt, pa, cornell, ocornell):
    view, proj = pa.reference_camera(48, 48)
    rt.buildBlas(cornell)
    rt.resize(48, 48)
    rt.setCamera(view, proj)
    rt.setParams(spp_per_frame=2, max_depth=4)
    rt.raytrace(2)
    rt.resize(20, 10)  # resize -> resetFrame (main.cpp:433-435)
    assert rt.frame == -1
    view, proj = pa.reference_camera(20, 10)
    rt.setCamera(view, proj)
    rt.raytrace(1)
    want, _ = ocornell.render(view, proj, 20, 10, frames=1, spp_per_frame=2, max_depth=4)
    assert np.array_equal(bits(rt.readImage()), bits(want))
    # growing the queues (more spp) after the fact
    rt.setParams(spp_per_frame=9, max_depth=4)
    rt.resetFrame()
    rt.raytrace(1)
    want, _ = ocornell.render(view, proj, 20, 10, frames=1, spp_per_frame=9, max_depth=4)
    assert np.array_equal(bits(rt.readImage()), bits(want))


def test_degenerate_triangles_and_nan_are_harmless(rt, pa, O):
    sc = pa.scenes.grid_mesh(6, 6)
    v = sc.vertices.copy()
    idx = np.concatenate([sc.indices, np.array([0, 0, 0, 1, 2, 1, 3, 3, 4], dtype=np.uint32)])  # zero-area triangles
    mid = np.concatenate([sc.mat_indices, np.zeros(3, dtype=np.uint32)])
    sc2 = pa.Scene(v, idx, sc.materials, mid)
    rt.buildBlas(sc2)
    rng = np.random.default_rng(3)
    rays = np.concatenate([rng.uniform(-1, 1, (20000, 3)), rng.normal(size=(20000, 3))], axis=1).astype(np.float32)
    rays[:, 1] = np.abs(rays[:, 1]) + 0.2
    rays[0, 3:] = 0.0          # zero direction
    rays[1, 3] = np.nan        # NaN direction
    osc = O.OracleScene(sc2.vertices, sc2.indices, sc2.materials, sc2.mat_indices, use_bvh=False)
    got, want = rt.debugTrace(rays), osc.trace(rays, brute=True)
    assert np.array_equal(got["prim"], want["prim"])
    assert got["prim"][0] == 0xFFFFFFFF and got["prim"][1] == 0xFFFFFFFF
    assert not np.isin(got["prim"], [sc.n_tris, sc.n_tris + 1, sc.n_tris + 2]).any()


def test_error_codes(rt, pa, cornell):
    with pytest.raises(pa.PathTracerError) as e:
        rt.raytrace(1)  # nothing uploaded
    assert e.value.code == -1
    bad = pa.Scene(cornell.vertices, cornell.indices.copy(), cornell.materials, cornell.mat_indices)
    bad.indices[5] = 1000
    with pytest.raises(pa.PathTracerError):
        rt.uploadScene(bad)
    bad2 = pa.Scene(cornell.vertices, cornell.indices, cornell.materials, cornell.mat_indices.copy())
    bad2.mat_indices[0] = 99
    with pytest.raises(pa.PathTracerError):
        rt.uploadScene(bad2)
    rt.uploadScene(cornell)
    with pytest.raises(pa.PathTracerError):
        rt.raytrace(1)  # accel not built
    rt.buildBlas()
    with pytest.raises(pa.PathTracerError):
        rt.raytrace(1)  # no image / camera
    with pytest.raises(pa.PathTracerError):
        rt.resize(0, 10)
    with pytest.raises(pa.PathTracerError) as e:
        rt.setParams(max_depth=65)
    assert e.value.code == -5
    with pytest.raises(pa.PathTracerError):
        rt.setParams(spp_per_frame=0)
    with pytest.raises(pa.PathTracerError):
        rt.setParams(tmin=-1.0)
    with pytest.raises(pa.PathTracerError):
        pa.HipRaytracing(4096)


def test_checkpoint_resume_is_bit_identical(pa, cornell, tmp_path):
    """pt_save_checkpoint / pt_load_checkpoint: 2 frames, save, a NEW context loads
    and renders 3 more == 5 frames in one go, bit for bit (frame index, seed, schedule, mode all restored)."""
    from pathtracer_amd.api import PT_ACCUM_SUM
    W, H = 80, 45
    view, proj = pa.reference_camera(W, H)

    def fresh():
        r = pa.HipRaytracing(0)
        r.buildBlas(cornell)
        r.setCamera(view, proj)
        return r

    a = fresh()
    a.resize(W, H)
    a.setParams(spp_per_frame=3, max_depth=6, seed=11, max_frames=0)
    a.setFrameSchedule(first=1, stride=2)
    a.setAccumMode(PT_ACCUM_SUM)
    a.raytrace(5)
    want = a.readImage()
    a.resetFrame()
    a.raytrace(2)
    ck = str(tmp_path / "acc.ckpt")
    a.saveCheckpoint(ck)
    a.cleanup()
    b = fresh()
    b.loadCheckpoint(ck)
    assert b.frame == 1
    b.raytrace(3)
    assert b.frame == 4
    assert np.array_equal(bits(b.readImage()), bits(want))
    with pytest.raises(pa.PathTracerError):
        b.loadCheckpoint(str(tmp_path / "missing.ckpt"))
    open(tmp_path / "bad.ckpt", "wb").write(b"not a checkpoint at all" * 10)
    with pytest.raises(pa.PathTracerError):
        b.loadCheckpoint(str(tmp_path / "bad.ckpt"))
    b.cleanup()


def test_camera_change_restarts_accumulation(rt, pa, cornell, ocornell):
    """VulkanRaytracing::updateFrame: "Resets the frame counter if the camera has changed" (its TODO, built here)."""
    W, H = 96, 64
    view, proj = pa.reference_camera(W, H)
    rt.buildBlas(cornell)
    rt.resize(W, H)
    rt.setCamera(view, proj)
    rt.setParams(spp_per_frame=4, max_depth=6, seed=2)
    rt.raytrace(2)
    rt.setCamera(view, proj)  # the per-frame updateUniformBuffer of the reference: same matrices, nothing happens
    assert rt.frame == 1
    rt.raytrace(1)
    want3, _ = ocornell.render(view, proj, W, H, 3, spp_per_frame=4, max_depth=6, seed=2)
    assert np.array_equal(bits(rt.readImage()), bits(want3))
    view2 = pa.look_at((0.3, 1.2, 5.5), (0.0, 1.0, 0.0), (0, 1, 0))
    rt.setCamera(view2, proj)
    assert rt.frame == -1
    rt.raytrace(2)
    want2, _ = ocornell.render(view2, proj, W, H, 2, spp_per_frame=4, max_depth=6, seed=2)
    assert np.array_equal(bits(rt.readImage()), bits(want2))


def test_tiled_checkpoint_resume_and_header_validation(pa, cornell, tmp_path):
    """a tile-split rank that resumes must still render only its tile (the checkpoint carries the tile), and a
    header with out-of-range fields must be refused without touching the context."""
    import struct
    from pathtracer_amd.api import PT_ACCUM_SUM
    W, H = 64, 40
    view, proj = pa.reference_camera(W, H)

    def fresh():
        r = pa.HipRaytracing(0)
        r.buildBlas(cornell)
        r.setCamera(view, proj)
        return r

    a = fresh()
    a.resize(W, H)
    a.setParams(spp_per_frame=2, max_depth=5, seed=4, max_frames=0)
    a.setAccumMode(PT_ACCUM_SUM)
    a.setTile(16, 0, 24, H)
    a.raytrace(4)
    want = a.readImage()
    assert not want[:, :16, :3].any() and not want[:, 40:, :3].any() and want[:, 16:40, :3].any()
    a.resetFrame()
    a.resize(W, H)
    a.setTile(16, 0, 24, H)
    a.raytrace(1)
    ck = str(tmp_path / "tile.ckpt")
    a.saveCheckpoint(ck)
    a.cleanup()
    b = fresh()
    b.loadCheckpoint(ck)
    b.raytrace(3)
    assert np.array_equal(bits(b.readImage()), bits(want))  # columns outside the strip are still zero
    # corrupt headers: frame index, stride, spp, tile -- each refused, the context keeps rendering as before
    blob = bytearray(open(ck, "rb").read())
    assert blob[:8] == b"PTCKPT03"  # (round 3: + block width and stride of an interleaved tile, 8 bytes behind the tile)
    fields = {"frame": (16, "<i", -7), "frame_big": (16, "<i", 0x7FFFFFFF), "stride": (28, "<I", 0), "spp": (36, "<i", 0),
              "tile_w": (64, "<I", 1000), "tile_block_without_stride": (72, "<I", 8)}
    for name, (off, fmt, val) in fields.items():
        bad = bytearray(blob)
        struct.pack_into(fmt, bad, off, val)
        p = tmp_path / f"bad_{name}.ckpt"
        open(p, "wb").write(bad)
        before = b.frame
        with pytest.raises(pa.PathTracerError):
            b.loadCheckpoint(str(p))
        assert b.frame == before, name
    open(tmp_path / "short.ckpt", "wb").write(blob[:-16])
    with pytest.raises(pa.PathTracerError):
        b.loadCheckpoint(str(tmp_path / "short.ckpt"))
    # a version-1 file (no tile) still loads, as the whole image
    v1 = b"PTCKPT01" + bytes(blob[8:56]) + bytes(blob[80:])
    open(tmp_path / "v1.ckpt", "wb").write(v1)
    b.loadCheckpoint(str(tmp_path / "v1.ckpt"))
    assert b.frame == 0
    # ... and a version-2 file (a rectangular tile, no block fields) as that tile: the resumed run equals the uninterrupted one
    v2 = b"PTCKPT02" + bytes(blob[8:72]) + bytes(blob[80:])
    open(tmp_path / "v2.ckpt", "wb").write(v2)
    b.loadCheckpoint(str(tmp_path / "v2.ckpt"))
    b.raytrace(3)
    assert np.array_equal(bits(b.readImage()), bits(want))
    b.cleanup()


@pytest.mark.parametrize("n_tris", [3, 6, 40, 300])
def test_tiny_far_geometry_matches_brute_force(rt, pa, O, n_tris):
    """the node step of the wave kernels has no explicit empty-slot test -- an empty slot is an inverted box that
    fails the interval test by itself, EXCEPT for a ray so far away that the node's whole extent (255 quantisation steps)
    vanishes in the test's 1e-6 relative slack; then PT_EMPTY_CHILD is pushed and the leaf step must skip it.  Triangles of
    size 1e-3 seen from a distance of 2e4 through a 2e-7 rad field of view make every node a stray case; 3 ... 300
    triangles cover the three node sources of k_paths (decoded LDS records <= 8 nodes, packed LDS records <= 16, memory),
    and the same scene is rendered instanced and on the other two pipelines.  Image and ray count == brute-force oracle."""
    import math
    from pathtracer_amd.api import PT_ACCEL_BVH4, PT_ACCEL_LEAF_SIZE_1, PT_PIPELINE_FUSED, PT_PIPELINE_PERSISTENT, PT_PIPELINE_SPLIT
    rng = np.random.default_rng(n_tris)
    ext, dist = 2e-3, 2e4
    c = rng.uniform(-ext / 2, ext / 2, (n_tris, 1, 3))
    size = min(8e-4, max(1.5e-4, 2.4e-3 / math.sqrt(n_tris)))
    tri = (c + rng.uniform(-size, size, (n_tris, 3, 3))).astype(np.float32)
    verts = np.zeros((n_tris * 3, 11), dtype=np.float32)
    verts[:, 0:3] = tri.reshape(-1, 3)
    verts[:, 3:6] = [0, 0, 1]
    verts[:, 6:9] = 1.0
    mats = np.array([[0.8, 0.8, 0.8, 0, 0, 0], [0, 0, 0, 15, 15, 15]], dtype=np.float32)
    sc = pa.Scene(verts, np.arange(n_tris * 3, dtype=np.uint32), mats, (np.arange(n_tris) % 2).astype(np.uint32))
    W, H = 64, 48
    view = pa.look_at((0.0, 0.0, dist), (0.0, 0.0, 0.0), (0, 1, 0))
    proj = pa.perspective(2.0 * math.atan(0.6 * ext / dist), W / H, 1.0, 4.0 * dist)
    proj[5] *= -1
    args = dict(spp_per_frame=2, max_depth=3, tmax=1e5, seed=3)
    osc = O.OracleScene(sc.vertices, sc.indices, sc.materials, sc.mat_indices, use_bvh=False)
    want, rays = osc.render(view, proj, W, H, frames=2, **args)
    assert want[..., :3].max() == 15.0 and 0.01 < (want[..., :3].sum(-1) > 0).mean() < 0.98
    for flags in (PT_ACCEL_BVH4 | PT_ACCEL_LEAF_SIZE_1, 2):  # LBVH with one triangle per leaf (3-child nodes), PLOC default
        rt.buildBlas(sc, flags)
        rt.resize(W, H)
        rt.setCamera(view, proj)
        rt.setParams(max_frames=0, **args)
        for pipe in (PT_PIPELINE_PERSISTENT, PT_PIPELINE_FUSED, PT_PIPELINE_SPLIT):
            rt.setPipeline(pipe)
            rt.resetFrame()
            rt.resetStats()
            rt.raytrace(2)
            assert np.array_equal(bits(rt.readImage()), bits(want)), (flags, pipe)
            assert rt.stats().rays_total == rays, (flags, pipe)
    # instanced: two copies side by side, far away as well
    xf = np.array([[1, 0, 0, -ext, 0, 1, 0, 0, 0, 0, 1, 0], [1, 0, 0, ext, 0, 1, 0, 0, 0, 0, 1, 0]], dtype=np.float32)
    proj2 = pa.perspective(2.0 * math.atan(1.8 * ext / dist), W / H, 1.0, 4.0 * dist)
    proj2[5] *= -1
    osc2 = O.OracleScene(sc.vertices, sc.indices, sc.materials, sc.mat_indices, use_bvh=False, instances=xf)
    want2, rays2 = osc2.render(view, proj2, W, H, frames=2, **args)
    assert want2[..., :3].max() == 15.0
    rt.setPipeline(PT_PIPELINE_PERSISTENT)
    rt.uploadScene(sc)
    rt.setInstances(xf)
    rt.buildBlas(None, PT_ACCEL_BVH4 | PT_ACCEL_LEAF_SIZE_1)
    rt.resize(W, H)
    rt.setCamera(view, proj2)
    rt.setParams(max_frames=0, **args)
    rt.resetStats()
    rt.raytrace(2)
    assert np.array_equal(bits(rt.readImage()), bits(want2))
    assert rt.stats().rays_total == rays2


@pytest.mark.parametrize("flat", [True, False])
def test_emission_that_does_not_end_the_path(rt, pa, cornell, O, flat):
    """rchit:56-58 ends a path only where the emission's components sum to more than zero.  A wall that emits (3, -3, 0)
    adds to the radiance AND lets the path go on, so a path can write its radiance record several times: the first write
    stores (the record is not zeroed beforehand, a bit in the launch's mask says it has been written), the later ones add.
    The oracle's image, bit for bit, in the flat form (path state travels with the ray) and the tree form (state table)."""
    from pathtracer_amd.api import PT_ACCEL_BVH4_PLOC
    import os
    mats = cornell.materials.copy()
    mats[2, 3:] = (3.0, -3.0, 0.0)  # the white walls, floor and ceiling
    mats[0, 3:] = (0.25, 0.25, -0.75)  # sum < 0
    sc = pa.Scene(cornell.vertices, cornell.indices, mats, cornell.mat_indices)
    osc = O.OracleScene(sc.vertices, sc.indices, sc.materials, sc.mat_indices)
    W, H = 96, 64
    view, proj = pa.reference_camera(W, H)
    want, rays = osc.render(view, proj, W, H, frames=3, spp_per_frame=4, max_depth=8, seed=5)
    old = os.environ.get("PT_NO_FLAT")
    if not flat:
        os.environ["PT_NO_FLAT"] = "1"
    try:
        rt.uploadScene(sc)
        rt.buildBlas(None, PT_ACCEL_BVH4_PLOC)
    finally:
        if not flat:
            if old is None:
                del os.environ["PT_NO_FLAT"]
            else:
                os.environ["PT_NO_FLAT"] = old
    rt.resize(W, H)
    rt.setCamera(view, proj)
    rt.setParams(spp_per_frame=4, max_depth=8, seed=5)
    rt.raytrace(3)
    img = rt.readImage()
    assert rt.stats().rays_total == rays
    assert np.array_equal(bits(img), bits(want))
    assert (want[..., 1] < 0).any() and (want[..., 0] > 0).any()  # the signed emission is in the image
