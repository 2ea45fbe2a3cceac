"""GPU: BASELINE.json's full size (Cornell 1920x1080, 16 spp per frame, depth 8) -- too big for a
whole-image oracle render in seconds, so: a random pixel sample against the oracle (bit-exact),
plus size-independent properties (determinism under nondeterministic queue order, linearity of the
sum accumulation across frame schedules, exact light pixels, ray accounting)."""
import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu
W, H, S, D = 1920, 1080, 16, 8


def setup(rt, pa, cornell, seed=0):
    view, proj = pa.reference_camera(W, H)
    rt.buildBlas(cornell)
    rt.resize(W, H)
    rt.setCamera(view, proj)
    rt.setParams(spp_per_frame=S, max_depth=D, seed=seed, max_frames=0)
    return view, proj


def test_fullsize_pixel_sample_bit_exact_and_deterministic(rt, pa, cornell, ocornell):
    view, proj = setup(rt, pa, cornell)
    rng = np.random.default_rng(7)
    px = np.unique(rng.integers(0, W * H, size=6000)).astype(np.uint32)
    for gframe in (0, 5):  # centred-jitter frame and a jittered one
        rt.setFrameSchedule(first=gframe, stride=1)
        rt.resetFrame()
        rt.raytrace(1)
        img = rt.readImage().reshape(-1, 4)
        want = ocornell.render_pixels(view, proj, W, H, px, gframe=gframe, spp_per_frame=S, max_depth=D)
        assert np.array_equal(bits(img[px, :3]), bits(want)), gframe
        # same frame again: queue order differs from run to run (atomics), the image must not
        rt.resetFrame()
        rt.raytrace(1)
        assert np.array_equal(bits(rt.readImage().reshape(-1, 4)), bits(img))
    st = rt.stats()
    assert st.paths_last_frame == W * H * S
    assert W * H * S <= st.rays_last_frame <= W * H * S * D


def test_fullsize_sum_linearity_across_schedules(rt, pa, cornell):
    """frames {0,2} + frames {1,3} (two 'ranks') == frames {0,1,2,3} up to fp32 summation order."""
    from pathtracer_amd.api import PT_ACCUM_SUM
    setup(rt, pa, cornell, seed=3)
    rt.setAccumMode(PT_ACCUM_SUM)
    parts = []
    for r in range(2):
        rt.setFrameSchedule(first=r, stride=2)
        rt.resetFrame()
        rt.raytrace(2)
        parts.append(rt.readImage()[..., :3].astype(np.float64))
    rt.setFrameSchedule(first=0, stride=1)
    rt.resetFrame()
    rt.raytrace(4)
    whole = rt.readImage()[..., :3].astype(np.float64)
    assert np.allclose(parts[0] + parts[1], whole, rtol=1e-6, atol=1e-6)
    assert whole.max() == 4 * 15.0  # a pixel that sees the light in all 4 frames


def test_fullsize_light_and_background_pixels(rt, pa, cornell):
    setup(rt, pa, cornell)
    hits = rt.debugPrimaryHits()["prim"].reshape(H, W)
    rt.raytrace(1)  # frame 0: all 16 samples share the pixel-centre primary ray
    img = rt.readImage()
    light = np.isin(hits, [30, 31])
    assert light.sum() > 1000 and np.all(img[light][:, :3] == 15.0)
    assert np.all(img[hits == 0xFFFFFFFF][:, :3] == 0.0)
    assert np.all(img[..., 3] == 1.0) and np.isfinite(img).all()


def test_config3_shape_4k_textured_pixel_sample(rt, pa, O):
    """BASELINE configs[2] in shape: 3840x2160, diffuse map of the reference texture's size (1024^2), depth 8.  The mesh is
    the textured Cornell stand-in (viking_room.obj/.png live in the reference tree and do not travel to the GPU box); a
    random pixel sample against the oracle, bit for bit, plus ray accounting."""
    Wc, Hc, Sc = 3840, 2160, 4
    sc = pa.scenes.textured_cornell()
    tex = np.random.default_rng(3).integers(0, 256, (1024, 1024, 4), dtype=np.uint8)
    view, proj = pa.reference_camera(Wc, Hc)
    rt.buildBlas(sc)
    rt.setTexture(tex)
    rt.resize(Wc, Hc)
    rt.setCamera(view, proj)
    rt.setParams(spp_per_frame=Sc, max_depth=D, seed=11, max_frames=0)
    rt.setFrameSchedule(first=3, stride=8)  # rank 3 of 8: global frame 3
    rt.raytrace(1)
    img = rt.readImage().reshape(-1, 4)
    px = np.unique(np.random.default_rng(4).integers(0, Wc * Hc, size=3000)).astype(np.uint32)
    osc = O.OracleScene(sc.vertices, sc.indices, sc.materials, sc.mat_indices, texture=tex)
    want = osc.render_pixels(view, proj, Wc, Hc, px, gframe=3, spp_per_frame=Sc, max_depth=D, seed=11)
    assert np.array_equal(bits(img[px, :3]), bits(want))
    st = rt.stats()
    assert st.paths_last_frame == Wc * Hc * Sc and Wc * Hc * Sc <= st.rays_last_frame <= Wc * Hc * Sc * D


def test_config4_shape_64x64_instances_pixel_sample(rt, pa, O):
    """BASELINE configs[3] in shape: one 6 274-triangle mesh (stand-in for teapot.obj) instanced on a 64 x 64 grid, spacing 8
    on XZ, random rotations about Y = 25.7 M instanced triangles behind a two-level structure; 1920x1080, depth 8.  A random
    pixel sample against the oracle (which has no top-level structure: every instance for every ray), bit for bit."""
    import math
    n = 64
    sc = pa.scenes.grid_mesh(56, 56, size=6.0, amplitude=0.8)
    rng = np.random.default_rng(1)
    xf = np.zeros((n * n, 12), dtype=np.float32)
    k = 0
    for i in range(n):
        for j in range(n):
            a = rng.uniform(0, 2 * math.pi)
            xf[k] = [math.cos(a), 0, math.sin(a), 8.0 * (i - (n - 1) / 2), 0, 1, 0, 0, -math.sin(a), 0, math.cos(a), 8.0 * (j - (n - 1) / 2)]
            k += 1
    ext = 8.0 * n
    view = pa.look_at((0.0, 0.45 * ext, 0.8 * ext), (0.0, 0.0, 0.0), (0, 1, 0))
    proj = pa.perspective(math.radians(45.0), W / H, 0.1, 10.0 * ext)
    proj[5] *= -1
    rt.uploadScene(sc)
    rt.setInstances(xf)
    rt.buildBlas()
    rt.resize(W, H)
    rt.setCamera(view, proj)
    rt.setParams(spp_per_frame=2, max_depth=D, tmax=1e6, seed=5, max_frames=0)
    rt.setFrameSchedule(first=1, stride=1)
    rt.raytrace(1)
    img = rt.readImage().reshape(-1, 4)
    px = np.unique(np.random.default_rng(6).integers(0, W * H, size=1500)).astype(np.uint32)
    osc = O.OracleScene(sc.vertices, sc.indices, sc.materials, sc.mat_indices, instances=xf)
    want = osc.render_pixels(view, proj, W, H, px, gframe=1, spp_per_frame=2, max_depth=D, tmax=1e6, seed=5)
    assert np.array_equal(bits(img[px, :3]), bits(want))
    assert (want.sum(axis=1) > 0).mean() > 0.02  # some of the sampled pixels do see light


def test_config5_shape_soup_tiles_pixel_sample(rt, pa, O):
    """BASELINE configs[4] in shape (1 M instead of 100 M triangles so that the oracle's own BVH builds in seconds): the
    procedural soup (tea/lcg generator of SURVEY.md 8(d)), LBVH, 1920x1080, depth 8, rendered as 8 column strips of 240
    pixels -- the per-GPU image tiles -- whose sum must equal the untiled frame bit for bit, and a pixel sample must equal
    the oracle."""
    import math
    from pathtracer_amd.api import PT_ACCEL_BVH4, PT_ACCUM_SUM
    n = 1_000_000
    extent = 1000.0 * (n / 1e8) ** (1.0 / 3.0)
    sc = pa.scenes.random_triangles(n, extent=extent, edge=1.0)
    c = extent * 0.5
    view = pa.look_at((c, c * 1.2, c + extent * 1.6), (c, c * 0.9, c), (0, 1, 0))
    proj = pa.perspective(math.radians(45.0), W / H, 0.1, extent * 10)
    proj[5] *= -1
    rt.buildBlas(sc, PT_ACCEL_BVH4)
    rt.resize(W, H)
    rt.setCamera(view, proj)
    rt.setParams(spp_per_frame=2, max_depth=D, tmax=1e6, seed=21, max_frames=0)
    rt.setAccumMode(PT_ACCUM_SUM)
    rt.setFrameSchedule(first=1, stride=1)
    rt.raytrace(1)
    whole = rt.readImage()
    total = np.zeros_like(whole)
    for g in range(8):
        rt.resize(W, H)  # zeroes the image, restores the whole-image tile
        rt.setTile(240 * g, 0, 240, H)
        rt.setFrameSchedule(first=1, stride=1)
        rt.raytrace(1)
        strip = rt.readImage()
        assert not strip[:, :240 * g, :3].any() and not strip[:, 240 * (g + 1):, :3].any()
        total[..., :3] += strip[..., :3]
    assert np.array_equal(bits(total[..., :3]), bits(whole[..., :3]))
    px = np.unique(np.random.default_rng(8).integers(0, W * H, size=2000)).astype(np.uint32)
    osc = O.OracleScene(sc.vertices, sc.indices, sc.materials, sc.mat_indices)
    want = osc.render_pixels(view, proj, W, H, px, gframe=1, spp_per_frame=2, max_depth=D, tmax=1e6, seed=21)
    assert np.array_equal(bits(whole.reshape(-1, 4)[px, :3]), bits(want))


def test_fullsize_whole_frame_bit_exact(rt, pa, cornell, ocornell):
    """The headline frame itself -- all 2 073 600 pixels of one 1920x1080 x 16 spp frame (a jittered one: global frame 2),
    every bit against the oracle, and the ray count (the oracle traces ~10^8 rays for it: a few seconds on the box's cores)."""
    view, proj = setup(rt, pa, cornell, seed=1)
    rt.setFrameSchedule(first=2, stride=1)
    rt.resetStats()
    rt.raytrace(1)
    img = rt.readImage()
    rays = rt.stats().rays_total
    px = np.arange(W * H, dtype=np.uint32)
    want = ocornell.render_pixels(view, proj, W, H, px, gframe=2, spp_per_frame=S, max_depth=D, seed=1)
    assert np.array_equal(bits(img.reshape(-1, 4)[:, :3]), bits(want))
    assert W * H * S < rays <= W * H * S * D
