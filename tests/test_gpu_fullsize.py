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
