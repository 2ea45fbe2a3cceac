"""GPU: pt_render_frames_noalloc -- the entry point for callers that cannot allocate (HIP graph capture, tight serving
loops, harnesses that forbid hipMalloc).  It has no counterpart in the reference.  Same frames as pt_render_frames, on a
file-scope device scratch array of fixed size (16 MiB + 34 KiB per GPU): up to 2^18 paths per frame (pixels rounded up to
64, times spp_per_frame) and 256 spp per frame.  Checked against pt_render_frames through the C ABI, bit for bit, on two
calls in a row, the second with a smaller input; oversize inputs are refused with PT_ELIMIT."""
import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu


def render(pa, scene, W, H, spp, depth, frames, noalloc, seed=3, tile=None, ctx=None):
    """Image and ray count of `frames` frames through pt_render_frames or pt_render_frames_noalloc."""
    from pathtracer_amd import _lib
    L = _lib.lib()
    rt = ctx or pa.HipRaytracing(0)
    view, proj = pa.reference_camera(W, H)
    rt.buildBlas(scene)
    rt.resize(W, H)
    rt.setCamera(view, proj)
    rt.setParams(spp_per_frame=spp, max_depth=depth, seed=seed, max_frames=0)
    if tile:
        rt.setTile(*tile)
    rt.resetStats()
    if noalloc:
        rc = L.pt_render_frames_noalloc(rt._ctx, frames)
        assert rc == 0, L.pt_last_error(rt._ctx).decode()
    else:
        rt.raytrace(frames)
    img, rays = rt.readImage(), rt.stats().rays_total
    if ctx is None:
        rt.cleanup()
    return img, rays


def test_noalloc_equals_render_frames_twice_in_a_row(pa, cornell):
    mesh = pa.scenes.grid_mesh(40, 40)  # 3 202 triangles: nodes from memory, a different scene in the second call
    # call 1: 400 x 300 x 2 spp = 240 000 paths per frame (close to the limit), 5 frames; call 2: smaller, other scene
    na = pa.HipRaytracing(0)
    got1 = render(pa, cornell, 400, 300, 2, 8, 5, True, ctx=na)
    got2 = render(pa, mesh, 97, 61, 3, 5, 4, True, seed=9, ctx=na)
    # and a third on yet another context of the same GPU: the scratch array is shared, stale contents must not matter
    got3 = render(pa, cornell, 64, 64, 16, 8, 3, True, seed=1)
    na.cleanup()
    want1 = render(pa, cornell, 400, 300, 2, 8, 5, False)
    want2 = render(pa, mesh, 97, 61, 3, 5, 4, False, seed=9)
    want3 = render(pa, cornell, 64, 64, 16, 8, 3, False, seed=1)
    for got, want in ((got1, want1), (got2, want2), (got3, want3)):
        assert got[1] == want[1]
        assert np.array_equal(bits(got[0]), bits(want[0]))
        assert got[0][..., :3].max() > 0


def test_noalloc_tiles_and_limits(pa, cornell):
    from pathtracer_amd import _lib
    L = _lib.lib()
    rt = pa.HipRaytracing(0)
    W, H = 1024, 512
    view, proj = pa.reference_camera(W, H)
    rt.buildBlas(cornell)
    rt.resize(W, H)
    rt.setCamera(view, proj)
    rt.setParams(spp_per_frame=2, max_depth=6, seed=5, max_frames=0)
    # the whole image is 2^20 paths per frame: refused, nothing rendered
    assert L.pt_render_frames_noalloc(rt._ctx, 1) == -5 and b"exceed" in L.pt_last_error(rt._ctx)
    assert rt.frame == -1
    # ... but it renders tile by tile (8 tiles of 256 x 256 x 2 = 2^17 paths), equal to one pt_render_frames call
    from pathtracer_amd.api import PT_ACCUM_SUM
    rt.setAccumMode(PT_ACCUM_SUM)
    total = np.zeros((H, W, 4), dtype=np.float32)
    for ty in range(2):
        for tx in range(4):
            rt.resize(W, H)
            rt.setTile(256 * tx, 256 * ty, 256, 256)
            assert L.pt_render_frames_noalloc(rt._ctx, 2) == 0, L.pt_last_error(rt._ctx).decode()
            total[..., :3] += rt.readImage()[..., :3]
    rt.resize(W, H)
    rt.raytrace(2)
    assert np.array_equal(bits(total[..., :3]), bits(rt.readImage()[..., :3]))
    rt.setParams(spp_per_frame=300, max_depth=6)
    rt.setTile(0, 0, 8, 8)
    assert L.pt_render_frames_noalloc(rt._ctx, 1) == -5
    rt.cleanup()
