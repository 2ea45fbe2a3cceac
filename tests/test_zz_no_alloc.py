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


# ------------------------------------------------------------------------------------------------------------------------
# The property the entry point exists for, checked directly: nothing is allocated, and it can be recorded into a HIP graph.
def _hip():
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    for name in ("hipStreamCreate", "hipStreamDestroy", "hipStreamSynchronize", "hipStreamBeginCapture", "hipStreamEndCapture",
                 "hipGraphInstantiate", "hipGraphLaunch", "hipGraphExecDestroy", "hipGraphDestroy", "hipMemGetInfo",
                 "hipDeviceSynchronize"):
        getattr(hip, name).restype = C.c_int
    hip.hipStreamBeginCapture.argtypes = [C.c_void_p, C.c_int]
    hip.hipStreamEndCapture.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    hip.hipGraphInstantiate.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    hip.hipGraphLaunch.argtypes = [C.c_void_p, C.c_void_p]
    hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
    for name in ("hipStreamDestroy", "hipStreamSynchronize", "hipGraphExecDestroy", "hipGraphDestroy"):
        getattr(hip, name).argtypes = [C.c_void_p]
    hip.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    return hip


def _free_bytes(hip):
    import ctypes as C
    f, t = C.c_size_t(), C.c_size_t()
    assert hip.hipDeviceSynchronize() == 0 and hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
    return f.value


def _setup(pa, scene, W, H, spp, depth, seed, instances=None):
    rt = pa.HipRaytracing(0)
    view, proj = pa.reference_camera(W, H)
    if instances is not None:
        rt.uploadScene(scene)
        rt.setInstances(instances)
        rt.buildBlas()
    else:
        rt.buildBlas(scene)
    rt.buildTlas()
    rt.resize(W, H)
    rt.setCamera(view, proj)
    rt.setParams(spp_per_frame=spp, max_depth=depth, seed=seed, max_frames=0)
    return rt


def _three_boxes():
    """object-to-world 3x4 of three Cornell boxes: shifted, rotated about y and scaled, mirrored"""
    import math
    c, s = math.cos(0.6), math.sin(0.6)
    return np.array([[1, 0, 0, -1.2, 0, 1, 0, 0, 0, 0, 1, -1.0],
                     [0.7 * c, 0, 0.7 * s, 1.1, 0, 0.8, 0, 0.1, -0.7 * s, 0, 0.7 * c, -1.5],
                     [-1, 0, 0, 0.2, 0, 1, 0, 1.9, 0, 0, 1, -3.0]], dtype=np.float32)


def test_noalloc_allocates_nothing_and_mixes_with_render_frames(pa, cornell):
    """Device memory free before == free after a pt_render_frames_noalloc call (hipMemGetInfo), on a single-level and on a
    two-level scene, also on a context whose pt_render_frames buffers already exist; and the two entry points interleaved on ONE
    context continue each other's accumulation exactly (image and ray count of raytrace, noalloc, raytrace == one raytrace)."""
    hip = _hip()
    for instances in (None, _three_boxes()):
        W, H, spp, depth = 160, 120, 3, 6
        rt = _setup(pa, cornell, W, H, spp, depth, 7, instances)
        rt.raytraceNoAlloc(1)  # (the first launch of a kernel loads its code object: not the entry point's allocation)
        rt.raytrace(2)         # pt_render_frames creates its own buffers: they must be neither used nor grown below
        rt.resetFrame()
        rt.resetStats()
        before = _free_bytes(hip)
        rt.raytraceNoAlloc(3)
        rt.raytraceNoAlloc(1)
        assert _free_bytes(hip) == before
        rt.raytrace(2)
        rt.raytraceNoAlloc(2)
        got, rays = rt.readImage(), rt.stats().rays_total
        rt.cleanup()
        ref = _setup(pa, cornell, W, H, spp, depth, 7, instances)
        ref.raytrace(8)
        want, wrays = ref.readImage(), ref.stats().rays_total
        ref.cleanup()
        assert rays == wrays and np.array_equal(bits(got), bits(want)) and got[..., :3].max() > 0


def test_noalloc_under_hip_graph_capture(pa, cornell):
    """pt_render_frames_noalloc recorded between hipStreamBeginCapture / hipStreamEndCapture (global mode: any call that is
    not allowed under capture fails it) on a caller's stream, instantiated and launched.  The frame numbers and seed words
    are kernel ARGUMENTS, fixed when the call is recorded: a graph replays exactly the frames it recorded -- launching the
    first graph (frames 0 .. F-1, which overwrite the image) twice leaves the image of F frames; a second graph recorded
    after it continues the accumulation with frames F .. 2F-1.  Both bit-identical to pt_render_frames."""
    import ctypes as C
    from pathtracer_amd import _lib
    L = _lib.lib()
    hip = _hip()
    W, H, spp, depth, F = 200, 150, 4, 8, 3
    rt = _setup(pa, cornell, W, H, spp, depth, 2)
    rt.raytraceNoAlloc(1)  # code objects loaded, scratch address known: nothing left to initialise under capture
    rt.resetFrame()
    rt.resetStats()
    stream = C.c_void_p()
    assert hip.hipStreamCreate(C.byref(stream)) == 0
    rt.set_stream(stream.value)
    graphs, execs = [], []
    for k in range(2):
        assert hip.hipStreamBeginCapture(stream, 0) == 0  # hipStreamCaptureModeGlobal
        rc = L.pt_render_frames_noalloc(rt._ctx, F)
        g = C.c_void_p()
        assert hip.hipStreamEndCapture(stream, C.byref(g)) == 0 and g.value
        assert rc == 0, L.pt_last_error(rt._ctx).decode()
        e = C.c_void_p()
        assert hip.hipGraphInstantiate(C.byref(e), g, None, None, 0) == 0 and e.value
        graphs.append(g)
        execs.append(e)
    assert rt.frame == 2 * F - 1  # the host-side counter advanced when the calls were recorded
    ref = _setup(pa, cornell, W, H, spp, depth, 2)
    ref.raytrace(F)
    want1 = ref.readImage()
    ref.raytrace(F)
    want2 = ref.readImage()
    ref.cleanup()
    assert hip.hipGraphLaunch(execs[0], stream) == 0 and hip.hipStreamSynchronize(stream) == 0
    got = rt.readImage()
    assert np.array_equal(bits(got), bits(want1)) and got[..., :3].max() > 0
    assert hip.hipGraphLaunch(execs[0], stream) == 0 and hip.hipStreamSynchronize(stream) == 0
    assert np.array_equal(bits(rt.readImage()), bits(want1))  # a replay renders the same frames again
    assert hip.hipGraphLaunch(execs[1], stream) == 0 and hip.hipStreamSynchronize(stream) == 0
    assert np.array_equal(bits(rt.readImage()), bits(want2))
    for e in execs:
        hip.hipGraphExecDestroy(e)
    for g in graphs:
        hip.hipGraphDestroy(g)
    rt.set_stream(None)
    hip.hipStreamDestroy(stream)
    rt.cleanup()
