"""CPU: the oracle's renderer (restating raytrace.rgen/.rchit/.rmiss) -- internal consistency,
committed golden images, and the analytic facts SURVEY.md section 4 lists."""
import os

import numpy as np

from conftest import GOLDEN, bits

KAT = np.load(os.path.join(GOLDEN, "oracle_kat.npz"))


def test_golden_primary_hits_and_images(O, ocornell):
    W, H = 64, 48
    view, proj = O.reference_camera(W, H)
    h = ocornell.primary_hits(view, proj, W, H)
    assert np.array_equal(h["prim"], KAT["cornell_primary_64x48_prim"])
    assert np.array_equal(bits(np.stack([h["t"], h["u"], h["v"]], axis=1)), bits(KAT["cornell_primary_64x48_tuv"]))
    img, rays = ocornell.render(view, proj, W, H, frames=3, spp_per_frame=4, max_depth=8, seed=0)
    assert rays == int(KAT["cornell_image_rays"][0])
    assert np.array_equal(bits(img), bits(KAT["cornell_image_64x48_f3_s4_d8"]))
    img2, _ = ocornell.render(view, proj, W, H, frames=2, spp_per_frame=10, max_depth=20, seed=7)
    assert np.array_equal(bits(img2), bits(KAT["cornell_image_64x48_f2_s10_d20_seed7"]))


def test_portable_and_fma_builds_agree_bit_for_bit(O, cornell):
    """liboracle.so (fmaf through libm) and liboracle_fma.so (-mfma) are the same function."""
    import ctypes as C
    port = O.lib(portable=True)
    W, H = 40, 30
    view, proj = O.reference_camera(W, H)
    sc = O.OracleScene(cornell.vertices, cornell.indices, cornell.materials, cornell.mat_indices)
    img_default, _ = sc.render(view, proj, W, H, frames=2, spp_per_frame=3, max_depth=6)
    img_port = np.zeros((H, W, 4), dtype=np.float32)
    pp = O.Params(3, 6, 1e-3, 1e4, 0)
    for k in range(2):
        port.pto_render_frame(C.byref(sc.c), None, view.ctypes.data_as(C.c_void_p), proj.ctypes.data_as(C.c_void_p), W, H,
                              C.byref(pp), k, k, 0, img_port.ctypes.data_as(C.c_void_p), 0)
    assert np.array_equal(bits(img_default), bits(img_port))


def test_light_pixel_is_exactly_ke(O, ocornell):
    """A frame-0 primary ray that hits the light returns emission * 1 = 15 for every sample."""
    W = H = 64
    view, proj = O.reference_camera(W, H)
    h = ocornell.primary_hits(view, proj, W, H)
    light = np.isin(h["prim"], [30, 31]).reshape(H, W)  # the two 'light' triangles (mat index 1)
    assert light.any()
    img, _ = ocornell.render(view, proj, W, H, frames=1, spp_per_frame=5, max_depth=8)
    assert np.all(img[light][:, :3] == 15.0) and np.all(img[..., 3] == 1.0)
    # rays that leave through the open front contribute nothing
    miss = (h["prim"] == O.NONE).reshape(H, W)
    assert np.all(img[miss][:, :3] == 0.0)


def test_progressive_mean_and_sum_modes(O, ocornell):
    W, H = 32, 24
    view, proj = O.reference_camera(W, H)
    kw = dict(spp_per_frame=4, max_depth=6, seed=3)
    frames = [ocornell.render(view, proj, W, H, frames=1, first_frame=f, **kw)[0] for f in range(4)]
    mean, _ = ocornell.render(view, proj, W, H, frames=4, **kw)
    ssum, _ = ocornell.render(view, proj, W, H, frames=4, accum_sum=True, **kw)
    ref = np.mean([f[..., :3].astype(np.float64) for f in frames], axis=0)
    assert np.allclose(mean[..., :3], ref, rtol=1e-5, atol=1e-6)
    assert np.allclose(ssum[..., :3] / 4.0, ref, rtol=1e-5, atol=1e-6)
    # running mean, literally: mix(old, new, 1/(frame+1)) (raytrace.rgen:88-95)
    acc = frames[0][..., :3].copy()
    for k in range(1, 4):
        a = np.float32(1.0) / np.float32(k + 1)
        acc = acc * (np.float32(1.0) - a) + frames[k][..., :3] * a
    assert np.array_equal(bits(acc), bits(mean[..., :3]))


def test_depth_one_sees_only_directly_visible_emission(O, ocornell):
    W = H = 48
    view, proj = O.reference_camera(W, H)
    img, rays = ocornell.render(view, proj, W, H, frames=1, spp_per_frame=2, max_depth=1)
    assert rays == W * H * 2
    assert set(np.unique(img[..., :3])) <= {0.0, 15.0}


def test_oracle_bvh_equals_brute_force(O, pa):
    sc = pa.scenes.grid_mesh(24, 20)
    osc = O.OracleScene(sc.vertices, sc.indices, sc.materials, sc.mat_indices, use_bvh=True)
    rng = np.random.default_rng(5)
    n = 4000
    o = rng.uniform(-1.2, 1.2, size=(n, 3)).astype(np.float32)
    o[:, 1] = rng.uniform(0.2, 1.4, size=n)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:50, 0] = 0.0  # axis-parallel components
    d[50:100, 2] = 0.0
    rays = np.concatenate([o, d], axis=1)
    a, b = osc.trace(rays), osc.trace(rays, brute=True)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert (a["prim"] != O.NONE).mean() > 0.1


def test_morton_definition(O, ocornell, cornell):
    codes, order, cmin, cmax = ocornell.morton()
    assert np.array_equal(codes, KAT["cornell_morton_codes"]) and np.array_equal(order, KAT["cornell_morton_order"])
    # independent numpy restatement of the definition
    tri = cornell.vertices[cornell.indices.reshape(-1, 3), :3]
    c = (tri.min(axis=1) + tri.max(axis=1)) * np.float32(0.5)
    lo, hi = c.min(axis=0), c.max(axis=0)
    assert np.array_equal(lo, cmin) and np.array_equal(hi, cmax)
    q = np.clip(((c - lo) / (hi - lo) * np.float32(1024.0)), 0, 1023).astype(np.uint32)

    def ex(v):
        out = np.zeros_like(v)
        for b in range(10):
            out |= ((v >> b) & 1) << (3 * b)
        return out
    want = (ex(q[:, 0]) << 2) | (ex(q[:, 1]) << 1) | ex(q[:, 2])
    assert np.array_equal(codes, want)
    assert np.array_equal(order, np.argsort(codes, kind="stable").astype(np.uint32))
