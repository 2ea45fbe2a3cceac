"""Two-level instancing (reference data model VulkanRaytracing.cpp:382-411,
raytrace.rchit:45-46).  CPU: the oracle's instanced closest hit against the same geometry flattened into one
mesh.  GPU: pt_set_instances + top-level BVH4 + two-level traversal against the oracle, bit for bit."""
import math

import numpy as np
import pytest

from conftest import bits


def xform(rot_y_deg=0.0, rot_x_deg=0.0, scale=(1.0, 1.0, 1.0), translate=(0.0, 0.0, 0.0)):
    """Object-to-world 3x4 row-major: T * Ry * Rx * S."""
    a, b = math.radians(rot_y_deg), math.radians(rot_x_deg)
    ry = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
    rx = np.array([[1, 0, 0], [0, math.cos(b), -math.sin(b)], [0, math.sin(b), math.cos(b)]])
    m = ry @ rx @ np.diag(scale)
    return np.concatenate([m, np.array(translate, dtype=np.float64).reshape(3, 1)], axis=1).astype(np.float32).reshape(12)


def three_boxes():
    return np.stack([xform(),
                     xform(rot_y_deg=30.0, translate=(2.6, 0.0, -0.4)),
                     xform(rot_y_deg=-20.0, rot_x_deg=10.0, scale=(0.8, 1.2, -0.9), translate=(-2.7, 0.2, 0.3))])  # mirrored


def wide_camera(pa_or_O, W, H, eye=(0.0, 1.2, 7.5), center=(0.0, 1.0, 0.0)):
    look = getattr(pa_or_O, "look_at", None) or pa_or_O.lookat
    view = look(eye, center, (0.0, 1.0, 0.0))
    proj = pa_or_O.perspective(np.float32(math.radians(45.0)), np.float32(W / H), np.float32(0.1), np.float32(100.0))
    proj = np.array(proj, dtype=np.float32)
    proj[5] *= -1
    return view, proj


def random_rays(n, seed, lo=(-4, 0, -2), hi=(4, 2.4, 8)):
    rng = np.random.default_rng(seed)
    o = rng.uniform(lo, hi, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o, d], axis=1).astype(np.float32)


def test_oracle_instances_equal_flattened_geometry(O, pa, cornell):
    """Semantics check without a GPU: tracing the instanced scene == tracing one mesh made of the transformed
    copies (same primitive, same distance up to transform rounding), and the world-to-object matrices really
    are inverses."""
    xf = three_boxes()
    osc = O.OracleScene(cornell.vertices, cornell.indices, cornell.materials, cornell.mat_indices, instances=xf)
    for i in range(len(xf)):
        a = np.vstack([xf[i].reshape(3, 4), [0, 0, 0, 1]]).astype(np.float64)
        b = np.vstack([osc.instances[i, 12:].reshape(3, 4), [0, 0, 0, 1]]).astype(np.float64)
        assert np.allclose(a @ b, np.eye(4), atol=1e-5)
    nv, nt = cornell.vertices.shape[0], cornell.n_tris
    verts, idx, mid = [], [], []
    for i in range(len(xf)):
        m = xf[i].reshape(3, 4).astype(np.float64)
        v = cornell.vertices.copy()
        v[:, 0:3] = (cornell.vertices[:, 0:3].astype(np.float64) @ m[:, :3].T + m[:, 3]).astype(np.float32)
        verts.append(v)
        idx.append(cornell.indices + np.uint32(i * nv))
        mid.append(cornell.mat_indices)
    flat = O.OracleScene(np.concatenate(verts), np.concatenate(idx), cornell.materials, np.concatenate(mid))
    rays = random_rays(60000, 5)
    hits, inst = osc.trace_inst(rays)
    want = flat.trace(rays)
    got_id = np.where(hits["prim"] == 0xFFFFFFFF, 0xFFFFFFFF, inst.astype(np.uint64) * nt + hits["prim"]).astype(np.uint64)
    agree = got_id == want["prim"].astype(np.uint64)
    assert agree.mean() > 0.999  # grazing rays may flip between neighbours: different rounding, same surface
    both = agree & (hits["prim"] != 0xFFFFFFFF)
    assert both.sum() > 8000 and set(np.unique(inst[both])) == {0, 1, 2}
    assert np.allclose(hits["t"][both], want["t"][both], rtol=2e-4, atol=2e-4)
    # brute force over instances == the per-instance BVH2 path, bit for bit
    hb, ib = osc.trace_inst(rays[:4000], brute=True)
    assert np.array_equal(hb.view(np.uint32), hits[:4000].view(np.uint32)) and np.array_equal(ib, inst[:4000])


def test_oracle_identity_instance_renders_like_no_instance(O, pa, cornell, ocornell):
    """One identity instance: same image as the single-level path except where -0 + 0 differs (nowhere here)."""
    osc = O.OracleScene(cornell.vertices, cornell.indices, cornell.materials, cornell.mat_indices, instances=[xform()])
    W, H = 40, 30
    view, proj = O.reference_camera(W, H)
    a, ra = osc.render(view, proj, W, H, frames=2, spp_per_frame=3, max_depth=6, seed=1)
    b, rb = ocornell.render(view, proj, W, H, frames=2, spp_per_frame=3, max_depth=6, seed=1)
    assert ra == rb and np.array_equal(bits(a), bits(b))


@pytest.mark.gpu
@pytest.mark.parametrize("builder", [0, 2])  # PT_ACCEL_BVH4, PT_ACCEL_BVH4_PLOC
def test_three_instanced_cornell_boxes_bit_exact(rt, pa, O, cornell, builder):
    xf = three_boxes()
    osc = O.OracleScene(cornell.vertices, cornell.indices, cornell.materials, cornell.mat_indices, instances=xf)
    W, H = 192, 80
    view, proj = wide_camera(pa, W, H)
    rt.setInstances(xf)
    rt.buildBlas(cornell, builder)
    rt.resize(W, H)
    rt.setCamera(view, proj)
    rt.setParams(spp_per_frame=4, max_depth=8, seed=3)
    rt.raytrace(2)
    want, rays = osc.render(view, proj, W, H, frames=2, spp_per_frame=4, max_depth=8, seed=3)
    assert rt.stats().rays_total == rays
    img = rt.readImage()
    assert np.array_equal(bits(img), bits(want)), int((bits(img) != bits(want)).sum())
    third = W // 3
    assert all(want[:, k * third:(k + 1) * third, :3].max() > 1.0 for k in range(3))  # every copy shows its light
    # back to the reference's single instance
    rt.setInstances(None)
    with pytest.raises(pa.PathTracerError):
        rt.raytrace(1)  # the acceleration structure has to be rebuilt
    rt.buildBlas(cornell, builder)
    view, proj = pa.reference_camera(64, 48)
    rt.resize(64, 48)
    rt.setCamera(view, proj)
    rt.raytrace(1)
    single = O.OracleScene(cornell.vertices, cornell.indices, cornell.materials, cornell.mat_indices)
    want, _ = single.render(view, proj, 64, 48, frames=1, spp_per_frame=4, max_depth=8, seed=3)
    assert np.array_equal(bits(rt.readImage()), bits(want))


@pytest.mark.gpu
def test_grid_of_rotated_meshes_bit_exact(rt, pa, O):
    """Config-4 shape in small: one mesh (height field + light, 258 triangles) instanced on a 6 x 6 grid with
    per-instance rotations and scales -- overlapping world boxes, a top-level tree of several levels."""
    sc = pa.scenes.grid_mesh(8, 16, size=2.0, amplitude=0.2)
    rng = np.random.default_rng(7)
    xf = np.stack([xform(rot_y_deg=float(rng.uniform(0, 360)), rot_x_deg=float(rng.uniform(-25, 25)),
                         scale=tuple(rng.uniform(0.6, 1.3, size=3)), translate=(2.4 * (i - 2.5), 0.3 * math.sin(i + j), 2.4 * (j - 2.5)))
                   for i in range(6) for j in range(6)])
    osc = O.OracleScene(sc.vertices, sc.indices, sc.materials, sc.mat_indices, instances=xf)
    W, H = 160, 100
    view, proj = wide_camera(pa, W, H, eye=(0.0, 7.0, 12.0), center=(0.0, 0.0, 0.0))
    rt.setInstances(xf)
    rt.buildBlas(sc)
    st = rt.stats()
    assert st.bvh_nodes > 36 // 4  # bottom-level + top-level nodes
    rt.resize(W, H)
    rt.setCamera(view, proj)
    rt.setParams(spp_per_frame=3, max_depth=5, seed=8)
    rt.raytrace(2)
    want, rays = osc.render(view, proj, W, H, frames=2, spp_per_frame=3, max_depth=5, seed=8)
    assert rt.stats().rays_total == rays
    assert np.array_equal(bits(rt.readImage()), bits(want))
    assert (want[..., :3].sum(axis=2) > 0).mean() > 0.2
    # The top-level tree is built with BOTH builders and the one with the smaller surface-area cost is kept (round 3);
    # PT_TLAS_BUILDER forces one.  Whatever tree is traversed, the image and the ray count are the oracle's.
    import os
    nodes = {}
    try:
        for forced in ("lbvh", "ploc"):
            os.environ["PT_TLAS_BUILDER"] = forced
            rt.buildBlas(sc)
            nodes[forced] = rt.stats().bvh_nodes
            rt.resize(W, H)
            rt.setCamera(view, proj)
            rt.resetStats()
            rt.raytrace(2)
            assert rt.stats().rays_total == rays, forced
            assert np.array_equal(bits(rt.readImage()), bits(want)), forced
    finally:
        os.environ.pop("PT_TLAS_BUILDER", None)
    assert st.bvh_nodes in nodes.values()  # the unforced build kept one of the two


@pytest.mark.gpu
def test_instance_errors(rt, pa, cornell):
    from pathtracer_amd.api import PT_PIPELINE_FUSED
    with pytest.raises(pa.PathTracerError):
        rt.setInstances(np.zeros((1, 12), dtype=np.float32))  # singular
    bad = three_boxes()
    bad[1, 3] = np.inf
    with pytest.raises(pa.PathTracerError):
        rt.setInstances(bad)
    rt.setInstances(three_boxes())
    rt.buildBlas(cornell)
    rt.resize(16, 16)
    rt.setCamera(*pa.reference_camera(16, 16))
    rt.setPipeline(PT_PIPELINE_FUSED)
    with pytest.raises(pa.PathTracerError):
        rt.raytrace(1)  # instanced scenes render on the persistent pipeline
    with pytest.raises(pa.PathTracerError):
        rt.debugPrimaryHits()
