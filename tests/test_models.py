"""Several Models in one scene (raytrace.rchit:27 `model.m[gl_InstanceCustomIndexEXT]`; struct Model main.cpp:46-52; the
reference fills in one Model, builds one BLAS and one instance -- VulkanRaytracing.cpp:158, 385 -- but instance i is wired
to m_blases[i] and custom index i).  pt_upload_models + PtInstance.custom_index: every Model gets its own bottom-level
tree, an instance shows the Model its custom_index names.  CPU: the oracle's definition against flattened geometry.
GPU: bit-exact against the oracle."""
import numpy as np
import pytest

from conftest import bits
from test_instancing import random_rays, wide_camera, xform


def _models(pa):
    """Cornell box (32 triangles, own light), a lit height field (130 triangles), a small quad mesh with a light."""
    a = pa.scenes.cornell_box()
    b = pa.scenes.grid_mesh(8, 8, size=2.0, amplitude=0.25)
    c = pa.scenes.with_light_quad(pa.scenes.grid_mesh(3, 2, size=1.5, amplitude=0.1),
                                  [[-0.4, 1.2, -0.4], [0.4, 1.2, -0.4], [0.4, 1.2, 0.4], [-0.4, 1.2, 0.4]], (0, -1, 0))
    return [a, b, c]


def _placement():
    xf = np.stack([xform(),
                   xform(rot_y_deg=25.0, translate=(2.7, 0.0, -0.3)),
                   xform(rot_y_deg=-35.0, rot_x_deg=8.0, scale=(0.9, 1.1, 0.8), translate=(-2.6, 0.1, 0.4)),
                   xform(rot_y_deg=80.0, scale=(0.7, 0.7, 0.7), translate=(0.0, 2.3, 0.0)),
                   xform(rot_y_deg=10.0, scale=(1.0, 1.0, -1.0), translate=(2.6, 2.2, 0.2))])  # mirrored
    custom = np.array([0, 1, 2, 1, 0], dtype=np.uint32)
    return xf, custom


def test_oracle_models_equal_flattened_geometry(O, pa):
    """Without a GPU: instances of different Models == one mesh made of the transformed copies of the right Model."""
    models = _models(pa)
    xf, custom = _placement()
    osc = O.OracleScene.from_models(models, xf, custom)
    first = np.cumsum([0] + [m.n_tris for m in models])
    verts, idx, mid, owner = [], [], [], []
    v0 = 0
    mats = np.concatenate([m.materials for m in models])
    m_first = np.cumsum([0] + [m.materials.shape[0] for m in models])
    for i, (t, ci) in enumerate(zip(xf, custom)):
        s = models[ci]
        m = t.reshape(3, 4).astype(np.float64)
        v = s.vertices.copy()
        v[:, 0:3] = (s.vertices[:, 0:3].astype(np.float64) @ m[:, :3].T + m[:, 3]).astype(np.float32)
        verts.append(v)
        idx.append(s.indices + np.uint32(v0))
        mid.append(s.mat_indices + np.uint32(m_first[ci]))
        owner.append(np.stack([np.full(s.n_tris, i), first[ci] + np.arange(s.n_tris)], axis=1))
        v0 += v.shape[0]
    flat = O.OracleScene(np.concatenate(verts), np.concatenate(idx), mats, np.concatenate(mid))
    owner = np.concatenate(owner)
    rays = random_rays(50000, 9, lo=(-4, 0, -2), hi=(4, 4, 8))
    hits, inst = osc.trace_inst(rays)
    want = flat.trace(rays)
    hit = want["prim"] != 0xFFFFFFFF
    agree = (hits["prim"] == 0xFFFFFFFF) == ~hit
    both = hit & (hits["prim"] != 0xFFFFFFFF)
    same = (owner[want["prim"][both], 0] == inst[both]) & (owner[want["prim"][both], 1] == hits["prim"][both])
    assert agree.mean() > 0.999 and same.mean() > 0.999 and both.sum() > 5000
    assert set(np.unique(inst[both])) == {0, 1, 2, 3, 4}
    assert np.allclose(hits["t"][both][same], want["t"][both][same], rtol=2e-4, atol=2e-4)
    hb, ib = osc.trace_inst(rays[:3000], brute=True)  # the BVH path only culls
    assert np.array_equal(hb.view(np.uint32), hits[:3000].view(np.uint32)) and np.array_equal(ib, inst[:3000])


@pytest.mark.gpu
@pytest.mark.parametrize("builder", [0, 2])  # PT_ACCEL_BVH4, PT_ACCEL_BVH4_PLOC
def test_three_models_five_instances_bit_exact(rt, pa, O, builder):
    models = _models(pa)
    xf, custom = _placement()
    osc = O.OracleScene.from_models(models, xf, custom)
    W, H = 176, 112
    view, proj = wide_camera(pa, W, H, eye=(0.0, 2.2, 8.5), center=(0.0, 1.8, 0.0))
    rt.uploadModels(models)
    rt.setInstances(xf, custom)
    rt.buildBlas(None, builder)
    st = rt.stats()
    assert st.bvh_tris == sum(m.n_tris for m in models)
    rt.resize(W, H)
    rt.setCamera(view, proj)
    rt.setParams(spp_per_frame=4, max_depth=7, seed=21)
    rt.raytrace(2)
    want, rays = osc.render(view, proj, W, H, frames=2, spp_per_frame=4, max_depth=7, seed=21)
    assert rt.stats().rays_total == rays
    img = rt.readImage()
    assert np.array_equal(bits(img), bits(want)), int((bits(img) != bits(want)).sum())
    assert (want[..., :3].sum(axis=2) > 0).mean() > 0.02


@pytest.mark.gpu
def test_models_without_instances_get_one_identity_instance_each(rt, pa, O):
    """The reference's createInstancesBuffer: instance i = Model i, identity transform.  Two Models that interpenetrate."""
    a = pa.scenes.cornell_box()
    b = pa.scenes.grid_mesh(6, 6, size=1.6, amplitude=0.3, height=0.9)
    ident = np.stack([xform(), xform()])
    osc = O.OracleScene.from_models([a, b], ident, [0, 1])
    W, H = 96, 72
    view, proj = pa.reference_camera(W, H)
    rt.uploadModels([a, b])
    rt.buildBlas(None)
    rt.resize(W, H)
    rt.setCamera(view, proj)
    rt.setParams(spp_per_frame=3, max_depth=6, seed=4)
    rt.raytrace(2)
    want, rays = osc.render(view, proj, W, H, frames=2, spp_per_frame=3, max_depth=6, seed=4)
    assert rt.stats().rays_total == rays and np.array_equal(bits(rt.readImage()), bits(want))
    # a single Model afterwards: back to the single-level scene, no instance left over
    rt.uploadScene(a)
    rt.buildBlas(None)
    rt.resetFrame()
    rt.raytrace(2)
    single = O.OracleScene(a.vertices, a.indices, a.materials, a.mat_indices)
    want, rays = single.render(view, proj, W, H, frames=2, spp_per_frame=3, max_depth=6, seed=4)
    assert np.array_equal(bits(rt.readImage()), bits(want))


@pytest.mark.gpu
def test_model_errors(rt, pa):
    from pathtracer_amd.api import PT_ACCEL_BRUTE_FORCE
    a, b = pa.scenes.cornell_box(), pa.scenes.grid_mesh(2, 2)
    rt.uploadModels([a, b])
    rt.setInstances(np.stack([xform(), xform()]), [0, 2])
    with pytest.raises(pa.PathTracerError, match="custom_index 2"):
        rt.buildBlas(None)
    rt.setInstances(np.stack([xform(), xform()]), [0, 1])
    with pytest.raises(pa.PathTracerError):
        rt.buildBlas(None, PT_ACCEL_BRUTE_FORCE)
    empty = pa.Scene(np.zeros((0, 11), np.float32), np.zeros(0, np.uint32), np.array([[.5, .5, .5, 0, 0, 0]], np.float32), np.zeros(0, np.uint32))
    with pytest.raises(pa.PathTracerError):
        rt.uploadModels([a, empty])
    with pytest.raises(pa.PathTracerError):
        rt.uploadModels([])
    rt.uploadScene(a)
    rt.setInstances(np.stack([xform()]), [1])
    with pytest.raises(pa.PathTracerError, match="custom_index 1"):
        rt.buildBlas(None)
