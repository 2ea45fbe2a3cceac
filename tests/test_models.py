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


# ------------------------------------------------------------------------- one diffuse map per Model ----
def _textured_models(pa):
    """Two Cornell boxes with texture coordinates (different ones) and an untextured height field between them."""
    a = pa.scenes.textured_cornell()
    b = pa.scenes.grid_mesh(6, 6, size=2.0, amplitude=0.25)
    c0 = pa.scenes.textured_cornell()
    v = c0.vertices.copy()
    v[:, 9], v[:, 10] = v[:, 10] * np.float32(0.7), v[:, 9] * np.float32(2.1)
    c = pa.Scene(v, c0.indices, c0.materials, c0.mat_indices)
    xf = np.stack([xform(translate=(-2.8, 0.0, 0.0)), xform(rot_y_deg=20.0, translate=(0.0, 0.3, 0.4)),
                   xform(rot_y_deg=-30.0, scale=(0.9, 1.0, 0.9), translate=(2.8, 0.0, 0.2))])
    return [a, b, c], xf, np.array([0, 1, 2], dtype=np.uint32)


def test_oracle_model_textures(O, pa):
    """Without a GPU: the same map in every slot == the scene-wide map; a Model without a map keeps its plain colours."""
    models, xf, custom = _textured_models(pa)
    tex = pa.scenes.random_texture(19, 11, seed=3)
    W, H = 56, 40
    view, proj = wide_camera(pa, W, H, eye=(0.0, 1.6, 8.0), center=(0.0, 1.0, 0.0))
    kw = dict(frames=1, spp_per_frame=2, max_depth=5, seed=8)
    whole, _ = O.OracleScene.from_models(models, xf, custom, texture=tex).render(view, proj, W, H, **kw)
    slots, _ = O.OracleScene.from_models(models, xf, custom, model_textures=[tex, tex, tex]).render(view, proj, W, H, **kw)
    assert np.array_equal(bits(whole), bits(slots))
    plain, _ = O.OracleScene.from_models(models, xf, custom).render(view, proj, W, H, **kw)
    none, _ = O.OracleScene.from_models(models, xf, custom, model_textures=[None, None, None]).render(view, proj, W, H, **kw)
    assert np.array_equal(bits(plain), bits(none))
    mixed, _ = O.OracleScene.from_models(models, xf, custom, model_textures=[tex, None, None]).render(view, proj, W, H, **kw)
    assert not np.array_equal(bits(mixed), bits(plain)) and not np.array_equal(bits(mixed), bits(whole))


@pytest.mark.gpu
def test_model_textures_bit_exact(rt, pa, O):
    """pt_set_model_texture: Model 0 and Model 2 with maps of different sizes, Model 1 without one -- against the oracle; then
    the maps changed and removed one by one, the scene-wide map replacing them and being replaced.  (Scenes of several
    Models are two-level scenes: the persistent pipeline.)"""
    models, xf, custom = _textured_models(pa)
    t0, t2 = pa.scenes.random_texture(37, 23, seed=7), pa.scenes.random_texture(16, 64, seed=11)
    W, H = 144, 88
    view, proj = wide_camera(pa, W, H, eye=(0.0, 1.6, 8.0), center=(0.0, 1.0, 0.0))
    kw = dict(frames=2, spp_per_frame=3, max_depth=6, seed=5)
    rt.uploadModels(models)
    rt.setInstances(xf, custom)
    rt.buildBlas(None)
    rt.resize(W, H)
    rt.setCamera(view, proj)
    rt.setParams(spp_per_frame=3, max_depth=6, seed=5)

    def check(model_textures=None, texture=None):
        rt.resetFrame()
        rt.resetStats()
        rt.raytrace(2)
        extra = {"model_textures": model_textures} if model_textures is not None else {"texture": texture}
        want, rays = O.OracleScene.from_models(models, xf, custom, **extra).render(view, proj, W, H, **kw)
        assert rt.stats().rays_total == rays
        img = rt.readImage()
        assert np.array_equal(bits(img), bits(want)), int((bits(img) != bits(want)).sum())
        return img

    rt.setModelTexture(0, t0)
    rt.setModelTexture(2, t2)
    both = check([t0, None, t2])
    rt.setModelTexture(1, t2)
    assert not np.array_equal(bits(check([t0, t2, t2])), bits(both))
    rt.setModelTexture(0, None)
    rt.setModelTexture(1, None)
    check([None, None, t2])
    rt.setTexture(t0)  # the scene-wide map replaces the per-Model ones ...
    check(texture=t0)
    rt.setModelTexture(2, t2)  # ... and is replaced by them
    check([None, None, t2])
    rt.setModelTexture(2, None)  # the last map gone: the plain colours
    check(texture=None)
    with pytest.raises(pa.PathTracerError, match="model 3 of 3"):
        rt.setModelTexture(3, t0)
    # a new scene drops the per-Model maps
    rt.setModelTexture(0, t0)
    rt.uploadModels(models)
    rt.setInstances(xf, custom)
    rt.buildBlas(None)
    check(texture=None)


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline", ["persistent", "fused", "split"])
def test_model_texture_of_a_single_model_scene(rt, pa, O, pipeline):
    """pt_upload_scene is one Model: its map through pt_set_model_texture(0, ...) == the scene-wide map, every pipeline."""
    from pathtracer_amd.api import PT_PIPELINE_FUSED, PT_PIPELINE_PERSISTENT, PT_PIPELINE_SPLIT
    sc, tex = pa.scenes.textured_cornell(), pa.scenes.random_texture(29, 31, seed=2)
    W, H = 80, 64
    view, proj = pa.reference_camera(W, H)
    rt.setPipeline({"persistent": PT_PIPELINE_PERSISTENT, "fused": PT_PIPELINE_FUSED, "split": PT_PIPELINE_SPLIT}[pipeline])
    rt.buildBlas(sc)
    rt.setModelTexture(0, tex)
    rt.resize(W, H)
    rt.setCamera(view, proj)
    rt.setParams(spp_per_frame=3, max_depth=6, seed=9)
    rt.raytrace(2)
    osc = O.OracleScene(sc.vertices, sc.indices, sc.materials, sc.mat_indices, texture=tex)
    want, rays = osc.render(view, proj, W, H, frames=2, spp_per_frame=3, max_depth=6, seed=9)
    assert rt.stats().rays_total == rays and np.array_equal(bits(rt.readImage()), bits(want))
