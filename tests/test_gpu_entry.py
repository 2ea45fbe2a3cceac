"""GPU: the entry node of the persistent pipeline (pt_wave_traverse.h) -- the top of the tree flattened into one 16-wide
node that is tested while a ray is staged -- in its three forms: none (PT_ACCEL_NO_ENTRY_NODE), tree (the node has inner
children; PT_NO_FLAT forces it), flat (every child is a leaf: no traversal loop for the first leaf, no stack).  A traversal
that only culls conservatively returns the brute-force answer, so every form must give the oracle's image and ray count
bit for bit; the statistics say which form ran."""
import os

import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu


def _render(rt, pa, scene, view, proj, W, H, flags, env=None, frames=2, spp=3, depth=6, seed=11, profile=True):
    from pathtracer_amd.api import PT_PROFILE_TRAVERSAL
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k)
        os.environ[k] = v
    try:
        rt.uploadScene(scene)
        rt.buildBlas(None, flags)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    rt.resize(W, H)
    rt.setCamera(view, proj)
    rt.setParams(spp_per_frame=spp, max_depth=depth, seed=seed)
    out = {}
    for prof in ((0, PT_PROFILE_TRAVERSAL) if profile else (0,)):
        rt.setProfiling(prof)
        rt.resetFrame()
        rt.resetStats()
        rt.raytrace(frames)
        out[prof] = (rt.readImage(), rt.stats())
    rt.setProfiling(0)
    return out


@pytest.mark.parametrize("nx,nz,leaf", [(1, 1, 0), (2, 1, 1), (2, 2, 0), (4, 2, 0), (4, 4, 0), (3, 3, 1), (5, 4, 0), (12, 9, 0)])
def test_entry_node_forms_equal_the_oracle(rt, pa, O, nx, nz, leaf):
    from pathtracer_amd.api import PT_ACCEL_BVH4, PT_ACCEL_BVH4_PLOC, PT_ACCEL_NO_ENTRY_NODE, PT_PROFILE_TRAVERSAL
    sc = pa.scenes.grid_mesh(nx, nz, size=2.0, amplitude=0.3)
    osc = O.OracleScene(sc.vertices, sc.indices, sc.materials, sc.mat_indices)
    W, H = 80, 56
    view = pa.look_at((0.1, 1.2, 2.5), (0.0, 0.1, 0.0), (0, 1, 0))
    proj = pa.perspective(np.radians(45.0), W / H, 0.1, 100.0)
    proj[5] *= -1
    want, rays = osc.render(view, proj, W, H, frames=2, spp_per_frame=3, max_depth=6, seed=11)
    for builder in (PT_ACCEL_BVH4, PT_ACCEL_BVH4_PLOC):
        base = builder | (leaf << 8)
        forms = {"default": (base, None), "tree": (base, {"PT_NO_FLAT": "1"}), "none": (base | PT_ACCEL_NO_ENTRY_NODE, None)}
        visits = {}
        for name, (flags, env) in forms.items():
            res = _render(rt, pa, sc, view, proj, W, H, flags, env)
            for prof, (img, st) in res.items():
                assert st.rays_total == rays, (name, prof)
                assert np.array_equal(bits(img), bits(want)), (name, prof, nx, nz)
            st = res[PT_PROFILE_TRAVERSAL][1]
            visits[name] = (st.trav_nodes, st.trav_tris, st.trav_hits)
        assert visits["default"][2] == visits["none"][2] == visits["tree"][2]  # the same hits
        n_leaves = rt.stats().bvh_tris  # (upper bound on leaves)
        if nx * nz * 2 <= 16:
            # at most 16 leaves whatever the builder does: the entry node is the whole scene and a ray visits exactly that node
            assert visits["default"][0] == rays, (visits, rays)
        assert visits["tree"][0] <= visits["none"][0]  # the wide first step never adds node visits
        assert n_leaves == sc.n_tris


def test_entry_node_far_from_the_origin(rt, pa, O):
    """The entry node stores its planes relative to its own origin: a scene (and camera) 10^4 units away from the world
    origin renders the oracle's image in every form."""
    from pathtracer_amd.api import PT_ACCEL_BVH4_PLOC, PT_ACCEL_NO_ENTRY_NODE
    sc0 = pa.scenes.cornell_box()
    off = np.array([12288.0, -4096.0, 8192.0], np.float32)
    v = sc0.vertices.copy()
    v[:, 0:3] += off
    sc = pa.Scene(v, sc0.indices, sc0.materials, sc0.mat_indices)
    osc = O.OracleScene(sc.vertices, sc.indices, sc.materials, sc.mat_indices)
    W, H = 64, 48
    view = pa.look_at(tuple(off + np.array([0.0, 1.0, 3.4], np.float32)), tuple(off + np.array([0.0, 1.0, 0.0], np.float32)), (0, 1, 0))
    proj = pa.perspective(np.radians(45.0), W / H, 0.1, 100.0)
    proj[5] *= -1
    want, rays = osc.render(view, proj, W, H, frames=2, spp_per_frame=3, max_depth=6, seed=11)
    assert want[..., :3].max() > 0
    for flags, env in ((PT_ACCEL_BVH4_PLOC, None), (PT_ACCEL_BVH4_PLOC, {"PT_NO_FLAT": "1"}), (PT_ACCEL_BVH4_PLOC | PT_ACCEL_NO_ENTRY_NODE, None)):
        res = _render(rt, pa, sc, view, proj, W, H, flags, env, profile=False)
        img, st = res[0]
        assert st.rays_total == rays and np.array_equal(bits(img), bits(want)), (flags, env)


def test_entry_node_spares_the_wall_a_ray_starts_on(rt, pa, cornell, ocornell):
    """Cornell box: the entry node carries the exact boxes of its leaves, so the slab of the wall a bounce ray starts on is
    left before tmin and is not tested; on the 255-step grid of the ordinary nodes it is.  Same image, fewer triangle tests."""
    from pathtracer_amd.api import PT_ACCEL_BVH4_PLOC, PT_ACCEL_NO_ENTRY_NODE, PT_PROFILE_TRAVERSAL
    W, H = 160, 120
    view, proj = pa.reference_camera(W, H)
    want, rays = ocornell.render(view, proj, W, H, frames=2, spp_per_frame=3, max_depth=6, seed=11)
    tris = {}
    for name, flags in (("entry", PT_ACCEL_BVH4_PLOC), ("none", PT_ACCEL_BVH4_PLOC | PT_ACCEL_NO_ENTRY_NODE)):
        res = _render(rt, pa, cornell, view, proj, W, H, flags)
        for prof, (img, st) in res.items():
            assert st.rays_total == rays and np.array_equal(bits(img), bits(want)), (name, prof)
        st = res[PT_PROFILE_TRAVERSAL][1]
        tris[name] = st.trav_tris / st.rays_total
        if name == "entry":
            assert st.trav_nodes == st.rays_total  # flat: one (wide) node visit per ray, no 4-wide node steps
    assert tris["entry"] < 0.85 * tris["none"], tris


def _random_scene(pa, seed):
    """1 .. 48 triangles of mixed kinds: axis-parallel quads (zero-thickness boxes), slivers, degenerate triangles (two equal
    vertices), duplicates (ties), tiny and room-sized ones; two materials, one emissive."""
    rng = np.random.default_rng(seed)
    tris = []
    n = int(rng.integers(1, 25))
    for _ in range(n):
        kind = rng.integers(0, 6)
        c = rng.uniform(-1.0, 1.0, 3)
        if kind == 0:  # axis-parallel quad as a fan
            ax = int(rng.integers(0, 3)); u, v = (ax + 1) % 3, (ax + 2) % 3
            e = rng.uniform(0.1, 1.5, 2)
            p = [c.copy() for _ in range(4)]
            p[1][u] += e[0]; p[2][u] += e[0]; p[2][v] += e[1]; p[3][v] += e[1]
            tris += [[p[0], p[1], p[2]], [p[0], p[2], p[3]]]
        elif kind == 1:  # degenerate: two equal vertices
            a = c + rng.uniform(-0.3, 0.3, 3)
            tris.append([c, a, a.copy()])
        elif kind == 2 and tris:  # duplicate of an earlier triangle (equal t: lowest primitive id wins)
            tris.append([x.copy() for x in tris[int(rng.integers(0, len(tris)))]])
        elif kind == 3:  # tiny
            tris.append([c, c + rng.uniform(-1e-3, 1e-3, 3), c + rng.uniform(-1e-3, 1e-3, 3)])
        elif kind == 4:  # room-sized
            tris.append([c + rng.uniform(-3, 3, 3) for _ in range(3)])
        else:
            tris.append([c + rng.uniform(-0.5, 0.5, 3) for _ in range(3)])
    t = np.array(tris, dtype=np.float32)
    nt = t.shape[0]
    verts = np.zeros((3 * nt, 11), dtype=np.float32)
    verts[:, 0:3] = t.reshape(-1, 3)
    nrm = rng.normal(size=(3 * nt, 3)).astype(np.float32)
    verts[:, 3:6] = nrm / np.linalg.norm(nrm, axis=1, keepdims=True)
    verts[:, 6:9] = 1.0
    idx = np.arange(3 * nt, dtype=np.uint32)
    mats = np.array([[0.8, 0.7, 0.6, 0, 0, 0], [0, 0, 0, 9, 8, 7]], dtype=np.float32)
    mid = (rng.uniform(size=nt) < 0.25).astype(np.uint32)
    return pa.Scene(verts, idx, mats, mid)


@pytest.mark.parametrize("seed", range(14))
def test_random_small_scenes_every_form_and_pipeline(rt, pa, O, seed):
    """Random scenes around the 16-leaf boundary of the flat form, with degenerate and coincident geometry: entry node
    (flat or tree), PT_NO_FLAT, no entry node, the fused and split pipelines and GPU brute force all give the oracle's image."""
    from pathtracer_amd.api import (PT_ACCEL_BRUTE_FORCE, PT_ACCEL_BVH4, PT_ACCEL_BVH4_PLOC, PT_ACCEL_NO_ENTRY_NODE,
                                    PT_PIPELINE_FUSED, PT_PIPELINE_PERSISTENT, PT_PIPELINE_SPLIT)
    sc = _random_scene(pa, seed)
    osc = O.OracleScene(sc.vertices, sc.indices, sc.materials, sc.mat_indices)
    W, H = 48, 36
    rng = np.random.default_rng(1000 + seed)
    eye = tuple(rng.uniform(-2.5, 2.5, 3))
    view = pa.look_at(eye, tuple(rng.uniform(-0.3, 0.3, 3)), (0, 1, 0))
    proj = pa.perspective(np.radians(60.0), W / H, 0.1, 100.0)
    proj[5] *= -1
    want, rays = osc.render(view, proj, W, H, frames=2, spp_per_frame=3, max_depth=6, seed=11)
    leaf = seed % 3  # default, 1, 2 triangles per leaf
    cases = [("flat/tree", PT_ACCEL_BVH4_PLOC | (leaf << 8), None, PT_PIPELINE_PERSISTENT),
             ("tree", PT_ACCEL_BVH4 | (leaf << 8), {"PT_NO_FLAT": "1"}, PT_PIPELINE_PERSISTENT),
             ("no quads", PT_ACCEL_BVH4_PLOC | (leaf << 8), {"PT_NO_QUADS": "1"}, PT_PIPELINE_PERSISTENT),
             ("none", PT_ACCEL_BVH4_PLOC | (leaf << 8) | PT_ACCEL_NO_ENTRY_NODE, None, PT_PIPELINE_PERSISTENT),
             ("fused", PT_ACCEL_BVH4_PLOC | (leaf << 8), None, PT_PIPELINE_FUSED),
             ("brute", PT_ACCEL_BRUTE_FORCE, None, PT_PIPELINE_SPLIT)]
    for name, flags, env, pipe in cases:
        rt.setPipeline(pipe)
        res = _render(rt, pa, sc, view, proj, W, H, flags, env, profile=False)
        img, st = res[0]
        assert st.rays_total == rays, (name, seed)
        assert np.array_equal(bits(img), bits(want)), (name, seed, sc.n_tris)
    rt.setPipeline(PT_PIPELINE_PERSISTENT)


def test_rays_along_shared_edges_do_not_leak(rt, pa, O):
    """Watertightness where it is hardest: with an odd image extent and an axis-aligned camera the centre column and the centre
    row of pixel-centre rays (global frame 0) have a direction component that is exactly zero and run exactly along the shared
    edges x = 0 and z = 0 of a flat, emissive grid mesh (and through its vertices): an edge function is exactly 0 there, the
    double-precision fall-back of the watertight test decides, and no ray may slip between two triangles.  Every pipeline and
    every entry-node form against the oracle, bit for bit; the centre lines inside the mesh are lit without a gap."""
    from pathtracer_amd.api import (PT_ACCEL_BVH4, PT_ACCEL_BVH4_PLOC, PT_ACCEL_NO_ENTRY_NODE, PT_PIPELINE_FUSED,
                                    PT_PIPELINE_PERSISTENT, PT_PIPELINE_SPLIT)
    g = pa.scenes.grid_mesh(8, 8, size=2.0, amplitude=0.0)
    n_mesh = 2 * 8 * 8
    mats = np.array([[0.5, 0.5, 0.5, 7.0, 7.0, 7.0]], dtype=np.float32)  # everything emits: a hit pixel is exactly 7
    sc = pa.Scene(g.vertices[:81], g.indices[:3 * n_mesh], mats, np.zeros(n_mesh, np.uint32))
    osc = O.OracleScene(sc.vertices, sc.indices, sc.materials, sc.mat_indices)
    W = H = 65
    view = pa.look_at((0.0, 3.0, 0.0), (0.0, 0.0, 0.0), (0.0, 0.0, -1.0))
    proj = pa.perspective(np.radians(45.0), 1.0, 0.1, 100.0)
    proj[5] *= -1
    want, rays = osc.render(view, proj, W, H, frames=1, spp_per_frame=2, max_depth=3, seed=0)
    ohits = osc.primary_hits(view, proj, W, H)
    prim = ohits["prim"].reshape(H, W)
    inside = np.abs(np.linspace(-1, 1, W) * 3.0 * np.tan(np.radians(22.5))) < 0.99  # pixel centres over the mesh
    assert inside.sum() > 40
    assert np.all(prim[H // 2, inside] != 0xFFFFFFFF) and np.all(prim[inside, W // 2] != 0xFFFFFFFF)
    assert np.all(want[H // 2, inside, :3] == 7.0) and np.all(want[inside, W // 2, :3] == 7.0)
    cases = [(PT_ACCEL_BVH4_PLOC, None, PT_PIPELINE_PERSISTENT), (PT_ACCEL_BVH4, {"PT_NO_QUADS": "1"}, PT_PIPELINE_PERSISTENT),
             (PT_ACCEL_BVH4_PLOC | (1 << 8), None, PT_PIPELINE_PERSISTENT),
             (PT_ACCEL_BVH4_PLOC | PT_ACCEL_NO_ENTRY_NODE, None, PT_PIPELINE_PERSISTENT),
             (PT_ACCEL_BVH4_PLOC, None, PT_PIPELINE_FUSED), (PT_ACCEL_BVH4, None, PT_PIPELINE_SPLIT)]
    for flags, env, pipe in cases:
        rt.setPipeline(pipe)
        res = _render(rt, pa, sc, view, proj, W, H, flags, env, frames=1, spp=2, depth=3, seed=0, profile=False)
        img, st = res[0]
        assert st.rays_total == rays, (flags, env, pipe)
        assert np.array_equal(bits(img), bits(want)), (flags, env, pipe)
        hits = rt.debugPrimaryHits()
        assert np.array_equal(hits["prim"], ohits["prim"]) and np.array_equal(bits(hits["t"]), bits(ohits["t"]))
    rt.setPipeline(PT_PIPELINE_PERSISTENT)


def test_flat_scene_skips_leaves_behind_the_hit(rt, pa, O):
    """Flat form with something to skip: 14 parallel quads one behind the other, of growing size (the builder orders the
    entry node's leaves by box area, so the small near ones are tested first), two of them coincident (equal t: the lower
    primitive id wins, so a leaf that begins exactly AT the hit must still be tested), one emissive; a leaf whose box begins
    behind the ray's hit is not tested.  Image and ray count equal the oracle's; fewer than two quad tests per ray although a
    ray through the middle crosses all 14 boxes."""
    from pathtracer_amd.api import PT_ACCEL_BVH4_PLOC, PT_PROFILE_TRAVERSAL
    quads = []
    for i in range(14):
        z = -0.25 * (i if i != 9 else 8)  # quads 8 and 9 coincide
        h = 0.2 + 0.12 * i
        quads.append([[-h, -h, z], [h, -h, z], [h, h, z], [-h, h, z]])
    t = []
    for q in quads:
        t += [[q[0], q[1], q[2]], [q[0], q[2], q[3]]]
    t = np.array(t, dtype=np.float32)
    nt = t.shape[0]
    verts = np.zeros((3 * nt, 11), dtype=np.float32)
    verts[:, 0:3] = t.reshape(-1, 3)
    verts[:, 5] = 1.0
    verts[:, 6:9] = 1.0
    mats = np.array([[0.7, 0.6, 0.5, 0, 0, 0], [0, 0, 0, 5, 6, 7]], dtype=np.float32)
    mid = np.zeros(nt, dtype=np.uint32)
    mid[2 * 13:] = 1  # the largest, farthest quad is the light
    sc = pa.Scene(verts, np.arange(3 * nt, dtype=np.uint32), mats, mid)
    osc = O.OracleScene(sc.vertices, sc.indices, sc.materials, sc.mat_indices)
    W, H = 96, 72
    view = pa.look_at((0.3, 0.2, 2.0), (0.0, 0.0, -1.0), (0, 1, 0))
    proj = pa.perspective(np.radians(50.0), W / H, 0.1, 100.0)
    proj[5] *= -1
    want, rays = osc.render(view, proj, W, H, frames=2, spp_per_frame=3, max_depth=6, seed=11)
    res = _render(rt, pa, sc, view, proj, W, H, PT_ACCEL_BVH4_PLOC)
    for prof, (img, st) in res.items():
        assert st.rays_total == rays and np.array_equal(bits(img), bits(want)), prof
    st = res[PT_PROFILE_TRAVERSAL][1]
    assert st.trav_nodes == st.rays_total and st.quad_leaves >= 12  # flat form; (the coincident pair may be paired crosswise)
    assert st.trav_tris / st.rays_total < 4.0, st.trav_tris / st.rays_total
