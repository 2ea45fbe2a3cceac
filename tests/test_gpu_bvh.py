"""GPU: the device-built acceleration structure -- structural validity of the quantised BVH4 and
hit-for-hit agreement of its traversal with the brute-force definition on incoherent rays."""
import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu

LEAF = 0x80000000
EMPTY = 0xFFFFFFFE


def random_rays(n, lo, hi, seed):
    rng = np.random.default_rng(seed)
    o = rng.uniform(lo, hi, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[: n // 50, 0] = 0.0  # axis-parallel / zero components
    d[n // 50: n // 25, 1] = 0.0
    d[n // 25: 3 * n // 50, 2] = -0.0
    return np.concatenate([o, d], axis=1).astype(np.float32)


def validate_bvh(nodes, tris, scene):
    """Every primitive in exactly one leaf; every child box (dequantised as the traversal does)
    contains its triangles / its sub-tree's boxes; all nodes reachable once."""
    n_tris = scene.n_tris
    assert sorted(tris["prim"].tolist()) == list(range(n_tris))
    pos = scene.vertices[scene.indices.reshape(-1, 3), :3]
    assert np.array_equal(tris["v0"], pos[tris["prim"], 0]) and np.array_equal(tris["v2"], pos[tris["prim"], 2])
    if n_tris == 0:
        return 0
    seen_nodes = np.zeros(len(nodes), dtype=np.int32)
    seen_tris = np.zeros(len(tris), dtype=np.int32)
    depth_max = 0
    stack = [(0, None, None, 1)]
    while stack:
        ni, plo, phi, depth = stack.pop()
        depth_max = max(depth_max, depth)
        seen_nodes[ni] += 1
        nd = nodes[ni]
        scale = np.array([nd["scale_x"], nd["scale_yz"][0], nd["scale_yz"][1]], dtype=np.float32).astype(np.float64)
        org = nd["origin"].astype(np.float64)
        for c in range(4):
            ref = int(nd["child"][c])
            if ref == EMPTY:
                assert np.all(nd["qlo"][:, c] == 255) and np.all(nd["qhi"][:, c] == 0)
                continue
            lo = org + nd["qlo"][:, c].astype(np.float64) * scale
            hi = org + nd["qhi"][:, c].astype(np.float64) * scale
            if plo is not None:
                # nested inside the box the parent stores for this node, up to one step of this node's
                # own (finer) quantisation grid: both boxes round the same true box outwards
                assert np.all(lo >= plo - scale) and np.all(hi <= phi + scale)
            if ref & LEAF:
                first, cnt = ref & 0x0FFFFFFF, ((ref >> 28) & 7) + 1
                assert 1 <= cnt <= 4 and first + cnt <= len(tris)
                seen_tris[first:first + cnt] += 1
                p = np.stack([tris["v0"][first:first + cnt], tris["v1"][first:first + cnt],
                              tris["v2"][first:first + cnt]], axis=1).astype(np.float64)
                assert np.all(p >= lo) and np.all(p <= hi), "triangle sticks out of its quantised leaf box"
            else:
                assert 0 < ref < len(nodes)
                stack.append((ref, lo, hi, depth + 1))
    assert np.all(seen_nodes == 1) and np.all(seen_tris == 1)
    return depth_max


BUILDERS = {"lbvh": 0, "ploc": 2}  # PT_ACCEL_BVH4, PT_ACCEL_BVH4_PLOC


@pytest.mark.parametrize("builder", list(BUILDERS))
@pytest.mark.parametrize("scene_name", ["cornell", "grid", "random", "single", "two", "five"])
def test_bvh_structure(rt, pa, scene_name, builder):
    sc = {"cornell": pa.scenes.cornell_box, "grid": lambda: pa.scenes.grid_mesh(40, 33),
          "random": lambda: pa.scenes.random_triangles(20000, extent=50.0, edge=1.0),
          "single": lambda: pa.scenes.random_triangles(0), "two": lambda: pa.scenes.random_triangles(0),
          "five": lambda: pa.scenes.random_triangles(3)}[scene_name]()
    if scene_name == "single":  # one triangle only
        sc = pa.Scene(sc.vertices, sc.indices[:3], sc.materials, sc.mat_indices[:1])
    rt.buildBlas(sc, BUILDERS[builder])
    nodes, tris = rt.debugReadBvh()
    depth = validate_bvh(nodes, tris, sc)
    st = rt.stats()
    assert st.bvh_nodes == len(nodes) and st.bvh_tris == sc.n_tris and st.bvh_depth == depth


@pytest.mark.parametrize("builder", list(BUILDERS))
@pytest.mark.parametrize("scene_name,n_rays", [("cornell", 400000), ("grid", 300000), ("random", 300000)])
def test_bvh_traversal_equals_brute_force_definition(rt, pa, O, scene_name, n_rays, builder):
    """Incoherent rays (secondary-bounce-like), zero direction components included: GPU BVH4
    traversal == oracle (which is itself checked against brute force) == GPU brute force, bit for bit
    (t, u, v, primitive id)."""
    from pathtracer_amd.api import PT_ACCEL_BRUTE_FORCE
    if scene_name == "cornell":
        sc, lo, hi = pa.scenes.cornell_box(), (-1.0, 0.01, -1.0), (1.0, 1.98, 3.0)
    elif scene_name == "grid":
        sc, lo, hi = pa.scenes.grid_mesh(90, 70), (-1.2, -0.3, -1.2), (1.2, 1.6, 1.2)
    else:
        sc, lo, hi = pa.scenes.random_triangles(60000, extent=30.0, edge=1.5), (0.0, 0.0, 0.0), (30.0, 30.0, 30.0)
    rays = random_rays(n_rays, lo, hi, seed=42)
    osc = O.OracleScene(sc.vertices, sc.indices, sc.materials, sc.mat_indices)
    want = osc.trace(rays)
    rt.buildBlas(sc, BUILDERS[builder])
    got = rt.debugTrace(rays)
    assert np.array_equal(got["prim"], want["prim"]), int((got["prim"] != want["prim"]).sum())
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert 0.05 < (got["prim"] != 0xFFFFFFFF).mean()
    if sc.n_tris <= 20000 and builder == "lbvh":
        rt.buildBlas(sc, PT_ACCEL_BRUTE_FORCE)
        sub = rays[:50000]
        assert np.array_equal(rt.debugTrace(sub).view(np.uint32), want[:50000].view(np.uint32))
        assert np.array_equal(osc.trace(sub[:5000], brute=True).view(np.uint32), want[:5000].view(np.uint32))


def test_ties_resolve_to_lowest_primitive_id(rt, pa, O):
    """Two coincident triangles and a shared edge: equal t -> lowest primitive id wins (the build's
    documented tie-break; the Vulkan spec leaves it open)."""
    v = np.zeros((6, 11), dtype=np.float32)
    v[:, 0:3] = [[-1, -1, 0], [1, -1, 0], [0, 1, 0], [-1, -1, 0], [1, -1, 0], [0, 1, 0]]
    v[:, 5] = 1.0
    sc = pa.Scene(v, np.array([3, 4, 5, 0, 1, 2], dtype=np.uint32), np.array([[.5, .5, .5, 0, 0, 0]], dtype=np.float32),
                  np.zeros(2, dtype=np.uint32))
    rays = np.array([[0, 0, 2, 0, 0, -1], [0.2, -0.3, 1, 0, 0, -1], [0, 0, -3, 0, 0, 1]], dtype=np.float32)
    rt.buildBlas(sc)
    got = rt.debugTrace(rays)
    assert list(got["prim"]) == [0, 0, 0]
    osc = O.OracleScene(sc.vertices, sc.indices, sc.materials, sc.mat_indices)
    assert np.array_equal(got.view(np.uint32), osc.trace(rays, brute=True).view(np.uint32))


@pytest.mark.parametrize("builder", list(BUILDERS))
def test_million_triangle_build_traversal_equals_oracle(rt, pa, O, builder):
    """1.1 M triangles: the builder's multi-tile paths (radix histogram scan over 2 levels, PLOC flag sums over
    more than one scan tile, LDS-blocked refit with thousands of block roots, 12+ collapse levels) against the
    oracle's own BVH (itself tested against brute force), bit for bit on 200 k incoherent rays."""
    sc = pa.scenes.random_triangles(1_100_000, extent=100.0, edge=1.5)
    rays = random_rays(200000, (0.0, 0.0, 0.0), (100.0, 100.0, 100.0), seed=9)
    want = O.OracleScene(sc.vertices, sc.indices, sc.materials, sc.mat_indices).trace(rays)
    rt.buildBlas(sc, BUILDERS[builder])
    st = rt.stats()
    assert st.bvh_tris == sc.n_tris and 0.15 * sc.n_tris < st.bvh_nodes < 0.6 * sc.n_tris
    got = rt.debugTrace(rays)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), int((got["prim"] != want["prim"]).sum())
    assert (got["prim"] != 0xFFFFFFFF).mean() > 0.5
    _, tris = rt.debugReadBvh()
    assert np.array_equal(np.sort(tris["prim"]), np.arange(sc.n_tris, dtype=np.uint32))  # every primitive in exactly one slot


def test_compaction_shrinks_the_node_array_and_changes_nothing_else(rt, pa, O):
    """buildBlas(allowCompaction) of the reference (VulkanRaytracing.cpp:150-256): same tree, same hits, less memory."""
    from pathtracer_amd._lib import PT_ACCEL_NO_COMPACTION
    sc = pa.scenes.random_triangles(50000, extent=40.0, edge=1.0)
    rays = random_rays(50000, (0.0, 0.0, 0.0), (40.0, 40.0, 40.0), seed=3)
    rt.buildBlas(sc, BUILDERS["lbvh"] | PT_ACCEL_NO_COMPACTION)
    a, hits_a = rt.stats(), rt.debugTrace(rays)
    nodes_a, _ = rt.debugReadBvh()
    assert a.accel_bytes == a.accel_bytes_uncompacted == sc.n_tris * (64 + 3 * 48 + 8)
    rt.buildBlas(sc, BUILDERS["lbvh"])
    b, hits_b = rt.stats(), rt.debugTrace(rays)
    nodes_b, _ = rt.debugReadBvh()
    assert b.accel_bytes_uncompacted == a.accel_bytes_uncompacted
    assert b.accel_bytes == sc.n_tris * (3 * 48 + 8) + b.bvh_nodes * 64 < 0.85 * b.accel_bytes_uncompacted
    assert np.array_equal(hits_a.view(np.uint32), hits_b.view(np.uint32))
    assert len(nodes_a) == len(nodes_b) == b.bvh_nodes


def _quad_mesh(nx, nz, reverse_every=0, drop_every=0, seed=0):
    """nx * nz separate quad faces (tiles on a jittered grid, tilted), every face fan-triangulated as an OBJ loader does:
    (a, b, c), (a, c, d).  reverse_every: every k-th face lists its second triangle first; drop_every: every k-th
    face keeps one triangle only.  An emissive quad hangs above."""
    rng = np.random.default_rng(seed)
    F = nx * nz
    verts = np.zeros((4 * F + 4, 11), dtype=np.float32)
    idx, face = [], 0
    for j in range(nz):
        for i in range(nx):
            cx = -1.0 + 2.0 * (i + 0.5) / nx + rng.uniform(-0.1, 0.1) / nx
            cz = -1.0 + 2.0 * (j + 0.5) / nz + rng.uniform(-0.1, 0.1) / nz
            cy = rng.uniform(0.0, 0.3)
            hx, hz = 0.7 / nx, 0.7 / nz
            tilt = rng.uniform(-0.3, 0.3, size=4) * min(hx, hz)
            base = 4 * face
            corners = [(-hx, -hz), (hx, -hz), (hx, hz), (-hx, hz)]
            for k, (dx, dz) in enumerate(corners):
                verts[base + k, 0:3] = (cx + dx, cy + tilt[k], cz + dz)
                verts[base + k, 3:6] = (0, 1, 0)
            t1, t2 = [base, base + 1, base + 2], [base, base + 2, base + 3]
            face += 1
            if drop_every and face % drop_every == 0:
                idx += t1
            elif reverse_every and face % reverse_every == 0:
                idx += t2 + t1
            else:
                idx += t1 + t2
    nv = 4 * F
    verts[nv:, 0:3] = [[-0.5, 1.5, -0.5], [0.5, 1.5, -0.5], [0.5, 1.5, 0.5], [-0.5, 1.5, 0.5]]
    verts[nv:, 3:6] = (0, -1, 0)
    verts[:, 6:9] = 1.0
    n_ground = len(idx) // 3
    idx += [nv, nv + 1, nv + 2, nv, nv + 2, nv + 3]
    mats = np.array([[0.7, 0.6, 0.5, 0, 0, 0], [0, 0, 0, 15, 15, 15]], dtype=np.float32)
    mid = np.concatenate([np.zeros(n_ground, np.uint32), np.ones(2, np.uint32)])
    return verts, np.array(idx, np.uint32), mats, mid


@pytest.mark.parametrize("nx,nz,reverse_every,drop_every", [(3, 2, 0, 0), (12, 9, 2, 0), (40, 31, 3, 17), (9, 9, 0, 2)])
def test_quad_leaves_bit_exact(rt, pa, O, nx, nz, reverse_every, drop_every):
    """Scenes made of fan-triangulated quad faces are traversed through quad leaves by the persistent pipeline
    (PtQuad64): both triangle orders inside a leaf, lone triangles among the quads, and a scene with too few quads to
    switch the format on; image, ray count, node visits and triangle tests against the oracle / the fused pipeline."""
    from pathtracer_amd.api import PT_PROFILE_TRAVERSAL
    v, i, m, mi = _quad_mesh(nx, nz, reverse_every, drop_every, seed=nx)
    sc = pa.Scene(v, i, m, mi)
    osc = O.OracleScene(sc.vertices, sc.indices, sc.materials, sc.mat_indices)
    W, H = 128, 72
    view = pa.look_at((0.2, 1.3, 2.4), (0.0, 0.2, 0.0), (0, 1, 0))
    proj = pa.perspective(np.radians(50.0), W / H, 0.1, 100.0)
    proj[5] *= -1
    want, rays = osc.render(view, proj, W, H, frames=2, spp_per_frame=3, max_depth=6, seed=5)
    quads = []
    for builder in (pa.api.PT_ACCEL_BVH4, pa.api.PT_ACCEL_BVH4_PLOC):
        # with the entry node (default) and without it (then node visits and triangle tests equal the fused pipeline's)
        for entry_flag in (0, pa.api.PT_ACCEL_NO_ENTRY_NODE):
            rt.uploadScene(sc)
            rt.buildBlas(None, builder | entry_flag)
            st = rt.stats()
            if entry_flag:
                quads.append(st.quad_leaves)
            if drop_every == 2:
                assert st.quad_leaves == 0  # half of the faces are lone triangles: below the 90 % share
            rt.resize(W, H)
            rt.setCamera(view, proj)
            rt.setParams(spp_per_frame=3, max_depth=6, seed=5)
            trav = {}
            for name, mode in (("persistent", 2), ("fused", 0)):
                rt.setPipeline(mode)
                for prof in (0, PT_PROFILE_TRAVERSAL):
                    rt.setProfiling(prof)
                    rt.resetFrame()
                    rt.resetStats()
                    rt.raytrace(2)
                    s2 = rt.stats()
                    assert s2.rays_total == rays, (name, prof)
                    assert np.array_equal(bits(rt.readImage()), bits(want)), (name, prof, builder, entry_flag)
                trav[name] = (s2.trav_nodes, s2.trav_tris, s2.trav_hits)
            rt.setProfiling(0)
            if entry_flag:
                assert trav["persistent"] == trav["fused"], trav
            else:
                assert trav["persistent"][2] == trav["fused"][2], trav
    print("quad leaves (lbvh, ploc):", quads, "of", sc.n_tris // 2, "faces")
    if drop_every != 2:
        assert max(quads) > 0, quads  # at least one builder keeps the faces together


def test_quad_leaves_of_a_big_scene_pass_the_sampled_check(rt, pa):
    """Trees above 2^19 nodes are first checked on a sample (their last 65 536 nodes) before every slot is flagged: a scene of
    2.25 M separate quad tiles (750 K nodes) still gets its quad leaves -- as many as the same tiles in one piece less would
    predict -- and a soup of as many random triangles gets none."""
    n = 1500
    rng = np.random.default_rng(3)
    gx, gz = np.meshgrid(np.arange(n, dtype=np.float32), np.arange(n, dtype=np.float32), indexing="xy")
    cx = (gx.reshape(-1) + 0.5 + rng.uniform(-0.1, 0.1, n * n).astype(np.float32)) * np.float32(2.0 / n) - np.float32(1.0)
    cz = (gz.reshape(-1) + 0.5 + rng.uniform(-0.1, 0.1, n * n).astype(np.float32)) * np.float32(2.0 / n) - np.float32(1.0)
    cy = rng.uniform(0.0, 0.3, n * n).astype(np.float32)
    h = np.float32(0.7 / n)
    F = n * n
    verts = np.zeros((4 * F, 11), dtype=np.float32)
    for k, (dx, dz) in enumerate([(-h, -h), (h, -h), (h, h), (-h, h)]):
        verts[k::4, 0] = cx + dx
        verts[k::4, 1] = cy + rng.uniform(-0.3, 0.3, F).astype(np.float32) * h
        verts[k::4, 2] = cz + dz
    verts[:, 4] = 1.0
    verts[:, 6:9] = 1.0
    base = (4 * np.arange(F, dtype=np.uint32))[:, None]
    idx = (base + np.array([0, 1, 2, 0, 2, 3], dtype=np.uint32)[None, :]).reshape(-1)
    mats = np.array([[0.7, 0.6, 0.5, 0, 0, 0]], dtype=np.float32)
    sc = pa.Scene(verts, idx, mats, np.zeros(2 * F, np.uint32))
    rt.buildBlas(sc, BUILDERS["ploc"])
    st = rt.stats()
    assert st.bvh_nodes > (1 << 19)
    assert st.quad_leaves >= 0.9 * F, (st.quad_leaves, F)
    soup = pa.scenes.random_triangles(2 * F, extent=300.0, edge=1.0)
    rt.buildBlas(soup, BUILDERS["lbvh"])
    st = rt.stats()
    assert st.bvh_nodes > (1 << 19) and st.quad_leaves == 0


@pytest.mark.parametrize("builder", sorted(BUILDERS))
def test_one_triangle_plane_equals_three(rt, pa, O, builder):
    """PT_ACCEL_TRI_PLANES_1 (what one Model of 2 M triangles or more gets by itself): the leaf triangles are stored once, in
    x, y, z order, and the leaf test permutes them -- a third of the bytes, the same hits bit for bit (incoherent rays against
    brute force and against the three-plane build), the same image and ray count in all three pipelines."""
    from pathtracer_amd.api import PT_ACCEL_TRI_PLANES_1, PT_ACCEL_TRI_PLANES_3
    sc = pa.scenes.with_light_quad(pa.scenes.grid_mesh(37, 29, size=2.0, height=0.2, amplitude=0.3),
                                   [[-0.5, 1.5, -0.5], [0.5, 1.5, -0.5], [0.5, 1.5, 0.5], [-0.5, 1.5, 0.5]], (0, -1, 0))
    osc = O.OracleScene(sc.vertices, sc.indices, sc.materials, sc.mat_indices)
    rng = np.random.default_rng(8)
    rays = np.zeros((20000, 6), dtype=np.float32)
    rays[:, :3] = rng.uniform(-1.2, 1.2, (20000, 3)) + (0, 0.8, 0)
    d = rng.normal(size=(20000, 3))
    rays[:, 3:] = d / np.linalg.norm(d, axis=1, keepdims=True)
    want_hits = osc.trace(rays, brute=True)
    W, H = 96, 64
    view = pa.look_at((0.2, 1.3, 2.4), (0.0, 0.2, 0.0), (0, 1, 0))
    proj = pa.perspective(np.radians(50.0), W / H, 0.1, 100.0)
    proj[5] *= -1
    want, nrays = osc.render(view, proj, W, H, frames=2, spp_per_frame=3, max_depth=6, seed=5)
    sizes = {}
    for planes in (PT_ACCEL_TRI_PLANES_3, PT_ACCEL_TRI_PLANES_1):
        rt.buildBlas(sc, BUILDERS[builder] | planes)
        st = rt.stats()
        sizes[planes] = st.accel_bytes
        assert np.array_equal(rt.debugTrace(rays).view(np.uint32), want_hits.view(np.uint32))
        nodes, tris = rt.debugReadBvh()
        assert len(tris) == sc.n_tris and sorted(tris["prim"].tolist()) == list(range(sc.n_tris))
        rt.resize(W, H)
        rt.setCamera(view, proj)
        rt.setParams(spp_per_frame=3, max_depth=6, seed=5)
        for mode in (2, 0, 1):
            rt.setPipeline(mode)
            rt.resetFrame()
            rt.resetStats()
            rt.raytrace(2)
            assert rt.stats().rays_total == nrays
            assert np.array_equal(bits(rt.readImage()), bits(want)), (planes, mode)
        rt.setPipeline(2)
    assert sizes[PT_ACCEL_TRI_PLANES_3] - sizes[PT_ACCEL_TRI_PLANES_1] == 2 * 48 * sc.n_tris
    with pytest.raises(pa.PathTracerError):
        rt.buildBlas(sc, BUILDERS[builder] | PT_ACCEL_TRI_PLANES_1 | PT_ACCEL_TRI_PLANES_3)
