"""GPU: the HIP path through the C ABI against the CPU oracle on the same seeded inputs.
Bar (north_star): RNG sequences, Morton codes and primary-ray hit primitive IDs bit-exact; images:
this build goes further and requires the WHOLE image bit-exact at oracle-sized cases, because both
sides implement one arithmetic specification -- tolerance 0 ulp."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, bits

pytestmark = pytest.mark.gpu
KAT = np.load(os.path.join(GOLDEN, "oracle_kat.npz"))


def test_rng_bit_exact(rt, O):
    rng = np.random.default_rng(0)
    v0 = np.concatenate([KAT["tea_v0"], rng.integers(0, 2 ** 32, size=3000, dtype=np.uint64).astype(np.uint32),
                         np.arange(2048, dtype=np.uint32)])
    v1 = np.concatenate([KAT["tea_v1"], rng.integers(0, 2 ** 32, size=3000, dtype=np.uint64).astype(np.uint32),
                         np.arange(2048, dtype=np.uint32) % 17])
    tea, lcg, unit = rt.debugRng(v0, v1, 16)
    n = len(KAT["tea_v0"])
    assert np.array_equal(tea[:n], KAT["tea_out"]) and np.array_equal(lcg[:n], KAT["lcg_out"])
    assert np.array_equal(bits(unit[:n]), bits(KAT["unit_out"]))
    for i in range(0, v0.size, 7):
        assert int(tea[i]) == O.tea(int(v0[i]), int(v1[i]))
        want, _ = O.lcg_stream(int(tea[i]), 16)
        assert [int(x) for x in lcg[i]] == want
        u, _ = O.rnd_unit_vector(int(tea[i]))
        assert np.array_equal(bits(unit[i]), bits(u))


@pytest.mark.parametrize("accel", ["bvh4", "brute"])
def test_cornell_primary_hits_bit_exact_512(rt, pa, O, cornell, ocornell, accel):
    from pathtracer_amd.api import PT_ACCEL_BRUTE_FORCE, PT_ACCEL_BVH4
    W = H = 512
    view, proj = pa.reference_camera(W, H)
    rt.buildBlas(cornell, PT_ACCEL_BVH4 if accel == "bvh4" else PT_ACCEL_BRUTE_FORCE)
    rt.resize(W, H)
    rt.setCamera(view, proj)
    hits = rt.debugPrimaryHits()
    want = ocornell.primary_hits(view, proj, W, H, brute=True)
    assert np.array_equal(hits["prim"], want["prim"])
    assert np.array_equal(hits.view(np.uint32), want.view(np.uint32))  # t, u, v too
    assert set(np.unique(hits["prim"])) >= {0, 1, 2, 3, 30, 31, 0xFFFFFFFF}


def test_golden_small_image(rt, pa, cornell):
    """Committed fixture (tests/golden/oracle_kat.npz, made by make_oracle_golden.py)."""
    W, H = 64, 48
    view, proj = pa.reference_camera(W, H)
    rt.buildBlas(cornell)
    rt.resize(W, H)
    rt.setCamera(view, proj)
    rt.setParams(spp_per_frame=4, max_depth=8, seed=0)
    hits = rt.debugPrimaryHits()
    assert np.array_equal(hits["prim"], KAT["cornell_primary_64x48_prim"])
    rt.raytrace(3)
    img = rt.readImage()
    assert np.array_equal(bits(img), bits(KAT["cornell_image_64x48_f3_s4_d8"]))
    assert rt.stats().rays_total == int(KAT["cornell_image_rays"][0])
    rt.setParams(spp_per_frame=10, max_depth=20, seed=7)  # the reference's own constants
    rt.resetFrame()
    rt.raytrace(2)
    assert np.array_equal(bits(rt.readImage()), bits(KAT["cornell_image_64x48_f2_s10_d20_seed7"]))


def test_config0_cornell_512_16spp_depth8_image_bit_exact(rt, pa, cornell, ocornell):
    """BASELINE.json configs[0]: Cornell 512x512, 16 spp (2 frames x 8), max depth 8."""
    W = H = 512
    view, proj = pa.reference_camera(W, H)
    rt.buildBlas(cornell)
    rt.resize(W, H)
    rt.setCamera(view, proj)
    rt.setParams(spp_per_frame=8, max_depth=8, seed=0)
    rt.raytrace(2)
    img = rt.readImage()
    want, rays = ocornell.render(view, proj, W, H, frames=2, spp_per_frame=8, max_depth=8, seed=0)
    assert rt.stats().rays_total == rays
    assert np.array_equal(bits(img), bits(want)), float(np.abs(img - want).max())
    # sanity of the converging picture: this Cornell box has its green wall at x = -1 (image left) and
    # the red one at x = +1 (image right) -- cornell.obj faces `usemtl green` / `usemtl red`
    assert img[256, 40, 1] > img[256, 40, 0] and img[256, 470, 0] > img[256, 470, 1]


def test_reference_constants_800x600_frame(rt, pa, cornell, ocornell):
    """The reference's own dispatch: 800x600, 10 spp, depth 20 (VulkanUtils.h:11-12, rgen:22-23); one
    frame 0 and one jittered frame, non-square aspect."""
    W, H = 800, 600
    view, proj = pa.reference_camera(W, H)
    rt.buildBlas(cornell)
    rt.resize(W, H)
    rt.setCamera(view, proj)
    rt.setParams(spp_per_frame=10, max_depth=20, seed=0)
    rt.raytrace(2)
    want, rays = ocornell.render(view, proj, W, H, frames=2, spp_per_frame=10, max_depth=20, seed=0)
    assert rt.stats().rays_total == rays
    assert np.array_equal(bits(rt.readImage()), bits(want))


def test_morton_codes_and_sort_bit_exact(rt, pa, O, cornell, ocornell):
    rt.buildBlas(cornell)
    codes, order, cmin, cmax = rt.debugMorton()
    oc, oo, omin, omax = ocornell.morton()
    assert np.array_equal(codes, oc) and np.array_equal(order, oo)
    assert np.array_equal(bits(cmin), bits(omin)) and np.array_equal(bits(cmax), bits(omax))
    assert np.array_equal(codes, KAT["cornell_morton_codes"])
    # many primitives, duplicate codes included (several radix-sort tiles, 4 passes)
    for sc in (pa.scenes.grid_mesh(64, 50), pa.scenes.random_triangles(30000, extent=20.0, edge=0.5)):
        osc = O.OracleScene(sc.vertices, sc.indices, sc.materials, sc.mat_indices)
        rt.buildBlas(sc)
        codes, order, cmin, cmax = rt.debugMorton()
        oc, oo, omin, omax = osc.morton()
        assert np.array_equal(codes, oc)
        assert np.array_equal(order, oo)
        assert len(np.unique(codes)) < codes.size  # ties exist: stability is exercised


def test_accumulation_modes_and_frame_schedule(rt, pa, cornell, ocornell):
    """PT_ACCUM_SUM + pt_set_frame_schedule (multi-GPU contract) against the oracle's same modes."""
    from pathtracer_amd.api import PT_ACCUM_MEAN, PT_ACCUM_SUM
    W, H = 96, 64
    view, proj = pa.reference_camera(W, H)
    rt.buildBlas(cornell)
    rt.resize(W, H)
    rt.setCamera(view, proj)
    rt.setParams(spp_per_frame=4, max_depth=8, seed=5)
    rt.setAccumMode(PT_ACCUM_SUM)
    rt.setFrameSchedule(first=1, stride=2)
    rt.raytrace(3)  # global frames 1, 3, 5
    want, _ = ocornell.render(view, proj, W, H, frames=3, spp_per_frame=4, max_depth=8, seed=5, first_frame=1,
                              frame_stride=2, accum_sum=True)
    assert np.array_equal(bits(rt.readImage()), bits(want))
    rt.setAccumMode(PT_ACCUM_MEAN)
    rt.setFrameSchedule(first=0, stride=1)
    rt.resetFrame()
    assert rt.frame == -1
    rt.raytrace(4)
    assert rt.frame == 3
    want, _ = ocornell.render(view, proj, W, H, frames=4, spp_per_frame=4, max_depth=8, seed=5)
    assert np.array_equal(bits(rt.readImage()), bits(want))


def test_srgb8_output(rt, pa, O, cornell):
    W, H = 64, 48
    view, proj = pa.reference_camera(W, H)
    rt.buildBlas(cornell)
    rt.resize(W, H)
    rt.setCamera(view, proj)
    rt.setParams(spp_per_frame=8, max_depth=8)
    rt.raytrace(2)
    img, px = rt.readImage(), rt.readImageSrgb8()
    want = O.srgb8(img).reshape(H, W, 4)
    # powf differs by an ulp between libm and the device: at most one 8-bit step, almost never
    diff = np.abs(px.astype(np.int32) - want.astype(np.int32))
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3
    assert np.all(px[..., 3] == 255)


def test_in_range_division_and_sqrt_forms_equal_the_operators(rt):
    """the shared-reciprocal division and the unscaled sqrt the path kernel issues for in-range operands
    are the compiler's own expansions minus identity steps -- 2^28 operand sets over the admitted exponent ranges
    (ends included), bit for bit against `/` and sqrtf on the device."""
    for seed in (1, 2):
        bad = rt.debugMath(1 << 27, seed)
        assert bad.tolist() == [0, 0, 0, 1 << 27], bad
