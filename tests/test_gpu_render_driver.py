"""GPU: the pt_render host driver (C++ over the C ABI; reference: main.cpp's call sequence without the window) and its
host-thread-per-GPU mode -- `-gpus G` runs one thread and one context per GPU with the sharding rule of
pathtracer_amd/multigpu.py.  One GPU rehearses the path by naming its ordinal twice (-devices 0,0)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PT_RENDER = os.path.join(ROOT, "pathtracer_amd", "pt_render")


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def read_pfm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"PF"
        w, h = map(int, f.readline().split())
        scale = float(f.readline())
        assert scale < 0  # little endian
        data = np.frombuffer(f.read(), dtype="<f4").reshape(h, w, 3)
    return data[::-1]  # PFM stores the bottom row first


def run(args):
    r = subprocess.run([PT_RENDER] + args, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def api_sum(pa, scene, W, H, spp, depth, seed, frames, first=0, stride=1, tile=None):
    from pathtracer_amd.api import PT_ACCUM_SUM
    rt = pa.HipRaytracing(0)
    rt.buildBlas(scene)
    rt.resize(W, H)
    view, proj = pa.reference_camera(W, H)
    rt.setCamera(view, proj)
    rt.setParams(spp_per_frame=spp, max_depth=depth, seed=seed, max_frames=frames)
    rt.setFrameSchedule(first=first, stride=stride)
    rt.setAccumMode(PT_ACCUM_SUM)
    if tile:
        rt.setTile(*tile)
    rt.raytrace(frames)
    img = rt.readImage()
    rays = rt.stats().rays_total
    rt.cleanup()
    return img, rays


def test_pt_render_one_and_two_contexts(pa, tmp_path):
    """-gpus 1: the image of the API's running mean.  -gpus 2 (sample streams): context r renders global frames r, r+2, ...
    as raw sums, the host adds them in rank order and divides by the frame count -- bit-identical to the same rule applied
    through the Python API.  -gpus 3 -split tiles: column strips, identical to ONE context's sum image divided the same way."""
    from pathtracer_amd import multigpu
    obj = pa.scenes.write_cornell_box(str(tmp_path))
    scene = pa.load_obj(obj)
    W, H, spp, depth, seed, frames = 72, 40, 2, 5, 9, 6
    common = [obj, "-w", str(W), "-h", str(H), "-spp", str(spp), "-depth", str(depth), "-seed", str(seed), "-frames", str(frames)]

    out1 = str(tmp_path / "one.pfm")
    log = run(common + ["-o", out1])
    assert "32 triangles" in log and "Reduced size of BLAS" in log
    rt = pa.HipRaytracing(0)
    rt.buildBlas(scene)
    rt.resize(W, H)
    rt.setCamera(*pa.reference_camera(W, H))
    rt.setParams(spp_per_frame=spp, max_depth=depth, seed=seed, max_frames=frames)
    rt.raytrace(frames)
    want = rt.readImage()
    rt.cleanup()
    assert np.array_equal(bits(read_pfm(out1)), bits(want[..., :3]))

    out2 = str(tmp_path / "two.pfm")
    log = run(common + ["-gpus", "2", "-devices", "0,0", "-o", out2])
    assert "on 2 GPUs (streams, host sum)" in log  # one ordinal named twice: RCCL wants a rank per device, so the host adds
    acc = None
    for r in range(2):
        first, stride = multigpu.frame_schedule(r, 2)
        img, _ = api_sum(pa, scene, W, H, spp, depth, seed, frames // 2, first, stride)
        acc = img.copy() if acc is None else acc + img
    want2 = multigpu.finalize_mean(acc, frames)
    assert np.array_equal(bits(read_pfm(out2)), bits(want2[..., :3]))
    assert float(np.abs(want2[..., :3] - want[..., :3]).max()) < 1e-4  # same frames, mean formed in a different order

    out3 = str(tmp_path / "tiles.png")
    outp = str(tmp_path / "tiles.pfm")
    run(common + ["-gpus", "3", "-devices", "0,0,0", "-split", "tiles", "-o", outp])
    whole, _ = api_sum(pa, scene, W, H, spp, depth, seed, frames)
    want3 = multigpu.finalize_mean(whole.copy(), frames)
    assert np.array_equal(bits(read_pfm(outp)), bits(want3[..., :3]))
    run(common + ["-gpus", "3", "-devices", "0,0,0", "-split", "tiles", "-o", out3])
    assert os.path.getsize(out3) > 100
    # two contexts: 2 x 32 columns fit the 72 of the image, so `tiles` deals blocks of 32 columns round-robin
    # (pt_set_tile_interleaved: context 0 renders columns 0-31 and 64-71, context 1 columns 32-63); `strips` = one strip each
    for split in ("tiles", "strips"):
        run(common + ["-gpus", "2", "-devices", "0,0", "-split", split, "-o", outp])
        assert np.array_equal(bits(read_pfm(outp)), bits(want3[..., :3]))


def test_pt_render_rccl_reduce_one_rank_and_device_side_srgb(pa, tmp_path):
    """-reduce rccl with one rank: ncclCommInitAll + ncclReduce(sum, fp32, 4*W*H) on the image where it lies
    (pt_image_device_ptr), then the mean -- the G > 1 code path with G = 1, which is all one GPU can run.  The image equals
    the sum image of the API divided by the frame count.  And -o x.png is the LIBRARY's sRGB encode (pt_read_image_srgb8 on the
    final image, whatever the number of GPUs): byte-identical to readImageSrgb8 of the same render."""
    from pathtracer_amd import multigpu
    obj = pa.scenes.write_cornell_box(str(tmp_path))
    scene = pa.load_obj(obj)
    W, H, spp, depth, seed, frames = 64, 48, 3, 6, 4, 5
    common = [obj, "-w", str(W), "-h", str(H), "-spp", str(spp), "-depth", str(depth), "-seed", str(seed), "-frames", str(frames)]
    out = str(tmp_path / "rccl.pfm")
    log = run(common + ["-gpus", "1", "-reduce", "rccl", "-o", out])
    assert "RCCL reduce" in log, log
    whole, _ = api_sum(pa, scene, W, H, spp, depth, seed, frames)
    want = multigpu.finalize_mean(whole.copy(), frames)
    assert np.array_equal(bits(read_pfm(out)), bits(want[..., :3])) and want[..., :3].max() > 0
    # -reduce host on the same job: the same image
    outh = str(tmp_path / "host.pfm")
    assert "host sum" in run(common + ["-gpus", "2", "-devices", "0,0", "-reduce", "host", "-o", outh])
    # PNG of a one-context render == the library's device-side encode of the same image
    png = str(tmp_path / "one.png")
    run(common + ["-o", png])
    rt = pa.HipRaytracing(0)
    rt.buildBlas(scene)
    rt.resize(W, H)
    rt.setCamera(*pa.reference_camera(W, H))
    rt.setParams(spp_per_frame=spp, max_depth=depth, seed=seed, max_frames=frames)
    rt.raytrace(frames)
    want8 = rt.readImageSrgb8()
    # ... and of the G-context mean written back with pt_write_image
    png2 = str(tmp_path / "two.png")
    run(common + ["-gpus", "2", "-devices", "0,0", "-frames", "6", "-o", png2])
    acc = None
    for r in range(2):
        first, stride = multigpu.frame_schedule(r, 2)
        img, _ = api_sum(pa, scene, W, H, spp, depth, seed, 3, first, stride)
        acc = img.copy() if acc is None else acc + img
    rt.writeImage(multigpu.finalize_mean(acc, 6))
    want8b = rt.readImageSrgb8()
    rt.cleanup()
    assert np.array_equal(pa.load_png(png).reshape(H, W, 4), want8.reshape(H, W, 4))
    assert np.array_equal(pa.load_png(png2).reshape(H, W, 4), want8b.reshape(H, W, 4))


def test_pt_render_loads_the_map_kd_of_the_mtl(pa, tmp_path):
    """An OBJ whose MTL names a diffuse map (map_Kd; tinyobj's diffuse_texname, which the reference parses and never uses):
    pt_render samples it without -tex, exactly as the API does after setTextureFromScene; `-tex none` renders the untextured
    image, which differs."""
    rng = np.random.default_rng(5)
    tex = rng.integers(0, 256, size=(8, 16, 4), dtype=np.uint8)
    tex[..., 3] = 255
    (tmp_path / "tex").mkdir()
    pa.write_png(str(tmp_path / "tex" / "checks.png"), tex)
    (tmp_path / "wall.mtl").write_text("newmtl lamp\nKd 0 0 0\nKe 4 4 4\nnewmtl wall\nKd 1 1 1\nmap_Kd -clamp off tex/checks.png\n")
    (tmp_path / "wall.obj").write_text(
        "mtllib wall.mtl\n"
        "v -1.5 0 -1\nv 1.5 0 -1\nv 1.5 2.5 -1\nv -1.5 2.5 -1\n"  # wall in front of pt_reference_camera's eye (0, 1, 3.5), facing it
        "v -2 -1 5\nv -2 3 5\nv 2 3 5\nv 2 -1 5\n"                # lamp behind the eye, facing the wall
        "vt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nvn 0 0 1\nvn 0 0 -1\n"
        "usemtl wall\nf 1/1/1 2/2/1 3/3/1 4/4/1\nusemtl lamp\nf 5/1/2 6/2/2 7/3/2 8/4/2\n")
    obj = str(tmp_path / "wall.obj")
    scene = pa.load_obj(obj)
    assert scene.diffuse_maps == ("", str(tmp_path) + "/tex/checks.png", "")
    W, H, spp, depth, seed, frames = 48, 32, 4, 4, 3, 2
    common = [obj, "-w", str(W), "-h", str(H), "-spp", str(spp), "-depth", str(depth), "-seed", str(seed), "-frames", str(frames)]
    out_t, out_n = str(tmp_path / "t.pfm"), str(tmp_path / "n.pfm")
    run(common + ["-o", out_t])
    run(common + ["-tex", "none", "-o", out_n])

    def api(textured):
        rt = pa.HipRaytracing(0)
        rt.buildBlas(scene)
        if textured:
            assert rt.setTextureFromScene(scene) == scene.diffuse_maps[1]
        rt.resize(W, H)
        rt.setCamera(*pa.reference_camera(W, H))
        rt.setParams(spp_per_frame=spp, max_depth=depth, seed=seed, max_frames=frames)
        rt.raytrace(frames)
        img = rt.readImage()
        rt.cleanup()
        return img

    want_t, want_n = api(True), api(False)
    assert np.array_equal(bits(read_pfm(out_t)), bits(want_t[..., :3]))
    assert np.array_equal(bits(read_pfm(out_n)), bits(want_n[..., :3]))
    assert want_n[..., :3].max() > 0 and not np.array_equal(want_t, want_n)


def test_pt_render_rejects_bad_arguments(tmp_path, pa):
    obj = pa.scenes.write_cornell_box(str(tmp_path))
    for extra in (["-gpus", "0"], ["-gpus", "2", "-devices", "0"], ["-split", "rows"], ["-gpus", "1", "-devices", "99"],
                  ["-reduce", "mpi"]):
        r = subprocess.run([PT_RENDER, obj, "-w", "16", "-h", "16", "-frames", "1"] + extra, capture_output=True, text=True, timeout=120)
        assert r.returncode != 0, extra
