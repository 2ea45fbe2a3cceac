"""Python host mirror of the reference's seam, over the C ABI (include/pt_hip.h).

`HipRaytracing` keeps the method names of `class VulkanRaytracing` (VulkanRaytracing.h:88-117)
for the calls that survive without Vulkan -- setContext, buildBlas, buildTlas, raytrace,
updateFrame/resetFrame, cleanup -- so that tests read like the reference's call sequence
(main.cpp:272-317, 1411-1412).  Everything here is argument marshalling; all computation is in
libpt_hip.so.  No function in this module has a CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import (PT_ACCEL_BRUTE_FORCE, PT_ACCEL_BVH4, PT_ACCEL_BVH4_PLOC, PT_ACCEL_LEAF_SIZE, PT_ACCEL_LEAF_SIZE_1, PT_ACCEL_NO_COMPACTION, PT_ACCEL_NO_ENTRY_NODE, PT_ACCEL_TRI_PLANES_1, PT_ACCEL_TRI_PLANES_3, PT_ACCUM_MEAN, PT_ACCUM_SUM, PT_NO_HIT, PT_PROFILE_KERNELS,
                   PT_PROFILE_TRAVERSAL, PT_PIPELINE_FUSED, PT_PIPELINE_SPLIT, PT_PIPELINE_PERSISTENT, PtStats)

HIT_DTYPE = np.dtype([("t", np.float32), ("u", np.float32), ("v", np.float32), ("prim", np.uint32)])
NODE_DTYPE = np.dtype([("origin", np.float32, 3), ("scale_x", np.float32), ("child", np.uint32, 4),
                       ("qlo", np.uint8, (3, 4)), ("qhi", np.uint8, (3, 4)), ("scale_yz", np.float32, 2)])
TRI_DTYPE = np.dtype([("v0", np.float32, 3), ("v1", np.float32, 3), ("v2", np.float32, 3), ("prim", np.uint32),
                      ("pad", np.uint32, 2)])
INSTANCE_DTYPE = np.dtype([("transform", np.float32, 12), ("custom_index", np.uint32), ("mask", np.uint32)])
assert INSTANCE_DTYPE.itemsize == 56 and HIT_DTYPE.itemsize == 16 and NODE_DTYPE.itemsize == 64 and TRI_DTYPE.itemsize == 48


class PathTracerError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[pt error {code}] {msg}")
        self.code = code


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


@dataclass
class Scene:
    """The four buffers of the reference's `Model` table (main.cpp:46-65, 1192-1209)."""
    vertices: np.ndarray      # (n, 11) float32: pos3 normal3 color3 uv2   (44 B records)
    indices: np.ndarray       # (m,) uint32
    materials: np.ndarray     # (k, 6) float32: diffuse3 emission3         (24 B records)
    mat_indices: np.ndarray   # (m/3,) uint32
    warnings: str = ""
    diffuse_maps: tuple = ()  # per material: path of its map_Kd image, or ""

    def __post_init__(self):
        self.vertices = np.ascontiguousarray(self.vertices, dtype=np.float32).reshape(-1, 11)
        self.indices = np.ascontiguousarray(self.indices, dtype=np.uint32).reshape(-1)
        self.materials = np.ascontiguousarray(self.materials, dtype=np.float32).reshape(-1, 6)
        self.mat_indices = np.ascontiguousarray(self.mat_indices, dtype=np.uint32).reshape(-1)

    @property
    def n_tris(self):
        return self.indices.size // 3


def load_obj(path: str) -> Scene:
    """PathTracer::loadModel (main.cpp:1101-1190) through pt_load_obj."""
    L = _lib.lib()
    h = C.c_void_p()
    rc = L.pt_load_obj(str(path).encode(), C.byref(h))
    if rc:
        raise PathTracerError(rc, L.pt_last_error(None).decode())
    return _scene_from_handle(L, h)


def scene_from_tinyobj(vertices, normals, texcoords, indices, material_ids, materials) -> Scene:
    """The de-duplication half of loadModel (main.cpp:1122-1189) over arrays tinyobj::LoadObj produced
    (pt_scene_from_tinyobj): attrib vertices / normals / texcoords, index_t triples (v, vn, vt), material ids, .mtl
    materials (k, 6)."""
    L = _lib.lib()
    v = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1)
    vn = np.ascontiguousarray(normals, dtype=np.float32).reshape(-1)
    vt = np.ascontiguousarray(texcoords, dtype=np.float32).reshape(-1)
    idx = np.ascontiguousarray(indices, dtype=np.int32).reshape(-1, 3)
    mid = np.ascontiguousarray(material_ids, dtype=np.int32).reshape(-1)
    mats = np.ascontiguousarray(materials, dtype=np.float32).reshape(-1, 6)
    if v.size % 3 or vn.size % 3 or vt.size % 2 or idx.shape[0] % 3 or mid.size != idx.shape[0] // 3:
        raise ValueError("scene_from_tinyobj: array sizes do not describe a triangulated tinyobj result")
    h = C.c_void_p()
    rc = L.pt_scene_from_tinyobj(_p(v), v.size // 3, _p(vn), vn.size // 3, _p(vt), vt.size // 2, _p(idx), idx.shape[0],
                                 _p(mid), _p(mats), mats.shape[0], C.byref(h))
    if rc:
        raise PathTracerError(rc, L.pt_last_error(None).decode())
    return _scene_from_handle(L, h)


def _scene_from_handle(L, h) -> Scene:
    try:
        nv, ni, nm = L.pt_scene_n_vertices(h), L.pt_scene_n_indices(h), L.pt_scene_n_materials(h)

        def view(ptr, dtype, count):
            if count == 0:
                return np.zeros(0, dtype=dtype)
            buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr)
            return np.frombuffer(buf, dtype=dtype, count=count).copy()

        return Scene(view(L.pt_scene_vertices(h), np.float32, nv * 11).reshape(-1, 11),
                     view(L.pt_scene_indices(h), np.uint32, ni),
                     view(L.pt_scene_materials(h), np.float32, nm * 6).reshape(-1, 6),
                     view(L.pt_scene_mat_indices(h), np.uint32, ni // 3),
                     L.pt_scene_warnings(h).decode(),
                     tuple(L.pt_scene_diffuse_map(h, m).decode() for m in range(nm)))
    finally:
        L.pt_scene_free(h)


def load_png(path: str) -> np.ndarray:
    """stb_image's role: 8-bit non-interlaced PNG -> (H, W, 4) uint8 RGBA."""
    L = _lib.lib()
    ptr, w, h = C.c_void_p(), C.c_uint32(), C.c_uint32()
    rc = L.pt_load_png(os.fsencode(path), C.byref(ptr), C.byref(w), C.byref(h))
    if rc != 0:
        raise PathTracerError(rc, (L.pt_last_error(None) or b"").decode())
    try:
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(h.value, w.value, 4)).copy()
    finally:
        L.pt_free(ptr)


def srgb8_to_linear(values) -> np.ndarray:
    v = np.ascontiguousarray(values, dtype=np.uint8)
    out = np.empty(v.shape, dtype=np.float32)
    _lib.lib().pt_srgb8_to_linear(_p(v), v.size, _p(out))
    return out


def look_at(eye, center, up):
    e, c, u = (np.ascontiguousarray(x, dtype=np.float32) for x in (eye, center, up))
    m = np.zeros(16, dtype=np.float32)
    _lib.lib().pt_look_at(_p(e), _p(c), _p(u), _p(m))
    return m


def perspective(fovy, aspect, z_near, z_far):
    m = np.zeros(16, dtype=np.float32)
    _lib.lib().pt_perspective(np.float32(fovy), np.float32(aspect), np.float32(z_near), np.float32(z_far), _p(m))
    return m


def inverse4(m):
    a = np.ascontiguousarray(m, dtype=np.float32).reshape(16)
    out = np.zeros(16, dtype=np.float32)
    _lib.lib().pt_inverse4(_p(a), _p(out))
    return out


def _write_image(fn, path, img, dtype):
    a = np.ascontiguousarray(img, dtype=dtype)
    if a.ndim != 3 or a.shape[2] != 4:
        raise ValueError("image must be (H, W, 4)")
    rc = fn(os.fsencode(path), _p(a), a.shape[1], a.shape[0])
    if rc:
        raise PathTracerError(rc, _lib.lib().pt_last_error(None).decode())


def write_png(path, rgba8):
    """8-bit RGBA PNG of `HipRaytracing.readImageSrgb8()` (pt_write_png)."""
    _write_image(_lib.lib().pt_write_png, path, rgba8, np.uint8)


def write_ppm(path, rgba8):
    _write_image(_lib.lib().pt_write_ppm, path, rgba8, np.uint8)


def write_pfm(path, rgba32f):
    _write_image(_lib.lib().pt_write_pfm, path, rgba32f, np.float32)


def reference_camera(width, height):
    """view, proj of main.cpp:1465-1480 (column-major float[16] each)."""
    v = np.zeros(16, dtype=np.float32)
    p = np.zeros(16, dtype=np.float32)
    _lib.lib().pt_reference_camera(width, height, _p(v), _p(p))
    return v, p


class HipRaytracing:
    """Drop-in for `VulkanRaytracing` on one MI355X (one context per GPU, one host thread)."""

    def __init__(self, device: int = 0):
        self._L = _lib.lib()
        self._ctx = None
        self.width = self.height = 0
        self.setContext(device)

    # ---- plumbing ----
    def _check(self, rc):
        if rc:
            raise PathTracerError(rc, self._L.pt_last_error(self._ctx).decode())

    def setContext(self, device: int):
        """VulkanRaytracing::setContext (VulkanRaytracing.cpp:9-12)."""
        if self._ctx:
            self.cleanup()
        ctx = self._L.pt_create(int(device))
        if not ctx:
            raise PathTracerError(-2, self._L.pt_last_error(None).decode())
        self._ctx = C.c_void_p(ctx)

    def cleanup(self):
        """VulkanRaytracing::cleanup (VulkanRaytracing.cpp:915-935)."""
        if self._ctx:
            self._L.pt_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.cleanup()
        except Exception:
            pass

    def set_stream(self, hip_stream: int | None):
        self._check(self._L.pt_set_stream(self._ctx, C.c_void_p(hip_stream or 0)))

    def synchronize(self):
        self._check(self._L.pt_synchronize(self._ctx))

    # ---- scene + acceleration structure ----
    def uploadScene(self, scene: Scene):
        """createModelBuffers (main.cpp:1192-1279)."""
        s = scene
        self._check(self._L.pt_upload_scene(self._ctx, _p(s.vertices), s.vertices.shape[0], _p(s.indices),
                                            s.indices.size, _p(s.materials), s.materials.shape[0], _p(s.mat_indices)))
        self._scene = s

    def uploadModels(self, scenes):
        """Several Models in one scene (raytrace.rchit:27 model.m[gl_InstanceCustomIndexEXT]): model m is shown by the
        instances whose custom_index is m (setInstances), each gets its own bottom-level structure at buildBlas."""
        import ctypes as C

        class PtModelDesc(C.Structure):
            _fields_ = [("vertices", C.c_void_p), ("n_vertices", C.c_uint32), ("indices", C.c_void_p), ("n_indices", C.c_uint32),
                        ("materials", C.c_void_p), ("n_materials", C.c_uint32), ("mat_indices", C.c_void_p)]
        scenes = list(scenes)
        arr = (PtModelDesc * max(1, len(scenes)))()
        for d, s in zip(arr, scenes):
            d.vertices, d.n_vertices = s.vertices.ctypes.data, s.vertices.shape[0]
            d.indices, d.n_indices = s.indices.ctypes.data, s.indices.size
            d.materials, d.n_materials = s.materials.ctypes.data, s.materials.shape[0]
            d.mat_indices = s.mat_indices.ctypes.data
        self._check(self._L.pt_upload_models(self._ctx, C.cast(arr, C.c_void_p) if scenes else None, len(scenes)))
        self._scene = scenes[0] if len(scenes) == 1 else None

    def buildBlas(self, scene: Scene | None = None, flags: int = PT_ACCEL_BVH4_PLOC):
        """VulkanRaytracing::buildBlas (VulkanRaytracing.cpp:147-262): here the BVH is built by our own
        kernels instead of the driver.  Includes buildTlas (single identity instance)."""
        if scene is not None:
            self.uploadScene(scene)
        self._check(self._L.pt_build_accel(self._ctx, int(flags)))

    def setInstances(self, transforms=None, custom_index=None):
        """createInstancesBuffer (VulkanRaytracing.cpp:382-411): `transforms` = (n, 12) object-to-world 3x4
        row-major matrices, or None/empty for the reference's own list (one identity instance per Model);
        custom_index = the Model of every instance (default 0).  Call before buildBlas."""
        if transforms is None or len(transforms) == 0:
            self._check(self._L.pt_set_instances(self._ctx, None, 0))
            return
        t = np.ascontiguousarray(transforms, dtype=np.float32).reshape(-1, 12)
        rec = np.zeros(t.shape[0], dtype=INSTANCE_DTYPE)
        rec["transform"] = t
        rec["mask"] = 0xFF
        if custom_index is not None:
            rec["custom_index"] = np.asarray(custom_index, dtype=np.uint32).reshape(-1)
        self._check(self._L.pt_set_instances(self._ctx, _p(rec), t.shape[0]))

    def setTexture(self, rgba8=None):
        """Diffuse map (H, W, 4) uint8 sRGB, or None to remove it (pt_set_texture)."""
        if rgba8 is None:
            self._check(self._L.pt_set_texture(self._ctx, None, 0, 0))
            return
        t = np.ascontiguousarray(rgba8, dtype=np.uint8)
        assert t.ndim == 3 and t.shape[2] == 4
        self._check(self._L.pt_set_texture(self._ctx, _p(t), t.shape[1], t.shape[0]))

    def setModelTexture(self, model: int, rgba8=None):
        """Diffuse map of ONE Model of uploadModels (pt_set_model_texture), or None to remove it; replaces a scene-wide map."""
        if rgba8 is None:
            self._check(self._L.pt_set_model_texture(self._ctx, int(model), None, 0, 0))
            return
        t = np.ascontiguousarray(rgba8, dtype=np.uint8)
        assert t.ndim == 3 and t.shape[2] == 4
        self._check(self._L.pt_set_model_texture(self._ctx, int(model), _p(t), t.shape[1], t.shape[0]))

    def setTextureFromScene(self, scene: "Scene"):
        """The first map_Kd image the scene's MTL file names and that exists (what pt_render does without -tex); returns
        its path, or None (texture left as it was) when there is none.  One texture slot: pt_set_texture."""
        for path in scene.diffuse_maps:
            if path and os.path.isfile(path):
                self.setTexture(load_png(path))
                return path
        return None

    def buildTlas(self):
        """VulkanRaytracing::buildTlas (VulkanRaytracing.cpp:448-527): one identity instance -- nothing
        left to do once the BLAS exists.  Kept so call sequences match the reference."""

    # ---- frame state ----
    def resize(self, width: int, height: int):
        """createColorResources + updateDescriptorSets + resetFrame (main.cpp:409-436, 908-931)."""
        self._check(self._L.pt_resize(self._ctx, width, height))
        self.width, self.height = width, height

    def setCamera(self, view, proj):
        """updateUniformBuffer (main.cpp:1465-1480)."""
        v = np.ascontiguousarray(view, dtype=np.float32).reshape(16)
        p = np.ascontiguousarray(proj, dtype=np.float32).reshape(16)
        self._check(self._L.pt_set_camera(self._ctx, _p(v), _p(p)))

    def setParams(self, spp_per_frame=10, max_depth=20, tmin=1e-3, tmax=1e4, seed=0, max_frames=1000):
        self._check(self._L.pt_set_params(self._ctx, spp_per_frame, max_depth, tmin, tmax, seed, max_frames))

    def setFrameSchedule(self, first=0, stride=1):
        self._check(self._L.pt_set_frame_schedule(self._ctx, first, stride))

    def setTile(self, x0=0, y0=0, w=0, h=0):
        """Restrict dispatches to a sub-rectangle (image-tile split); w = h = 0: whole image."""
        self._check(self._L.pt_set_tile(self._ctx, x0, y0, w, h))

    def setTileInterleaved(self, rank: int, world: int, block_w: int = 32):
        """pt_set_tile_interleaved: blocks of `block_w` columns dealt round-robin, this context renders those of `rank`."""
        self._check(self._L.pt_set_tile_interleaved(self._ctx, rank, world, block_w))

    def setAccumMode(self, mode):
        self._check(self._L.pt_set_accum_mode(self._ctx, mode))

    def setPipeline(self, mode):
        self._check(self._L.pt_set_pipeline(self._ctx, mode))

    def raytrace(self, frames: int = 1):
        """VulkanRaytracing::raytrace (VulkanRaytracing.cpp:868-901), `frames` times.  Asynchronous."""
        self._check(self._L.pt_render_frames(self._ctx, frames))

    def raytraceNoAlloc(self, frames: int = 1):
        """pt_render_frames_noalloc: the same frames on the GPU's fixed scratch array, without any allocation (limits in
        include/pt_hip.h).  No counterpart in the reference."""
        self._check(self._L.pt_render_frames_noalloc(self._ctx, frames))

    def updateFrame(self):
        raise NotImplementedError("updateFrame() is folded into raytrace(), as in VulkanRaytracing::raytrace")

    def resetFrame(self):
        """VulkanRaytracing::resetFrame (VulkanRaytracing.cpp:910-913)."""
        self._check(self._L.pt_reset_accumulation(self._ctx))

    @property
    def frame(self) -> int:
        return self._L.pt_frame_index(self._ctx)

    def readImage(self) -> np.ndarray:
        img = np.zeros((self.height, self.width, 4), dtype=np.float32)
        self._check(self._L.pt_read_image(self._ctx, _p(img)))
        return img

    def readImageSrgb8(self) -> np.ndarray:
        img = np.zeros((self.height, self.width, 4), dtype=np.uint8)
        self._check(self._L.pt_read_image_srgb8(self._ctx, _p(img)))
        return img

    def imageDevicePtr(self) -> int:
        p = C.c_void_p()
        self._check(self._L.pt_image_device_ptr(self._ctx, C.byref(p)))
        return p.value

    def writeImageFromDevice(self, dev_ptr: int):
        self._check(self._L.pt_write_image_from_device(self._ctx, C.c_void_p(dev_ptr)))

    def writeImage(self, rgba32f):
        """Overwrites the accumulation image from host memory (pt_write_image): H x W x 4 float32."""
        a = np.ascontiguousarray(rgba32f, dtype=np.float32)
        if a.size != 4 * self.width * self.height:
            raise ValueError("writeImage: expected H x W x 4 floats")
        self._check(self._L.pt_write_image(self._ctx, _p(a)))

    def saveCheckpoint(self, path: str):
        """Accumulation image + frame index + sampling parameters (pt_save_checkpoint)."""
        self._check(self._L.pt_save_checkpoint(self._ctx, os.fsencode(path)))

    def loadCheckpoint(self, path: str):
        self._check(self._L.pt_load_checkpoint(self._ctx, os.fsencode(path)))
        w, h = C.c_uint32(0), C.c_uint32(0)
        self._check(self._L.pt_image_extent(self._ctx, C.byref(w), C.byref(h)))
        self.width, self.height = int(w.value), int(h.value)  # the checkpoint's extent replaces the context's

    def setProfiling(self, flags=PT_PROFILE_KERNELS):
        self._check(self._L.pt_set_profiling(self._ctx, flags))

    def stats(self) -> PtStats:
        st = PtStats()
        self._check(self._L.pt_get_stats(self._ctx, C.byref(st)))
        return st

    def resetStats(self):
        self._check(self._L.pt_reset_stats(self._ctx))

    # ---- parity / inspection ----
    def debugRng(self, v0, v1, n_steps=8):
        v0 = np.ascontiguousarray(v0, dtype=np.uint32)
        v1 = np.ascontiguousarray(v1, dtype=np.uint32)
        n = v0.size
        tea = np.zeros(n, dtype=np.uint32)
        lcg = np.zeros((n, n_steps), dtype=np.uint32)
        unit = np.zeros((n, 3), dtype=np.float32)
        self._check(self._L.pt_debug_rng(self._ctx, n, _p(v0), _p(v1), n_steps, _p(tea), _p(lcg), _p(unit)))
        return tea, lcg, unit

    def debugMath(self, n, seed=1):
        """pt_debug_math: mismatch counts (quotients, roots, reciprocal roots) of the in-range IEEE forms and the number
        of operand sets checked."""
        bad = np.zeros(4, dtype=np.uint64)
        self._check(self._L.pt_debug_math(self._ctx, int(n), int(seed), _p(bad)))
        return bad

    def debugPrimaryHits(self) -> np.ndarray:
        hits = np.zeros(self.width * self.height, dtype=HIT_DTYPE)
        self._check(self._L.pt_debug_primary_hits(self._ctx, _p(hits)))
        return hits

    def debugTrace(self, rays, tmin=1e-3, tmax=1e4) -> np.ndarray:
        r = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
        hits = np.zeros(r.shape[0], dtype=HIT_DTYPE)
        self._check(self._L.pt_debug_trace(self._ctx, _p(r), r.shape[0], tmin, tmax, _p(hits)))
        return hits

    def debugMorton(self):
        n = self._scene.n_tris
        codes = np.zeros(n, dtype=np.uint32)
        order = np.zeros(n, dtype=np.uint32)
        cmin = np.zeros(3, dtype=np.float32)
        cmax = np.zeros(3, dtype=np.float32)
        self._check(self._L.pt_debug_morton(self._ctx, _p(codes), _p(order), _p(cmin), _p(cmax)))
        return codes, order, cmin, cmax

    def debugReadBvh(self):
        nn, nt = C.c_uint32(), C.c_uint32()
        self._check(self._L.pt_debug_read_bvh(self._ctx, None, C.byref(nn), None, C.byref(nt)))
        nodes = np.zeros(nn.value, dtype=NODE_DTYPE)
        tris = np.zeros(nt.value, dtype=TRI_DTYPE)
        self._check(self._L.pt_debug_read_bvh(self._ctx, _p(nodes) if nn.value else None, C.byref(nn),
                                              _p(tris) if nt.value else None, C.byref(nt)))
        return nodes, tris


__all__ = ["HipRaytracing", "PathTracerError", "Scene", "load_obj", "scene_from_tinyobj", "look_at", "perspective", "inverse4",
           "reference_camera", "write_png", "write_ppm", "write_pfm", "HIT_DTYPE", "NODE_DTYPE", "TRI_DTYPE", "PT_ACCEL_BVH4", "PT_ACCEL_BVH4_PLOC", "PT_ACCEL_BRUTE_FORCE",
           "PT_ACCUM_MEAN", "PT_ACCUM_SUM", "PT_NO_HIT", "PT_PROFILE_KERNELS", "PT_PROFILE_TRAVERSAL", "PT_PIPELINE_FUSED", "PT_PIPELINE_SPLIT", "PT_PIPELINE_PERSISTENT", "PtStats"]
