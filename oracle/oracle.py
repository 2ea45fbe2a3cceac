"""ctypes front-end of the CPU oracle (oracle/pt_oracle.c).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by
the product package."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
NONE = 0xFFFFFFFF


class Scene(C.Structure):
    _fields_ = [("verts", C.c_void_p), ("nVerts", C.c_uint32), ("indices", C.c_void_p), ("nIdx", C.c_uint32),
                ("materials", C.c_void_p), ("nMat", C.c_uint32), ("matIdx", C.c_void_p),
                ("instances", C.c_void_p), ("nInst", C.c_uint32),
                ("texture", C.c_void_p), ("texW", C.c_uint32), ("texH", C.c_uint32), ("instRange", C.c_void_p),
                ("texSlots", C.c_void_p), ("nTexSlots", C.c_uint32)]


class TexSlot(C.Structure):
    _fields_ = [("texels", C.c_void_p), ("w", C.c_uint32), ("h", C.c_uint32), ("primEnd", C.c_uint32), ("pad", C.c_uint32)]


class Params(C.Structure):
    _fields_ = [("spp_per_frame", C.c_int32), ("max_depth", C.c_int32), ("tmin", C.c_float), ("tmax", C.c_float),
                ("seed", C.c_uint32)]


HIT_DTYPE = np.dtype([("t", np.float32), ("u", np.float32), ("v", np.float32), ("prim", np.uint32)])


def _cpu_has_fma():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    fl = line.split()
                    return "fma" in fl and "avx2" in fl
    except OSError:
        pass
    return False


def build(force=False):
    """Compile the oracle if the shared objects are missing (gcc only; no GPU needed)."""
    if force or not os.path.exists(os.path.join(_HERE, "liboracle.so")):
        subprocess.check_call(["make", "-C", _HERE, "all"], stdout=subprocess.DEVNULL)


_lib = None


def lib(portable=False):
    global _lib
    if portable:
        build()
        return _bind(C.CDLL(os.path.join(_HERE, "liboracle.so")))
    if _lib is None:
        build()
        name = "liboracle_fma.so" if _cpu_has_fma() and os.path.exists(os.path.join(_HERE, "liboracle_fma.so")) \
            else "liboracle.so"
        _lib = _bind(C.CDLL(os.path.join(_HERE, name)))
    return _lib


def _bind(L):
    u32, f32, vp = C.c_uint32, C.c_float, C.c_void_p
    L.pto_tea.restype = u32
    L.pto_tea.argtypes = [u32, u32]
    L.pto_lcg.restype = u32
    L.pto_lcg.argtypes = [C.POINTER(u32)]
    L.pto_rnd.restype = f32
    L.pto_rnd.argtypes = [C.POINTER(u32)]
    L.pto_sincos_2pi.argtypes = [f32, C.POINTER(f32), C.POINTER(f32)]
    L.pto_rnd_unit_vector.argtypes = [C.POINTER(u32), vp]
    L.pto_lookat.argtypes = [vp, vp, vp, vp]
    L.pto_perspective.argtypes = [f32, f32, f32, f32, vp]
    L.pto_inverse4.argtypes = [vp, vp]
    L.pto_reference_camera.argtypes = [u32, u32, vp, vp]
    L.pto_camera_ray.argtypes = [vp, vp, u32, u32, u32, u32, f32, f32, vp, vp]
    L.pto_intersect_brute.argtypes = [C.POINTER(Scene), vp, vp, f32, f32, vp]
    L.pto_bvh_build.restype = vp
    L.pto_bvh_build.argtypes = [C.POINTER(Scene)]
    L.pto_bvh_free.argtypes = [vp]
    L.pto_render_frame.restype = C.c_uint64
    L.pto_render_frame.argtypes = [C.POINTER(Scene), vp, vp, vp, u32, u32, C.POINTER(Params), u32, u32, C.c_int, vp,
                                   C.c_int]
    L.pto_render_pixels.restype = C.c_uint64
    L.pto_render_pixels.argtypes = [C.POINTER(Scene), vp, vp, vp, u32, u32, C.POINTER(Params), u32, vp, u32, vp]
    L.pto_primary_hits.argtypes = [C.POINTER(Scene), vp, vp, vp, u32, u32, f32, f32, vp]
    L.pto_trace_rays.argtypes = [C.POINTER(Scene), vp, vp, u32, f32, f32, vp]
    L.pto_trace_rays_inst.argtypes = [C.POINTER(Scene), vp, vp, u32, f32, f32, vp, vp]
    L.pto_morton_codes.argtypes = [C.POINTER(Scene), vp, vp, vp]
    L.pto_morton_order.argtypes = [vp, u32, vp]
    L.pto_srgb8.argtypes = [vp, u32, vp]
    L.pto_sample_texture.argtypes = [vp, u32, u32, f32, f32, vp]
    L.pto_max_threads.restype = C.c_int
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def tea(v0, v1):
    return int(lib().pto_tea(v0 & 0xFFFFFFFF, v1 & 0xFFFFFFFF))


def lcg_stream(state, n):
    """Returns (list of n lcg outputs, final state)."""
    s = C.c_uint32(state)
    out = [int(lib().pto_lcg(C.byref(s))) for _ in range(n)]
    return out, int(s.value)


def rnd_stream(state, n):
    s = C.c_uint32(state)
    out = np.array([lib().pto_rnd(C.byref(s)) for _ in range(n)], dtype=np.float32)
    return out, int(s.value)


def sincos_2pi(u):
    s, c = C.c_float(), C.c_float()
    lib().pto_sincos_2pi(np.float32(u), C.byref(s), C.byref(c))
    return np.float32(s.value), np.float32(c.value)


def rnd_unit_vector(state):
    s = C.c_uint32(state)
    out = np.zeros(3, dtype=np.float32)
    lib().pto_rnd_unit_vector(C.byref(s), _p(out))
    return out, int(s.value)


def lookat(eye, center, up):
    m = np.zeros(16, dtype=np.float32)
    e, c, u = (np.ascontiguousarray(x, dtype=np.float32) for x in (eye, center, up))
    lib().pto_lookat(_p(e), _p(c), _p(u), _p(m))
    return m


def perspective(fovy, aspect, near, far):
    m = np.zeros(16, dtype=np.float32)
    lib().pto_perspective(np.float32(fovy), np.float32(aspect), np.float32(near), np.float32(far), _p(m))
    return m


def inverse4(m):
    a = np.ascontiguousarray(m, dtype=np.float32).reshape(16)
    out = np.zeros(16, dtype=np.float32)
    lib().pto_inverse4(_p(a), _p(out))
    return out


def reference_camera(W, H):
    v = np.zeros(16, dtype=np.float32)
    p = np.zeros(16, dtype=np.float32)
    lib().pto_reference_camera(W, H, _p(v), _p(p))
    return v, p


def instance_world_to_object(o2w):
    """World-to-object 3x4 (row-major) of an object-to-world 3x4 (row-major): glm::inverse of the 4x4 whose
    last row is (0,0,0,1), i.e. what gl_WorldToObjectEXT holds (the Vulkan spec leaves the precision of that
    inverse to the implementation; this build DEFINES it as glm's cofactor algorithm in fp32)."""
    a = np.ascontiguousarray(o2w, dtype=np.float32).reshape(3, 4)
    m = np.zeros((4, 4), dtype=np.float32)  # column-major float[16] as glm: m[c][r]
    m[:, :3] = a.T
    m[3, 3] = 1.0
    inv = inverse4(m.reshape(16)).reshape(4, 4)  # inv[c][r]
    return np.ascontiguousarray(inv[:, :3].T, dtype=np.float32).reshape(12)


class OracleScene:
    """Holds the reference-layout buffers (Vertex 44 B, Material 24 B) and an optional BVH."""

    @classmethod
    def from_models(cls, models, instances, custom_index, **kw):
        """Several Models (each with .vertices .indices .materials .mat_indices) shown by instances: instance i = Model
        custom_index[i] under transform instances[i].  The buffers are concatenated (indices / material indices rebased),
        primitive ids run through all models in order -- within one instance that is the order of the local ids."""
        v0 = m0 = t0 = 0
        vs, ids, ms, mis, first, count = [], [], [], [], [], []
        for s in models:
            vs.append(np.asarray(s.vertices, np.float32).reshape(-1, 11))
            ids.append(np.asarray(s.indices, np.uint32).reshape(-1) + np.uint32(v0))
            ms.append(np.asarray(s.materials, np.float32).reshape(-1, 6))
            mis.append(np.asarray(s.mat_indices, np.uint32).reshape(-1) + np.uint32(m0))
            first.append(t0)
            count.append(ids[-1].size // 3)
            v0 += vs[-1].shape[0]
            m0 += ms[-1].shape[0]
            t0 += count[-1]
        ci = np.asarray(custom_index, np.uint32).reshape(-1)
        rng = np.stack([np.asarray(first, np.uint32)[ci], np.asarray(count, np.uint32)[ci]], axis=1)
        if "model_textures" in kw:  # one diffuse map (or None) per Model -> (first primitive behind the Model, map)
            mt = kw.pop("model_textures")
            assert len(mt) == len(models)
            kw["texture_slots"] = [(first[m] + count[m], mt[m]) for m in range(len(models))]
        return cls(np.concatenate(vs), np.concatenate(ids), np.concatenate(ms), np.concatenate(mis), instances=instances,
                   inst_range=rng, **kw)

    def __init__(self, vertices, indices, materials, mat_indices, use_bvh=None, instances=None, texture=None, inst_range=None,
                 texture_slots=None):
        self.vertices = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 11)
        self.indices = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
        self.materials = np.ascontiguousarray(materials, dtype=np.float32).reshape(-1, 6)
        self.mat_indices = np.ascontiguousarray(mat_indices, dtype=np.uint32).reshape(-1)
        assert self.indices.size % 3 == 0 and self.mat_indices.size == self.indices.size // 3
        # instances: (n, 12) object-to-world 3x4 row-major transforms; the world-to-object halves are derived
        # here with the oracle's glm::inverse restatement, as the product derives them with its own
        self.instances = None
        if instances is not None and len(instances):
            o2w = np.ascontiguousarray(instances, dtype=np.float32).reshape(-1, 12)
            self.instances = np.zeros((o2w.shape[0], 24), dtype=np.float32)
            self.instances[:, :12] = o2w
            for i in range(o2w.shape[0]):
                self.instances[i, 12:] = instance_world_to_object(o2w[i])
        self.c = Scene(_p(self.vertices), self.vertices.shape[0], _p(self.indices), self.indices.size,
                       _p(self.materials), self.materials.shape[0], _p(self.mat_indices),
                       _p(self.instances) if self.instances is not None else None,
                       0 if self.instances is None else self.instances.shape[0])
        self.inst_range = None
        if inst_range is not None:
            self.inst_range = np.ascontiguousarray(inst_range, dtype=np.uint32).reshape(-1, 2)
            assert self.instances is not None and self.inst_range.shape[0] == self.instances.shape[0]
            self.c.instRange = self.inst_range.ctypes.data
        # texture: (H, W, 4) uint8 sRGB diffuse map -> linear float texels (IEC 61966-2-1, evaluated in double)
        self.texture = None
        if texture is not None:
            t = np.ascontiguousarray(texture, dtype=np.uint8)
            assert t.ndim == 3 and t.shape[2] == 4
            self.texture = np.ascontiguousarray(srgb8_to_linear_lut()[t])
            self.texture[..., 3] = t[..., 3].astype(np.float32) / np.float32(255.0)
            self.c.texture, self.c.texW, self.c.texH = _p(self.texture).value, t.shape[1], t.shape[0]
        # texture_slots: [(prim_end, (H, W, 4) uint8 map or None), ...] in primitive order -- one diffuse map per Model
        self.texture_slots = None
        if texture_slots is not None:
            assert texture is None
            self._slot_texels = []
            self.texture_slots = (TexSlot * len(texture_slots))()
            for i, (prim_end, tex) in enumerate(texture_slots):
                self.texture_slots[i].primEnd = int(prim_end)
                if tex is not None:
                    t = np.ascontiguousarray(tex, dtype=np.uint8)
                    assert t.ndim == 3 and t.shape[2] == 4
                    lin = np.ascontiguousarray(srgb8_to_linear_lut()[t])
                    lin[..., 3] = t[..., 3].astype(np.float32) / np.float32(255.0)
                    self._slot_texels.append(lin)
                    self.texture_slots[i].texels = _p(lin).value
                    self.texture_slots[i].w, self.texture_slots[i].h = t.shape[1], t.shape[0]
            self.c.texSlots = C.cast(self.texture_slots, C.c_void_p).value
            self.c.nTexSlots = len(texture_slots)
        if use_bvh is None:
            use_bvh = self.indices.size // 3 > 64
        self.bvh = lib().pto_bvh_build(C.byref(self.c)) if use_bvh else None

    def __del__(self):
        try:
            if self.bvh:
                lib().pto_bvh_free(self.bvh)
        except Exception:
            pass

    @property
    def n_tris(self):
        return self.indices.size // 3

    def primary_hits(self, view, proj, W, H, tmin=1e-3, tmax=1e4, brute=False):
        hits = np.zeros(W * H, dtype=HIT_DTYPE)
        v = np.ascontiguousarray(view, dtype=np.float32)
        p = np.ascontiguousarray(proj, dtype=np.float32)
        lib().pto_primary_hits(C.byref(self.c), None if brute else self.bvh, _p(v), _p(p), W, H, tmin, tmax, _p(hits))
        return hits

    def trace(self, rays, tmin=1e-3, tmax=1e4, brute=False):
        r = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
        hits = np.zeros(r.shape[0], dtype=HIT_DTYPE)
        lib().pto_trace_rays(C.byref(self.c), None if brute else self.bvh, _p(r), r.shape[0], tmin, tmax, _p(hits))
        return hits

    def trace_inst(self, rays, tmin=1e-3, tmax=1e4, brute=False):
        """(hits, instance index per hit) for an instanced scene."""
        r = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
        hits = np.zeros(r.shape[0], dtype=HIT_DTYPE)
        inst = np.zeros(r.shape[0], dtype=np.uint32)
        lib().pto_trace_rays_inst(C.byref(self.c), None if brute else self.bvh, _p(r), r.shape[0], tmin, tmax, _p(hits),
                                  _p(inst))
        return hits, inst

    def render(self, view, proj, W, H, frames, spp_per_frame=10, max_depth=20, tmin=1e-3, tmax=1e4, seed=0,
               first_frame=0, frame_stride=1, accum_sum=False, nthreads=0, image=None, lframe0=0):
        """Renders `frames` dispatches (global frame = first_frame + k*frame_stride) into a progressive
        image exactly as raytrace.rgen does.  Returns (image HxWx4 f32, rays traced)."""
        if image is None:
            image = np.zeros((H, W, 4), dtype=np.float32)
        v = np.ascontiguousarray(view, dtype=np.float32)
        p = np.ascontiguousarray(proj, dtype=np.float32)
        pp = Params(spp_per_frame, max_depth, tmin, tmax, seed)
        rays = 0
        for k in range(frames):
            rays += lib().pto_render_frame(C.byref(self.c), self.bvh, _p(v), _p(p), W, H, C.byref(pp),
                                           first_frame + k * frame_stride, lframe0 + k, int(accum_sum), _p(image),
                                           nthreads)
        return image, int(rays)

    def render_pixels(self, view, proj, W, H, pixels, gframe=0, spp_per_frame=10, max_depth=20, tmin=1e-3, tmax=1e4,
                      seed=0, return_rays=False):
        """Per-dispatch colour of the listed pixels (linear index y*W+x) for global frame `gframe` (return_rays: and the
        number of rays traced for them)."""
        px = np.ascontiguousarray(pixels, dtype=np.uint32)
        out = np.zeros((px.size, 3), dtype=np.float32)
        v = np.ascontiguousarray(view, dtype=np.float32)
        p = np.ascontiguousarray(proj, dtype=np.float32)
        pp = Params(spp_per_frame, max_depth, tmin, tmax, seed)
        rays = lib().pto_render_pixels(C.byref(self.c), self.bvh, _p(v), _p(p), W, H, C.byref(pp), gframe, _p(px), px.size,
                                       _p(out))
        return (out, int(rays)) if return_rays else out

    def morton(self):
        n = self.n_tris
        codes = np.zeros(n, dtype=np.uint32)
        cmin = np.zeros(3, dtype=np.float32)
        cmax = np.zeros(3, dtype=np.float32)
        lib().pto_morton_codes(C.byref(self.c), _p(codes), _p(cmin), _p(cmax))
        order = np.zeros(n, dtype=np.uint32)
        lib().pto_morton_order(_p(codes), n, _p(order))
        return codes, order, cmin, cmax


def camera_ray(inv_view, inv_proj, W, H, px, py, jx=0.5, jy=0.5):
    o = np.zeros(3, dtype=np.float32)
    d = np.zeros(3, dtype=np.float32)
    iv = np.ascontiguousarray(inv_view, dtype=np.float32)
    ip = np.ascontiguousarray(inv_proj, dtype=np.float32)
    lib().pto_camera_ray(_p(iv), _p(ip), W, H, px, py, np.float32(jx), np.float32(jy), _p(o), _p(d))
    return o, d


def srgb8_to_linear_lut():
    """256-entry sRGB decode table: c = i/255; c <= 0.04045 ? c/12.92 : ((c+0.055)/1.055)**2.4 in double, one rounding to float."""
    c = np.arange(256, dtype=np.float64) / 255.0
    return np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4).astype(np.float32)


def sample_texture(texture_linear, u, v):
    t = np.ascontiguousarray(texture_linear, dtype=np.float32)
    out = np.zeros(3, dtype=np.float32)
    lib().pto_sample_texture(_p(t), t.shape[1], t.shape[0], C.c_float(u), C.c_float(v), _p(out))
    return out


def srgb8(rgba):
    a = np.ascontiguousarray(rgba, dtype=np.float32).reshape(-1, 4)
    out = np.zeros((a.shape[0], 4), dtype=np.uint8)
    lib().pto_srgb8(_p(a), a.shape[0], _p(out))
    return out


def max_threads():
    return int(lib().pto_max_threads())
