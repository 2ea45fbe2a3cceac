"""ctypes binding of libpt_hip.so (the C ABI declared in include/pt_hip.h).

PyTorch is not involved: the library is plain HIP.  Loading fails loudly if the shared object is
missing -- there is no CPU fallback path in this package."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PT_HIP_LIB selects an alternative build of the same library (kernel tuning experiments)
LIB_PATH = os.environ.get("PT_HIP_LIB") or os.path.join(_HERE, "libpt_hip.so")

PT_OK = 0
PT_NO_HIT = 0xFFFFFFFF
PT_ACCEL_BVH4 = 0
PT_ACCEL_BRUTE_FORCE = 1
PT_ACCEL_BVH4_PLOC = 2
PT_ACCEL_NO_COMPACTION = 0x1000
PT_ACCEL_NO_ENTRY_NODE = 0x2000
PT_ACCEL_TRI_PLANES_1 = 0x4000
PT_ACCEL_TRI_PLANES_3 = 0x8000
PT_ACCEL_LEAF_SIZE_1 = 1 << 8  # PT_ACCEL_LEAF_SIZE(1)


def PT_ACCEL_LEAF_SIZE(n):
    return int(n) << 8


PT_ACCUM_MEAN = 0
PT_ACCUM_SUM = 1
PT_PROFILE_KERNELS = 1
PT_PROFILE_TRAVERSAL = 2
PT_PIPELINE_FUSED = 0
PT_PIPELINE_SPLIT = 1
PT_PIPELINE_PERSISTENT = 2


class PtStats(C.Structure):
    _fields_ = [("rays_total", C.c_uint64), ("rays_last_frame", C.c_uint64), ("paths_last_frame", C.c_uint64),
                ("frames_rendered", C.c_uint64), ("extend_ms_total", C.c_double), ("extend_launches", C.c_uint64),
                ("shade_ms_total", C.c_double), ("shade_launches", C.c_uint64), ("bvh_nodes", C.c_uint32),
                ("bvh_depth", C.c_uint32), ("bvh_tris", C.c_uint32), ("reserved", C.c_uint32),
                ("build_ms", C.c_double), ("trav_nodes", C.c_uint64), ("trav_tris", C.c_uint64),
                ("trav_hits", C.c_uint64), ("trav_leaf_visits", C.c_uint64), ("trav_node_steps", C.c_uint64),
                ("trav_leaf_steps", C.c_uint64), ("accel_bytes", C.c_uint64), ("accel_bytes_uncompacted", C.c_uint64), ("quad_leaves", C.c_uint64)]


# every symbol include/pt_hip.h declares: name -> (restype, argtypes)
_vp, _u32, _i, _f = C.c_void_p, C.c_uint32, C.c_int, C.c_float
SYMBOLS = {
    "pt_create": (_vp, [_i]),
    "pt_destroy": (None, [_vp]),
    "pt_last_error": (C.c_char_p, [_vp]),
    "pt_set_stream": (_i, [_vp, _vp]),
    "pt_synchronize": (_i, [_vp]),
    "pt_upload_scene": (_i, [_vp, _vp, _u32, _vp, _u32, _vp, _u32, _vp]),
    "pt_upload_models": (_i, [_vp, _vp, _u32]),
    "pt_build_accel": (_i, [_vp, _i]),
    "pt_set_instances": (_i, [_vp, _vp, _u32]),
    "pt_resize": (_i, [_vp, _u32, _u32]),
    "pt_set_camera": (_i, [_vp, _vp, _vp]),
    "pt_set_params": (_i, [_vp, _i, _i, _f, _f, _u32, _i]),
    "pt_set_frame_schedule": (_i, [_vp, _u32, _u32]),
    "pt_set_accum_mode": (_i, [_vp, _i]),
    "pt_set_pipeline": (_i, [_vp, _i]),
    "pt_render_frame": (_i, [_vp]),
    "pt_render_frames": (_i, [_vp, _i]),
    "pt_render_frames_noalloc": (_i, [_vp, _i]),
    "pt_reset_accumulation": (_i, [_vp]),
    "pt_frame_index": (_i, [_vp]),
    "pt_read_image": (_i, [_vp, _vp]),
    "pt_image_device_ptr": (_i, [_vp, C.POINTER(_vp)]),
    "pt_write_image_from_device": (_i, [_vp, _vp]),
    "pt_write_image": (_i, [_vp, _vp]),
    "pt_read_image_srgb8": (_i, [_vp, _vp]),
    "pt_set_profiling": (_i, [_vp, _i]),
    "pt_get_stats": (_i, [_vp, C.POINTER(PtStats)]),
    "pt_reset_stats": (_i, [_vp]),
    "pt_debug_rng": (_i, [_vp, _u32, _vp, _vp, _u32, _vp, _vp, _vp]),
    "pt_debug_math": (_i, [_vp, C.c_uint64, _u32, _vp]),
    "pt_debug_primary_hits": (_i, [_vp, _vp]),
    "pt_debug_trace": (_i, [_vp, _vp, _u32, _f, _f, _vp]),
    "pt_debug_morton": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "pt_debug_read_bvh": (_i, [_vp, _vp, C.POINTER(_u32), _vp, C.POINTER(_u32)]),
    "pt_load_obj": (_i, [C.c_char_p, C.POINTER(_vp)]),
    "pt_scene_from_tinyobj": (_i, [_vp, _u32, _vp, _u32, _vp, _u32, _vp, _u32, _vp, _vp, _u32, C.POINTER(_vp)]),
    "pt_scene_free": (None, [_vp]),
    "pt_scene_n_vertices": (_u32, [_vp]),
    "pt_scene_n_indices": (_u32, [_vp]),
    "pt_scene_n_materials": (_u32, [_vp]),
    "pt_scene_vertices": (_vp, [_vp]),
    "pt_scene_indices": (_vp, [_vp]),
    "pt_scene_materials": (_vp, [_vp]),
    "pt_scene_mat_indices": (_vp, [_vp]),
    "pt_scene_warnings": (C.c_char_p, [_vp]),
    "pt_scene_diffuse_map": (C.c_char_p, [_vp, C.c_uint32]),
    "pt_look_at": (None, [_vp, _vp, _vp, _vp]),
    "pt_perspective": (None, [_f, _f, _f, _f, _vp]),
    "pt_inverse4": (None, [_vp, _vp]),
    "pt_reference_camera": (None, [_u32, _u32, _vp, _vp]),
    "pt_write_pfm": (_i, [C.c_char_p, _vp, _u32, _u32]),
    "pt_write_ppm": (_i, [C.c_char_p, _vp, _u32, _u32]),
    "pt_image_extent": (_i, [_vp, _vp, _vp]),
    "pt_set_tile": (_i, [_vp, _u32, _u32, _u32, _u32]),
    "pt_set_tile_interleaved": (_i, [_vp, _u32, _u32, _u32]),
    "pt_write_png": (_i, [C.c_char_p, _vp, _u32, _u32]),
    "pt_save_checkpoint": (_i, [_vp, C.c_char_p]),
    "pt_load_checkpoint": (_i, [_vp, C.c_char_p]),
    "pt_set_texture": (_i, [_vp, _vp, _u32, _u32]),
    "pt_set_model_texture": (_i, [_vp, _u32, _vp, _u32, _u32]),
    "pt_load_png": (_i, [C.c_char_p, C.POINTER(_vp), C.POINTER(_u32), C.POINTER(_u32)]),
    "pt_free": (None, [_vp]),
    "pt_srgb8_to_linear": (None, [_vp, _u32, _vp]),
}

_lib = None


def lib():
    """Loads libpt_hip.so once.  Raises if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built (run `make -C pathtracer_amd/csrc` "
                "or __graft_entry__.build()).  There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib
