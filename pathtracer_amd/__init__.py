"""pathtracer_amd -- MI355X-native replacement for the Vulkan KHR ray-tracing hot path of
clarlzx/PathTracer.  The product is the C-ABI library libpt_hip.so (include/pt_hip.h); this
package is the thin Python host mirror used by the tests and bench.py."""
from .api import (HipRaytracing, PathTracerError, Scene, load_obj, scene_from_tinyobj, look_at, perspective, inverse4, reference_camera,
                  write_png, write_ppm, write_pfm, load_png, srgb8_to_linear)
from . import scenes

__all__ = ["load_png", "srgb8_to_linear", "write_png", "write_ppm", "write_pfm", "HipRaytracing", "PathTracerError", "Scene", "load_obj", "scene_from_tinyobj", "look_at", "perspective", "inverse4",
           "reference_camera", "scenes"]
