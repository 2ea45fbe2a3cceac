"""Attributes device-memory deltas (hipMemGetInfo) to API calls: 30 fresh contexts per row."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pathtracer_amd as pa
hip = ctypes.CDLL("libamdhip64.so")
def free_mem():
    f, t = ctypes.c_size_t(), ctypes.c_size_t(); hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)); return f.value
sc = pa.scenes.cornell_box()
v, p = pa.reference_camera(120, 80)
def cycle(what):
    rt = pa.HipRaytracing(0)
    if "upload" in what: rt.uploadScene(sc)
    if "build" in what: rt.buildBlas(None)
    if "resize" in what: rt.resize(120, 80); rt.setCamera(v, p)
    if "render" in what: rt.raytrace(2)
    if "read" in what: rt.readImage()
    if "stats" in what: rt.stats()
    rt.cleanup()
for what in ["", "upload", "upload build", "upload build resize", "upload build resize render", "upload build resize render read stats"]:
    for i in range(3): cycle(what)
    b = free_mem()
    for i in range(30): cycle(what)
    d = (b - free_mem()) / 2**20 / 30
    print(f"{what or 'create/destroy':45s} delta per cycle {d:.2f} MiB", flush=True)
    assert d < 0.1, "device memory is not returned"
