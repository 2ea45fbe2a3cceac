"""Where the time of bench.py's timed region goes when it runs through torch.distributed (one rank, RCCL): the render, the
stream sync, the image reduce, the barrier.  usage (under torch.distributed.run or with --standalone env): see below."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
import torch, torch.distributed as dist
import bench
import pathtracer_amd as pa
from pathtracer_amd import multigpu
from pathtracer_amd.api import PT_ACCUM_SUM
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
W, H = 1920, 1080
rt = pa.HipRaytracing(0)
rt.buildBlas(pa.scenes.cornell_box()); rt.resize(W, H); rt.setCamera(*pa.reference_camera(W, H))
rt.setParams(spp_per_frame=16, max_depth=8, max_frames=0); rt.setAccumMode(PT_ACCUM_SUM)
img_t = torch.as_tensor(bench._DevArray(rt.imageDevicePtr(), (H, W, 4)), device="cuda:0")
def once(frames):
    T = {}
    t = time.perf_counter(); rt.synchronize(); dist.barrier(); torch.cuda.synchronize(); T["barrier0"] = time.perf_counter() - t
    t0 = t = time.perf_counter(); rt.raytrace(frames); T["raytrace(call)"] = time.perf_counter() - t
    t = time.perf_counter(); rt.synchronize(); T["rt.synchronize"] = time.perf_counter() - t
    t = time.perf_counter(); multigpu.reduce_sum_image(img_t, dist, dst=0); T["reduce(call)"] = time.perf_counter() - t
    t = time.perf_counter(); torch.cuda.synchronize(); T["cuda.synchronize"] = time.perf_counter() - t
    t = time.perf_counter(); rt.synchronize(); dist.barrier(); torch.cuda.synchronize(); T["barrier1"] = time.perf_counter() - t
    T["total"] = time.perf_counter() - t0
    return T
from pathtracer_amd.api import PT_PROFILE_KERNELS
rt.setProfiling(PT_PROFILE_KERNELS)
for rep in range(4):
    rt.resetFrame(); rt.resetStats()
    T = once(20)
    st = rt.stats()
    print(" ".join(f"{k} {v*1e3:.3f} ms |" for k, v in T.items()), f"k_paths {st.extend_ms_total / max(1, st.extend_launches):.3f} ms", flush=True)
dist.destroy_process_group()
