#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (+ samples/px/s) of the path tracer on the Cornell box, 1920x1080, max depth 8
(BASELINE.json metric; workload = configs[1], 16 spp per frame so that the default 64 steps are the 1024 spp of
that config).

  python bench.py --gpus N --steps K --warmup W [--split streams|tiles] [--scene cornell|soup]

One "step" = one pass of the hot path over one batch = one frame: W*H*spp_per_frame paths traced from the camera
ray to termination by the persistent path kernel k_paths (generate -> traverse -> shade -> ... inside one launch,
several frames per launch), then folded into the accumulation image.  A "ray" is one closest-hit query (one
traceRayEXT of the reference).

N > 1: one process per GPU over RCCL.  When the process was NOT started by torch.distributed.run (no WORLD_SIZE in
the environment) it starts `python -m torch.distributed.run --nproc-per-node N ... bench.py <same arguments>` as a
child process BEFORE anything touches the GPU, relays rank 0's JSON line and exits with the child's code.
  --split streams (default; configs 2-4): rank r renders the global frames r, r+N, ... against a replicated
      scene/BVH into raw sums; one RCCL reduce(sum) of the image inside the timed region.  Weak scaling.
  --split tiles (config 5): every rank renders the SAME frames on its own share of the image -- blocks of --tile-block
      columns dealt round-robin (0: one column strip per rank); same reduce.  Strong scaling.
  --scaling strong (with --split streams): --steps is the TOTAL number of frames, split across the ranks.

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel (k_paths) against the bound that binds it --
VALU issue on the cache-resident Cornell box -- and `roofline_hbm` keeps the HBM figure: achieved = algorithmic
units per launch / mean launch duration, both measured in this run (HIP events on the library's stream around
every launch of the timed region).  The per-ray constants that only hardware counters can give (VALU wave
instructions per ray, HBM bytes per ray) come from the committed rocprofv3 --pmc summaries under profiles/.
`cpu_baseline` is the CPU oracle (oracle/pt_oracle.c, "port") timed on a bounded sample of the same workload on
rank 0 at N=1; `reference_lavapipe` says why the reference-on-lavapipe leg cannot run on this box.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
CUS, SIMDS_PER_CU, MAX_CLOCK_GHZ = 256, 4, 2.4
# a wave64 VALU instruction issues over 2 cycles on a SIMD-32 (MI355X_MICROARCH.md, wave scheduling)
VALU_PEAK_GINST = CUS * SIMDS_PER_CU * MAX_CLOCK_GHZ / 2.0  # 1228.8 G wave-instructions/s


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp-per-frame", type=int, default=16)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-frames", type=int, default=1)
    ap.add_argument("--out-image", default="")
    ap.add_argument("--pipeline", choices=["persistent", "fused", "split"], default="persistent")
    ap.add_argument("--builder", choices=["ploc", "lbvh"], default="ploc", help="BVH2 construction before the BVH4 collapse")
    ap.add_argument("--split", choices=["streams", "tiles"], default="streams",
                    help="multi-GPU partitioning: disjoint sample streams (frames) or image tiles (column strips)")
    ap.add_argument("--scene", choices=["cornell", "soup", "viking", "teapot"], default="cornell",
                    help="cornell = BASELINE configs[1] (the metric); soup = the procedural triangles of configs[4]; viking = "
                         "configs[2] (viking_room.obj with a 1024^2 diffuse map, use --width 3840 --height 2160); teapot = configs[3] "
                         "(teapot.obj x 64x64 instances) -- both meshes from tests/golden/ref_seam.npz, one light quad added")
    ap.add_argument("--tris", type=float, default=1e6, help="--scene soup: number of triangles")
    ap.add_argument("--tile-block", type=int, default=32,
                    help="--split tiles: columns per block of the interleaved schedule (blocks dealt round-robin to the ranks: "
                         "pt_set_tile_interleaved); 0 = one column strip per rank (max/mean strip time 2.1 on configs[4], "
                         "profiles/r03j_tile_balance.log)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="--split streams only.  weak: every rank renders --steps frames (N x the samples).  strong: --steps is the "
                         "TOTAL number of frames (configs[1]: a fixed 1024 spp), rank r renders the global frames r, r+N, ... below it")
    ap.add_argument("--cpu-sample-pixels", type=int, default=0,
                    help="cpu_baseline of the scenes other than cornell: pixels of the first timed sample (0 = 2048; a second, "
                         "larger sample is sized from it to ~8 s of CPU work)")
    ap.add_argument("--force-dist", action="store_true",
                    help="go through torch.distributed/RCCL even for one rank (rehearses the N>1 path on one GPU)")
    ap.add_argument("--ramp-ms", type=float, default=150.0,
                    help="untimed frames of the same workload rendered BEFORE the --warmup steps until this much wall time has "
                         "passed (0: none): a GPU that has idled through scene set-up starts below its sustained clock, and a "
                         "short --warmup (5 frames = 9 ms on the headline config) leaves that ramp inside the timed region "
                         "(-2.5 % on a 20-step line).  The timed region itself is unchanged; the line reports the frames spent")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="nccl = RCCL over xGMI, one GPU per rank (the measured path).  gloo = REHEARSAL of the N>1 code path "
                         "on a box with fewer GPUs than ranks: rank r renders on GPU r %% device_count, the image reduce goes "
                         "through host memory -- the line says so and its value is not a scaling number")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------ self-launch ----
def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_command(n_ranks, argv, port):
    """The torch.distributed.run command line that starts `n_ranks` copies of this script (one per GPU)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]


def self_launch(n_ranks, argv):
    """Runs the ranks as CHILD processes (this process has not touched the GPU and never will), relays the one JSON
    line rank 0 prints to stdout and everything else to stderr, returns the children's exit code."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = launch_command(n_ranks, argv, free_port())
    print("[bench] starting: " + " ".join(cmd), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    lines = 0
    for line in proc.stdout:
        if line.startswith('{"metric"'):
            lines += 1
            sys.stdout.write(line)
            sys.stdout.flush()
        else:
            sys.stderr.write(line)
    rc = proc.wait()
    if rc == 0 and lines != 1:
        print(f"[bench] expected one JSON line from rank 0, saw {lines}", file=sys.stderr)
        return 1
    return rc


# ------------------------------------------------------------------------------------------------ helpers ----
class _DevArray:
    """Exposes a raw device pointer to torch through __cuda_array_interface__ (zero copy)."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f4", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def probe_lavapipe():
    """north_star asks for the reference pipeline headless on Mesa lavapipe beside the GPU number.  That needs a Vulkan
    loader, a lavapipe ICD with VK_KHR_ray_tracing_pipeline + VK_KHR_shader_clock, and a headless harness for a program
    that requires a GLFW window surface (main.cpp:239-279).  Reports what is there."""
    import ctypes.util
    import glob
    import shutil
    loader = ctypes.util.find_library("vulkan")
    icds = [p for d in ("/usr/share/vulkan/icd.d", "/etc/vulkan/icd.d", "/usr/local/share/vulkan/icd.d")
            for p in glob.glob(os.path.join(d, "*.json"))]
    lvp = [p for p in icds if "lvp" in os.path.basename(p)]
    tools = [t for t in ("glslangValidator", "glslc", "vulkaninfo") if shutil.which(t)]
    if loader and lvp:
        return f"present but not driven: loader {loader}, ICD {lvp[0]} (the reference needs a window surface and a swapchain)"
    return ("unavailable: Vulkan loader " + (loader or "not found (ctypes.util.find_library('vulkan') is None)") +
            "; lavapipe ICD " + (lvp[0] if lvp else f"not found ({len(icds)} ICD manifests under */vulkan/icd.d)") +
            "; GLSL/Vulkan tools on PATH: " + (", ".join(tools) if tools else "none") +
            "; no package index to install them; the reference additionally needs a GLFW window surface (main.cpp:239-279)")


KERNEL_SOURCES = ("pt_paths.hip", "pt_wave_traverse.h", "pt_device_common.h", "pt_trace.h")


def kernel_source_hash():
    """sha256 over the sources of k_paths (scripts/make_pmc_json.py stamps the same value into the PMC summaries)."""
    import hashlib
    h = hashlib.sha256()
    try:
        for name in KERNEL_SOURCES:
            with open(os.path.join(ROOT, "pathtracer_amd", "csrc", name), "rb") as f:
                h.update(f.read())
    except OSError:
        return None
    return h.hexdigest()[:16]


# the per-ray counter constants are a property of (kernel instantiation, workload): one pair of summaries per scene
PROFILE_SUFFIX = {"cornell": "", "viking": "_viking", "teapot": "_teapot", "soup": "_soup"}


def profile_suffix(a):
    """profiles/valu_paths<suffix>.json / traffic_paths<suffix>.json of this workload; the soup has one pair per size
    (configs[4]'s 100 M triangles: _soup; bench.py's quick default of 1 M: _soup_1000000)"""
    if a.scene == "soup" and int(a.tris) != 100_000_000:
        return f"_soup_{int(a.tris)}"
    return PROFILE_SUFFIX[a.scene]


def workload_key(a):
    """What a PMC summary must have been taken on for its per-ray constants to price this run (the number of frames does
    not enter: the constants are per ray)."""
    k = f"{a.scene} {a.width}x{a.height} spp{a.spp_per_frame} depth{a.depth} {a.builder} {a.pipeline}"
    if a.scene == "soup":
        k += f" tris{int(a.tris)}"
    return k


def load_profile_json(name, workload=None):
    """A committed PMC summary (profiles/), or None when there is none -- or when it was taken on other kernel sources than
    the ones in this tree, or on another workload: counter constants of another kernel must not price this one."""
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            d = json.load(f)
    except Exception:
        return None
    if d.get("kernel_sources_sha") and d["kernel_sources_sha"] != kernel_source_hash():
        sys.stderr.write(f"bench.py: profiles/{name} was taken on other kernel sources ({d.get('commit')}); its counter "
                         "constants are not used (regenerate: scripts/gpu_round_profile.sh + scripts/collect_profile.sh)\n")
        return None
    if workload and d.get("workload") and d["workload"] != workload:
        sys.stderr.write(f"bench.py: profiles/{name} was taken on '{d['workload']}', this run is '{workload}'; its counter "
                         "constants are not used\n")
        return None
    return d


def reference_mesh(pa, name):
    """The reference's models/<name>.obj as its own tinyobjloader read it (oracle/_ref dump in tests/golden/ref_seam.npz),
    through the product's loadModel half."""
    import numpy as np
    z = np.load(os.path.join(ROOT, "tests", "golden", "ref_seam.npz"))
    return pa.scene_from_tinyobj(*(z[f"{name}_{k}"] for k in ("v", "vn", "vt", "idx", "mid", "mat")))


def build_scene(a, pa):
    """(scene, view, proj, tmax, description); a.instances / a.texture are set for the scenes that have them"""
    import math
    import numpy as np
    W, H = a.width, a.height
    a.instances = a.texture = None
    if a.scene == "cornell":
        view, proj = pa.reference_camera(W, H)
        return pa.scenes.cornell_box(), view, proj, 1e4, "Cornell box"
    if a.scene == "viking":
        # neither mesh comes with a .mtl or an emitter in the reference (all grey, black image): one light quad is added
        sc = pa.scenes.with_light_quad(reference_mesh(pa, "viking_room"),
                                       [[-2.0, -2.0, 2.0], [2.0, -2.0, 2.0], [2.0, 2.0, 2.0], [-2.0, 2.0, 2.0]], (0, 0, -1))
        # the texels of the reference's textures/viking_room.png (decoded in the build container: tests/golden/make_viking_texture.py)
        a.texture = np.load(os.path.join(ROOT, "tests", "golden", "viking_room_texels.npz"))["rgba8"]
        view = pa.look_at((1.3, 1.3, 1.2), (0.05, 0.0, 0.3), (0.0, 0.0, 1.0))
        proj = pa.perspective(math.radians(45.0), W / H, 0.1, 10.0)
        proj[5] *= -1
        return sc, view, proj, 1e4, f"viking_room.obj ({sc.n_tris} triangles) with its 1024x1024 diffuse map (textures/viking_room.png)"
    if a.scene == "teapot":
        sc = pa.scenes.with_light_quad(reference_mesh(pa, "teapot"),
                                       [[-3.0, 5.0, -3.0], [3.0, 5.0, -3.0], [3.0, 5.0, 3.0], [-3.0, 5.0, 3.0]], (0, -1, 0))
        n = 64
        ext = 8.0 * n
        a.instances = pa.scenes.instance_grid(n, 8.0)
        view = pa.look_at((0.0, 0.45 * ext, 0.8 * ext), (0.0, 0.0, 0.0), (0, 1, 0))
        proj = pa.perspective(math.radians(45.0), W / H, 0.1, 10.0 * ext)
        proj[5] *= -1
        return sc, view, proj, 1e6, f"teapot.obj ({sc.n_tris} triangles) x {n}x{n} instances = {n * n * sc.n_tris / 1e6:.1f} M instanced triangles"
    n = int(a.tris)
    extent = 1000.0 * (n / 1e8) ** (1.0 / 3.0)  # the density of configs[4]: 1e8 triangles in a 1000^3 cube
    sc = pa.scenes.random_triangles(n, extent=extent, edge=1.0)
    c = extent * 0.5
    view = pa.look_at((c, c * 1.2, c + extent * 1.6), (c, c * 0.9, c), (0, 1, 0))
    proj = pa.perspective(math.radians(45.0), W / H, 0.1, extent * 10)
    proj[5] *= -1  # main.cpp:1475
    return sc, view, proj, 1e6, f"procedural soup of {n} triangles"


# --------------------------------------------------------------------------------------------------- main ----
def main():
    a = parse()
    n_gpus = a.gpus
    launched = "WORLD_SIZE" in os.environ
    if (n_gpus > 1 or a.force_dist) and not launched:
        # plain `python bench.py --gpus N`: become the launcher.  Nothing above has imported torch or the library.
        sys.exit(self_launch(max(1, n_gpus), sys.argv[1:]))

    import numpy as np
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    torch = None
    rccl = None
    device = local_rank  # GPU of this rank
    if n_gpus > 1 or a.force_dist:
        import torch
        import torch.distributed as dist
        if world != n_gpus and not a.force_dist:
            raise SystemExit(f"--gpus {n_gpus} but WORLD_SIZE={world}")
        if a.dist_backend == "gloo":
            device = local_rank % max(1, torch.cuda.device_count())  # rehearsal: ranks may share a GPU
            torch.cuda.set_device(device)
            dist.init_process_group("gloo")
        else:
            # one GPU per rank; a launcher that narrows each rank's visibility to its own GPU (HIP_VISIBLE_DEVICES per rank)
            # leaves that GPU as device 0
            ndev = torch.cuda.device_count()
            device = local_rank if local_rank < ndev else local_rank % max(1, ndev)
            torch.cuda.set_device(device)
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        world = dist.get_world_size()  # as the backend sees it
        try:
            rccl = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception:
            rccl = "unknown"

    import pathtracer_amd as pa
    from pathtracer_amd.api import (PT_ACCUM_MEAN, PT_ACCUM_SUM, PT_PIPELINE_FUSED, PT_PIPELINE_SPLIT, PT_PROFILE_KERNELS,
                                    PT_PROFILE_TRAVERSAL)
    from pathtracer_amd import multigpu

    W, H, S, D = a.width, a.height, a.spp_per_frame, a.depth
    scene, view, proj, tmax, scene_name = build_scene(a, pa)
    rt = pa.HipRaytracing(device)
    if a.instances is not None:
        rt.setInstances(a.instances)
    rt.buildBlas(scene, pa.api.PT_ACCEL_BVH4_PLOC if a.builder == "ploc" else pa.api.PT_ACCEL_BVH4)
    rt.buildTlas()
    if a.texture is not None:
        rt.setTexture(a.texture)
    rt.resize(W, H)
    rt.setCamera(view, proj)
    rt.setParams(spp_per_frame=S, max_depth=D, tmin=1e-3, tmax=tmax, seed=0, max_frames=0)
    tiles = a.split == "tiles"
    strong = a.scaling == "strong" and not tiles
    if strong and a.steps < world:
        raise SystemExit(f"--scaling strong: {a.steps} frames cannot be split over {world} ranks")

    def local_frames(total):
        """frames this rank renders of `total`: all of them (weak scaling, tiles), or its share r, r+N, ... < total (strong)"""
        return multigpu.strong_frame_count(rank, world, total) if strong else total

    if tiles:
        rt.setFrameSchedule(first=0, stride=1)  # every rank renders the same frames ...
        if a.tile_block > 0:
            rt.setTileInterleaved(rank, world, a.tile_block)  # ... on its own share of the column blocks
        else:
            rt.setTile(*multigpu.tile_schedule(rank, world, W, H))  # ... on its own column strip
    else:
        first, stride = multigpu.frame_schedule(rank, world)
        rt.setFrameSchedule(first=first, stride=stride)
    rt.setAccumMode(PT_ACCUM_SUM if dist is not None else PT_ACCUM_MEAN)
    rt.setPipeline({"fused": PT_PIPELINE_FUSED, "split": PT_PIPELINE_SPLIT, "persistent": pa.api.PT_PIPELINE_PERSISTENT}[a.pipeline])

    def barrier():
        rt.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    img_t = None
    if dist is not None:
        img_t = torch.as_tensor(_DevArray(rt.imageDevicePtr(), (H, W, 4)), device=f"cuda:{device}")

    def run(frames):
        rt.raytrace(frames)
        rt.synchronize()
        if dist is not None:
            multigpu.reduce_sum_image(img_t, dist, dst=0)
            torch.cuda.synchronize()

    # ---- untimed: traversal statistics (node visits / triangle tests per ray, lane utilisation of the steps) ----
    rt.setProfiling(PT_PROFILE_TRAVERSAL)
    rt.raytrace(2)  # frame 0 has centred jitter; include a jittered one
    rt.synchronize()
    ts = rt.stats()
    rays_stat = max(1, ts.rays_total)
    nodes_per_ray = ts.trav_nodes / rays_stat
    tris_per_ray = ts.trav_tris / rays_stat
    p_hit = ts.trav_hits / rays_stat
    # the flat form of k_paths (every child of the entry node is a leaf): one wide node visit per ray, no node step
    flat_form = a.pipeline == "persistent" and ts.trav_node_steps == 0 and ts.trav_nodes == ts.rays_total and ts.bvh_tris < 255
    divergence = None
    if ts.trav_leaf_steps:
        # steps of the traversal loop only: with an entry node (pt_wave_traverse.h) a ray's first node visit -- on a flat
        # scene like the Cornell box also its first leaf test -- happens in the staging phase, with all 64 lanes
        divergence = {"node_step_lane_util": (round(ts.trav_nodes / (64.0 * ts.trav_node_steps), 4)
                                              if ts.trav_node_steps else None),
                      # (flat scenes: trav_leaf_visits includes the first leaf of every ray, tested while staging -- the ratio
                      # says nothing about the loop then)
                      "leaf_step_lane_util": (round(ts.trav_leaf_visits / (64.0 * ts.trav_leaf_steps), 4)
                                              if ts.trav_leaf_visits <= 64.0 * ts.trav_leaf_steps else None),
                      "node_steps_per_64_rays": round(64.0 * ts.trav_node_steps / rays_stat, 3),
                      "leaf_steps_per_64_rays": round(64.0 * ts.trav_leaf_steps / rays_stat, 3),
                      "leaf_visits_per_ray": round(ts.trav_leaf_visits / rays_stat, 3),
                      "note": "with an entry node (default for trees up to 2^20 nodes) trav_nodes includes one visit per ray made "
                              "in the staging phase, outside the counted node steps"}
    rt.setProfiling(0)

    # the FIRST barrier of a process group sets its communicator up (~20 ms of host time with an idle GPU, after which a
    # launch starts at a ramped-down clock: measured 5 % on the 20-step line): take it here, not in front of the timed region
    barrier()

    # ---- clock ramp (untimed, before the warm-up steps): the same frames until --ramp-ms have passed ----
    ramp_frames = 0
    if a.ramp_ms > 0:
        t_ramp = time.perf_counter()
        while (time.perf_counter() - t_ramp) * 1e3 < a.ramp_ms and ramp_frames < 256:
            rt.raytrace(4)
            rt.synchronize()
            ramp_frames += 4

    # ---- warmup ----
    rt.resetFrame()
    rt.resetStats()
    if a.warmup > 0:
        run(max(1, local_frames(a.warmup)))
    rt.resetFrame()
    rt.resetStats()
    rt.setProfiling(PT_PROFILE_KERNELS)
    if dist is not None and tiles:
        # the warm-up's reduce left the other ranks' tiles in rank 0's image; a rank's first frame overwrites its OWN tile only,
        # and "zero outside the tile" is what makes reduce(sum) a gather (pt_resize zeroes the image, nothing after it does)
        img_t.zero_()
        torch.cuda.synchronize()

    # ---- timed region: exactly K steps ----
    barrier()
    t0 = time.perf_counter()
    run(local_frames(a.steps))
    barrier()
    dt = time.perf_counter() - t0

    st = rt.stats()
    rays_local = float(st.rays_total)
    if dist is not None:
        v = torch.tensor([dt, rays_local], dtype=torch.float64, device="cpu" if a.dist_backend == "gloo" else f"cuda:{device}")
        mx = v.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = v.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        dt_max, rays_all = float(mx[0]), float(sm[1])
    else:
        dt_max, rays_all = dt, rays_local

    if a.out_image and rank == 0:
        img = rt.readImage()
        if dist is not None:
            multigpu.finalize_mean(img, a.steps if (tiles or strong) else world * a.steps)
        np.save(a.out_image, img)

    if rank == 0:
        mrays = rays_all / dt_max / 1e6
        total_spp = a.steps * S * (1 if (tiles or strong) else world)
        scene_bytes = st.bvh_nodes * 64.0 + 3 * st.bvh_tris * 48.0
        paths_local = float(st.paths_last_frame) * local_frames(a.steps)
        p_alive = max(0.0, (rays_local - paths_local) / max(1.0, rays_local))  # rays of bounce >= 1 = survivors
        p_prim = paths_local / max(1.0, rays_local)
        kernel = {"persistent": "k_paths", "fused": "k_bounce"}.get(a.pipeline, "k_extend")
        cache_bytes_per_ray = nodes_per_ray * 64.0 + tris_per_ray * 48.0
        if a.pipeline == "persistent":
            # k_paths: a new path writes its 16 B state record; a hit reads the state
            # record; a survivor writes it back and makes one round trip (32 B out, 32 B in) through the wave's ring.
            # Camera rays and hit records never exist in memory.
            # The radiance record is written (16 B) only by a path that reaches a light -- at most the hits that end a path --
            # and a new path clears one bit of the launch's mask (8 B per 64 paths) instead of zeroing it.
            p_end = max(0.0, p_hit - p_alive)
            bytes_per_ray = (16.0 + 0.125) * p_prim + 16.0 * p_hit + 80.0 * p_alive + 16.0 * p_end
            if flat_form:
                # flat scenes (k_paths<.., PT_ENTRY_FLAT>): throughput and RNG state travel with the ray -- no state record;
                # a survivor makes one round trip of 48 B out + 48 B in through the wave's ring
                bytes_per_ray = 0.125 * p_prim + 96.0 * p_alive + 16.0 * p_end
        elif a.pipeline == "fused":
            # k_bounce (one launch per bounce): 32 B ray record read, 16 B throughput record per hit, 48 B per survivor
            bytes_per_ray = 32.0 + 16.0 * p_hit + 48.0 * p_alive
        else:
            # k_extend: 32 B ray record read + 16 B hit record written
            bytes_per_ray = 48.0
        # whole-step figure of SURVEY.md 8(d): 160 B queue traffic + traversal + 172 B closest-hit gather
        bytes_per_ray_step = 160.0 + nodes_per_ray * 64.0 + tris_per_ray * 48.0 + 172.0 * p_hit
        launches = max(1, st.extend_launches)
        ext_ms = st.extend_ms_total / launches
        rays_per_launch = rays_local / launches
        sec = ext_ms * 1e-3
        # A scene whose node + triangle arrays exceed the 256 MiB Infinity Cache cannot be cache-resident: its traversal
        # fetches (n_nodes x 64 B + n_tris x 48 B per ray) are HBM traffic and count as algorithmic bytes;
        # below that size they are served by L1/L2/MALL and are reported separately (cache_served_*).
        traversal_from_hbm = scene_bytes > 256.0 * 2 ** 20
        if traversal_from_hbm:
            bytes_per_ray += cache_bytes_per_ray
        hbm_achieved = (rays_per_launch * bytes_per_ray + (0.0 if traversal_from_hbm else scene_bytes)) / sec / 1e9 if sec > 0 else 0.0
        # hardware-counter constants of this kernel on this workload, from the committed --pmc summaries (per RAY, so that
        # they compare with bytes_per_ray whatever the launch length)
        sfx, wkey = profile_suffix(a), workload_key(a)
        tj = load_profile_json({"persistent": f"traffic_paths{sfx}.json", "fused": "traffic_bounce.json"}.get(a.pipeline, "none"), wkey)
        traffic_per_ray = tj.get("hbm_bytes_per_ray") if tj else None
        traffic = round(traffic_per_ray * rays_per_launch) if traffic_per_ray else None
        hbm = {"bound": "hbm", "achieved": round(hbm_achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": round(hbm_achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
               "traffic_bytes_per_ray": round(traffic_per_ray, 2) if traffic_per_ray else None,
               "bytes_per_ray": round(bytes_per_ray, 2),
               "traffic_over_algorithmic": round(traffic_per_ray / bytes_per_ray, 3) if traffic_per_ray else None,
               # what the counters say left the L2 per second (FETCH_SIZE counts Infinity-Cache hits too: an upper bound on HBM)
               "traffic_gbs": round(traffic_per_ray * rays_per_launch / sec / 1e9, 1) if (traffic_per_ray and sec > 0) else None,
               "traffic_frac_of_hbm_peak": round(traffic_per_ray * rays_per_launch / sec / 1e9 / HBM_PEAK_GBS, 4) if (traffic_per_ray and sec > 0) else None,
               "kernel": kernel, "launch_ms": round(ext_ms, 4), "rays_per_launch": rays_per_launch,
               "p_alive": round(p_alive, 4), "p_hit": round(p_hit, 4),
               "traversal_bytes_counted": traversal_from_hbm, "scene_bytes": scene_bytes,
               "profile": tj.get("source") if tj else None,
               "cache_served_bytes_per_ray": round(cache_bytes_per_ray, 2),
               "cache_served_gbs": round(rays_per_launch * cache_bytes_per_ray / sec / 1e9, 1) if sec > 0 else 0.0,
               "nodes_per_ray": round(nodes_per_ray, 3), "tris_per_ray": round(tris_per_ray, 3)}
        vj = load_profile_json(f"valu_paths{sfx}.json", wkey) if a.pipeline == "persistent" else None
        valu = None
        if vj and vj.get("valu_wave_insts_per_ray"):
            ipr = float(vj["valu_wave_insts_per_ray"])
            ginst = rays_per_launch * ipr / sec / 1e9 if sec > 0 else 0.0
            held = float(vj.get("clock_ghz") or MAX_CLOCK_GHZ)
            valu = {"bound": "valu", "achieved": round(ginst, 2), "peak": VALU_PEAK_GINST, "unit": "Ginst/s",
                    "frac": round(ginst / VALU_PEAK_GINST, 5), "traffic": traffic,
                    "note": "VALU wave-instruction issue: achieved = SQ_INSTS_VALU per ray (committed rocprofv3 --pmc pass, "
                            f"profiles/valu_paths{sfx}.json) x rays per launch / launch duration (live); peak = 256 CUs x 4 SIMDs x "
                            "2.4 GHz / 2 cycles per wave64 instruction.  Cheapest-instruction peak: cmp/cndmask/cvt/min/max "
                            "issue at ~1.5x and rcp/sqrt at ~3x that cost, so 1.0 is not reachable by this mix",
                    "kernel": kernel, "launch_ms": round(ext_ms, 4), "rays_per_launch": rays_per_launch,
                    "valu_wave_insts_per_ray": round(ipr, 3), "valu_lane_util": vj.get("valu_lane_util"),
                    "wait_any_share_of_wave_cycles": vj.get("wait_any_share"), "l2_hit_rate": vj.get("l2_hit_rate"),
                    "kernel_instantiation": vj.get("kernel"),
                    "clock_ghz_held_in_profile": held,
                    "frac_at_held_clock": round(ginst / (CUS * SIMDS_PER_CU * held / 2.0), 5),
                    "extend_mrays_per_s": round(rays_per_launch / sec / 1e6, 1) if sec > 0 else 0.0,
                    "profile": vj.get("source")}
        # the bound that binds = the larger fraction of its peak
        if valu and valu["frac"] >= hbm["frac"]:
            roofline, other_key, other = valu, "roofline_hbm", hbm
        else:
            roofline, other_key, other = hbm, "roofline_valu", valu
        out = {
            "metric": "Mrays/s", "value": round(mrays, 2), "unit": "Mrays/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(dt_max / a.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong" if (tiles or strong) else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{scene_name} {W}x{H}, {S} spp/frame x {a.steps} frames"
                                   f"{'' if (tiles or strong) else '/GPU'}, max depth {D}, persistent wavefront path tracer, quantised BVH4 "
                                   f"({a.builder.upper()})" + {"cornell": " (BASELINE.json configs[1])", "soup": " (BASELINE.json configs[4] generator)",
                                                             "viking": " (BASELINE.json configs[2])", "teapot": " (BASELINE.json configs[3])"}[a.scene],
                       "width": W, "height": H, "spp_per_frame": S, "max_depth": D, "total_spp": total_spp,
                       "parallelism": ((f"image tiles (blocks of {a.tile_block} columns, round-robin) x{world}" if a.tile_block > 0
                                        else f"image tiles (column strips) x{world}") if tiles else
                                       f"sample-stream x{world}" + (" (frames of one fixed total split across the ranks)" if strong else ""))},
            "samples_per_px_per_s": round(total_spp / dt_max, 3),
            "rays_per_step_per_gpu": rays_local / max(1, local_frames(a.steps)),
            "roofline": roofline, other_key: other,
            "roofline_step": {"note": "SURVEY.md 8(d) whole-step formula; its traversal/gather terms are L1/L2 traffic on a "
                                      "cache-resident scene, so frac may exceed 1 -- HBM-only accounting is roofline_hbm",
                              "bytes_per_ray": round(bytes_per_ray_step, 2),
                              "achieved": round(mrays / world * 1e6 * bytes_per_ray_step / 1e9, 2), "unit": "GB/s",
                              "frac": round(mrays / world * 1e6 * bytes_per_ray_step / 1e9 / HBM_PEAK_GBS, 5),
                              "queue_only_frac": round(mrays / world * 1e6 * 160.0 / 1e9 / HBM_PEAK_GBS, 5)},
            "kernel_ms": {{"persistent": "paths_mean", "fused": "bounce_mean"}.get(a.pipeline, "extend_mean"): round(ext_ms, 4),
                          ("accumulate_mean" if a.pipeline == "persistent" else "shade_mean"):
                          round(st.shade_ms_total / max(1, st.shade_launches), 4),
                          "launches": int(st.extend_launches)},
            "pipeline": a.pipeline, "builder": a.builder, "split": a.split, "divergence": divergence,
            "profile_workload": workload_key(a), "bench_args": " ".join(sys.argv[1:]),
            "clock_ramp": {"frames": ramp_frames, "ms_asked": a.ramp_ms,
                           "note": "untimed frames rendered before the --warmup steps so that the timed region starts at the "
                                   "sustained clock (--ramp-ms 0: none)"},
            "bvh": {"nodes": st.bvh_nodes, "depth": st.bvh_depth, "tris": st.bvh_tris, "build_ms": round(st.build_ms, 3)},
            "distributed": ({"backend": "nccl (RCCL)", "rccl_version": rccl, "world_size": world,
                             "launcher": "torch.distributed.run"} if (dist is not None and a.dist_backend == "nccl") else
                            {"backend": "gloo (REHEARSAL: ranks share GPUs, host-memory reduce; not a scaling measurement)",
                             "world_size": world, "gpus_visible": torch.cuda.device_count(),
                             "launcher": "torch.distributed.run"} if dist is not None else None),
            "reference_lavapipe": probe_lavapipe(),
        }
        if world == 1 and not a.no_cpu_baseline:
            if a.scene == "cornell":
                out["cpu_baseline"] = cpu_baseline(scene, view, proj, W, H, S, D, a.cpu_sample_frames)
            else:
                out["cpu_baseline"] = cpu_baseline_pixels(a, scene, view, proj, tmax)
        print(json.dumps(out), flush=True)

    rt.cleanup()
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(scene, view, proj, W, H, S, D, frames):
    """CPU oracle (a port: plain-C restatement of the reference shaders, OpenMP over rows) on a
    bounded sample of the same workload: `frames` jittered frames of the full-size image."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    osc = O.OracleScene(scene.vertices, scene.indices, scene.materials, scene.mat_indices)
    threads = O.max_threads()
    t0 = time.perf_counter()
    _, rays = osc.render(view, proj, W, H, frames=frames, spp_per_frame=S, max_depth=D, seed=0, first_frame=1)
    dt = time.perf_counter() - t0
    return {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"{frames} frame(s) of {W}x{H} x {S} spp, depth {D} (global frame 1..), {rays} rays in {dt:.2f} s",
            "samples_per_px_per_s": round(frames * S / dt, 4)}


def cpu_baseline_pixels(a, scene, view, proj, tmax, budget_s=8.0):
    """The same oracle on a bounded PIXEL sample of the workload (the scenes whose full frame would take the CPU minutes):
    a seeded random set of pixels of global frame 1 (jittered), all spp and bounces of each; rays/s = rays traced / wall.
    A first sample sizes a second one to ~budget_s of CPU work; the second is the one reported.  For a two-level scene
    the oracle has no top-level structure (its definition is "every instance": one root-box test per instance and ray),
    which is what its rate then measures."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    n_tris = scene.n_tris
    if a.instances is None and n_tris > 2_000_000:
        return {"value": None, "unit": "Mrays/s", "cores": O.max_threads(), "kind": "port",
                "sample": f"not run: the oracle's single-threaded BVH2 build over {n_tris} triangles takes ~7 us per triangle "
                          f"(~{7e-6 * n_tris / 60:.0f} min here); --scene soup --tris 1e6 times the CPU on the same generator"}
    t0 = time.perf_counter()
    osc = O.OracleScene(scene.vertices, scene.indices, scene.materials, scene.mat_indices, instances=a.instances,
                        texture=a.texture)
    build_s = time.perf_counter() - t0
    W, H, S, D = a.width, a.height, a.spp_per_frame, a.depth
    rng = np.random.default_rng(5)

    def sample(n):
        px = rng.choice(W * H, size=min(n, W * H), replace=False).astype(np.uint32)
        t = time.perf_counter()
        _, rays = osc.render_pixels(view, proj, W, H, px, gframe=1, spp_per_frame=S, max_depth=D, tmin=1e-3, tmax=tmax,
                                    seed=0, return_rays=True)
        return px.size, rays, time.perf_counter() - t

    n, rays, dt = sample(a.cpu_sample_pixels or 2048)
    if dt < 0.5 * budget_s and n < W * H:
        n, rays, dt = sample(int(n * budget_s / max(dt, 1e-3)))
    return {"value": round(rays / dt / 1e6, 4), "unit": "Mrays/s", "cores": O.max_threads(), "kind": "port",
            "sample": f"{n} random pixels of {W}x{H} (global frame 1) x {S} spp, depth {D}: {rays} rays in {dt:.2f} s; "
                      f"oracle BVH2 built in {build_s:.2f} s (not timed)"
                      + ("; no top-level structure in the oracle: every ray tests every instance's root box" if a.instances is not None else ""),
            "samples_per_px_per_s": round(S * n / (W * H) / dt, 6)}


if __name__ == "__main__":
    main()
