#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (+ samples/px/s) of the wavefront path tracer on the Cornell box,
1920x1080, max depth 8 (BASELINE.json metric; workload = configs[1], 16 spp per frame so that the
default 64 steps are the 1024 spp of that config).

  python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch = one frame (W*H*spp_per_frame paths pushed
through generate -> (extend -> shade) x depth -> accumulate).  A "ray" is one closest-hit query
(one traceRayEXT of the reference).  With N > 1 (launched by torch.distributed.run, one rank per
GPU) every rank renders K disjoint global frames (rank r renders frames r, r+N, ...: weak
scaling) against a replicated scene/BVH into raw sums, and the images are combined by one RCCL
reduce(sum) over xGMI inside the timed region.

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel (k_extend) against HBM:
achieved = algorithmic bytes per launch / mean launch duration, both measured in this run (HIP
events on the library's stream, averaged over the timed steps).  `cpu_baseline` is the CPU oracle
(oracle/pt_oracle.c, "port") timed on a bounded sample of the same workload on rank 0 at N=1.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp-per-frame", type=int, default=16)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-frames", type=int, default=1)
    ap.add_argument("--out-image", default="")
    ap.add_argument("--pipeline", choices=["persistent", "fused", "split"], default="persistent")
    ap.add_argument("--builder", choices=["ploc", "lbvh"], default="ploc", help="BVH2 construction before the BVH4 collapse")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed/RCCL even for one rank (rehearses the N>1 path on one GPU)")
    return ap.parse_args()


class _DevArray:
    """Exposes a raw device pointer to torch through __cuda_array_interface__ (zero copy)."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f4", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def main():
    a = parse()
    n_gpus = a.gpus
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    torch = None
    if n_gpus > 1 or a.force_dist:
        import torch
        import torch.distributed as dist
        if world != n_gpus and not a.force_dist:
            raise SystemExit(f"--gpus {n_gpus} needs WORLD_SIZE={n_gpus} (launch with torch.distributed.run); got {world}")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import pathtracer_amd as pa
    from pathtracer_amd.api import (PT_ACCUM_MEAN, PT_ACCUM_SUM, PT_PIPELINE_FUSED, PT_PIPELINE_SPLIT, PT_PROFILE_KERNELS,
                                    PT_PROFILE_TRAVERSAL)
    from pathtracer_amd import multigpu

    W, H, S, D = a.width, a.height, a.spp_per_frame, a.depth
    scene = pa.scenes.cornell_box()
    view, proj = pa.reference_camera(W, H)
    rt = pa.HipRaytracing(local_rank)
    rt.buildBlas(scene, pa.api.PT_ACCEL_BVH4_PLOC if a.builder == "ploc" else pa.api.PT_ACCEL_BVH4)
    rt.buildTlas()
    rt.resize(W, H)
    rt.setCamera(view, proj)
    rt.setParams(spp_per_frame=S, max_depth=D, tmin=1e-3, tmax=1e4, seed=0, max_frames=0)
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
        img_t = torch.as_tensor(_DevArray(rt.imageDevicePtr(), (H, W, 4)), device=f"cuda:{local_rank}")

    def run(frames):
        rt.raytrace(frames)
        rt.synchronize()
        if dist is not None:
            multigpu.reduce_sum_image(img_t, dist, dst=0)
            torch.cuda.synchronize()

    # ---- untimed: traversal statistics for the roofline (node visits / triangle tests per ray) ----
    rt.setProfiling(PT_PROFILE_TRAVERSAL)
    rt.raytrace(2)  # frame 0 has centred jitter; include a jittered one
    rt.synchronize()
    ts = rt.stats()
    rays_stat = max(1, ts.rays_total)
    nodes_per_ray = ts.trav_nodes / rays_stat
    tris_per_ray = ts.trav_tris / rays_stat
    p_hit = ts.trav_hits / rays_stat
    divergence = None
    if ts.trav_node_steps and ts.trav_leaf_steps:
        divergence = {"node_step_lane_util": round(ts.trav_nodes / (64.0 * ts.trav_node_steps), 4),
                      "leaf_step_lane_util": round(ts.trav_leaf_visits / (64.0 * ts.trav_leaf_steps), 4),
                      "node_steps_per_64_rays": round(64.0 * ts.trav_node_steps / rays_stat, 3),
                      "leaf_steps_per_64_rays": round(64.0 * ts.trav_leaf_steps / rays_stat, 3),
                      "leaf_visits_per_ray": round(ts.trav_leaf_visits / rays_stat, 3)}
    rt.setProfiling(0)

    # ---- warmup ----
    rt.resetFrame()
    rt.resetStats()
    if a.warmup > 0:
        run(a.warmup)
    rt.resetFrame()
    rt.resetStats()
    rt.setProfiling(PT_PROFILE_KERNELS)

    # ---- timed region: exactly K steps ----
    barrier()
    t0 = time.perf_counter()
    run(a.steps)
    barrier()
    dt = time.perf_counter() - t0

    st = rt.stats()
    rays_local = float(st.rays_total)
    if dist is not None:
        v = torch.tensor([dt, rays_local], dtype=torch.float64, device=f"cuda:{local_rank}")
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
            multigpu.finalize_mean(img, world * a.steps)
        np.save(a.out_image, img)

    if rank == 0:
        mrays = rays_all / dt_max / 1e6
        total_spp = a.steps * S * world
        # dominant kernel: k_extend.  Algorithmic HBM bytes per ray: 32 B ray record read
        # + 16 B hit record written; the scene (2 KB for Cornell) is read once per launch.  Node/triangle
        # fetches (nodes*64 + tris*48 B/ray) are cache traffic on this scene and reported separately.
        # Fused pipeline (default): the dominant kernel is k_bounce = traverse + shade + compact; per ray it
        # must read the 32 B ray record, the 16 B throughput record of every ray that hits (p_hit) and write
        # a 48 B record for every survivor (p_alive); the hit record never leaves the chip.
        scene_bytes = st.bvh_nodes * 64.0 + 3 * st.bvh_tris * 48.0
        paths_total = float(W) * H * S * a.steps
        p_alive = max(0.0, (rays_local - paths_total) / max(1.0, rays_local))  # rays of bounce >= 1 = survivors
        p_prim = paths_total / max(1.0, rays_local)
        if a.pipeline == "persistent":
            # k_paths: a new path writes its 16 B state record and zeroes its 16 B radiance record; a hit reads
            # the state record; a survivor writes it back and makes one round trip (32 B out, 32 B in) through
            # the wave's ring.  Camera rays never exist in memory.
            bytes_per_ray_extend = 32.0 * p_prim + 16.0 * p_hit + 80.0 * p_alive
        elif a.pipeline == "fused":
            bytes_per_ray_extend = 32.0 + 16.0 * p_hit + 48.0 * p_alive
        else:
            bytes_per_ray_extend = 48.0
        cache_bytes_per_ray = nodes_per_ray * 64.0 + tris_per_ray * 48.0
        # whole-step figure of SURVEY.md 8(d): 160 B queue traffic + traversal + 172 B closest-hit gather
        bytes_per_ray_step = 160.0 + nodes_per_ray * 64.0 + tris_per_ray * 48.0 + 172.0 * p_hit
        ext_ms = st.extend_ms_total / max(1, st.extend_launches)
        rays_per_launch = rays_local / max(1, st.extend_launches)
        achieved = (rays_per_launch * bytes_per_ray_extend + scene_bytes) / (ext_ms * 1e-3) / 1e9 if ext_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", {"persistent": "traffic_paths.json", "fused": "traffic_bounce.json"}.get(a.pipeline, "traffic_extend.json"))
        if os.path.exists(tpath):
            try:
                with open(tpath) as f:
                    traffic = json.load(f).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Mrays/s", "value": round(mrays, 2), "unit": "Mrays/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(dt_max / a.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"Cornell box {W}x{H}, {S} spp/frame x {a.steps} frames/GPU, max depth {D}, "
                                   f"wavefront path tracer, quantised BVH4 ({a.builder.upper()}) (BASELINE.json configs[1])",
                       "width": W, "height": H, "spp_per_frame": S, "max_depth": D,
                       "total_spp": total_spp, "parallelism": f"sample-stream x{world}"},
            "samples_per_px_per_s": round(total_spp / dt_max, 3),
            "rays_per_step_per_gpu": rays_local / a.steps,
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "kernel": {"persistent": "k_paths", "fused": "k_bounce"}.get(a.pipeline, "k_extend"), "p_alive": round(p_alive, 4),
                         "launch_ms": round(ext_ms, 4), "rays_per_launch": rays_per_launch,
                         "bytes_per_ray": round(bytes_per_ray_extend, 2),
                         "cache_served_bytes_per_ray": round(cache_bytes_per_ray, 2),
                         "cache_served_gbs": round(rays_per_launch * cache_bytes_per_ray / (ext_ms * 1e-3) / 1e9, 1)
                         if ext_ms > 0 else 0.0,
                         "extend_mrays_per_s": round(rays_per_launch / (ext_ms * 1e-3) / 1e6, 1) if ext_ms > 0 else 0.0,
                         "nodes_per_ray": round(nodes_per_ray, 3),
                         "tris_per_ray": round(tris_per_ray, 3), "p_hit": round(p_hit, 4)},
            "roofline_step": {"note": "SURVEY.md 8(d) whole-step formula; its traversal/gather terms are L1/L2 traffic on this "
                                      "2.5 KB scene, so frac may exceed 1 -- HBM-only accounting is `roofline`",
                              "bytes_per_ray": round(bytes_per_ray_step, 2),
                              "achieved": round(mrays / world * 1e6 * bytes_per_ray_step / 1e9, 2), "unit": "GB/s",
                              "frac": round(mrays / world * 1e6 * bytes_per_ray_step / 1e9 / HBM_PEAK_GBS, 5),
                              "queue_only_frac": round(mrays / world * 1e6 * 160.0 / 1e9 / HBM_PEAK_GBS, 5)},
            "kernel_ms": {{"persistent": "paths_mean", "fused": "bounce_mean"}.get(a.pipeline, "extend_mean"): round(ext_ms, 4),
                          ("accumulate_mean" if a.pipeline == "persistent" else "shade_mean"):
                          round(st.shade_ms_total / max(1, st.shade_launches), 4)},
            "pipeline": a.pipeline, "builder": a.builder, "divergence": divergence,
            "bvh": {"nodes": st.bvh_nodes, "depth": st.bvh_depth, "tris": st.bvh_tris, "build_ms": round(st.build_ms, 3)},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, view, proj, W, H, S, D, a.cpu_sample_frames)
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


if __name__ == "__main__":
    main()
