"""What would an 8-wide node buy?  Surface-area cost (expected wide-node visits of a random ray through the root box) of the
device's BVH4 beside the same tree merged to 8-wide on the host: from the root down, a node opens its inner children -- the
one with the largest box first -- while the children still fit into eight slots; the children that stay are the next heads.
usage: gpu_bvh8_probe.py viking|teapot|soup"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import bench
import pathtracer_amd as pa
from pathtracer_amd.api import PT_ACCEL_BVH4, PT_ACCEL_BVH4_PLOC

LEAF, EMPTY = 0x80000000, 0xFFFFFFFE
name = sys.argv[1]
a = bench.parse(["--scene", name] + (["--tris", "2e5"] if name == "soup" else []))
sc, view, proj, tmax, _ = bench.build_scene(a, pa)
rt = pa.HipRaytracing(0)
rt.buildBlas(sc, PT_ACCEL_BVH4 if name == "soup" else PT_ACCEL_BVH4_PLOC)
nodes, tris = rt.debugReadBvh()
tlo = np.minimum(np.minimum(tris["v0"], tris["v1"]), tris["v2"]).astype(np.float64)
thi = np.maximum(np.maximum(tris["v0"], tris["v1"]), tris["v2"]).astype(np.float64)
sys.setrecursionlimit(100000)
boxes = {}


def area(b):
    d = np.maximum(b[1] - b[0], 0)
    return 2.0 * (d[0] * d[1] + d[1] * d[2] + d[2] * d[0])


def box(ref):
    if ref in boxes:
        return boxes[ref]
    if ref & LEAF:
        cnt, first = ((ref >> 28) & 7) + 1, ref & 0x0FFFFFFF
        b = (tlo[first:first + cnt].min(0), thi[first:first + cnt].max(0))
    else:
        lo, hi = np.full(3, np.inf), np.full(3, -np.inf)
        for c in nodes["child"][ref]:
            if int(c) != EMPTY:
                l, h = box(int(c))
                lo, hi = np.minimum(lo, l), np.maximum(hi, h)
        b = (lo, hi)
    boxes[ref] = b
    return b


def kids(ref):
    return [int(c) for c in nodes["child"][ref] if int(c) != EMPTY]


root_area = area(box(0))
cost4 = sum(area(box(i)) for i in range(len(nodes))) / root_area
fill4 = np.mean([len(kids(i)) for i in range(len(nodes))])
for width in (8, 16):
    heads, cost, fill, stack = [0], 0.0, [], 0
    while heads:
        h = heads.pop()
        ks = kids(h)
        while True:
            cand = [(area(box(k)), k) for k in ks if not (k & LEAF) and len(ks) - 1 + len(kids(k)) <= width]
            if not cand:
                break
            _, k = max(cand)
            ks.remove(k)
            ks += kids(k)
        cost += area(box(h))
        fill.append(len(ks))
        heads += [k for k in ks if not (k & LEAF)]
    print(f"{name}: BVH4 {len(nodes)} nodes, {fill4:.2f} children per node, SA cost {cost4:.3f} visits per ray | merged to {width}-wide: "
          f"{len(fill)} nodes, {np.mean(fill):.2f} children per node, SA cost {cost / root_area:.3f} = {cost / root_area / cost4:.2f} of the BVH4's", flush=True)
rt.cleanup()
