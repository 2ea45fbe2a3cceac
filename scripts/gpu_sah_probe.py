"""How far is the device-built tree from a full-sweep SAH tree?  Surface-area cost (expected wide nodes visited and triangles
tested per random ray that hits the root box) of the BVH4 the library builds for a mesh, beside the same cost of a top-down
full-sweep SAH BVH2 built here in numpy and collapsed to four-wide the same way (largest child opened first).  An upper bound
on what a better builder could buy, before writing one.
usage: gpu_sah_probe.py viking|teapot"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import bench
import pathtracer_amd as pa
from pathtracer_amd.api import PT_ACCEL_BVH4, PT_ACCEL_BVH4_PLOC

LEAF = 0x80000000
EMPTY = 0xFFFFFFFE


def area(lo, hi):
    d = np.maximum(hi - lo, 0)
    return 2.0 * (d[0] * d[1] + d[1] * d[2] + d[2] * d[0])


def device_tree_cost(nodes, tris):
    tlo = np.minimum(np.minimum(tris["v0"], tris["v1"]), tris["v2"]).astype(np.float64)
    thi = np.maximum(np.maximum(tris["v0"], tris["v1"]), tris["v2"]).astype(np.float64)
    inner_area, leaf_area, leaves, leaf_tris = [0.0], [0.0], [0], [0]

    def box(ref):
        if ref & LEAF:
            cnt = ((ref >> 28) & 7) + 1
            first = ref & 0x0FFFFFFF
            lo, hi = tlo[first:first + cnt].min(0), thi[first:first + cnt].max(0)
            leaf_area[0] += area(lo, hi) * cnt
            leaves[0] += 1
            leaf_tris[0] += cnt
            return lo, hi
        lo, hi = np.full(3, np.inf), np.full(3, -np.inf)
        for c in nodes["child"][ref]:
            c = int(c)
            if c == EMPTY:
                continue
            l, h = box(c)
            lo, hi = np.minimum(lo, l), np.maximum(hi, h)
        inner_area[0] += area(lo, hi)
        return lo, hi

    sys.setrecursionlimit(10000)
    lo, hi = box(0)
    ra = area(lo, hi)
    return inner_area[0] / ra, leaf_area[0] / ra, len(nodes), leaves[0], leaf_tris[0]


def sah_tree_cost(tlo, thi, leaf_max, ct=1.0, cn=1.0):
    """full-sweep SAH BVH2, then collapsed to BVH4; returns the same pair of costs"""
    cen = 0.5 * (tlo + thi)

    def sweep_areas(lo, hi):
        # prefix (left) and suffix (right) box areas of a sorted run
        plo, phi = np.minimum.accumulate(lo, 0), np.maximum.accumulate(hi, 0)
        slo, shi = np.minimum.accumulate(lo[::-1], 0)[::-1], np.maximum.accumulate(hi[::-1], 0)[::-1]
        d = np.maximum(phi - plo, 0)
        la = 2 * (d[:, 0] * d[:, 1] + d[:, 1] * d[:, 2] + d[:, 2] * d[:, 0])
        d = np.maximum(shi - slo, 0)
        ra = 2 * (d[:, 0] * d[:, 1] + d[:, 1] * d[:, 2] + d[:, 2] * d[:, 0])
        return la, ra

    # BVH2 as nested tuples: ("leaf", lo, hi, n) | ("inner", lo, hi, left, right)
    def build(idx):
        lo, hi = tlo[idx].min(0), thi[idx].max(0)
        n = len(idx)
        if n <= 1:
            return ("leaf", lo, hi, n)
        best = (np.inf, None, None)
        for ax in range(3):
            o = idx[np.argsort(cen[idx, ax], kind="stable")]
            la, ra = sweep_areas(tlo[o], thi[o])
            k = np.arange(1, n)
            cost = la[:-1] * k + ra[1:] * (n - k)
            j = int(np.argmin(cost))
            if cost[j] < best[0]:
                best = (cost[j], o, j + 1)
        split_cost = cn + ct * best[0] / max(area(lo, hi), 1e-30)
        if n <= leaf_max and ct * n <= split_cost:
            return ("leaf", lo, hi, n)
        o, j = best[1], best[2]
        return ("inner", lo, hi, build(o[:j]), build(o[j:]))

    sys.setrecursionlimit(100000)
    root = build(np.arange(len(tlo)))
    inner_area, leaf_area, nn, nl = [0.0], [0.0], [0], [0]

    def collapse(node):
        if node[0] == "leaf":
            leaf_area[0] += area(node[1], node[2]) * node[3]
            nl[0] += 1
            return
        kids = [node[3], node[4]]
        while len(kids) < 4:
            cand = [(area(k[1], k[2]), i) for i, k in enumerate(kids) if k[0] == "inner"]
            if not cand:
                break
            _, i = max(cand)
            k = kids.pop(i)
            kids += [k[3], k[4]]
        inner_area[0] += area(node[1], node[2])
        nn[0] += 1
        for k in kids:
            collapse(k)

    collapse(root)
    ra = area(root[1], root[2])
    return inner_area[0] / ra, leaf_area[0] / ra, nn[0], nl[0]


def main():
    name = sys.argv[1]
    argv = ["--scene", name]
    a = bench.parse(argv)
    sc, view, proj, tmax, _ = bench.build_scene(a, pa)
    rt = pa.HipRaytracing(0)
    for label, flags in (("PLOC", PT_ACCEL_BVH4_PLOC), ("LBVH", PT_ACCEL_BVH4)):
        rt.buildBlas(sc, flags)
        nodes, tris = rt.debugReadBvh()
        nv, tt, nn, nl, nt = device_tree_cost(nodes, tris)
        print(f"{name} {label}: nodes {nn} leaves {nl} tris {nt}  SA cost: {nv:.3f} wide nodes + {tt:.3f} triangle tests per ray", flush=True)
    tlo = np.minimum(np.minimum(tris["v0"], tris["v1"]), tris["v2"]).astype(np.float64)
    thi = np.maximum(np.maximum(tris["v0"], tris["v1"]), tris["v2"]).astype(np.float64)
    for leaf_max in (2, 4):
        nv, tt, nn, nl = sah_tree_cost(tlo, thi, leaf_max)
        print(f"{name} sweep SAH (leaf <= {leaf_max}): nodes {nn} leaves {nl}  SA cost: {nv:.3f} wide nodes + {tt:.3f} triangle tests per ray", flush=True)
    rt.cleanup()


main()
