"""CPU probe: greedy (largest child opened first) vs DP-optimal BVH2 -> BVH4 collapse, surface-area cost of node visits,
on a sweep-SAH BVH2 over the reference meshes (leaf <= 2 triangles)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.setrecursionlimit(100000)
z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "ref_seam.npz"))
def mesh(name):
    v, idx = z[f"{name}_v"].reshape(-1, 3), z[f"{name}_idx"].reshape(-1, 3)[:, 0].reshape(-1, 3)
    p = v[idx]
    return p.min(1).astype(np.float64), p.max(1).astype(np.float64)
def area(lo, hi):
    d = np.maximum(hi - lo, 0); return 2.0 * (d[0]*d[1] + d[1]*d[2] + d[2]*d[0])
def build(tlo, thi, leaf_max=2):
    cen = 0.5 * (tlo + thi)
    def rec(idx):
        lo, hi = tlo[idx].min(0), thi[idx].max(0); n = len(idx)
        if n <= leaf_max: return ("leaf", lo, hi, n)
        best = (np.inf, None, None)
        for ax in range(3):
            o = idx[np.argsort(cen[idx, ax], kind="stable")]
            plo, phi = np.minimum.accumulate(tlo[o], 0), np.maximum.accumulate(thi[o], 0)
            slo, shi = np.minimum.accumulate(tlo[o][::-1], 0)[::-1], np.maximum.accumulate(thi[o][::-1], 0)[::-1]
            d = np.maximum(phi - plo, 0); la = 2*(d[:,0]*d[:,1]+d[:,1]*d[:,2]+d[:,2]*d[:,0])
            d = np.maximum(shi - slo, 0); ra = 2*(d[:,0]*d[:,1]+d[:,1]*d[:,2]+d[:,2]*d[:,0])
            k = np.arange(1, n); cost = la[:-1]*k + ra[1:]*(n-k); j = int(np.argmin(cost))
            if cost[j] < best[0]: best = (cost[j], o, j+1)
        o, j = best[1], best[2]
        return ("inner", lo, hi, rec(o[:j]), rec(o[j:]))
    return rec(np.arange(len(tlo)))
def greedy(root):
    tot, n, fill = [0.0], [0], []
    def col(node):
        if node[0] == "leaf": return
        kids = [node[3], node[4]]
        while len(kids) < 4:
            cand = [(area(k[1], k[2]), i) for i, k in enumerate(kids) if k[0] == "inner"]
            if not cand: break
            _, i = max(cand); k = kids.pop(i); kids += [k[3], k[4]]
        tot[0] += area(node[1], node[2]); n[0] += 1; fill.append(len(kids))
        for k in kids: col(k)
    col(root); return tot[0], n[0], np.mean(fill)
def dp(root):
    # T[k] = min sum of wide-node areas when the subtree appears as at most k slots in its parent (k = 1..4)
    cnt = [0]
    def rec(node):
        if node[0] == "leaf": return [0.0, 0.0, 0.0, 0.0], [0, 0, 0, 0]
        (L, nl), (R, nr) = rec(node[3]), rec(node[4])
        T, N = [0.0]*4, [0]*4
        split = {}
        for k in (2, 3, 4):
            best = (np.inf, 0)
            for i in range(1, k):
                c = L[i-1] + R[k-i-1]
                if c < best[0]: best = (c, nl[i-1] + nr[k-i-1])
            split[k] = best
        T[0] = area(node[1], node[2]) + split[4][0]; N[0] = 1 + split[4][1]
        for k in (2, 3, 4):
            if T[k-2] <= split[k][0]: T[k-1], N[k-1] = T[k-2], N[k-2]
            else: T[k-1], N[k-1] = split[k]
        return T, N
    T, N = rec(root)
    return T[0], N[0]
for name in ("viking_room", "teapot"):
    tlo, thi = mesh(name)
    root = build(tlo, thi)
    ra = area(root[1], root[2])
    g, gn, gf = greedy(root)
    d, dn = dp(root)
    print(f"{name}: {len(tlo)} triangles | greedy collapse: {gn} nodes, {gf:.2f} children per node, SA cost {g/ra:.3f} | DP-optimal collapse: {dn} nodes, SA cost {d/ra:.3f} = {d/g:.3f} of greedy", flush=True)
