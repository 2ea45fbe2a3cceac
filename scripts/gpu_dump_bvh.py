import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import pathtracer_amd as pa
sc = pa.scenes.cornell_box()
rt = pa.HipRaytracing(0); rt.buildBlas(sc)
nodes, tris = rt.debugReadBvh()
for i, nd in enumerate(nodes):
    scale = np.array([nd["scale_x"], nd["scale_yz"][0], nd["scale_yz"][1]], dtype=np.float32)
    print("node", i, "origin", nd["origin"])
    for c in range(4):
        ref = int(nd["child"][c])
        if ref == 0xFFFFFFFE: continue
        lo = nd["origin"] + nd["qlo"][:, c] * scale; hi = nd["origin"] + nd["qhi"][:, c] * scale
        if ref & 0x80000000:
            f, n = ref & 0x0FFFFFFF, ((ref >> 28) & 7) + 1
            desc = "leaf prims " + str(tris["prim"][f:f + n].tolist())
        else: desc = "node " + str(ref)
        print("   ", desc, "lo", np.round(lo, 2), "hi", np.round(hi, 2))
