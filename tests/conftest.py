import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))  # tests are allowed to use the oracle (the checker)

GOLDEN = os.path.join(ROOT, "tests", "golden")
REFERENCE_MODELS = "/root/reference/models"  # only in the build container; CPU tests skip without it


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def bits(a):
    """uint32 view of a float32 array for bit-exact comparisons."""
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="session")
def O():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def pa():
    import pathtracer_amd
    return pathtracer_amd


@pytest.fixture(scope="session")
def cornell(pa):
    return pa.scenes.cornell_box()


@pytest.fixture(scope="session")
def ocornell(O, cornell):
    return O.OracleScene(cornell.vertices, cornell.indices, cornell.materials, cornell.mat_indices)


@pytest.fixture()
def rt(pa):
    """A fresh context on GPU 0 (GPU tests only)."""
    r = pa.HipRaytracing(0)
    yield r
    r.cleanup()


def reference_mesh_arrays(name):
    """tinyobj::LoadObj's output for the reference's models/<name>.obj, as dumped by oracle/_ref (the reference's own vendored
    tinyobjloader) into tests/golden/ref_seam.npz: (vertices, normals, texcoords, index_t triples, material ids, materials)."""
    z = np.load(os.path.join(GOLDEN, "ref_seam.npz"))
    return tuple(z[f"{name}_{k}"] for k in ("v", "vn", "vt", "idx", "mid", "mat"))


@pytest.fixture(scope="session")
def ref_mesh(pa):
    """name -> Scene: the reference's mesh through the product's own loadModel half (pt_scene_from_tinyobj).  Travels to the
    GPU box as data (the fixture), where /root/reference does not exist."""
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = pa.scene_from_tinyobj(*reference_mesh_arrays(name))
        return cache[name]
    return get
