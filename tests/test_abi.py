"""CPU: the C-ABI library loads and exports every symbol include/pt_hip.h declares; struct layouts
are the reference's (Vertex 44 B, Material 24 B).  No compute calls (no GPU here)."""
import ctypes
import os
import re
import subprocess

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "pt_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(pa):
    from pathtracer_amd import _lib
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 40
    for n in names:
        assert hasattr(L, n), f"libpt_hip.so does not export {n}"
    # the Python binding covers the whole header, nothing more, nothing less
    assert sorted(_lib.SYMBOLS) == names


def test_header_is_plain_c_and_layouts_match_reference(tmp_path):
    src = tmp_path / "abi_check.c"
    src.write_text('#include "pt_hip.h"\n'
                   "_Static_assert(sizeof(PtVertex) == 44, \"Vertex is 44 bytes (main.cpp:60-65)\");\n"
                   "_Static_assert(sizeof(PtMaterial) == 24, \"Material is 24 bytes (main.cpp:54-58)\");\n"
                   "_Static_assert(sizeof(PtHit) == 16, \"hit record is 16 bytes\");\n"
                   "int main(void) { return 0; }\n")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-fsyntax-only",
                           str(src)])


def test_missing_library_fails_loudly(monkeypatch, pa):
    from pathtracer_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libpt_hip.so")
    try:
        _lib.lib()
    except RuntimeError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("loading a missing extension must raise")


def test_product_never_imports_the_oracle():
    """Source level: no product file imports, includes or names the oracle outside comments.  Binary level: libpt_hip.so and
    pt_render neither export nor import an oracle symbol (pto_*) and do not depend on liboracle*.so."""
    pkg = os.path.join(ROOT, "pathtracer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower() or all(
                    ("import oracle" not in line and "liboracle" not in line and "pt_oracle" not in line
                     and "obj_oracle" not in line) or line.lstrip().startswith(("//", "#", "*"))
                    for line in text.splitlines()), f"{f} references the oracle"
    from pathtracer_amd import _lib
    for binary in (_lib.LIB_PATH, os.path.join(pkg, "pt_render")):
        if not os.path.exists(binary):
            continue
        syms = subprocess.run(["nm", "-D", binary], capture_output=True, text=True, check=True).stdout
        assert not re.search(r"\bpto_", syms), f"{binary} links an oracle symbol"
        needed = subprocess.run(["readelf", "-d", binary], capture_output=True, text=True, check=True).stdout
        assert "oracle" not in needed.lower(), f"{binary} depends on an oracle library"


def test_create_without_a_gpu_fails_cleanly(pa):
    """On a machine without a HIP device pt_create returns NULL and says why (no crash, no CPU fallback); with a device it
    returns a context.  Context-free host-seam calls work either way."""
    from pathtracer_amd import _lib
    L = _lib.lib()
    ctx = L.pt_create(0)
    if ctx:
        L.pt_destroy(ctx)
    else:
        msg = L.pt_last_error(None).decode()
        assert "pt_create" in msg and len(msg) > 12
    view, proj = pa.reference_camera(800, 600)
    assert view.shape == (16,) and abs(float(view[14]) + 3.5) < 1e-6  # eye at z = 3.5 (main.cpp:1469)
