"""CPU: the C-ABI library loads (no GPU needed) and exports every function include/rolo_hip.h declares."""
import ctypes
import os
import re

import pytest

from rolo_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header="rolo_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rolo_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    names = declared_functions()
    assert len(names) >= 40
    assert sorted(_lib.SYMBOLS) == names
    assert sorted(_lib.FUSION_SYMBOLS) == declared_functions("rolo_fusion.h")


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        pytest.fail("rolo_amd/librolo_hip.so not built: run `python -m rolo_amd.build`")
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_functions() + declared_functions("rolo_fusion.h"):
        assert hasattr(L, name), name
    _lib.lib()  # binds argtypes for all of them


def test_no_gpu_is_a_loud_error_not_a_fallback():
    L = _lib.lib()
    if L.rolo_device_count() > 0:
        pytest.skip("a GPU is visible")
    h = ctypes.c_void_p()
    rc = L.rolo_ctx_create(0, ctypes.byref(h))
    assert rc == -6 and b"no HIP device" in L.rolo_last_error()
    with pytest.raises(_lib.RoloError):
        from rolo_amd.rotvgicp import RotVGICP
        RotVGICP()


@pytest.mark.parametrize("demo", ["shim_demo", "nodes_demo"])
def test_cpp_host_mirrors_compile_and_link(demo):
    """The header-only C++ host mirrors — fast_gicp::RotVGICP (include/rot_vgicp_hip.hpp) and the node cores
    (include/rolo_nodes_hip.hpp) — build with plain g++ against the C ABI."""
    import subprocess, tempfile
    out = os.path.join(tempfile.mkdtemp(), demo)
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", demo + ".cpp"), "-o", out,
           "-L", os.path.join(ROOT, "rolo_amd"), "-lrolo_hip", "-Wl,-rpath," + os.path.join(ROOT, "rolo_amd"), "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
