"""The shared library loads without a GPU and exports every function include/*.h declares."""
import ctypes
import os
import re

from conftest import ROOT


def declared_functions():
    names = []
    inc = os.path.join(ROOT, "include")
    for f in sorted(os.listdir(inc)):
        if not f.endswith(".h"):
            continue
        text = open(os.path.join(inc, f)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names += re.findall(r"\b(pt_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_every_declared_symbol_is_exported():
    from raytracing_amd import _capi
    lib = ctypes.CDLL(_capi.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "library does not export " + n


def test_binding_covers_the_header():
    from raytracing_amd import _capi
    assert sorted(_capi.SIGNATURES) == declared_functions()


def test_version_and_error_channel():
    from raytracing_amd import _capi
    assert b"gfx950" in _capi.lib.pt_version()
    out = (ctypes.c_float * 16)()
    rc = _capi.lib.pt_host_mat4_inverse(None, out)
    assert rc == -1 and b"null" in _capi.lib.pt_last_error()


def test_param_struct_sizes_match_between_binding_and_oracle():
    from raytracing_amd import _capi
    # pt_params: 2+1+1+16+4+1+2+1+1+2+16+4+1+1+7 = 60 words
    assert ctypes.sizeof(_capi.PtParams) == 60 * 4
    assert ctypes.sizeof(_capi.PtStats) == 15 * 8


def test_cpp_facade_compiles():
    """include/raytracing_amd.hpp (the reference's class names over the C ABI) and its C++ user
    compile as C++17 with warnings as errors; running it needs a GPU (tests/test_gpu_facade.py)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-I" + os.path.join(root, "include"),
                        os.path.join(root, "tests", "cpp", "facade_frame.cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
