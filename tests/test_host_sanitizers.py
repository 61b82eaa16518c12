"""The host half of the library (BVH build incl. one large mesh on several threads, scene level,
traversal re-layout and its checker, image quantisation, the Radiance and glTF-binary readers on
truncated and damaged files) under AddressSanitizer + UndefinedBehaviorSanitizer, and its threaded parts
under ThreadSanitizer: tests/cpp/host_sanitize.cpp feeds it random and degenerate inputs; any
report makes the run fail.  Host code only (sanitizers are not available for GPU code here)."""
import glob
import os
import subprocess

import pytest

from conftest import ROOT

SOURCES = [os.path.join(ROOT, "tests", "cpp", "host_sanitize.cpp"), os.path.join(ROOT, "raytracing_amd", "csrc", "capi_host.cpp")] + \
    sorted(glob.glob(os.path.join(ROOT, "raytracing_amd", "csrc", "host", "*.cpp")))


@pytest.mark.parametrize("flags", ["-fsanitize=address,undefined -fno-sanitize-recover=undefined", "-fsanitize=thread"])
def test_host_code_under_sanitizers(tmp_path, flags):
    exe = str(tmp_path / "host_sanitize")
    cmd = ["g++", "-std=c++17", "-O1", "-g"] + flags.split() + ["-I" + os.path.join(ROOT, "include"),
                                                                  "-I" + os.path.join(ROOT, "raytracing_amd", "csrc")] + SOURCES + ["-lpthread", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    r = subprocess.run([exe, os.path.join(ROOT, "assets", "casa_obj.glb")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "host sanitize run ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    assert "runtime error" not in r.stderr and "Sanitizer" not in r.stderr, r.stderr[-4000:]
