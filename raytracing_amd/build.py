"""Builds raytracing_amd/libraytracing_amd.so in-tree with hipcc for gfx950.

Usage: python -m raytracing_amd.build [--force]

Host sources are compiled as plain C++ (no device pass); .hip sources are compiled for
--offload-arch=gfx950.  All translation units use -ffp-contract=off: scene arrays and
closest-hit results are specified bit-exactly in fp32, and a fused multiply-add changes
them.  hipcc cross-compiles without a GPU, so this runs anywhere ROCm is installed.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libraytracing_amd.so")

HOST_SOURCES = [
    "capi_host.cpp",
    "host/bvh_builder.cpp",
    "host/scene_host.cpp",
    "host/sky_host.cpp",
    "host/accel_layout.cpp",
]
DEVICE_SOURCES = [
    "capi_device.hip",
    "device/trace_kernels.hip",
    "device/path_kernels.hip",
]

COMMON = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-result"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.sep not in c or os.path.exists(c)):
            return c
    raise RuntimeError("hipcc not found")


def _newer(src, dst, extra=()):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(s) > t for s in (src, *extra))


def _headers():
    out = [os.path.join(HERE, "..", "include", "raytracing_amd.h")]
    for root, _, files in os.walk(CSRC):
        out += [os.path.join(root, f) for f in files if f.endswith(".h")]
    return out


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    hdrs = _headers()
    jobs = []
    objs = []
    for rel in HOST_SOURCES + DEVICE_SOURCES:
        src = os.path.join(CSRC, rel)
        if not os.path.exists(src):
            raise RuntimeError("missing source " + src)
        obj = os.path.join(OBJ, rel.replace("/", "_") + ".o")
        objs.append(obj)
        if force or _newer(src, obj, hdrs):
            if rel.endswith(".hip"):
                cmd = [hipcc, "--offload-arch=gfx950", *COMMON, "-c", src, "-o", obj]
            else:
                cmd = [hipcc, "-x", "c++", *COMMON, "-c", src, "-o", obj]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("compile failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr, flush=True)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or force or not os.path.exists(LIB):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-lpthread"])
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
