"""Builds raytracing_amd/libraytracing_amd.so in-tree with hipcc for gfx950.

Usage: python -m raytracing_amd.build [--force] [--variant NAME -DMACRO=VALUE ...]

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
    "host/png_writer.cpp",
    "host/glb_loader.cpp",
]
DEVICE_SOURCES = [
    "capi_device.hip",
    "capi_comm.hip",
    "device/trace_kernels.hip",
    "device/path_kernels.hip",
    "device/tail_kernels.hip",
    "device/post_kernels.hip",
    "device/denoise_kernels.hip",
]

# Per-source flags.  The traversal kernels lose to the SLP vectoriser: it pairs scalar fp32 operations into
# v_pk_* instructions and pays for each pair with register moves (measured on MI355X: 16 Mi random rays
# against casa 8.2 -> 9.1 Grays/s without it; path tracing unchanged).
import json as _json
EXTRA = _json.loads(os.environ.get("RTA_EXTRA_FLAGS", "null")) or {
    "device/trace_kernels.hip": ["-fno-slp-vectorize"],
}

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


def build(force=False, verbose=True, variant=None, defines=()):
    """variant/defines: a tuning build, e.g. variant="r48", defines=["-DRT_FAST_REFILL=48"], written
    to libraytracing_amd.<variant>.so next to the real one (objects under _obj/<variant>/)."""
    obj_dir = os.path.join(OBJ, variant) if variant else OBJ
    lib = os.path.join(HERE, "libraytracing_amd.%s.so" % variant) if variant else LIB
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = _hipcc()
    hdrs = _headers()
    jobs = []
    objs = []
    common = [*COMMON, *defines]
    for rel in HOST_SOURCES + DEVICE_SOURCES:
        src = os.path.join(CSRC, rel)
        if not os.path.exists(src):
            raise RuntimeError("missing source " + src)
        obj = os.path.join(obj_dir, rel.replace("/", "_") + ".o")
        objs.append(obj)
        if force or _newer(src, obj, hdrs):
            if rel.endswith(".hip"):
                cmd = [hipcc, "--offload-arch=gfx950", *common, *EXTRA.get(rel, []), "-c", src, "-o", obj]
            else:
                cmd = [hipcc, "-x", "c++", *common, "-c", src, "-o", obj]
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
    if jobs or force or not os.path.exists(lib):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs, "-lpthread", "-ldl"])
    return lib


def build_all(verbose=True):
    """The library, its small-stack twin (three stack entries per ray in LDS; tests/test_gpu_errors.py
    uses it to make rays run out of stack) and the C++ demo of the facade."""
    lib = build(verbose=verbose)
    build(variant="smallstack", defines=["-DRT_CORE_LDS_STACK=3", "-DRT_FAST_LDS_STACK=3"], verbose=False)
    build_facade_demo(verbose=verbose)
    return lib


def kernel_sources_digest():
    """sha256 over the sources that decide what the traversal and shading kernels execute and what
    they read: the device directory and the host re-layout.  Counter summaries under profiles/ carry
    the digest of the tree they were measured on; bench.py refuses to quote one whose digest differs."""
    import hashlib
    h = hashlib.sha256()
    files = []
    for sub in ("device",):
        for f in sorted(os.listdir(os.path.join(CSRC, sub))):
            if f.endswith((".h", ".hip", ".inc")):
                files.append(os.path.join(CSRC, sub, f))
    files += [os.path.join(CSRC, "host", "accel_layout.cpp"), os.path.join(CSRC, "host", "accel_layout.h")]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def build_facade_demo(verbose=True):
    """tests/cpp/facade_frame: the C++ program that drives include/raytracing_amd.hpp."""
    root = os.path.dirname(HERE)
    src = os.path.join(root, "tests", "cpp", "facade_frame.cpp")
    exe = os.path.join(root, "tests", "cpp", "facade_frame")
    hdrs = [os.path.join(root, "include", "raytracing_amd.hpp"), os.path.join(root, "include", "raytracing_amd.h"), LIB]
    if _newer(src, exe, hdrs):
        cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(root, "include"), src, "-L" + HERE, "-lraytracing_amd",
               "-Wl,-rpath,$ORIGIN/../../raytracing_amd", "-o", exe]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("compile failed:\n" + r.stdout + r.stderr)
    return exe


if __name__ == "__main__":
    argv = sys.argv[1:]
    var = argv[argv.index("--variant") + 1] if "--variant" in argv else None
    if var is None and "--force" not in argv:
        print(build_all())
    else:
        print(build(force="--force" in argv, variant=var, defines=[a for a in argv if a.startswith("-") and a not in ("--variant", "--force")]))
