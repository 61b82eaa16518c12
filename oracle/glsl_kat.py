#!/usr/bin/env python3
"""TEST INFRASTRUCTURE, build container only.  Makes tests/golden/glsl_bsdf_kat.npz: known-answer
vectors of the reference's BSDF library and Sobol sampler computed from the reference's OWN GLSL
text (shader/basic/{math,sobol,sobol_matrix}.glsl, shader/materials/*.glsl under /root/reference),
compiled as C++ against the glm headers the reference vendors (oracle/glsl_harness.cpp).

The shader text is read where it lies and never copied into the repository; the expanded
translation unit goes to oracle/_ref/glsl_materials.inc (git-ignored).  The only changes made to
the text, all mechanical and listed here:
  * `#include shader/...` lines are replaced by the named file (the reference's read_shader,
    src/tool/tool.cpp:42-68, does the same at run time);
  * parameter qualifiers: `out float` -> `float &`, `inout T` -> `T &`, `in T` -> `T`;
  * the direction-number table's storage qualifiers `const uniform` / `shared` -> `static const` / `static`.
Usage: python oracle/glsl_kat.py [--out tests/golden/glsl_bsdf_kat.npz]"""
import os
import re
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"


def expand(rel, seen=None):
    seen = seen if seen is not None else []
    assert rel not in seen, "include cycle"
    out = []
    for line in open(os.path.join(REF, rel), encoding="utf-8").read().split("\n"):
        m = re.match(r"\s*#include\s+(shader/\S+)", line)
        if m:
            out.append(expand(m.group(1), seen + [rel]))
        else:
            out.append(line)
    return "\n".join(out)


def translate(text):
    text = re.sub(r"\bout\s+float\b", "float &", text)
    text = re.sub(r"\binout\s+(\w+)\s+", r"\1 &", text)
    text = re.sub(r"\bin\s+(BSDFQueryRecord)\b", r"\1", text)
    text = text.replace("const uniform uint V", "static const uint V").replace("shared uint V", "static uint V")
    return text


def build():
    os.makedirs(os.path.join(HERE, "_ref"), exist_ok=True)
    unit = "\n".join(translate(expand(f)) for f in ("shader/basic/math.glsl", "shader/basic/sobol.glsl", "shader/materials/materials.glsl"))
    open(os.path.join(HERE, "_ref", "glsl_materials.inc"), "w", encoding="utf-8").write(unit)
    exe = os.path.join(HERE, "_ref", "glsl_harness")
    cmd = ["g++", "-std=c++17", "-O1", "-fsingle-precision-constant", "-ffp-contract=off", "-w", "-I" + os.path.join(REF, "include"),
           "-I" + HERE, os.path.join(HERE, "glsl_harness.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("the reference's GLSL did not compile as C++:\n" + r.stderr[-4000:])
    return exe


def unit(v):
    v = np.asarray(v, np.float64)
    return (v / np.linalg.norm(v)).astype(np.float32)


def make_queries(rng):
    """Material records and queries covering both BSDFs: reflection and transmission, wi on either
    side, grazing directions, the parameter ranges the loader and the synthetic scenes produce."""
    records = []
    mptrs = []

    def add(rec):
        mptrs.append(len(records))
        records.extend(rec)
    for alpha, metallic in ((0.001, 1.0), (0.05, 1.0), (0.35, 0.6), (0.8, 0.0), (1.0, 0.3)):
        add([1.0, *(0.2 + 0.75 * rng.random(3)), alpha, metallic, -1.0])
    n_cond = len(mptrs)
    for alpha, ior in ((0.02, 1.5), (0.2, 1.33), (0.3, 1.8), (0.6, 1.0001)):
        add([2.0, *(0.5 + 0.5 * rng.random(3)), alpha, 0.0, -1.0, ior])
    q = []
    for k, mp in enumerate(mptrs):
        diel = k >= n_cond
        for _ in range(220):
            wi = unit(rng.normal(size=3))
            if rng.random() < 0.15:
                wi = unit([rng.normal(), rng.normal(), 0.02 * rng.normal()])      # grazing
            if not diel and rng.random() < 0.8:
                wi[2] = abs(wi[2])
            wo = unit(rng.normal(size=3))
            if rng.random() < 0.5:
                # near the mirror direction, where the GGX lobe lives
                wo = unit(np.array([-wi[0], -wi[1], wi[2]]) + 0.3 * rng.normal(size=3))
            if diel and rng.random() < 0.4:
                wo = unit(-wi + 0.4 * rng.normal(size=3))                            # transmission side
            seed = int(rng.integers(1, 2 ** 32 - 1)) | 1
            dim = int(rng.integers(1, 16))
            for which in (0, 1, 2, 3):
                q.append((which, mp, *wi, *wo, seed, dim))
    for d in range(20):
        for i in (1, 2, 3, 0x80000000, 0xFFFFFFFF, 0x12345679, int(rng.integers(1, 2 ** 32 - 1))):
            q.append((4, 0, 0, 0, 1, 0, 0, 1, i, d))
    return np.array(records, np.float32), q


def main():
    out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else os.path.join(ROOT, "tests", "golden", "glsl_bsdf_kat.npz")
    exe = build()
    rng = np.random.default_rng(20261111)
    records, queries = make_queries(rng)
    text = "%d\n%s\n" % (len(records), " ".join("%.9g" % x for x in records))
    text += "\n".join("%d %d %.9g %.9g %.9g %.9g %.9g %.9g %d %d" % tuple(q) for q in queries) + "\n"
    r = subprocess.run([exe], input=text, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("glsl_harness failed: " + r.stderr[-2000:])
    lines = r.stdout.strip().split("\n")
    assert len(lines) == len(queries), (len(lines), len(queries))
    qa = np.array([q for q in queries if q[0] != 4], np.float64)
    res = np.array([[float(x) for x in l.split()] for l, q in zip(lines, queries) if q[0] != 4], np.float64)
    sob_q = np.array([(q[8], q[9]) for q in queries if q[0] == 4], np.uint32)
    sob_r = np.array([float(l) for l, q in zip(lines, queries) if q[0] == 4], np.float32)
    np.savez_compressed(out, records=records, which=qa[:, 0].astype(np.int32), mptr=qa[:, 1].astype(np.int32),
                        wi=qa[:, 2:5].astype(np.float32), wo=qa[:, 5:8].astype(np.float32), seed=qa[:, 8].astype(np.uint32),
                        dim=qa[:, 9].astype(np.uint32), value=res[:, 0:3].astype(np.float32), pdf=res[:, 3].astype(np.float32),
                        wo_out=res[:, 4:7].astype(np.float32), eta=res[:, 7].astype(np.float32), dim_after=res[:, 8].astype(np.uint32),
                        sobol_index_dim=sob_q, sobol_value=sob_r)
    print("wrote %s: %d BSDF queries, %d sampler values" % (out, len(qa), len(sob_q)))


if __name__ == "__main__":
    main()
