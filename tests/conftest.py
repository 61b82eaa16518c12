import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CASA = os.path.join(ROOT, "assets", "casa_obj.glb")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return os.path.exists("/dev/kfd")


@pytest.fixture(scope="session")
def casa():
    from raytracing_amd import scene
    return scene.casa_scene(CASA)


@pytest.fixture(scope="session")
def small_sky():
    from raytracing_amd import scene
    return scene.procedural_sky(256, 128)


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def gpu_ctx():
    from raytracing_amd import device
    ctx = device.Context(0)
    yield ctx
    ctx.destroy()


def random_rays(lo, hi, n, seed):
    from raytracing_amd import scene
    return scene.random_rays(lo, hi, n, seed)


def _soup(rng, n, scale=1.0, offset=(0, 0, 0)):
    t = np.zeros((n, 24), np.float32)
    c = rng.uniform(-1, 1, (n, 1, 3)) * scale
    t[:, :9] = (c + rng.normal(0, 0.05 * scale, (n, 3, 3)) + np.asarray(offset, np.float32)).reshape(n, 9)
    t[:, 9:18] = np.tile([0, 0, 1], 3)
    return t


def hostile_scene():
    """Scales from 1e-6 to 1e6, a mesh 1e7 from the origin, coplanar triangles, every triangle of
    one mesh twice (exact ties in t), rotated / non-uniformly scaled / mirrored instances that
    overlap, a one- and a two-triangle mesh."""
    import numpy as np
    from raytracing_amd import scene
    rng = np.random.default_rng(17)
    b = scene.SceneBuilder()
    meshes = [_soup(rng, 400, 1e-6), _soup(rng, 400, 1e6), _soup(rng, 300, 1.0, (1e7, -1e7, 3e6)), _soup(rng, 1), _soup(rng, 2)]
    flat = _soup(rng, 200)
    flat[:, 2:9:3] = 0.25
    flat[100:] = flat[:100]
    meshes.append(flat)
    dup = _soup(rng, 300, 2.0)
    meshes.append(np.concatenate([dup, dup]))
    ids = [b.add_mesh(m) for m in meshes]
    for i in ids:
        b.add_instance(i, material=dict(type=1 + (i % 2), albedo=(0.7, 0.6, 0.5), roughness=0.3, metallic=0.5, ior=1.4))
    b.add_instance(ids[6], scene.transform_matrix((0.5, 0.2, -0.3), (30, 45, 60), (1.0, 0.2, 3.0)))
    b.add_instance(ids[6], scene.transform_matrix((-0.5, 0, 0), (0, 0, 0), (-1.0, 1.0, 1.0)))
    b.add_instance(ids[0], scene.transform_matrix((3, 0, 0), (10, 20, 30), (1e3, 1e3, 1e3)))
    return b.finish()
