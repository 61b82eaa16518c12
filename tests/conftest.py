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
