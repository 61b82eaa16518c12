"""GPU parity: closest-hit IDs must be bit-exact with the oracle (CPU restatement of
intersect_sceneBVH, shader/pathtracing.glsl:292-316); t, u, v come out of the same fp32
expressions, so they are compared bit for bit as well (north_star asks 1e-5 relative)."""
import numpy as np
import pytest

from conftest import random_rays

pytestmark = pytest.mark.gpu


def _check(hits, ref):
    assert np.array_equal(hits["triangle"], ref["triangle"])
    assert np.array_equal(hits["instance"], ref["instance"])
    assert np.array_equal(hits["t"].view(np.uint32), ref["t"].view(np.uint32))
    h = ref["triangle"] >= 0
    assert np.array_equal(hits["u"][h].view(np.uint32), ref["u"][h].view(np.uint32))
    assert np.array_equal(hits["v"][h].view(np.uint32), ref["v"][h].view(np.uint32))


def test_casa_random_rays(casa, oracle_mod, gpu_ctx):
    from raytracing_amd import device
    lo, hi = casa.world_box()
    rays = random_rays(lo, hi, 1 << 20, 7)
    sb = device.SceneBuffers(gpu_ctx, casa)
    p = device.PathTracingPass(gpu_ctx, 16, 16)
    p.bind_scene(sb)
    hits = device.trace_rays(p, rays)
    p.stats()
    ref, cnt = oracle_mod.OracleScene(casa).intersect(rays)
    assert 0.2 < (ref["triangle"] >= 0).mean() < 0.3
    _check(hits, ref)
    p.destroy()
    sb.release()
