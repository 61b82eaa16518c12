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
    assert p.stats()["reference_order_forced"] == 0
    ref, cnt = oracle_mod.OracleScene(casa).intersect(rays)
    assert 0.2 < (ref["triangle"] >= 0).mean() < 0.3
    _check(hits, ref)
    p.destroy()
    sb.release()


@pytest.mark.parametrize("fmt", [0, 1])
def test_both_wide_node_formats(casa, oracle_mod, gpu_ctx, fmt):
    """pt_pass_set_wide_nodes: full-precision and quantised four-wide nodes give the same hits;
    switching the format on a live pass rebuilds the mesh level."""
    from raytracing_amd import device
    lo, hi = casa.world_box()
    rays = random_rays(lo, hi, 1 << 19, 23 + fmt)
    sb = device.SceneBuffers(gpu_ctx, casa)
    p = device.PathTracingPass(gpu_ctx, 16, 16)
    p.bind_scene(sb)
    p.set_wide_nodes(fmt)
    hits = device.trace_rays(p, rays)
    ref, _ = oracle_mod.OracleScene(casa).intersect(rays)
    _check(hits, ref)
    builds = p.stats()["mesh_level_builds"]
    p.set_wide_nodes(1 - fmt)
    hits = device.trace_rays(p, rays)
    _check(hits, ref)
    assert p.stats()["mesh_level_builds"] == builds + 1
    with pytest.raises(Exception):
        p.set_wide_nodes(2)
    p.destroy()
    sb.release()


@pytest.mark.parametrize("fmt", [0, 1])
def test_non_finite_and_huge_node_boxes(casa, oracle_mod, gpu_ctx, fmt):
    """Boxes with NaN, infinite or overflowing coordinates: the reference's min/max drop NaNs
    (pathtracing.glsl:208-221), so such a box passes more rays; results must still agree."""
    import copy
    from raytracing_amd import device
    sd = copy.copy(casa)
    nodes = casa.mesh_nodes.copy()
    inner = np.flatnonzero(nodes["tri"] < 0)
    rng = np.random.default_rng(3)
    pick = rng.choice(inner[19:], 400, replace=False)
    nodes["aa"][pick[:100], 0] = np.nan
    nodes["bb"][pick[100:200], 1] = np.inf
    nodes["aa"][pick[200:300], 2] = -3e38
    nodes["bb"][pick[200:300], 2] = 3e38
    nodes["bb"][pick[300:350], :3] = np.nan
    nodes["bb"][pick[350:], :3] -= 0.25 * (nodes["bb"][pick[350:], :3] - nodes["aa"][pick[350:], :3])  # shrunk: children stick out
    sd.mesh_nodes = nodes
    lo, hi = casa.world_box()
    rays = random_rays(lo, hi, 1 << 18, 41)
    sb = device.SceneBuffers(gpu_ctx, sd)
    p = device.PathTracingPass(gpu_ctx, 16, 16)
    p.bind_scene(sb)
    p.set_wide_nodes(fmt)
    hits = device.trace_rays(p, rays)
    ref, _ = oracle_mod.OracleScene(sd).intersect(rays)
    assert (ref["triangle"] >= 0).mean() > 0.1
    _check(hits, ref)
    # such buffers are traced in the reference's order: the ordered traversal relies on nesting
    st = p.stats()
    assert st["reference_order_forced"] == 1 and st["rays_retraced"] == 0
    p.destroy()
    sb.release()


def test_reference_order_mode_matches_too(casa, oracle_mod, gpu_ctx):
    """pt_pass_set_traversal(1): every ray in the reference's visiting order."""
    from raytracing_amd import device
    lo, hi = casa.world_box()
    rays = random_rays(lo, hi, 1 << 18, 11)
    sb = device.SceneBuffers(gpu_ctx, casa)
    p = device.PathTracingPass(gpu_ctx, 16, 16)
    p.bind_scene(sb)
    p.set_traversal(1)
    hits = device.trace_rays(p, rays)
    ref, _ = oracle_mod.OracleScene(casa).intersect(rays)
    _check(hits, ref)
    p.destroy()
    sb.release()


def test_degenerate_rays(casa, oracle_mod, gpu_ctx):
    """Axis-parallel directions (zero components: infinities and NaNs in the slab test),
    origins on box planes, rays starting on surfaces, unnormalised and tiny directions."""
    from raytracing_amd import device
    lo, hi = casa.world_box()
    rng = np.random.default_rng(5)
    n = 20000
    o = lo + (hi - lo) * rng.random((n, 3))
    d = np.zeros((n, 3))
    ax = rng.integers(0, 3, n)
    d[np.arange(n), ax] = rng.choice([-1.0, 1.0], n)
    two = rng.random(n) < 0.3                      # a second non-zero component on some
    d[two, (ax[two] + 1) % 3] = rng.standard_normal(two.sum())
    rays = np.concatenate([o, d], 1).astype(np.float32)
    rays[:200, :3] = lo                            # origins exactly on the scene box corner
    rays[200:400, 3:] *= 1e-3                      # short direction vectors: t scales up
    rays[400:600, 3:] *= 250.0
    sb = device.SceneBuffers(gpu_ctx, casa)
    p = device.PathTracingPass(gpu_ctx, 16, 16)
    p.bind_scene(sb)
    osc = oracle_mod.OracleScene(casa)
    ref, _ = osc.intersect(rays)
    _check(device.trace_rays(p, rays), ref)
    # second generation: start exactly on the surfaces just hit, bounce along the normal-ish axis
    h = ref["triangle"] >= 0
    pos = rays[h, :3] + ref["t"][h, None] * rays[h, 3:]
    d2 = rng.standard_normal((h.sum(), 3))
    rays2 = np.concatenate([pos, d2], 1).astype(np.float32)
    ref2, _ = osc.intersect(rays2)
    _check(device.trace_rays(p, rays2), ref2)
    # NaN / Inf / zero-vector rays (the reference's refract() returns vec3(0) and NaNs): both sides
    # finish NaN rays as misses without walking the scene; the others go through the slab arithmetic
    rays3 = rays[:4000].copy()
    for k in range(6):
        rays3[k::40, k] = np.nan
    rays3[5::40, 3:] = 0.0
    rays3[7::40, 3] = np.inf
    rays3[9::40, 4] = -np.inf
    rays3[11::40, 0] = np.inf
    for mode in (0, 1):
        p.set_traversal(mode)
        ref3, _ = osc.intersect(rays3)
        got3 = device.trace_rays(p, rays3)
        _check(got3, ref3)
        nan_rows = np.isnan(rays3).any(axis=1)
        assert nan_rows.sum() >= 500 and (got3["triangle"][nan_rows] == -1).all()
    p.destroy()
    sb.release()


@pytest.mark.parametrize("fmt", [0, 1])
def test_hostile_geometry(oracle_mod, gpu_ctx, fmt):
    """Scales from 1e-6 to 1e6, a mesh 1e7 from the origin (an ulp there is 1), coplanar triangles,
    every triangle twice (exact ties in t: the visiting order decides), rotated / non-uniformly
    scaled / mirrored instances that overlap, a one- and a two-triangle mesh.  Rays are aimed at
    the instances from nearby, so that most of them hit something."""
    from raytracing_amd import device
    from conftest import hostile_scene
    rng = np.random.default_rng(18)
    sd = hostile_scene()
    rays = []
    for k in range(len(sd.instances)):
        leaf = np.flatnonzero(sd.scene_nodes["inst"] == k)[0]
        lo, hi = sd.scene_nodes["aa"][leaf][:3].astype(np.float64), sd.scene_nodes["bb"][leaf][:3].astype(np.float64)
        ext = np.maximum(hi - lo, 1e-30)
        n = 20000
        target = lo + rng.uniform(0, 1, (n, 3)) * ext
        origin = (lo + hi) / 2 + rng.normal(0, 1, (n, 3)) * ext * 1.5
        d = target - origin
        d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-300)
        rays.append(np.concatenate([origin, d], 1))
    rays = np.ascontiguousarray(np.concatenate(rays), np.float32)
    sb = device.SceneBuffers(gpu_ctx, sd)
    p = device.PathTracingPass(gpu_ctx, 16, 16)
    p.bind_scene(sb)
    p.set_wide_nodes(fmt)
    hits = device.trace_rays(p, rays)
    ref, _ = oracle_mod.OracleScene(sd).intersect(rays)
    assert (ref["triangle"] >= 0).mean() > 0.2
    _check(hits, ref)
    st = p.stats()
    assert st["reference_order_forced"] == 0 and st["rays_retraced"] > 1000
    # the doubled mesh: both copies of a triangle are hit at the same t, the reference's order picks one
    dup_inst = ref["instance"] == 6
    assert dup_inst.sum() > 1000 and len(np.unique(ref["triangle"][dup_inst] >= sd.tri_count[:6].sum() + 300)) == 2
    p.set_traversal(1)
    _check(device.trace_rays(p, rays), ref)
    p.destroy()
    sb.release()


@pytest.mark.parametrize("release_shadow", [False, True])
def test_instances_rooted_inside_a_tree(casa, oracle_mod, gpu_ctx, monkeypatch, release_shadow):
    """meshIndex may name any node of the mesh BVH (pathtracing.glsl:254-264 just starts there):
    an inner node of another mesh's tree, and a single leaf.  release_shadow: the library gives the
    host copy of the triangle buffer back after every mesh-level re-layout (as it does for large
    scenes), so the re-layouts that follow -- roots inside a tree, the other node format -- read the
    device copy."""
    import copy
    if release_shadow:
        monkeypatch.setenv("RTA_SHADOW_RELEASE_MIN", "0")
    from raytracing_amd import device
    sd = copy.copy(casa)
    inst = casa.instances.copy()
    inner = np.flatnonzero(casa.mesh_nodes["tri"] < 0)
    leaf = np.flatnonzero(casa.mesh_nodes["tri"] >= 0)
    inst["meshIndex"][0] = inner[40]
    inst["meshIndex"][1] = leaf[len(leaf) // 2]
    sd.instances = inst
    lo, hi = casa.world_box()
    rays = random_rays(lo, hi, 1 << 18, 77)
    sb = device.SceneBuffers(gpu_ctx, sd)
    p = device.PathTracingPass(gpu_ctx, 16, 16)
    p.bind_scene(sb)
    for fmt in (1, 0):
        p.set_wide_nodes(fmt)
        hits = device.trace_rays(p, rays)
        ref, _ = oracle_mod.OracleScene(sd).intersect(rays)
        _check(hits, ref)
    assert p.stats()["reference_order_forced"] == 0
    p.destroy()
    sb.release()


def test_random_scene_sweep(oracle_mod, gpu_ctx):
    """Twelve seeded random scenes (1..5 meshes of 1..2000 triangles at magnitudes 1e-4..1e4,
    extra instances with random rotations, non-uniform and mirrored scales), rays aimed at the
    instances and rays from far outside: ids, t, u, v word for word in both node formats."""
    from raytracing_amd import device, scene
    from conftest import _soup
    p = device.PathTracingPass(gpu_ctx, 16, 16)
    total_hits = 0
    for seed in range(12):
        rng = np.random.default_rng(1000 + seed)
        b = scene.SceneBuilder()
        ids = []
        for _ in range(int(rng.integers(1, 6))):
            mag = 10.0 ** rng.uniform(-4, 4)
            ids.append(b.add_mesh(_soup(rng, int(rng.integers(1, 2001)), mag, rng.uniform(-3, 3, 3) * mag)))
            b.add_instance(ids[-1])
        for _ in range(int(rng.integers(0, 5))):
            s = 10.0 ** rng.uniform(-2, 2, 3) * rng.choice([-1.0, 1.0], 3)
            b.add_instance(int(rng.choice(ids)), scene.transform_matrix(tuple(rng.uniform(-2, 2, 3)), tuple(rng.uniform(0, 360, 3)), tuple(s)))
        sd = b.finish()
        rays = []
        for k in range(len(sd.instances)):
            leaf = np.flatnonzero(sd.scene_nodes["inst"] == k)[0]
            lo, hi = sd.scene_nodes["aa"][leaf][:3].astype(np.float64), sd.scene_nodes["bb"][leaf][:3].astype(np.float64)
            ext = np.maximum(hi - lo, 1e-30)
            n = 4000
            target = lo + rng.uniform(0, 1, (n, 3)) * ext
            origin = (lo + hi) / 2 + rng.normal(0, 1, (n, 3)) * ext * rng.choice([0.3, 1.5, 50.0], (n, 1))
            d = target - origin
            d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-300)
            rays.append(np.concatenate([origin, d], 1))
        rays = np.ascontiguousarray(np.concatenate(rays), np.float32)
        ref, _ = oracle_mod.OracleScene(sd).intersect(rays)
        total_hits += int((ref["triangle"] >= 0).sum())
        sb = device.SceneBuffers(gpu_ctx, sd)
        p.bind_scene(sb)
        for fmt in (0, 1):
            p.set_wide_nodes(fmt)
            _check(device.trace_rays(p, rays), ref)
        assert p.stats()["reference_order_forced"] == 0
        sb.release()
    assert total_hits > 20000
