"""pt_pass_intersect_noalloc (the entry point for callers that cannot allocate) against
pt_pass_intersect: the same closest hits, bit for bit, on two calls in a row, the second with
a smaller input (the scratch array keeps the first call's contents).  Also: its limits are
reported as errors."""
import numpy as np
import pytest

from conftest import random_rays

pytestmark = pytest.mark.gpu


def _same(a, b):
    for f in ("instance", "triangle"):
        assert np.array_equal(a[f], b[f]), f
    for f in ("t", "u", "v"):
        assert np.array_equal(a[f].view(np.uint32), b[f].view(np.uint32)), f


def test_noalloc_matches_intersect_twice(casa, gpu_ctx):
    from raytracing_amd import device
    lo, hi = casa.world_box()
    sb = device.SceneBuffers(gpu_ctx, casa)
    p = device.PathTracingPass(gpu_ctx, 16, 16)
    p.bind_scene(sb)
    n_big, n_small = 1 << 20, 77_777
    rays_big = random_rays(lo, hi, n_big, 101)
    rays_small = random_rays(lo, hi, n_small, 202)
    rb_big, rb_small = device.Buffer(gpu_ctx, rays_big), device.Buffer(gpu_ctx, rays_small)
    hb_ref = device.Buffer(gpu_ctx, nbytes=n_big * device.HIT_DTYPE.itemsize)
    hb_new = device.Buffer(gpu_ctx, nbytes=n_big * device.HIT_DTYPE.itemsize)

    # before anything laid the scene out, the entry point refuses (laying out allocates)
    with pytest.raises(Exception, match="laid out"):
        p.intersect_noalloc(rb_small, n_small, hb_new)

    p.intersect(rb_big, n_big, hb_ref)
    ref_big = hb_ref.download(np.uint8).view(device.HIT_DTYPE)[:n_big].copy()
    p.intersect(rb_small, n_small, hb_ref)
    ref_small = hb_ref.download(np.uint8).view(device.HIT_DTYPE)[:n_small].copy()
    assert 0.2 < (ref_big["triangle"] >= 0).mean() < 0.3
    retraced_before = p.stats()["rays_retraced"]

    hb_new.fill(0xFFFFFFFF, np.uint32)
    p.intersect_noalloc(rb_big, n_big, hb_new)
    _same(hb_new.download(np.uint8).view(device.HIT_DTYPE)[:n_big], ref_big)
    hb_new.fill(0xFFFFFFFF, np.uint32)
    p.intersect_noalloc(rb_small, n_small, hb_new)     # second call in a row, smaller input
    got = hb_new.download(np.uint8).view(device.HIT_DTYPE)
    _same(got[:n_small], ref_small)
    # nothing behind the n records was touched
    assert np.all(got[n_small:].view(np.uint32) == 0xFFFFFFFF)
    # the undecided rays went through the scratch array's redo list as they do through the allocated one
    assert p.stats()["rays_retraced"] >= retraced_before

    # reference-order mode shares the scratch too
    p.set_traversal(1)
    hb_new.fill(0xFFFFFFFF, np.uint32)
    p.intersect_noalloc(rb_small, n_small, hb_new)
    _same(hb_new.download(np.uint8).view(device.HIT_DTYPE)[:n_small], ref_small)
    p.set_traversal(0)

    # limits
    with pytest.raises(Exception, match="exceed the limit"):
        p.intersect_noalloc(rb_big, n_big + 1, hb_new)
    p.intersect_noalloc(rb_small, 0, hb_new)  # nothing to do is fine
    for b in (rb_big, rb_small, hb_ref, hb_new):
        b.release()
    p.destroy()
    sb.release()


def test_two_contexts_on_one_gpu_take_turns(casa, gpu_ctx):
    """Every context on a GPU shares that GPU's scratch array and has a stream of its own: calls issued back to back
    from two contexts, with no wait in between, must not touch each other's counters, redo list or spill entries."""
    from raytracing_amd import device
    lo, hi = casa.world_box()
    ctx2 = device.Context(0)
    state = []
    for ctx, n, seed in ((gpu_ctx, 1 << 20, 303), (ctx2, 200_000, 404)):
        sb = device.SceneBuffers(ctx, casa)
        p = device.PathTracingPass(ctx, 16, 16)
        p.bind_scene(sb)
        rb = device.Buffer(ctx, random_rays(lo, hi, n, seed))
        ref_b = device.Buffer(ctx, nbytes=n * device.HIT_DTYPE.itemsize)
        new_b = device.Buffer(ctx, nbytes=n * device.HIT_DTYPE.itemsize)
        p.intersect(rb, n, ref_b)
        ref = ref_b.download(np.uint8).view(device.HIT_DTYPE).copy()
        state.append((sb, p, rb, ref_b, new_b, n, ref))
    for rep in range(3):
        for sb, p, rb, ref_b, new_b, n, ref in state:     # queued on two streams, nobody waits
            p.intersect_noalloc(rb, n, new_b)
        for sb, p, rb, ref_b, new_b, n, ref in state:
            _same(new_b.download(np.uint8).view(device.HIT_DTYPE)[:n], ref)
    for sb, p, rb, ref_b, new_b, n, ref in state:
        for b in (rb, ref_b, new_b):
            b.release()
        p.destroy()
        sb.release()
    ctx2.destroy()
