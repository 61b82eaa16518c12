"""bench.py's N > 1 control flow with real GPU renders: two ranks share the one GPU of the
test box (--rehearse: gloo backend, collectives staged through the host), each renders its
slice of the samples as sums, the slices are combined, and the frame must equal the frame one
rank renders alone (running mean) up to fp32 summation order; last-write outputs identical."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_on_one_gpu_equal_one_rank(tmp_path):
    common = ["--steps", "3", "--warmup", "1", "--width", "192", "--height", "108", "--depth", "3", "--no-cpu-baseline", "--no-profile"]
    one = str(tmp_path / "one.npz")
    two = str(tmp_path / "two.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    # 6 samples in both runs: 1 rank x 6 steps, 2 ranks x 3 steps
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", *common[:1], "6", *common[2:], "--dump", one],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    # (no launcher: bench.py --gpus 2 starts its two ranks itself)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse", "--scaling", "weak", *common, "--dump", two],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    import json
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak"
    assert out["allreduce_ms"] > 0 and len(out["per_rank"]) == 2 and all(r["samples"] == 3 and r["draw_ms"] > 0 for r in out["per_rank"])
    # the same job cut the other way rides along in the line: 3 samples in all over the two ranks
    o = out["other_scaling"]
    assert o["scaling"] == "strong" and o["total_spp"] == 3 and o["value"] > 0 and len(o["draw_ms_per_rank"]) == 2
    # strong scaling (the default with more than one rank): the same 6 samples, now as --steps 6 cut over the two ranks
    three = str(tmp_path / "three.npz")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse",
                        *common[:1], "6", *common[2:], "--dump", three], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    out3 = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out3["scaling"] == "strong" and out3["config"]["total_spp"] == 6
    assert out3["other_scaling"]["scaling"] == "weak" and out3["other_scaling"]["total_spp"] == 12
    c = np.load(three)
    b2 = np.load(two)
    for name in b2.files:
        assert np.array_equal(b2[name].view(np.uint32), c[name].view(np.uint32)), name
    # a world size that does not match --gpus is an error, not a silent single-GPU run
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True,
                       timeout=120, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
    a, b = np.load(one), np.load(two)
    for name in ("motion", "depth", "normal", "instanceIndex"):
        assert np.array_equal(a[name].view(np.uint32), b[name].view(np.uint32)), name
    for name in ("directLum", "indirectLum", "albedo"):
        x, y = a[name].astype(np.float64), b[name].astype(np.float64)
        both_nan = np.isnan(x) & np.isnan(y)
        err = np.abs(x - y) / np.maximum(np.maximum(np.abs(x), np.abs(y)), 1e-3)
        assert np.all((err < 1e-5) | both_nan), (name, float(np.nanmax(err)))


def test_capi_allreduce_world_of_one(gpu_ctx, casa, small_sky):
    """pt_comm_* / pt_pass_allreduce over RCCL with the one GPU of the test box as a world of one
    rank: sums of 4 samples / 4 must equal the running mean of the same samples up to fp32
    summation order, and the last-write outputs are untouched.  (More than one rank per GPU is
    refused by RCCL, so N > 1 runs only where there are N GPUs: bench.py --gpus N.  That leg --
    pt_comm_create with world > 1, pt_comm_create_all, pt_pass_allreduce_all -- has not run on hardware from this test.)"""
    from raytracing_amd import _capi, device, multigpu, scene
    try:
        uid = device.Comm.unique_id()
    except _capi.PtError as e:
        if "librccl" in str(e) or "RCCL" in str(e):
            pytest.skip("RCCL is not installed on this box: %s" % e)
        raise
    w, h, spp = 256, 144, 4
    sb = device.SceneBuffers(gpu_ctx, casa, small_sky)
    p = device.PathTracingPass(gpu_ctx, w, h)
    p.bind_scene(sb)
    jit = scene.mt19937_jitters(spp)

    def frame(accum_mode):
        for s in range(1, spp + 1):
            p.set_params(scene.make_params(w, h, scene.casa_camera(), small_sky, spp=s, max_depth=3, jitter=jit[s - 1],
                                           accum_mode=accum_mode))
            p.draw()

    frame(0)
    mean = p.download()
    for name in multigpu.ACCUMULATED:
        p.outputs[name].fill(0.0)
    frame(1)
    comm = device.Comm(gpu_ctx, uid, 0, 1)
    multigpu.allreduce_outputs_capi(comm, p, spp)        # queued behind the draws, no wait in between
    got = p.download()
    for name in multigpu.LAST_WRITE:
        assert np.array_equal(mean[name].view(np.uint32), got[name].view(np.uint32)), name
    for name in multigpu.ACCUMULATED:
        x, y = mean[name].astype(np.float64), got[name].astype(np.float64)
        err = np.abs(x - y) / np.maximum(np.maximum(np.abs(x), np.abs(y)), 1e-3)
        assert np.all((err < 1e-5) | (np.isnan(x) & np.isnan(y))), (name, float(np.nanmax(err)))
    with pytest.raises(_capi.PtError):
        comm.allreduce(p, 0, 0)
    with pytest.raises(_capi.PtError):
        comm.allreduce(p, spp, 1)
    comm.destroy()
    # the single-process form (one host thread, several contexts: the reference's main.cpp shape) with its one context
    import ctypes as C
    ctxs = (C.c_void_p * 1)(gpu_ctx.h)
    comms = (C.c_void_p * 1)()
    _capi.check(_capi.lib.pt_comm_create_all(ctxs, 1, comms))
    passes = (C.c_void_p * 1)(p.h)
    for name in multigpu.ACCUMULATED:
        p.outputs[name].fill(0.0)
    frame(1)
    _capi.check(_capi.lib.pt_pass_allreduce_all(passes, comms, 1, spp, 0))
    got_all = p.download()
    for name in got:
        assert np.array_equal(got_all[name].view(np.uint32), got[name].view(np.uint32)), name
    # its arguments are checked before anything is queued
    for bad in (lambda: _capi.lib.pt_pass_allreduce_all(passes, comms, 1, 0, 0), lambda: _capi.lib.pt_pass_allreduce_all(passes, comms, 1, spp, 1),
                lambda: _capi.lib.pt_pass_allreduce_all(passes, comms, 0, spp, 0), lambda: _capi.lib.pt_pass_allreduce_all(None, comms, 1, spp, 0),
                lambda: _capi.lib.pt_pass_allreduce_all((C.c_void_p * 1)(None), comms, 1, spp, 0)):
        with pytest.raises(_capi.PtError):
            _capi.check(bad())
    two = (C.c_void_p * 2)(gpu_ctx.h, gpu_ctx.h)
    with pytest.raises(_capi.PtError, match="one rank per GPU"):
        _capi.check(_capi.lib.pt_comm_create_all(two, 2, (C.c_void_p * 2)()))
    _capi.check(_capi.lib.pt_comm_destroy(comms[0]))
    p.destroy()
    sb.release()
