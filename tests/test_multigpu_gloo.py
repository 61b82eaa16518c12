"""The N > 1 path (sample slicing + one sum all-reduce + broadcast of the last-write
outputs) with world_size 2 over gloo on the CPU.  The per-rank renderer is the oracle
(the HIP pass needs a GPU); the slicing / combining code under test is
raytracing_amd.multigpu, the same functions bench.py drives over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import CASA, ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total_spp, w, h, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import oracle
    from raytracing_amd import multigpu, scene
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sd = scene.casa_scene(CASA, n_threads=1)
    sky = scene.procedural_sky(128, 64)
    osc = oracle.OracleScene(sd, sky)
    v2w = scene.casa_camera()
    jit = scene.mt19937_jitters(total_spp)
    acc = {k: np.zeros((h, w, c), np.float32) for k, c in multigpu.CHANNELS.items()}
    for s in multigpu.sample_slice(total_spp, rank, world):
        prm = scene.make_params(w, h, v2w, sky, spp=s, frame=1, jitter=jit[s - 1], max_depth=2, accum_mode=1)
        acc, _ = osc.render(prm, acc, n_threads=1)
        if s == 1:
            pass  # sample 1 overwrites (currentspp == 1), the buffers were zero anyway
    tensors = {k: torch.from_numpy(v.reshape(-1).copy()) for k, v in acc.items()}
    multigpu.combine(dist, tensors, total_spp, world, torch)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **{k: t.numpy() for k, t in tensors.items()})
    dist.destroy_process_group()


def test_sample_slices_partition_the_frame():
    from raytracing_amd import multigpu
    for world in (1, 2, 3, 8):
        for spp in (1, 5, 64):
            got = sorted(s for r in range(world) for s in multigpu.sample_slice(spp, r, world))
            assert got == list(range(1, spp + 1))
            assert spp in multigpu.sample_slice(spp, multigpu.owner_of_last_sample(spp, world), world)


def test_two_ranks_equal_one_rank(tmp_path, oracle_mod):
    import torch.multiprocessing as mp
    from raytracing_amd import multigpu, scene
    total_spp, w, h = 5, 24, 14
    port = _free_port()
    mp.spawn(_worker, args=(2, port, total_spp, w, h, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    for k in multigpu.CHANNELS:
        assert np.array_equal(r0[k], r1[k]), k        # every rank ends with the same frame
    # single-process reference: the reference's own running mean over the same 5 samples
    sd = scene.casa_scene(CASA, n_threads=1)
    sky = scene.procedural_sky(128, 64)
    osc = oracle_mod.OracleScene(sd, sky)
    jit = scene.mt19937_jitters(total_spp)
    acc = None
    for s in range(1, total_spp + 1):
        prm = scene.make_params(w, h, scene.casa_camera(), sky, spp=s, frame=1, jitter=jit[s - 1], max_depth=2)
        acc, _ = osc.render(prm, acc, n_threads=2)
    for k in multigpu.ACCUMULATED:
        a, b = r0[k].astype(np.float64), acc[k].reshape(-1).astype(np.float64)
        assert np.allclose(a, b, rtol=1e-5, atol=1e-6), k      # differs by fp32 summation order only
    for k in multigpu.LAST_WRITE:
        assert np.array_equal(r0[k], acc[k].reshape(-1)), k    # what the last sample left behind
