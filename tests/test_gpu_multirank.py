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
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse", *common, "--dump", two],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    import json
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["steps"] == 3
    a, b = np.load(one), np.load(two)
    for name in ("motion", "depth", "normal", "instanceIndex"):
        assert np.array_equal(a[name].view(np.uint32), b[name].view(np.uint32)), name
    for name in ("directLum", "indirectLum", "albedo"):
        x, y = a[name].astype(np.float64), b[name].astype(np.float64)
        both_nan = np.isnan(x) & np.isnan(y)
        err = np.abs(x - y) / np.maximum(np.maximum(np.abs(x), np.abs(y)), 1e-3)
        assert np.all((err < 1e-5) | both_nan), (name, float(np.nanmax(err)))
