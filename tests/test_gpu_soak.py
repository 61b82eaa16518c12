"""A few seconds of tools/soak.py: random sequences of settings on one pass object (draws in flight, tail
threshold, traversal mode, node format, material grouping, MAX_DEPTH 0..16, sampling switches, lights on and
off, resolutions with edge tiles down to one pixel), every frame compared with the oracle bit for bit.
Catches state carried from one configuration into the next."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 12])
def test_random_setting_sequences_stay_bit_equal(seed):
    env = dict(os.environ, SOAK_SEED=str(seed))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py"), "6"], capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0 and "soak ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
