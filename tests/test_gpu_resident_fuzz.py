"""Seeded random shapes / channel counts / flavours / run lengths: the resident-plane kernels against the per-launch flows
(tools/fuzz_resident.py: Perona-Malik planes identical byte for byte, CSV level sets within 1e-9 and the same iteration count)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_resident_kernels_against_the_per_launch_flows_on_random_shapes():
    env = dict(os.environ, CASES="16", SEED="3", MAXDIM="520")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_resident.py")], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "cases 16 bad 0" in r.stdout
