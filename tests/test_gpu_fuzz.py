"""Randomised parity sweeps kept short enough for the GPU suite (the long versions are tests/fuzz_cpd_stats.py and
tests/fuzz_parity.py): odd sizes around the chunk / tile / launch-round boundaries of the all-pairs passes, sigma2 from the dense
regime down to heavy exact-zero culling, and whole updates with random transforms / landmarks / step lengths; random kernel models through the set-up path (tests/fuzz_model_setup.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,args", [("fuzz_cpd_stats.py", ["24", "11"]), ("fuzz_parity.py", ["30", "5"]),
                                         ("fuzz_model_setup.py", ["15", "3"]), ("fuzz_surface.py", ["12", "21"])])
def test_fuzz_sweep(script, args):
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", script)] + args, env=env, cwd=ROOT, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert "MISMATCH" not in r.stdout and "worst " in r.stdout.splitlines()[-1]
