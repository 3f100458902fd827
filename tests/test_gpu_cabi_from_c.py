"""The boundary used the way the reference's host would use it: a plain C program (gcc, no Python, no torch in the process) linked
against libgingr_hip.so.  Its output is compared with the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import gingr_oracle as go

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_consumer(tmp_path):
    exe = str(tmp_path / "cabi_driver")
    libdir = os.path.join(ROOT, "gingr_amd")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "cabi_driver.c"),
                           "-o", exe, "-L", libdir, "-lgingr_hip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    rng = np.random.default_rng(3)
    q = rng.normal(0, 5, (300, 3))
    t = q[rng.permutation(300)[:250]] + rng.normal(0, 0.3, (250, 3))
    text = f"{q.shape[0]} {t.shape[0]}\n" + "\n".join(" ".join(repr(float(v)) for v in row) for row in np.concatenate([q, t]))
    out = subprocess.run([exe], input=text, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    idx = np.array([int(v) for v in lines[:300]])
    oidx, od2, omean = go.icp_closest_point(q, t)
    assert np.array_equal(idx, oidx)
    Np, s2n, mean = (float(v) for v in lines[300].split())
    st = go.cpd_stats_dense(q, t, 4.0, 0.1)
    assert abs(Np - st.Np) < 1e-9 * st.Np and abs(s2n - st.sigma2_next) < 1e-9 * abs(st.sigma2_next)
    assert abs(mean - omean) < 1e-12 * max(1.0, omean)
