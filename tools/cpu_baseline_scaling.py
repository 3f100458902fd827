"""Thread scaling of the optimised CPU baseline (oracle/cpd_baseline.c) on THIS host: pairs/s of both all-pairs passes for
several OMP_NUM_THREADS (each in a fresh process).  Explains the `cores` / `value` of bench.py's cpu_baseline line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, numpy as np
sys.path.insert(0, %r)
from oracle import c_baseline as cb
rng = np.random.default_rng(0)
N = 50000; m = int(sys.argv[1])
x = cb.soa(rng.normal(0, 50, (N, 3))); y = cb.soa(rng.normal(0, 50, (m, 3)))
cb.colsum(y[:, :64].copy(), x, 400.0)
t = time.perf_counter(); den = cb.colsum(y, x, 400.0); t1 = time.perf_counter() - t
t = time.perf_counter(); cb.rowstats(y, x, 400.0, 1.0 / (den + 1.0)); t2 = time.perf_counter() - t
print(cb.num_threads(), m * N / t1 / 1e9, m * N / t2 / 1e9)
''' % ROOT

if __name__ == "__main__":
    out = {"lscpu": subprocess.run("lscpu | grep -E 'Model name|^CPU\\(s\\)|Thread|Core|Socket|Flags' | cut -c1-200", shell=True,
                                   capture_output=True, text=True).stdout.splitlines(),
           "nproc": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)), "runs": []}
    for nt in [1, 8, 16, 32, 64, 128, 256]:
        if nt > 2 * (os.cpu_count() or 1):
            break
        env = dict(os.environ, OMP_NUM_THREADS=str(nt), OMP_PROC_BIND="false")
        rows = max(256, 96 * nt)
        r = subprocess.run([sys.executable, "-c", CHILD, str(rows)], env=env, capture_output=True, text=True)
        try:
            th, c, rs = r.stdout.split()
            out["runs"].append({"threads": int(th), "rows": rows, "colsum_Gpair_s": float(c), "rowstats_Gpair_s": float(rs)})
        except Exception:
            out["runs"].append({"threads": nt, "error": r.stderr[-300:]})
    print(json.dumps(out, indent=1))
