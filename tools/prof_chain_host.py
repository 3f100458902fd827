"""cProfile of the Metropolis-Hastings chain's host side (tools/bench_mh_chain.py 300): the functions of gingr_amd/ by own time."""
import cProfile, pstats, sys, os, runpy, io
sys.argv = ["bench_mh_chain.py", "300"]
import torch  # noqa: F401  (outside the profile)
import numpy  # noqa: F401
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tools", "bench_mh_chain.py"), run_name="__main__")
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
st = pstats.Stats(pr, stream=s)
st.sort_stats("tottime").print_stats("gingr_amd|ctypes|numpy", 40)
print(s.getvalue()[:12000])
