import cProfile, pstats, sys, os, runpy, io
sys.argv = ["bench_mh_chain.py", "300"]
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tools", "bench_mh_chain.py"), run_name="__main__")
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
