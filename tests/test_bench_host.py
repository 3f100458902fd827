"""Host-side pieces of bench.py that need no GPU: the environment handed to child runs, the lookup of tracked profile artefacts
(rocprof averages, PMC traffic) for the workload at hand, and the wide-rank plan's arithmetic as DESIGN.md states it."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_child_runs_do_not_inherit_the_launchers_rendezvous(monkeypatch):
    b = _bench()
    for k, v in {"RANK": "0", "WORLD_SIZE": "2", "MASTER_PORT": "1234", "TORCHELASTIC_USE_AGENT_STORE": "True", "TORCHELASTIC_RUN_ID": "x",
                 "GINGR_BENCH_SHARED_DEVICE": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}.items():
        monkeypatch.setenv(k, v)
    env = b._child_env()
    assert not any(k.startswith("TORCHELASTIC_") for k in env) and "RANK" not in env and "WORLD_SIZE" not in env and "MASTER_PORT" not in env
    assert env["GINGR_BENCH_SHARED_DEVICE"] == "1" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"     # what the child does need stays


def test_tracked_profiles_are_found_for_the_metric_workload_only():
    b = _bench()
    ms, src = b.load_rocprof_avg_ms("cpd_rowstats_kernel", 50000, 100)
    assert ms is not None and 1.0 < ms < 1.5 and src.startswith("profiles/r")
    ms256, src256 = b.load_rocprof_avg_ms("gram_wide_kernel", 50000, 256)
    assert ms256 is not None and 0.1 < ms256 < 0.3 and "r256" in src256
    assert b.load_rocprof_avg_ms("cpd_rowstats_kernel", 15000, 100) == (None, None)            # another workload: never borrowed
    tr, src = b.load_pmc_traffic("cpd_rowstats_kernel", 50000, 100)
    assert tr is not None and 4e7 < tr < 8e7 and "pmc_traffic" in src
    tr256, src256 = b.load_pmc_traffic("gram_wide_kernel", 50000, 256)
    assert tr256 is not None and 3e8 < tr256 < 5e8 and "pmc_wide_r256" in src256
    assert b.load_pmc_traffic("cpd_rowstats_kernel", 50000, 64) == (None, None)


def test_wide_gram_plan_as_documented():
    """The tile dealing of gp_wide.hip restated: T tiles per wave <= 17, parts = workgroups per slab, every tile exactly once."""
    for nt in range(8, 33):
        total = nt * (nt + 1) // 2
        parts = -(-total // (8 * 17))
        tpp = -(-total // parts)
        T = -(-tpp // 8)
        assert T <= 17 and parts * 8 * T >= total
        seen = []
        for p in range(parts):
            pend = min(total, (p + 1) * tpp)
            for w in range(8):
                g0 = p * tpp + w * T
                seen += list(range(g0, g0 + max(0, min(T, pend - g0))))
        assert seen == list(range(total)), nt
        assert (nt, T, parts) != (16, 17, 1) or total == 136


def test_a_hung_child_run_is_stopped_with_everything_it_started(monkeypatch, tmp_path):
    """A child launcher whose own child keeps running (and keeps the output open) must not hold the headline line back: the whole
    session is stopped at the time-out and the budget of all child runs together is one."""
    import sys
    import time
    import pytest
    b = _bench()
    monkeypatch.setenv("GINGR_BENCH_CHILD_BUDGET_S", "24")
    b._CHILD_DEADLINE[0] = None
    rc, out, err = b._run_child([sys.executable, "-c", "import sys; print('{\"a\": 1}'); print('note', file=sys.stderr)"], dict(os.environ), 60.0)
    assert rc == 0 and out.strip() == '{"a": 1}' and err.strip() == "note"
    pidfile = tmp_path / "grandchild.pid"
    hang = ("import subprocess, sys, time; p = subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(600)']); "
            f"open({str(pidfile)!r}, 'w').write(str(p.pid)); time.sleep(600)")
    t0 = time.monotonic()
    with pytest.raises(TimeoutError):
        b._run_child([sys.executable, "-c", hang], dict(os.environ), 600.0)    # capped by what is left of the 24 s budget
    assert time.monotonic() - t0 < 40
    gpid = int(pidfile.read_text())
    for _ in range(50):
        try:
            os.kill(gpid, 0)
        except ProcessLookupError:
            break
        time.sleep(0.1)
    else:
        raise AssertionError("the hung child's own child survived the time-out")
    with pytest.raises(TimeoutError):                                           # nothing left: refused at once
        b._run_child([sys.executable, "-c", "print(1)"], dict(os.environ), 60.0)
    b._CHILD_DEADLINE[0] = None
