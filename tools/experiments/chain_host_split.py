"""Where the ~50 us of Python between two native Metropolis-Hastings steps go (femur chain of tools/bench_mh_chain.py): wall-clock
shares of the pieces of MetropolisHastings.next and of the run loop, measured with perf_counter around them (600 steps)."""
import os, sys, time, math
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.argv = ["x", "600", "0"]
from gingr_amd import sampling as sp
from gingr_amd import api
T = {}
def add(k, dt):
    T[k] = T.get(k, 0.0) + dt
orig_mh = api.GingrAlgorithm._mh_step
def mh_step(self, *a, **kw):
    t0 = time.perf_counter()
    lib_call = self._lib.gingr_fitter_mh_step
    def timed(*x):
        t = time.perf_counter()
        r = lib_call(*x)
        add("native call", time.perf_counter() - t)
        return r
    self._lib.__dict__["gingr_fitter_mh_step"] = timed
    try:
        return orig_mh(self, *a, **kw)
    finally:
        del self._lib.__dict__["gingr_fitter_mh_step"]
        add("_mh_step total", time.perf_counter() - t0)
api.GingrAlgorithm._mh_step = mh_step
def next2(self, current, logger=None):
    t0 = time.perf_counter()
    proposal = self.generator.propose(current)
    t1 = time.perf_counter()
    currentP = self.evaluator.logValue(current)
    proposalP = self.evaluator.logValue(proposal)
    t2 = time.perf_counter()
    t = self.logTransitionRatio(current, proposal)
    t3 = time.perf_counter()
    a = proposalP - currentP - t
    acc = a > 0.0 or self.rnd.nextDouble() < math.exp(a)
    if logger is not None:
        (logger.accept if acc else logger.reject)(current, proposal, self.generator, self.evaluator)
    t4 = time.perf_counter()
    add("propose (incl. native)", t1 - t0); add("logValue x2", t2 - t1); add("logTransitionRatio", t3 - t2); add("accept + logger", t4 - t3)
    add("next total", t4 - t0)
    return proposal if acc else current
sp.MetropolisHastings.next = next2
t0 = time.perf_counter()
exec(compile(open(os.path.join(ROOT, "tools", "bench_mh_chain.py")).read(), os.path.join(ROOT, "tools", "bench_mh_chain.py"), "exec"),
     {"__file__": os.path.join(ROOT, "tools", "bench_mh_chain.py"), "__name__": "__main__"})
n = 599
for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
    print(f"{k:28s} {1e6 * v / n:8.1f} us per step")
