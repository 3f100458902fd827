"""How often does a Metropolis-Hastings step of the femur chain accept WITHOUT consuming the accept test's uniform draw (a > 0 in
`a > 0 || rnd.nextDouble() < exp(a)`), and where does the host time of a step go?  (why a batch of k steps per native
call cannot be exact.)   python tools/chain_probe.py      -> COUNTS {steps, a_pos, acc} + a cProfile of the chain"""
import cProfile, pstats, io, sys, os, math, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = ["x", "600", "0"]
sys.path.insert(0, ROOT)
from gingr_amd import sampling as sp
cnt = {"steps": 0, "a_pos": 0, "acc": 0}
orig_next = sp.MetropolisHastings.next
def next2(self, current, logger=None):
    proposal = self.generator.propose(current)
    currentP = self.evaluator.logValue(current)
    proposalP = self.evaluator.logValue(proposal)
    t = self.logTransitionRatio(current, proposal)
    a = proposalP - currentP - t
    cnt["steps"] += 1
    if a > 0.0:
        cnt["a_pos"] += 1
    if a > 0.0 or self.rnd.nextDouble() < math.exp(a):
        cnt["acc"] += 1
        if logger is not None:
            logger.accept(current, proposal, self.generator, self.evaluator)
        return proposal
    if logger is not None:
        logger.reject(current, proposal, self.generator, self.evaluator)
    return current
sp.MetropolisHastings.next = next2
pr = cProfile.Profile()
pr.enable()
exec(compile(open(os.path.join(ROOT, "tools", "bench_mh_chain.py")).read(), os.path.join(ROOT, "tools", "bench_mh_chain.py"), "exec"), {"__file__": os.path.join(ROOT, "tools", "bench_mh_chain.py"), "__name__": "__main__"})
pr.disable()
print("COUNTS", cnt)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats("gingr_amd|numpy|method|built-in", 60)
print(s.getvalue()[:12000])
