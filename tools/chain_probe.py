import cProfile, pstats, io, sys, os, math, time
sys.argv = ["x", "600", "0"]
sys.path.insert(0, "/root/repo")
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
exec(open("/root/repo/tools/bench_mh_chain.py").read())
pr.disable()
print("COUNTS", cnt)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
