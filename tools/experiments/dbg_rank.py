import sys, numpy as np
sys.path.insert(0, '.')
import gingr_amd as ga
from oracle import gingr_oracle as go
def rel(a,b): return float(np.linalg.norm(np.asarray(a)-np.asarray(b))/max(np.linalg.norm(b),1e-300))
ctx = ga.Context(0)
for (M,N,rank) in [(700,650,112),(700,650,120),(700,650,127),(700,650,128),(700,650,129),(700,650,144),(800,700,200)]:
    rng = np.random.default_rng(M+rank)
    ref = rng.normal(0,30,(M,3)); U,_ = np.linalg.qr(rng.normal(0,1,(3*M,rank))); lam=np.sort(rng.uniform(1,400,rank))[::-1].copy()
    mo = go.PDM(ref=ref, mean=rng.normal(0,0.1,(M,3)), U=U, lam=lam)
    target = rng.normal(0,30,(N,3))
    algo = ga.CpdRegistration(ctx)
    state = algo.createInitialState(ga.PointDistributionModel(mo.ref,mo.mean,mo.U,mo.lam), target, ga.CpdConfiguration(maxIterations=10,w=0.2,initialSigma=400.0))
    st = go.initial_state(mo, 400.0)
    out=[]
    for _ in range(3):
        state = algo.update(state); st = go.cpd_update(mo, target, st, w=0.2)
        out.append((state.general.status, st.status, rel(state.general.fit, st.fit), rel(state.general.modelParameters.shape, st.alpha), abs(state.general.sigma2-st.sigma2)/st.sigma2))
    print(rank, out)
    algo.close()
