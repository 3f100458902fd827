"""Oracle-backed restatement of the six update phases for ONE row shard (test infrastructure).

Mirrors gingr_amd/csrc/fitter.hip phase by phase with numpy so that the sharding algebra -- which partial sums are
exchanged, in which order, and what is replicated -- can be exercised across real processes (gloo) without a GPU.
Segment layout = gingr_fitter_exchange: [den N] [G rp*rp | rhs rp | 8 scalars] [proj rp] [umeyama 24] [proj rp],
with rp = r here (no MFMA padding needed on the CPU).
"""
import numpy as np

from oracle import c_oracle as co
from oracle import gingr_oracle as go

EPS = 1e-5


class OracleShard:
    def __init__(self, model: go.PDM, target, begin, end, global_transform=go.RIGID_TRANSFORMS, step_length=1.0, w=0.0, lam=1.0):
        self.m, self.x = model, np.asarray(target, dtype=np.float64)
        self.b, self.e = begin, end
        self.gt, self.step, self.w, self.lam = global_transform, step_length, w, lam
        r, N = model.rank, self.x.shape[0]
        self.counts = [N, r * r + r + 8, r, 24, r]
        self.offsets = list(np.cumsum([0] + self.counts[:-1]))
        self.xch = np.zeros(sum(self.counts))
        rows = slice(3 * begin, 3 * end)
        self.Q0 = model.U[rows] * np.sqrt(model.lam)[None, :]         # local rows of Q0
        self.ref, self.mean = model.ref[begin:end], model.mean[begin:end]
        self.c0 = model.ref.mean(0)
        self.S_local = self.Q0.T @ self.Q0                             # all-reduced once -> Binv
        self.Binv = None

    def seg(self, k):
        return self.xch[self.offsets[k]: self.offsets[k] + self.counts[k]]

    def finalize(self, S_total):
        self.Binv = np.linalg.inv(S_total / EPS + np.eye(self.m.rank))

    def set_state(self, st: go.State):
        self.st = st
        self.fit = go.model_instance_shape_pose_scale(self.m, st)[self.b:self.e]

    def phase(self, ph):
        m, st, r = self.m, self.st, self.m.rank
        M_total, N = m.M, self.x.shape[0]
        R = st.rotation()
        if ph == 0:
            self.seg(0)[:] = co.cpd_colsum_partial(self.fit, self.x, st.sigma2, 0, self.fit.shape[0])
        elif ph == 1:
            c = go.cpd_outlier_constant(M_total, N, st.sigma2, self.w)
            colsum = self.seg(0).copy()
            den = colsum + c
            Pt1 = colsum / den
            P1, PX = co.cpd_rowstats_partial(self.fit, self.x, st.sigma2, den, 0, self.fit.shape[0])
            yhat = self.fit + (PX * (1.0 / P1)[:, None] - self.fit)
            wgt = 1.0 / (st.sigma2 * self.lam * (1.0 / P1))
            e = wgt[:, None] * ((yhat - st.center - st.translation) @ R - (self.ref - st.center) - self.mean)
            w3 = np.repeat(wgt, 3)
            G = self.Q0.T @ (self.Q0 * w3[:, None])
            rhs = self.Q0.T @ e.reshape(-1)
            s = self.seg(1)
            s[: r * r] = G.reshape(-1)
            s[r * r: r * r + r] = rhs
            xpx = float(Pt1 @ (self.x ** 2).sum(1)) if self.b == 0 else 0.0   # replicated quantity: counted once
            s[r * r + r:] = [P1.sum(), xpx, float((self.fit * PX).sum()), float(P1 @ (self.fit ** 2).sum(1)), 0, 0, 0, 0]
        elif ph == 2:
            s = self.seg(1)
            G, rhs = s[: r * r].reshape(r, r), s[r * r: r * r + r]
            self.a = np.linalg.solve(np.eye(r) + G, rhs)
            self.seg(2)[:] = self.Q0.T @ (self.Q0 @ self.a)
        elif ph == 3:
            alpha1 = self.Binv @ (self.seg(2) / EPS)
            self.alpha_c = st.alpha + (alpha1 - st.alpha) * self.step
            inst = self.ref + self.mean + (self.Q0 @ self.alpha_c).reshape(-1, 3)
            self.newshape = (inst - st.center) @ R.T + st.center + st.translation
            cur0 = self.ref + self.mean + (self.Q0 @ st.alpha).reshape(-1, 3)
            xt, yt = cur0 - self.c0, self.newshape - self.c0
            s = self.seg(3)
            s[:] = 0
            s[0:3], s[3:6] = xt.sum(0), yt.sum(0)
            s[6:15] = (yt.T @ xt).reshape(-1)
            s[15] = (xt ** 2).sum()
        elif ph == 4:
            s, n = self.seg(3), float(M_total)
            if self.gt == go.NO_TRANSFORMS:
                self.R2, self.t2, self.s2 = np.eye(3), np.zeros(3), 1.0
            else:
                mux, muy = s[0:3] / n, s[3:6] / n
                Sxy = s[6:15].reshape(3, 3) / n - np.outer(muy, mux)
                sig2x = s[15] / n - mux @ mux
                U, D, Vt = np.linalg.svd(Sxy)
                S = np.eye(3)
                if np.linalg.det(Sxy) < 0:
                    S[2, 2] = -1
                Rr = U @ S @ Vt
                c = float(np.trace(np.diag(D) @ S) / sig2x) if self.gt == go.SIMILARITY_TRANSFORMS else 1.0
                self.t2 = (muy + self.c0) - c * (Rr @ (mux + self.c0))
                self.R2 = go.euler_to_rot(*go.rot_to_euler(Rr))
                self.s2 = c
            e = (self.newshape - self.t2) @ self.R2 - self.ref - self.mean
            self.seg(4)[:] = self.Q0.T @ e.reshape(-1)
        elif ph == 5:
            alpha = self.Binv @ (self.seg(4) / EPS)
            sc = self.seg(1)[r * r + r:]
            s2n = (sc[1] - 2 * sc[2] + sc[3]) / (sc[0] * 3.0)
            new = go.State(alpha=alpha, euler=go.rot_to_euler(self.R2), center=np.zeros(3), translation=self.t2, scale=self.s2,
                           sigma2=float(s2n), fit=np.zeros((m.M, 3)), iteration=st.iteration + 1,
                           global_transformation=st.global_transformation, step_length=st.step_length)
            self.set_state(new)
