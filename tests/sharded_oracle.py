"""Oracle-backed restatement of the three update phases for ONE row shard (test infrastructure).

Mirrors gingr_amd/csrc/fitter.hip phase by phase with numpy so that the sharding algebra -- which partial sums are
exchanged, in which order, and what is replicated -- can be exercised across real processes (gloo) without a GPU.
Segment layout = gingr_fitter_exchange: [den N] [G rp*rp | rhs rp | 8 scalars | Q0^T e rp] [full fit 3 M_total] [reversal sums 4 M_total], with rp = r
here (no MFMA padding on the CPU).  Flavours as in gingr_fitter_update_sharded_async: 0 CPD, 1 ICP with the point-cloud closest
point, 2 ICP with the surface correspondence (phase 3 = GINGR_PHASE_GATHER writes the shard's rows of the fit into segment 2,
whose sum all-reduce is the all-gather); z = the replicated standard-normal draw of posterior.sample(); logpdf_* = the transition
density of gingr_fitter_posterior_logpdf_sharded (Q0^T e rides in the tail of segment 1).  Phase 2 is the "moment form": everything after the posterior solve is computed from one-off moments of the
basis (summed over all shards once, like gingr_model_gram_exchange + gingr_model_finalize) with no further exchange.
"""
import numpy as np

from oracle import c_oracle as co
from oracle import gingr_oracle as go

EPS = 1e-5


class OracleShard:
    def __init__(self, model: go.PDM, target, begin, end, global_transform=go.RIGID_TRANSFORMS, step_length=1.0, w=0.0, lam=1.0,
                 flavour=0, icp=(1.0, 1.0, 1), tmpl_tris=None, tgt_tris=None, reversed_direction=False):
        self.m, self.x = model, np.asarray(target, dtype=np.float64)
        self.b, self.e = begin, end
        self.gt, self.step, self.w, self.lam = global_transform, step_length, w, lam
        self.flavour, self.icp, self.tmpl_tris, self.tgt_tris = flavour, icp, tmpl_tris, tgt_tris
        self.reversed = bool(reversed_direction)                     # ICP.scala:46-48; gathers the fit like flavour 2
        self.z = None                                                 # set for ONE sampled proposal, the same on every shard
        r, N = model.rank, self.x.shape[0]
        # segments 0, 1 (gingr_fitter_exchange), 2 = GINGR_SEGMENT_FULLFIT, 3 = GINGR_SEGMENT_REVSUM ([4][M_total] sums of the reversed direction)
        self.counts = [N, r * r + r + 8 + r, 3 * model.M, 4 * model.M]
        self.offsets = [0, N, N + self.counts[1], N + self.counts[1] + 3 * model.M]
        self.xch = np.zeros(sum(self.counts))
        rows = slice(3 * begin, 3 * end)
        self.Q0 = model.U[rows] * np.sqrt(model.lam)[None, :]         # local rows of Q0
        self.ref, self.mean = model.ref[begin:end], model.mean[begin:end]
        self.c0 = model.ref.mean(0)
        # local moments (MomentLayout of gp.h): S_tot, S[d][e], V[d][e], W[d]
        Q3 = self.Q0.reshape(-1, 3, r)
        pt = self.ref + self.mean - self.c0
        self.mom_local = np.concatenate([
            (self.Q0.T @ self.Q0).reshape(-1),
            np.einsum("idk,iel->dekl", Q3, Q3).reshape(-1),
            np.einsum("idk,ie->dek", Q3, pt).reshape(-1),
            Q3.sum(0).reshape(-1)])
        pt_all = model.ref + model.mean - self.c0                       # host moments use the FULL model
        self.Pp, self.Ps = pt_all.T @ pt_all, pt_all.sum(0)
        self.Binv = None

    def seg(self, k):
        return self.xch[self.offsets[k]: self.offsets[k] + self.counts[k]]

    def finalize(self, mom_total):
        r = self.m.rank
        o = 0
        self.S_tot = mom_total[o:o + r * r].reshape(r, r); o += r * r
        self.S = mom_total[o:o + 9 * r * r].reshape(3, 3, r, r); o += 9 * r * r
        self.V = mom_total[o:o + 9 * r].reshape(3, 3, r); o += 9 * r
        self.W = mom_total[o:o + 3 * r].reshape(3, r)
        self.Binv = np.linalg.inv(self.S_tot / EPS + np.eye(r))

    def set_state(self, st: go.State):
        self.st = st
        self.fit = go.model_instance_shape_pose_scale(self.m, st)[self.b:self.e]

    def phase(self, ph):
        m, st, r = self.m, self.st, self.m.rank
        M_total, N = m.M, self.x.shape[0]
        R = st.rotation()
        if ph == 3:      # GINGR_PHASE_GATHER: own rows into zeros, [3][M_total] planes in the original vertex order
            full = self.seg(2).reshape(3, M_total)
            full[:] = 0.0
            full[:, self.b:self.e] = self.fit.T
        elif ph == 0 and self.reversed and self.flavour != 0:
            # the reversed correspondence, sharded by QUERY range (round 5): the target's vertices look for their match on the GATHERED
            # template (any shard's rows); this shard answers for its index range of the target -- the fraction of the cloud its rows are
            # of the template -- and leaves, per template vertex of the whole template, the sum of the accepted target points and their
            # number (segment 3, summed across the shards between phases 0 and 1)
            full = self.seg(2).reshape(3, M_total).T.copy()
            method = "TriangularClosestPoint" if self.flavour == 2 else "PointcloudClosestPoint"
            tid, pts, w = go.correspondence_reversal(full, self.tmpl_tris, self.x, self.tgt_tris, method)
            q0, q1 = N * self.b // M_total, N * self.e // M_total
            mine = np.zeros(N, dtype=bool)
            mine[q0:q1] = True
            keep = (w == 1.0) & mine
            sums = self.seg(3).reshape(4, M_total)
            sums[:] = 0.0
            np.add.at(sums[0], tid[keep], pts[keep, 0])
            np.add.at(sums[1], tid[keep], pts[keep, 1])
            np.add.at(sums[2], tid[keep], pts[keep, 2])
            np.add.at(sums[3], tid[keep], 1.0)
        elif ph == 0 and self.flavour == 1:
            idx, _, _ = go.icp_closest_point(self.fit, self.x)          # the shard's own rows against the replicated target
            self.obs, self.acc = self.x[idx], np.ones(self.fit.shape[0])
        elif ph == 0 and self.flavour == 2:
            self.obs, self.acc = self._surface_rows(self.seg(2).reshape(3, M_total).T.copy())
        elif ph == 0:
            self.seg(0)[:] = co.cpd_colsum_partial(self.fit, self.x, st.sigma2, 0, self.fit.shape[0])
        elif ph == 1 and self.flavour != 0 and self.reversed:
            # the totals are in place: this shard's rows of them -- k accepted targets of a vertex = one observation of their mean with
            # k-fold precision (algebraically the reference's k observations)
            sums = self.seg(3).reshape(4, M_total)[:, self.b:self.e]
            rows = np.flatnonzero(sums[3] > 0)
            k = sums[3, rows]
            pts = (sums[:3, rows] / k).T
            Q3 = self.Q0.reshape(-1, 3, r)[rows]
            wgt = k / st.sigma2
            e = wgt[:, None] * ((pts - st.center - st.translation) @ R - (self.ref[rows] - st.center) - self.mean[rows])
            s = self.seg(1)
            s[: r * r] = np.einsum("i,idk,idl->kl", wgt, Q3, Q3).reshape(-1)
            s[r * r: r * r + r] = np.einsum("idk,id->k", Q3, e)
            s[r * r + r:] = 0.0
        elif ph == 1 and self.flavour != 0:
            # uniform weight 1 / sigma2 on the accepted correspondences (ICP.scala:90-92), nothing for the sigma^2 update
            wgt = self.acc / st.sigma2
            e = wgt[:, None] * ((self.obs - st.center - st.translation) @ R - (self.ref - st.center) - self.mean)
            s = self.seg(1)
            s[: r * r] = (self.Q0.T @ (self.Q0 * np.repeat(wgt, 3)[:, None])).reshape(-1)
            s[r * r: r * r + r] = self.Q0.T @ e.reshape(-1)
            s[r * r + r:] = 0.0
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
            s[r * r + r: r * r + r + 8] = [P1.sum(), xpx, float((self.fit * PX).sum()), float(P1 @ (self.fit ** 2).sum(1)), 0, 0, 0, 0]
            s[r * r + r + 8:] = 0.0
        elif ph == 2:
            s = self.seg(1)
            G, rhs = s[: r * r].reshape(r, r), s[r * r: r * r + r]
            a = np.linalg.solve(np.eye(r) + G, rhs)
            if self.z is not None:                                           # posterior.sample(): a + L^-T z, replicated
                a = a + np.linalg.solve(np.linalg.cholesky(np.eye(r) + G).T, np.asarray(self.z, dtype=np.float64))
            alpha = st.alpha
            alpha1 = self.Binv @ (self.S_tot @ a) / EPS                      # Q^T (Q a) = S_tot a
            ac = alpha + (alpha1 - alpha) * self.step
            n = float(M_total)
            # u~ = p~ + Q0 alpha, v~ = p~ + Q0 alpha_c, newshape - c0 = R v~ + g~
            su = self.Ps + self.W @ alpha
            sv = self.Ps + self.W @ ac
            Mvu = self.Pp + np.einsum("bdk,k->db", self.V, alpha) + np.einsum("dbk,k->db", self.V, ac) \
                + np.einsum("k,dbkl,l->db", ac, self.S, alpha)
            gt = R @ (self.c0 - st.center) + st.center + st.translation - self.c0
            sy = R @ sv + n * gt
            Syx = R @ Mvu + np.outer(gt, su)
            sxx = float(np.trace(self.Pp) + 2 * sum(self.V[d, d] @ alpha for d in range(3))
                        + sum(alpha @ self.S[d, d] @ alpha for d in range(3)))
            if self.gt == go.NO_TRANSFORMS:
                R2, t2, s2 = np.eye(3), np.zeros(3), 1.0
            else:
                mux, muy = su / n, sy / n
                Sxy = Syx / n - np.outer(muy, mux)
                sig2x = sxx / n - mux @ mux
                U, D, Vt = np.linalg.svd(Sxy)
                Sg = np.eye(3)
                if np.linalg.det(Sxy) < 0:
                    Sg[2, 2] = -1
                Rr = U @ Sg @ Vt
                s2 = float(np.trace(np.diag(D) @ Sg) / sig2x) if self.gt == go.SIMILARITY_TRANSFORMS else 1.0
                t2 = (muy + self.c0) - s2 * (Rr @ (mux + self.c0))
                R2 = go.euler_to_rot(*go.rot_to_euler(Rr))
            # e_i = R2^T (newshape_i - t2) - p_i = (B - I) p~_i + B Q0_i alpha_c + h
            B = R2.T @ R
            h = R2.T @ (gt + self.c0 - t2) - self.c0
            proj = np.einsum("de,dek->k", B - np.eye(3), self.V) + np.einsum("de,dekl,l->k", B, self.S, ac) + h @ self.W
            alpha_new = self.Binv @ proj / EPS
            sc = s[r * r + r: r * r + r + 8]
            if self.flavour == 0:
                s2n = (sc[1] - 2 * sc[2] + sc[3]) / (sc[0] * 3.0)
            else:
                s2n = go.icp_update_sigma2(st.sigma2, *self.icp)
            new = go.State(alpha=alpha_new, euler=go.rot_to_euler(R2), center=np.zeros(3), translation=t2, scale=s2,
                           sigma2=float(s2n), fit=np.zeros((m.M, 3)), iteration=st.iteration + 1,
                           global_transformation=st.global_transformation, step_length=st.step_length)
            self.set_state(new)

    def _surface_rows(self, full_fit):
        """ClosestPointTriangleMesh3D.closestPointCorrespondence (ClosestPointRegistrator.scala:75-100) for the shard's rows: the
        queries are the shard's own fit vertices, the template mesh (vertex normals, self-intersection test) is the GATHERED fit."""
        import math
        tgt, tt, mt = self.x, self.tgt_tris, self.tmpl_tris
        cp, _ = go.mesh_closest_point(self.fit, tgt, tt)
        nn_idx, _, _ = go.icp_closest_point(cp, tgt)
        bnd = go.boundary_vertices(tgt.shape[0], tt)
        n_tmpl, n_tgt = go.vertex_normals(full_fit, mt)[self.b:self.e], go.vertex_normals(tgt, tt)
        w = np.ones(self.fit.shape[0])
        for i in range(self.fit.shape[0]):
            j, p = int(nn_idx[i]), full_fit[self.b + i]
            if bnd[j] or float(n_tmpl[i] @ n_tgt[j]) < 0:
                w[i] = 0.0
                continue
            v = p - cp[i]
            ips = go.line_mesh_intersections(p, v, full_fit, mt)
            keep = np.any(ips != p, axis=1)
            if keep.any():
                dd = ips[keep] - p
                if math.sqrt(float((dd * dd).sum(1).min())) < math.sqrt(float(v @ v)):
                    w[i] = 0.0
        return cp, w

    # ---- transition density: posterior(state).gp.logpdf(posterior.coefficients(mesh)) (GeneratorWrapperStochastic.scala:42-63)
    def logpdf_prepare(self, mesh_full):
        """after phase 1, before the exchange of segment 1: the shard's part of Q0^T e for the OTHER state's mesh"""
        st, r = self.st, self.m.rank
        R = st.rotation()
        e = (np.asarray(mesh_full, dtype=np.float64)[self.b:self.e] - st.center - st.translation) @ R - (self.ref - st.center) - self.mean
        self.seg(1)[r * r + r + 8:] = self.Q0.T @ e.reshape(-1)

    def logpdf_finish(self):
        """segment 1 summed: replicated.  Posterior basis Q_p = Q L^-T (any orthogonal mixing leaves c.c unchanged), L L^T = I + G"""
        r = self.m.rank
        s = self.seg(1)
        G, rhs, qte = s[: r * r].reshape(r, r), s[r * r: r * r + r], s[r * r + r + 8:]
        L = np.linalg.cholesky(np.eye(r) + G)
        a = np.linalg.solve(L.T, np.linalg.solve(L, rhs))
        Li = np.linalg.inv(L)
        A = Li @ self.S_tot @ Li.T
        c = np.linalg.solve(A / EPS + np.eye(r), Li @ (qte - self.S_tot @ a) / EPS)
        return go.gp_logpdf(c)
