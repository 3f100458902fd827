"""CPU restatements of three model set-up kernels of gingr_amd/csrc/gpmm.hip whose round-6 forms claim to give the bits of the forms they
replaced.  The claims rest on small pieces of index arithmetic and algebra; these tests state them in numpy / plain Python so that
they can be checked without a GPU (the GPU side is tools/experiments/gpmm_build_ab.py: eleven models, two libraries, bit for bit).

* dist_extrema_kernel: the tile pairs on and above the diagonal, dealt to workgroups (tile row, chunk of `ch` column tiles), visit
  every unordered pair of points exactly once and the pairs of a diagonal tile from both sides -- so maximum and minimum are the full
  scan's (GPMMHelper.scala:75-87).
* log_position / log_occupant: the tie rule of the pivoted Cholesky (scalismo takes the first maximum in the current PERMUTED order)
  replays a log of swaps; the branch-free form (the step that pivoted e, else the last step that displaced it) equals the literal replay.
* jacobi_eig_grid_kernel: one round of the two-sided Jacobi method formed entry by entry, A'[i][j] from the four entries at
  (i | partner of i, j | partner of j) -- column combination first, row combination second -- equals the two passes of the
  one-workgroup kernel (all columns, then all rows) bit for bit.
"""
import math

import numpy as np


def _extrema_plan(n, tile=256):
    nt = -(-n // tile)
    ch = max(8, -(-nt // 64))
    gy = -(-nt // ch)
    return nt, ch, gy


def test_distance_extrema_visits_every_pair_once():
    for n in (1, 2, 255, 256, 257, 2049, 4500, 70000):
        nt, ch, gy = _extrema_plan(n)
        assert gy <= 64 and gy * ch >= nt
        seen = np.zeros((nt, nt), dtype=np.int32)
        for bi in range(nt):
            for c in range(gy):
                jt0 = bi + c * ch
                for jt in range(jt0, min(nt, jt0 + ch)):
                    seen[bi, jt] += 1
        assert np.array_equal(seen, np.triu(np.ones((nt, nt), dtype=np.int32))), n   # each tile pair on / above the diagonal: once


def test_distance_extrema_of_the_triangle_scan_are_the_full_scans():
    rng = np.random.default_rng(7)
    for n in (2, 257, 700):
        P = rng.normal(0.0, 100.0, (n, 3))
        if n > 300:
            P[500] = P[3]                                    # coincident points in different tiles: the minimum is 0
        tile = 256
        nt, ch, gy = _extrema_plan(n, tile)
        mx, mn = 0.0, math.inf
        for bi in range(nt):
            A = P[bi * tile:(bi + 1) * tile]
            for c in range(gy):
                for jt in range(bi + c * ch, min(nt, bi + (c + 1) * ch)):
                    B = P[jt * tile:(jt + 1) * tile]
                    d = A[:, None, :] - B[None, :, :]
                    d2 = d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1] + d[..., 2] * d[..., 2]
                    mx = max(mx, float(d2.max()))
                    if jt == bi:
                        d2 = d2.copy()
                        np.fill_diagonal(d2, math.inf)
                    mn = min(mn, float(d2.min()))
        full = P[:, None, :] - P[None, :, :]
        f2 = full[..., 0] * full[..., 0] + full[..., 1] * full[..., 1] + full[..., 2] * full[..., 2]
        fmx = float(f2.max())
        np.fill_diagonal(f2, math.inf)
        assert mx == fmx and mn == float(f2.min())


def _literal_position(piv, dis, old, e, virtual=None):
    q = e
    for s in range(len(piv)):
        if piv[s] == e:
            return s
        if dis[s] == e:
            q = old[s]
    if virtual is not None:
        vp, vd, vo = virtual
        if vp == e:
            return len(piv)
        if vd == e:
            q = vo
    return q


def _select_position(piv, dis, old, e, virtual=None):
    sp = sl = -1
    for s in range(len(piv)):
        sp = s if piv[s] == e else sp
        sl = s if dis[s] == e else sl
    if sp >= 0:
        return sp
    q = old[sl] if sl >= 0 else e
    if virtual is not None:
        vp, vd, vo = virtual
        if vp == e:
            return len(piv)
        if vd == e:
            q = vo
    return q


def _literal_occupant(dis, old, slot):
    e = slot
    for s in range(len(old)):
        if old[s] == slot:
            e = dis[s]
    return e


def _select_occupant(dis, old, slot):
    sl = -1
    for s in range(len(old)):
        sl = s if old[s] == slot else sl
    return dis[sl] if sl >= 0 else slot


def test_swap_log_replay_without_branches_equals_the_literal_replay():
    rng = np.random.default_rng(11)
    for n, steps in [(5, 5), (40, 17), (300, 120), (64, 64)]:
        slot_of = list(range(n))                 # element -> slot (what the log stands for)
        elem_at = list(range(n))                 # slot -> element
        piv, dis, old = [], [], []
        for k in range(steps):
            for e in range(n):                   # every element, against the true permutation and against each other
                want = slot_of[e]
                assert _literal_position(piv, dis, old, e) == want == _select_position(piv, dis, old, e)
            for slot in range(k, n):
                assert _literal_occupant(dis, old, slot) == elem_at[slot] == _select_occupant(dis, old, slot)
            p = int(rng.choice([e for e in range(n) if slot_of[e] >= k]))      # any element not pivoted yet
            d, o = elem_at[k], slot_of[p]
            # the step being executed as the virtual entry (not in the log yet)
            for e in (p, d, int(rng.integers(n))):
                a, b = _literal_position(piv, dis, old, e, (p, d, o)), _select_position(piv, dis, old, e, (p, d, o))
                assert a == b
            piv.append(p), dis.append(d), old.append(o)
            elem_at[k], elem_at[o] = p, d
            slot_of[p], slot_of[d] = k, o


def _round_pairs(n, rd):
    """the round-robin pairing of jacobi_eig_kernel / jacobi_eig_grid_kernel: (a, b) with a < b, b = -1 for the bye of an odd n"""
    np_ = (n + 1) & ~1
    out = []
    for t in range(np_ // 2):
        a = np_ - 1 if t == 0 else (rd + t) % (np_ - 1)
        b = (rd + np_ - 1 - t) % (np_ - 1)
        if a > b:
            a, b = b, a
        out.append((a, b if b < n else -1))
    return out


def test_one_jacobi_round_entry_by_entry_equals_columns_then_rows():
    rng = np.random.default_rng(5)
    for n in (2, 7, 16, 33):
        X = rng.normal(size=(n, n))
        A0 = X @ X.T
        for rd in range(((n + 1) & ~1) - 1):
            pairs = _round_pairs(n, rd)
            touched = sorted(i for a, b in pairs for i in (a, b) if b >= 0)
            assert len(set(touched)) == len(touched)                           # disjoint pairs
            rot = []
            for a, b in pairs:
                c, s = 1.0, 0.0
                if b >= 0:
                    app, aqq, apq = A0[a, a], A0[b, b], A0[a, b]
                    if abs(apq) > 1.1102230246251565e-16 * math.sqrt(abs(app) * abs(aqq)) and apq != 0.0:
                        theta = (aqq - app) / (2.0 * apq)
                        tt = (1.0 if theta >= 0.0 else -1.0) / (abs(theta) + math.sqrt(theta * theta + 1.0))
                        c = 1.0 / math.sqrt(tt * tt + 1.0)
                        s = tt * c
                rot.append((c, s))
            # the one-workgroup kernel: all columns, then all rows, in place
            A = A0.copy()
            for (p, q), (c, s) in zip(pairs, rot):
                if q < 0 or s == 0.0:
                    continue
                aip, aiq = A[:, p].copy(), A[:, q].copy()
                A[:, p] = c * aip - s * aiq
                A[:, q] = s * aip + c * aiq
            for (p, q), (c, s) in zip(pairs, rot):
                if q < 0 or s == 0.0:
                    continue
                apj, aqj = A[p, :].copy(), A[q, :].copy()
                A[p, :] = c * apj - s * aqj
                A[q, :] = s * apj + c * aqj
            # the grid kernel: every entry from the source buffer
            pair_of = [-1] * n
            for m, ((p, q), (c, s)) in enumerate(zip(pairs, rot)):
                if q >= 0 and s != 0.0:
                    pair_of[p] = pair_of[q] = m
            B = np.empty_like(A0)
            for i in range(n):
                for j in range(n):
                    mj = pair_of[j]

                    def col(r):
                        if mj < 0:
                            return A0[r, j]
                        p, q = pairs[mj]
                        c, s = rot[mj]
                        return c * A0[r, p] - s * A0[r, q] if j == p else s * A0[r, p] + c * A0[r, q]
                    mi = pair_of[i]
                    if mi < 0:
                        B[i, j] = col(i)
                    else:
                        p, q = pairs[mi]
                        c, s = rot[mi]
                        B[i, j] = c * col(p) - s * col(q) if i == p else s * col(p) + c * col(q)
            assert np.array_equal(A, B), (n, rd)
            A0 = A


def test_blocked_factor_gram_writes_every_entry_once():
    """pc_gram_kernel: 4 x 4 blocks (a0, b0) with a0 <= b0; of a block only the entries a <= b inside the matrix are stored (and
    mirrored) -- every entry of the upper triangle exactly once, whatever k is modulo 4."""
    tile = 4
    for k in (1, 3, 4, 5, 34, 171):
        nb = -(-k // tile)
        written = np.zeros((k, k), dtype=np.int32)
        for bx in range(nb):
            for by in range(nb):
                a0, b0 = bx * tile, by * tile
                if a0 > b0:
                    continue
                for u in range(tile * tile):
                    a, b = a0 + u // tile, b0 + u % tile
                    if a < k and b < k and a <= b:
                        written[a, b] += 1
                        if a != b:
                            written[b, a] += 1
        assert np.array_equal(written, np.ones((k, k), dtype=np.int32)), k
