"""NumPy restatement of the sort-free z-step for piecewise-constant rank weights (csrc/zband.hip), so that the
structure the HIP kernels rely on can be validated on the CPU against the exact PAV (oracle/pav.py:pav_exact, itself
pinned by the reference's goldens).  Reference path being replaced: src/optim/algorithms.py:96-104 (argsort m,
PAV_solver, unsort) for sigma from src/optim/objective.py:108-136.

    bands      sigma = [s_0 x n_0 | s_1 x n_1 | ...]; single-rank bands carry the fractional weight
    structure  inside a band the element prox u_i is non-decreasing in m_i, so PAV pools only ACROSS a band edge, and
               what it pools there is one block {band below: u_i > x} + {single-rank bands} + {band above: u_i < x}
               whose value x is the root of the pooled derivative psi(x) = sum_block sigma_i l'(x) + rho (x - m_i)
    result     z_i = clamp(u_i, lo_band, hi_band) with the block values as clamps - in ROW order, no sort, no unsort

The device finds the band-edge keys with a radix select and the root with multi-candidate passes; here the edge values
come from np.partition and the root from the same monotone psi evaluated at the undecided elements' own prox values
(what k_zb_finish does with the last <= 2048 of them).  The certification rules are the device's: anything else is
reported (status != OK) and left to the sort path.
Test infrastructure only - see oracle/__init__.py.
"""
import numpy as np

from . import pav as _pav
from .prox import prox_exact, sigmoid

OK, TIE, SWALLOW_L, SWALLOW_R, ONESIDED, OVERLAP, UNSUPPORTED = 0, 1, 6, 7, 8, 9, 100


def bands_of(sigma):
    """(starts, values): start rank of every band of equal weights (+ n), the weights"""
    sigma = np.asarray(sigma, dtype=np.float64)
    edges = np.flatnonzero(sigma[1:] != sigma[:-1]) + 1
    starts = np.concatenate(([0], edges, [sigma.size])).astype(np.int64)
    return starts, sigma[starts[:-1]].copy()


def clusters_of(starts, values):
    """[(L, R, can_pool)]: a band of >= 2 ranks, single-rank bands, the next band of >= 2 ranks; None: unsupported"""
    nb = values.size
    size = np.diff(starts)
    if nb < 2 or size[0] < 2 or size[-1] < 2:
        return None
    out, L = [], 0
    for j in range(1, nb):
        if size[j] == 1:
            continue
        if j - L > 3:
            return None                    # more than two single-rank bands in a row: left to the sort
        out.append((L, j, bool(np.any(np.diff(values[L:j + 1]) > 0))))
        L = j
    return out


def z_step(loss, sigma, rho, m):
    """-> (z in row order or None, status).  sigma[k] is the weight of rank k (ascending m, ties by row)."""
    m = np.asarray(m, dtype=np.float64)
    n = m.size
    starts, values = bands_of(sigma)
    clusters = clusters_of(starts, values)
    if clusters is None:
        return None, UNSUPPORTED
    nb = values.size
    # 1. select: the values at the last rank of every band and the first rank of the next one
    ranks = sorted({int(starts[j + 1] - 1) for j in range(nb - 1)} | {int(starts[j]) for j in range(1, nb)})
    part = np.partition(m, ranks)
    at = {r: part[r] for r in ranks}
    for r in ranks:
        if r + 1 in at and not at[r] < at[r + 1]:
            return None, TIE                                   # band membership must be a value comparison
    hi_val = [at[int(starts[j + 1] - 1)] for j in range(nb - 1)] + [np.inf]
    band = np.zeros(n, dtype=np.int64)
    for j in range(nb - 1):
        band += m > hi_val[j]
    u = np.empty(n)
    for j in range(nb):
        sel = band == j
        u[sel] = prox_exact(loss, np.full(int(sel.sum()), values[j]), rho, m[sel])
    lo = np.full(nb, -np.inf)
    hi = np.full(nb, np.inf)
    # 2. one block per cluster whose weights increase somewhere and whose prox chain across the edge decreases
    for (L, R, can_pool) in clusters:
        if not can_pool:
            continue
        chain = [u[band == L].max()] + [u[band == j][0] for j in range(L + 1, R)] + [u[band == R].min()]
        if all(b >= a for a, b in zip(chain, chain[1:])):
            continue
        o = np.argsort(u[band == L], kind="stable")
        top_u, top_m = u[band == L][o], m[band == L][o]
        o = np.argsort(u[band == R], kind="stable")
        bot_u, bot_m = u[band == R][o], m[band == R][o]
        top_suffix = np.concatenate((np.cumsum(top_m[::-1])[::-1], [0.0]))     # sum of m over top positions >= i
        bot_prefix = np.concatenate(([0.0], np.cumsum(bot_m)))                  # sum of m over bottom positions < i
        At = float(sum(values[L + 1:R]))
        Mt = float(sum(m[band == j][0] for j in range(L + 1, R)))
        nt = float(R - L - 1)

        def sets(x, ties_on_top):
            """top part {u > x} (or {u >= x}), bottom part {u < x}: counts and sums of m"""
            i = np.searchsorted(top_u, x, side="left" if ties_on_top else "right")
            j = np.searchsorted(bot_u, x, side="left")
            return top_u.size - i, top_suffix[i], j, bot_prefix[j]

        # psi is non-decreasing; its sign changes at (or between) prox values of elements near the edge
        cand = np.unique(np.concatenate((top_u[top_u >= min(chain)], bot_u[bot_u <= max(chain)], chain)))
        cT, mT, cB, mB = sets(cand, False)
        A, M, cnt = values[L] * cT + At + values[R] * cB, mT + Mt + mB, cT + nt + cB
        if loss == "binary_cross_entropy":
            vals = A * sigmoid(cand) + rho * (cnt * cand - M)
        else:
            vals = cand - prox_exact(loss, A / cnt, rho, M / cnt)
        k = int(np.argmax(vals >= 0)) if np.any(vals >= 0) else cand.size
        if k < cand.size and vals[k] == 0.0:
            x = float(cand[k])
            cT, mT, cB, mB = (float(v[k]) for v in (cT, mT, cB, mB))
        else:
            # the root lies before cand[k]: elements AT cand[k] still belong to the top side, none of them to the bottom
            xr = cand[k] if k < cand.size else np.inf
            cT, mT, cB, mB = (float(v) for v in sets(xr, True))
            x = _pav.block_value(loss, values[L] * cT + At + values[R] * cB, mT + Mt + mB, cT + nt + cB, rho)
            lo_x = cand[k - 1] if k > 0 else -np.inf
            if not (lo_x <= x <= xr):
                return None, UNSUPPORTED
        if not (cT < top_u.size or L == 0):
            return None, SWALLOW_L
        if not (cB < bot_u.size or R == nb - 1):
            return None, SWALLOW_R
        # the root was computed for "top of L + every single-rank band + bottom of R"; it is pav.py's block iff every
        # prefix of it pools to a value >= x (csrc/zband.hip: zb_accept)
        ntiny = R - L - 1
        if ntiny == 0:
            ok = cT > 0 and cB > 0
        else:
            s1, m1, u1 = values[L + 1], m[band == L + 1][0], u[band == L + 1][0]
            s2, m2, u2 = values[R - 1], m[band == R - 1][0], u[band == R - 1][0]
            if cT > 0 and cB > 0:
                ok = True
                if ntiny == 2:
                    xl = _pav.block_value(loss, values[L] * cT + s1, mT + m1, cT + 1.0, rho)
                    xr = _pav.block_value(loss, s2 + values[R] * cB, m2 + mB, 1.0 + cB, rho)
                    ok = xl >= x and xr <= x
            elif cT > 0:
                ok = u2 <= x
            else:
                ok = u1 >= x
        if not ok:
            return None, ONESIDED
        hi[L] = x
        lo[R] = x
        for j in range(L + 1, R):
            lo[j] = hi[j] = x
    if np.any(lo > hi):
        return None, OVERLAP
    return np.minimum(np.maximum(u, lo[band]), hi[band]), OK


# ======================================================================================================================
# The same z-step as the device runs it across GPUs (csrc/zband.hip one step at a time, include/rbl.h: rbl_zbd_*):
# every rank holds its own rows; what depends on all rows is SUMMED (digit histograms of the radix select, block sums
# of the root passes) or GATHERED (the last undecided elements) by the driver in between.  One `Passes` object per
# rank; all of them hold the same state after every step.  tests/_numpy_engine.py wraps it for the gloo tests of
# admm-for-rank-based-loss_amd/dist.py: _z_banded.
BITS = (11, 11, 11, 11, 11, 9)
SHIFT = (53, 42, 31, 20, 9, 0)
NBINS, MAX_GROUPS, NCAND, ROOT_PASSES, GCAP = 2048, 6, 16, 4, 2048
GROUPS, BAD, BRACKET, UNRESOLVED = 3, 2, 4, 5


def flip_key(x):
    b = np.ascontiguousarray(x, dtype=np.float64).view(np.uint64)
    top = np.uint64(1) << np.uint64(63)
    return np.where(b & top != 0, ~b, b | top)


def unflip_key(k):
    k = np.asarray(k, dtype=np.uint64)
    top = np.uint64(1) << np.uint64(63)
    return np.where(k & top != 0, k & ~top, ~k).view(np.float64)


def _theta_gt(loss, s_over_rho, x):            # u > x  <=>  m > theta_gt(x)
    if loss == "binary_cross_entropy":
        return x + s_over_rho * sigmoid(x)
    return x + s_over_rho if x >= -1.0 else x


def _theta_lt(loss, s_over_rho, x):            # u < x  <=>  m < theta_lt(x)
    if loss == "binary_cross_entropy":
        return x + s_over_rho * sigmoid(x)
    return x + s_over_rho if x > -1.0 else x


def _psi(loss, A, M, cnt, rho, x):
    if not cnt > 0:
        return 0.0
    if loss == "binary_cross_entropy":
        return A * float(sigmoid(x)) + rho * (cnt * x - M)
    return x - _pav.block_value(loss, A, M, cnt, rho)


def candidates(loss, a, b):
    """zb_candidates: a = x_0 < ... < x_15 = b, the hinge kink and its two neighbours among them when it lies inside"""
    c = [b if i == NCAND - 1 else a + (b - a) * (i / (NCAND - 1)) for i in range(NCAND)]
    if loss == "hinge" and a <= -1.0 <= b and a < b:
        km, kp = np.nextafter(-1.0, -2.0), np.nextafter(-1.0, 0.0)
        if a == -1.0:
            if kp < c[2]:
                c[1] = kp
        elif b == -1.0:
            if km > c[NCAND - 3]:
                c[NCAND - 2] = km
        else:
            i = int((-1.0 - a) / (b - a) * (NCAND - 1) + 0.5)
            i = min(max(i, 2), NCAND - 3)
            if c[i - 2] <= km and kp <= c[i + 2]:
                c[i - 1], c[i], c[i + 1] = max(km, c[i - 2]), -1.0, min(kp, c[i + 2])
    return c


class Passes:
    def __init__(self, loss, sigma):
        self.loss = loss
        self.starts, self.values = bands_of(sigma)
        self.clusters = clusters_of(self.starts, self.values)          # [(L, R, can_pool)] or None
        self.nb = self.values.size
        if self.clusters is not None:
            ranks = sorted({int(self.starts[j + 1] - 1) for j in range(self.nb - 1)} | {int(self.starts[j]) for j in range(1, self.nb)})
            self.ranks = ranks
            self.last_t = {j: ranks.index(int(self.starts[j + 1] - 1)) for j in range(self.nb - 1)}
            self.first_t = {j: ranks.index(int(self.starts[j])) for j in range(1, self.nb)}

    # ---- rbl_zbd_begin
    def begin(self, m_local, rho):
        self.m, self.rho = np.asarray(m_local, dtype=np.float64), rho
        self.keys = flip_key(self.m)
        T = len(self.ranks)
        self.prefix, self.rem, self.group, self.gprefix = [0] * T, list(self.ranks), [0] * T, [0]
        self.key, self.status = [0] * T, OK
        K = len(self.clusters)
        self.done, self.has_block, self.x = [False] * K, [False] * K, [0.0] * K
        self.und, self.cand, self.br, self.frozen = [1e300] * K, [None] * K, [None] * K, [None] * K

    # ---- rbl_zbd_hist: digit histograms of this rank's keys that still share a target's prefix (to be summed)
    def hist(self, p):
        out = np.zeros((MAX_GROUPS, NBINS), dtype=np.int32)
        if self.status != OK:
            return out.reshape(-1)
        digit = ((self.keys >> np.uint64(SHIFT[p])) & np.uint64((1 << BITS[p]) - 1)).astype(np.int64)
        for g, gp in enumerate(self.gprefix):
            sel = slice(None) if p == 0 else (self.keys >> np.uint64(SHIFT[p] + BITS[p])) == np.uint64(gp)
            out[g] = np.bincount(digit[sel], minlength=NBINS)
        return out.reshape(-1)

    # ---- rbl_zbd_scan: every target's digit inside its bucket, regrouping; after the last pass the keys and the clusters
    def scan(self, p, hist_sum):
        if self.status != OK:
            return
        h = np.asarray(hist_sum, dtype=np.int64).reshape(MAX_GROUPS, NBINS)
        for t in range(len(self.ranks)):
            cum = np.cumsum(h[self.group[t]])
            b = int(np.searchsorted(cum, self.rem[t], side="right"))
            if b >= NBINS:
                self.status = BAD
                return
            self.rem[t] -= int(cum[b - 1]) if b > 0 else 0
            self.prefix[t] = (self.prefix[t] << BITS[p]) | b
        self.gprefix = []
        for t in range(len(self.ranks)):
            if self.prefix[t] not in self.gprefix:
                if len(self.gprefix) == MAX_GROUPS:
                    self.status = GROUPS
                    return
                self.gprefix.append(self.prefix[t])
            self.group[t] = self.gprefix.index(self.prefix[t])
        if p == 5:
            self.key = list(self.prefix)
            self._cluster_setup()

    def _mkey(self, t):
        return float(unflip_key(np.array([self.key[t]], dtype=np.uint64))[0])

    def _prox1(self, j, m):
        return float(prox_exact(self.loss, np.array([self.values[j]]), self.rho, np.array([m]))[0])

    def _cluster_setup(self):
        for t in range(len(self.ranks) - 1):
            if self.ranks[t + 1] == self.ranks[t] + 1 and not self.key[t] < self.key[t + 1]:
                self.status = TIE
                return
        for k, (L, R, can_pool) in enumerate(self.clusters):
            if not can_pool:
                self.done[k] = True
                continue
            chain = [self._prox1(L, self._mkey(self.last_t[L]))] + [self._prox1(j, self._mkey(self.first_t[j])) for j in range(L + 1, R + 1)]
            if all(b >= a for a, b in zip(chain, chain[1:])):
                self.done[k] = True
                continue
            self.cand[k] = candidates(self.loss, min(chain), max(chain))

    def root_clusters(self):
        return [k for k, c in enumerate(self.clusters) if c[2]]

    def _band_mask(self, j):
        lo = self.key[self.last_t[j - 1]] if j > 0 else None
        hi = self.key[self.last_t[j]] if j < self.nb - 1 else None
        sel = np.ones(self.keys.size, dtype=bool)
        if lo is not None:
            sel &= self.keys > np.uint64(lo)
        if hi is not None:
            sel &= self.keys <= np.uint64(hi)
        return sel

    def _idle(self, k):
        return self.status != OK or self.done[k] or self.und[k] <= GCAP

    # ---- rbl_zbd_eval: (sum m, count) of the top part of L / the bottom part of R for 16 candidates (to be summed)
    def eval(self, k):
        tot = np.zeros(4 * NCAND)
        if self._idle(k):
            return tot
        L, R, _ = self.clusters[k]
        mt, mb = self.m[self._band_mask(L)], self.m[self._band_mask(R)]
        for c, x in enumerate(self.cand[k]):
            t = mt > _theta_gt(self.loss, self.values[L] / self.rho, x)
            b = mb < _theta_lt(self.loss, self.values[R] / self.rho, x)
            tot[c], tot[NCAND + c] = mt[t].sum(), t.sum()
            tot[2 * NCAND + c], tot[3 * NCAND + c] = mb[b].sum(), b.sum()
        return tot

    def _tiny(self, k):
        L, R, _ = self.clusters[k]
        js = range(L + 1, R)
        return (float(sum(self.values[j] for j in js)), float(sum(self._mkey(self.first_t[j]) for j in js)), float(len(js)))

    # ---- rbl_zbd_decide (k_zb_refine's second half)
    def decide(self, k, tot, last):
        if self._idle(k):
            return
        L, R, _ = self.clusters[k]
        MT, NT, MB, NB = (np.asarray(tot[i * NCAND:(i + 1) * NCAND], dtype=np.float64) for i in range(4))
        At, Mt, nt = self._tiny(k)
        sL, sR, rho, cand = self.values[L], self.values[R], self.rho, self.cand[k]
        psi = [_psi(self.loss, sL * NT[c] + At + sR * NB[c], MT[c] + Mt + MB[c], NT[c] + nt + NB[c], rho, cand[c]) for c in range(NCAND)]
        c1 = next((c for c in range(NCAND) if psi[c] >= 0.0), -1)
        if c1 < 0 or (c1 == 0 and psi[0] > 0.0):
            self.status = BRACKET
            return
        km, kp = np.nextafter(-1.0, -2.0), np.nextafter(-1.0, 0.0)
        found = None
        if (self.loss == "hinge" and c1 > 0 and psi[c1] > 0.0 and
                ((cand[c1 - 1] == -1.0 and cand[c1] == kp) or (cand[c1 - 1] == km and cand[c1] == -1.0))):
            found = (-1.0, NT[c1 - 1], MT[c1 - 1], NB[c1], MB[c1])
        elif psi[c1] == 0.0:
            found = (cand[c1], NT[c1], MT[c1], NB[c1], MB[c1])
        else:
            c0 = c1 - 1
            undecided = (NT[c0] - NT[c1]) + (NB[c1] - NB[c0])
            if undecided == 0.0:
                cT, mT, cB, mB = NT[c1], MT[c1], NB[c0], MB[c0]
                x = _pav.block_value(self.loss, sL * cT + At + sR * cB, mT + Mt + mB, cT + nt + cB, rho)
                if not (cand[c0] <= x <= cand[c1]):
                    self.status = BRACKET
                    return
                found = (x, cT, mT, cB, mB)
            else:
                a, b = cand[c0], cand[c1]
                self.br[k], self.frozen[k], self.und[k] = (a, b), (NT[c1], MT[c1], NB[c0], MB[c0]), undecided
                if undecided > GCAP:
                    if last or not b > a:
                        self.status = UNRESOLVED
                        return
                    self.cand[k] = candidates(self.loss, a, b)
        if found:
            self._accept(k, *found)

    # ---- zb_accept
    def _accept(self, k, x, cT, mT, cB, mB):
        L, R, _ = self.clusters[k]
        sizeL, sizeR = self.starts[L + 1] - self.starts[L], self.starts[R + 1] - self.starts[R]
        if not (cT < sizeL or L == 0):
            self.status = SWALLOW_L
            return
        if not (cB < sizeR or R == self.nb - 1):
            self.status = SWALLOW_R
            return
        ntiny = R - L - 1
        if ntiny == 0:
            ok = cT > 0 and cB > 0
        else:
            s1, m1 = self.values[L + 1], self._mkey(self.first_t[L + 1])
            s2, m2 = self.values[R - 1], self._mkey(self.first_t[R - 1])
            u1, u2 = self._prox1(L + 1, m1), self._prox1(R - 1, m2)
            if cT > 0 and cB > 0:
                ok = True
                if ntiny == 2:
                    xl = _pav.block_value(self.loss, self.values[L] * cT + s1, mT + m1, cT + 1.0, self.rho)
                    xr = _pav.block_value(self.loss, s2 + self.values[R] * cB, m2 + mB, 1.0 + cB, self.rho)
                    ok = xl >= x and xr <= x
            elif cT > 0:
                ok = u2 <= x
            else:
                ok = u1 >= x
        if not ok:
            self.status = ONESIDED
            return
        self.x[k], self.has_block[k], self.done[k] = float(x), True, True

    # ---- rbl_zbd_gather: [count | this rank's undecided elements] (to be all-gathered)
    def gather(self, k):
        pack = np.zeros(GCAP + 1)
        if self.status != OK or self.done[k] or self.und[k] > GCAP:
            return pack
        L, R, _ = self.clusters[k]
        a, b = self.br[k]
        sl, sr = self.values[L] / self.rho, self.values[R] / self.rho
        mt, mb = self.m[self._band_mask(L)], self.m[self._band_mask(R)]
        top = mt[(mt > _theta_gt(self.loss, sl, a)) & ~(mt > _theta_gt(self.loss, sl, b))]
        bot = mb[(mb < _theta_lt(self.loss, sr, b)) & ~(mb < _theta_lt(self.loss, sr, a))]
        vals = np.concatenate((top, bot))[:GCAP]
        pack[0] = vals.size
        pack[1:1 + vals.size] = vals
        return pack

    # ---- rbl_zbd_finish: the union of the gathered elements settles the block exactly (k_zb_union + k_zb_finish)
    def finish(self, k, packs_all, world):
        if self.status != OK or self.done[k]:
            return
        if self.und[k] > GCAP:
            self.status = UNRESOLVED
            return
        packs = np.asarray(packs_all, dtype=np.float64).reshape(world, GCAP + 1)
        ms = np.concatenate([packs[r, 1:1 + int(packs[r, 0])] for r in range(world)])
        if ms.size != self.und[k]:
            self.status = BAD
            return
        L, R, _ = self.clusters[k]
        sL, sR, rho = self.values[L], self.values[R], self.rho
        mLhi = self._mkey(self.last_t[L])
        is_top = ms <= mLhi
        u = prox_exact(self.loss, np.where(is_top, sL, sR), rho, ms)
        order = np.lexsort((ms, u))
        ms, u, is_top = ms[order], u[order], is_top[order]
        fTc, fTm, fBc, fBm = self.frozen[k]
        At, Mt, nt = self._tiny(k)
        a, b = self.br[k]

        def sets(x, ties_on_top):
            t = is_top & ((u >= x) if ties_on_top else (u > x))
            bt = ~is_top & (u < x)
            return fTc + t.sum(), fTm + ms[t].sum(), fBc + bt.sum(), fBm + ms[bt].sum()

        def psi_of(s, x):
            cT, mT, cB, mB = s
            return _psi(self.loss, sL * cT + At + sR * cB, mT + Mt + mB, cT + nt + cB, rho, x)

        cand = np.unique(u)
        vals = [psi_of(sets(x, False), x) for x in cand]
        kk = next((i for i, v in enumerate(vals) if v >= 0.0), cand.size)
        if kk < cand.size and vals[kk] == 0.0:
            x, s, lo_x, hi_x = float(cand[kk]), sets(cand[kk], False), cand[kk], cand[kk]
        else:
            xr = cand[kk] if kk < cand.size else np.inf
            s = sets(xr, True)
            cT, mT, cB, mB = s
            x = _pav.block_value(self.loss, sL * cT + At + sR * cB, mT + Mt + mB, cT + nt + cB, rho)
            lo_x, hi_x = (cand[kk - 1] if kk > 0 else a), (xr if kk < cand.size else b)
        if not (lo_x <= x <= hi_x and a <= x <= b):
            self.status = BRACKET
            return
        self._accept(k, x, *s)

    # ---- rbl_zbd_apply: z of the local rows (None: not certified)
    def apply(self):
        status = self.status
        if status == OK and not all(self.done):
            status = UNRESOLVED
        lo, hi = np.full(self.nb, -np.inf), np.full(self.nb, np.inf)
        for k, (L, R, _) in enumerate(self.clusters):
            if self.has_block[k]:
                hi[L], lo[R] = self.x[k], self.x[k]
                for j in range(L + 1, R):
                    lo[j] = hi[j] = self.x[k]
        if status == OK and np.any(lo > hi):
            status = OVERLAP
        if status != OK:
            return None, status
        band = np.zeros(self.keys.size, dtype=np.int64)
        for j in range(self.nb - 1):
            band += self.keys > np.uint64(self.key[self.last_t[j]])
        u = prox_exact(self.loss, self.values[band], self.rho, self.m)
        return np.minimum(np.maximum(u, lo[band]), hi[band]), OK
