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
