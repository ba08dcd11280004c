"""Generalised pool-adjacent-violators for the z-step.

Solves   min_{u_1<=...<=u_n}  sum_i  sigma_i*loss(u_i) + rho/2 (u_i - m_i)^2
for m sorted ascending (the reference sorts first: src/optim/algorithms.py:92-93).
A block's value is the root of  mean(sigma)*loss'(x) + rho*(x - mean(m)) = 0
(src/util/pav.py:134-140).

* ``pav_exact``      - classic stack PAV with exact block solves (unique answer).
* ``pav_tree_exact`` - the merge-tree formulation the HIP kernels use, restated in
                       NumPy so the GPU algorithm can be validated on the CPU.
* ``pav_faithful``   - the reference's sweep PAV (src/util/pav.py:93-178) restated,
                       with its batch prox solves (oracle/prox.py faithful solvers).
* ``ehrm_*``         - the EHRM/CPT two-branch variant (src/util/PAV_cpt.py:169-293).
Test infrastructure only - see oracle/__init__.py.
"""
import ctypes
import math
import os
import numpy as np

from . import prox as _prox

_HERE = os.path.dirname(os.path.abspath(__file__))
_LOSS_ID = {"binary_cross_entropy": 0, "hinge": 1}


# ------------------------------------------------------------ scalar block solve
def _sig(x):
    if x > 0:
        return 1.0 / (1.0 + math.exp(-x))
    e = math.exp(x)
    return e / (1.0 + e)


def block_value(loss, s_sigma, s_m, cnt, rho, lo=None, hi=None):
    """Root of  s_sigma*loss'(x) + rho*(cnt*x - s_m) = 0  (block sums, not means:
    same equation as pav.py:134-140 multiplied by the block length)."""
    mbar = s_m / cnt
    sbar = s_sigma / cnt
    if loss == "hinge":
        a = mbar - sbar / rho
        if a >= -1.0:
            return a
        return mbar if mbar <= -1.0 else -1.0
    blo = mbar - sbar / rho
    bhi = mbar
    if lo is not None and lo > blo:
        blo = lo
    if hi is not None and hi < bhi:
        bhi = hi
    if not blo <= bhi:
        blo, bhi = mbar - sbar / rho, mbar
    x = bhi
    dxold = bhi - blo
    dx = dxold
    s = _sig(x)
    g = sbar * s + rho * (x - mbar)
    h = sbar * s * (1.0 - s) + rho
    for _ in range(200):
        if g == 0:
            break
        if ((x - bhi) * h - g) * ((x - blo) * h - g) > 0 or abs(2.0 * g) > abs(dxold * h):
            dxold = dx
            dx = 0.5 * (bhi - blo)
            xn = blo + dx
        else:
            dxold = dx
            dx = g / h
            xn = x - dx
        if xn == x:
            break
        x = xn
        s = _sig(x)
        g = sbar * s + rho * (x - mbar)
        h = sbar * s * (1.0 - s) + rho
        if g < 0:
            blo = x
        else:
            bhi = x
    return x


# ------------------------------------------------------------------ exact (stack)
def pav_exact_py(loss, sigma, rho, m):
    """Stack PAV, pure Python (small n only).  Returns (u, n_blocks)."""
    sigma = np.asarray(sigma, dtype=np.float64).reshape(-1)
    m = np.asarray(m, dtype=np.float64).reshape(-1)
    n = m.shape[0]
    x0 = _prox.prox_exact(loss, sigma, rho, m)
    S, M, C, X = [], [], [], []
    for i in range(n):
        s, mm, c, x = float(sigma[i]), float(m[i]), 1, float(x0[i])
        while X and X[-1] > x:
            s += S.pop()
            mm += M.pop()
            c += C.pop()
            X.pop()
            x = block_value(loss, s, mm, c, rho)
        S.append(s); M.append(mm); C.append(c); X.append(x)
    return np.repeat(np.array(X), np.array(C)), len(X)


_clib = None


def _load_c():
    global _clib
    if _clib is None:
        path = os.path.join(_HERE, "_build", "liboracle_pav.so")
        if not os.path.exists(path):
            raise FileNotFoundError(
                f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
        lib = ctypes.CDLL(path)
        lib.oracle_pav_exact.restype = ctypes.c_long
        lib.oracle_pav_exact.argtypes = [
            ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long,
            ctypes.c_double, ctypes.c_void_p]
        _clib = lib
    return _clib


def pav_exact(loss, sigma, rho, m):
    """Stack PAV via the C restatement (oracle/pav_c.c); same algorithm as
    ``pav_exact_py``.  Returns (u, n_blocks)."""
    sigma = np.ascontiguousarray(sigma, dtype=np.float64).reshape(-1)
    m = np.ascontiguousarray(m, dtype=np.float64).reshape(-1)
    out = np.empty_like(m)
    lib = _load_c()
    nb = lib.oracle_pav_exact(_LOSS_ID[loss], sigma.ctypes.data, m.ctypes.data,
                              m.shape[0], float(rho), out.ctypes.data)
    if nb < 0:
        raise RuntimeError("oracle_pav_exact failed")
    return out, int(nb)


# ------------------------------------------------- merge-tree form (GPU algorithm)
def _lossprime(loss, t):
    if loss == "hinge":
        # subgradient selection irrelevant away from the kink; at the kink see _F
        return 1.0 if t > -1.0 else 0.0
    return _sig(t)


def pav_tree_exact(loss, sigma, rho, m, stats=None):
    """Bottom-up merge tree.  Invariant: after level L every aligned segment of
    2^L sorted positions holds its own isotonic solution in u (non-decreasing).
    Merging two solved neighbours pools exactly one block around the seam: the left
    elements with value > x* and the right elements with value < x*, where x* is the
    root of the continuous increasing function
        Psi(t) = sum_{i in A(t)} f_i'(t),  A(t) = {left: u_i > t} U {right: u_i < t}.
    s* / e* are found by binary search on the sign of Psi at existing values; the
    pooled block is then solved once.  Uses prefix sums of sigma and m."""
    sigma = np.asarray(sigma, dtype=np.float64).reshape(-1)
    m = np.asarray(m, dtype=np.float64).reshape(-1)
    n = m.shape[0]
    u = _prox.prox_exact(loss, sigma, rho, m).copy()
    PA = np.concatenate([[0.0], np.cumsum(sigma.astype(np.longdouble))]).astype(np.longdouble)
    PM = np.concatenate([[0.0], np.cumsum(m.astype(np.longdouble))]).astype(np.longdouble)
    merges = 0

    def sums(s, e):
        return float(PA[e + 1] - PA[s]), float(PM[e + 1] - PM[s]), e + 1 - s

    def psi_sign(s, e, t):
        """sign of the pooled derivative at t for block [s,e] (block value vs t)."""
        if e < s:
            return 0.0
        A, M, c = sums(s, e)
        xb = block_value(loss, A, M, c, rho)
        return t - xb  # >0  <=>  derivative at t positive  <=>  block value < t

    size = 1
    while size < n:
        for seam in range(size, n, 2 * size):
            L0, R1 = seam - size, min(seam + size, n)
            if u[seam - 1] <= u[seam]:
                continue
            merges += 1
            left, right = u[L0:seam], u[seam:R1]

            def psi_at(t):
                s = L0 + int(np.searchsorted(left, t, side="right"))
                e = seam + int(np.searchsorted(right, t, side="left")) - 1
                return psi_sign(s, e, t)

            lo, hi = L0, seam - 1            # first i with Psi(u[i]) > 0 ; pred(hi) true
            while lo < hi:
                mid = (lo + hi) // 2
                if psi_at(u[mid]) > 0:
                    hi = mid
                else:
                    lo = mid + 1
            s_star = lo
            lo, hi = seam, R1 - 1            # last j with Psi(u[j]) < 0 ; pred(lo) true
            while lo < hi:
                mid = (lo + hi + 1) // 2
                if psi_at(u[mid]) < 0:
                    lo = mid
                else:
                    hi = mid - 1
            e_star = lo
            A, M, c = sums(s_star, e_star)
            u[s_star:e_star + 1] = block_value(loss, A, M, c, rho)
        size *= 2
    if stats is not None:
        stats["merges"] = merges
    nblocks = 1 + int(np.count_nonzero(np.diff(u) != 0)) if n else 0
    return u, nblocks


# ----------------------------------------------------------------- faithful sweeps
def _seq_runsum(a, idx, runlen):
    """Sum of every run a[idx[k] : idx[k]+runlen[k]] accumulated strictly left to right,
    ((a0+a1)+a2)+..., the order in which the reference's merge_block adds block sums
    (src/util/pav.py:22-29,113-119).  np.add.reduceat rounds differently, and the hinge
    bisection's early exit amplifies last-bit differences into visible ones."""
    acc = a[idx].copy()
    k = 1
    live = np.flatnonzero(runlen > 1)
    while live.size:
        acc[live] = acc[live] + a[idx[live] + k]
        k += 1
        live = live[runlen[live] > k]
    return acc


def pav_faithful(loss, sigma, rho, m, maxiter=None, stats=None):
    """src/util/pav.py:54-68 (initial batch prox) and :93-178 (get_opt): each sweep
    merges every maximal strictly-decreasing run into one block, re-solves ONLY the
    merged blocks as one batch with (sum sigma/len, sum m/len) (:134-146), repeats
    until no violation or ``maxiter`` sweeps (:127; the caller passes n,
    src/optim/algorithms.py:70,101)."""
    sigma = np.asarray(sigma, dtype=np.float64).reshape(-1)
    m = np.asarray(m, dtype=np.float64).reshape(-1)
    n = m.shape[0]
    if maxiter is None:
        maxiter = 10000
    x = np.asarray(_prox.prox_faithful(loss, sigma, rho, m), dtype=np.float64).reshape(-1)
    S, M, C = sigma.copy(), m.copy(), np.ones(n, dtype=np.int64)
    count = 0
    while True:
        count += 1
        if x.shape[0] < 2:
            break
        viol = x[:-1] > x[1:]
        if not viol.any() or count >= maxiter:
            break
        starts = np.concatenate([[True], ~viol])
        idx = np.flatnonzero(starts)
        runlen = np.diff(np.concatenate([idx, [x.shape[0]]]))
        S = _seq_runsum(S, idx, runlen)
        M = _seq_runsum(M, idx, runlen)
        C = np.add.reduceat(C, idx)
        newx = x[idx].copy()
        vio = runlen > 1
        newx[vio] = np.asarray(
            _prox.prox_faithful(loss, S[vio] / C[vio], rho, M[vio] / C[vio])).reshape(-1)
        x = newx
    if stats is not None:
        stats["sweeps"] = count
    return np.repeat(x, C), x.shape[0]


# ------------------------------------------------------------------------- EHRM
def _cpt_fval(sigma, rho, m, z):
    # PAV_cpt.py:41-43 (func_value): a SCALAR
    return float(np.sum(sigma * _prox._log1exp_ref(z)) + rho / 2 * np.dot(z - m, z - m))


def ehrm_branch_exact(sigma_a, sigma_b, B, rho, m):
    """Singleton-stage scalar test of PAV_cpt.py:205-226 with exact element solves.
    Returns 'a' (all z <= B, sigma_a) or 'b' (all z >= B, sigma_b)."""
    sigma_a = np.asarray(sigma_a, dtype=np.float64).reshape(-1)
    sigma_b = np.asarray(sigma_b, dtype=np.float64).reshape(-1)
    m = np.asarray(m, dtype=np.float64).reshape(-1)
    o1 = np.minimum(_prox.prox_bce_exact(sigma_a, rho, m), B)
    o2 = np.maximum(_prox.prox_bce_exact(sigma_b, rho, m), B)
    f1 = _cpt_fval(sigma_a, rho, m, o1)
    f2 = _cpt_fval(sigma_b, rho, m, o2)
    return "a" if f1 <= f2 else "b"


def ehrm_exact(sigma_a, sigma_b, B, rho, m, branch=None, use_c=True):
    """EHRM z-step in its clean form: z = min(B, PAV(sigma_a, m)) or
    max(B, PAV(sigma_b, m)); the branch is chosen by the singleton-stage scalar test
    unless given.  Returns (z_sorted, branch)."""
    if branch is None:
        branch = ehrm_branch_exact(sigma_a, sigma_b, B, rho, m)
    f = pav_exact if use_c else pav_exact_py
    if branch == "a":
        u, _ = f("binary_cross_entropy", sigma_a, rho, m)
        return np.minimum(u, B), branch
    u, _ = f("binary_cross_entropy", sigma_b, rho, m)
    return np.maximum(u, B), branch


def _newton_system(sigma, rho, m, tol=1e-4, maxiter=50):
    """PAV_cpt.py:72-94 (newton_system) with :47-63 pieces: global Armijo with
    shrink 0.5, stop on batch ||delta||_2 < 1e-4."""
    def fun(z):
        return _cpt_fval(sigma, rho, m, z)

    x = np.asarray(m, dtype=np.float64).copy()
    for _ in range(maxiter):
        sg = _prox.sigmoid(x)
        rtx = sigma * sg + rho * (x - m)
        with np.errstate(over="ignore"):
            invg = 1.0 / (sigma * sg / (1 + np.exp(x)) + rho)
        delta = -rtx * invg
        alpha = 1.0
        tempx = x + alpha * delta
        slope = np.dot(rtx, delta)
        fx = fun(x)
        while fun(tempx) > fx + 0.0001 * alpha * slope:
            alpha *= 0.5
            tempx = x + alpha * delta
        x = tempx
        nd = np.linalg.norm(delta)
        if nd < tol:
            return x
        if nd > 1e10:
            return _prox.prox_bce_exact(sigma, rho, m)
    return x


def ehrm_faithful(sigma_a, sigma_b, B, rho, m, stats=None):
    """PAV_cpt.py:169-293 restated: two Newton systems, clip, WHOLE-VECTOR branch
    pick by the scalar objective (:222-226, :287-288), sweeps that re-solve every
    block with np.mean parameters (:122-123, :258-265).  Returns (z_sorted, picks)."""
    sigma_a = np.asarray(sigma_a, dtype=np.float64).reshape(-1)
    sigma_b = np.asarray(sigma_b, dtype=np.float64).reshape(-1)
    m = np.asarray(m, dtype=np.float64).reshape(-1)
    n = m.shape[0]
    picks = []

    def solve(sa, sb, mm):
        o1 = _newton_system(sa, rho, mm)
        o1 = np.where(o1 > B, B, o1)
        o2 = _newton_system(sb, rho, mm)
        o2 = np.where(o2 <= B, B, o2)
        f1 = _cpt_fval(sa, rho, mm, o1)
        f2 = _cpt_fval(sb, rho, mm, o2)
        picks.append("a" if f1 <= f2 else "b")
        return o1 if f1 <= f2 else o2

    x = solve(sigma_a, sigma_b, m)
    Sa, Sb, M, C = sigma_a.copy(), sigma_b.copy(), m.copy(), np.ones(n, dtype=np.int64)
    sweeps = 0
    while x.shape[0] >= 2:
        viol = x[:-1] > x[1:]
        if not viol.any():
            break
        sweeps += 1
        idx = np.flatnonzero(np.concatenate([[True], ~viol]))
        Sa = np.add.reduceat(Sa, idx)
        Sb = np.add.reduceat(Sb, idx)
        M = np.add.reduceat(M, idx)
        C = np.add.reduceat(C, idx)
        x = solve(Sa / C, Sb / C, M / C)
    if stats is not None:
        stats["sweeps"] = sweeps
    return np.repeat(x, C), picks
