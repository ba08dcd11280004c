"""Element-wise proximal solves  x = argmin_x  sigma*loss(x) + rho/2 (x-m)^2.

Restates src/util/individual_solver.py of the reference.  ``*_exact`` solve each
element to machine precision; ``*_faithful`` restate the reference's own solvers
including their global (whole-batch) step-size and stopping tests.
Test infrastructure only - see oracle/__init__.py.
"""
import numpy as np


# ---------------------------------------------------------------- stable pieces
def sigmoid(x):
    # individual_solver.py:44-49 (safe_1divexp): exp(x)/(1+exp(x)) without overflow
    x = np.asarray(x, dtype=np.float64)
    e = np.exp(-np.abs(x))
    return np.where(x > 0, 1.0 / (1.0 + e), e / (1.0 + e))


def softplus(x):
    # individual_solver.py:52-57 (log1exp): log(1+exp(x)) without overflow
    x = np.asarray(x, dtype=np.float64)
    return np.maximum(x, 0.0) + np.log1p(np.exp(-np.abs(x)))


def _log1exp_ref(x):
    # same two-branch formula as the reference, but log(1+e) exactly as it writes it
    x = np.asarray(x, dtype=np.float64)
    e = np.exp(-np.abs(x))
    return np.where(x > 0, x + np.log(1 + e), np.log(1 + e))


def dsigmoid(x):
    # individual_solver.py:72-76 (safe_expdivexp2): exp(x)/(1+exp(x))^2
    x = np.asarray(x, dtype=np.float64)
    e = np.exp(-np.abs(x))
    return e / (1.0 + e) ** 2


# ------------------------------------------------------------------- exact mode
def prox_bce_exact(sigma, rho, m, iters=200):
    """Root of g(x) = sigma*sigmoid(x) + rho*(x-m)  (individual_solver.py:68-70),
    per element.  g is increasing with g(m - sigma/rho) <= 0 <= g(m) because
    0 <= sigmoid <= 1, so the root is bracketed; Newton steps are taken only while
    they stay inside the bracket and at least halve the previous step, otherwise the
    bracket is bisected (plain Newton cycles between the two flat tails of the
    sigmoid when sigma/rho is large).  This is the algorithm the HIP kernel runs per
    lane."""
    sigma = np.asarray(sigma, dtype=np.float64)
    m = np.asarray(m, dtype=np.float64)
    sigma, m = np.broadcast_arrays(sigma, m)
    lo = m - sigma / rho
    hi = m.copy()
    x = hi.copy()
    dxold = hi - lo
    dx = dxold.copy()
    g = sigma * sigmoid(x) + rho * (x - m)
    h = sigma * dsigmoid(x) + rho
    active = np.ones(x.shape, dtype=bool)
    for _ in range(iters):
        bis = (((x - hi) * h - g) * ((x - lo) * h - g) > 0) | (np.abs(2.0 * g) > np.abs(dxold * h))
        dxold = dx
        dx_b = 0.5 * (hi - lo)
        with np.errstate(divide="ignore", invalid="ignore"):
            dx_n = g / h
        dx = np.where(bis, dx_b, dx_n)
        xn = np.where(bis, lo + dx_b, x - dx_n)
        active = active & (xn != x) & (g != 0)
        if not active.any():
            break
        x = np.where(active, xn, x)
        g = sigma * sigmoid(x) + rho * (x - m)
        h = sigma * dsigmoid(x) + rho
        lo = np.where(active & (g < 0), x, lo)
        hi = np.where(active & (g >= 0), x, hi)
    return x


def prox_hinge_exact(sigma, rho, m):
    """Exact minimiser of sigma*max(0,1+x) + rho/2 (x-m)^2: what the reference's
    bisection of individual_solver.py:11-42 converges to.
    x = m - sigma/rho if that is >= -1;  m if m <= -1;  else -1."""
    sigma = np.asarray(sigma, dtype=np.float64)
    m = np.asarray(m, dtype=np.float64)
    a = m - sigma / rho
    return np.where(a >= -1.0, a, np.where(m <= -1.0, m, -1.0))


def prox_exact(loss, sigma, rho, m):
    if loss == "binary_cross_entropy":
        return prox_bce_exact(sigma, rho, m)
    if loss == "hinge":
        return prox_hinge_exact(sigma, rho, m)
    raise ValueError(
        f"Unrecognized loss '{loss}'! Options: ['binary_cross_entropy', 'multinomial_cross_entropy','hinge']"
    )


# ---------------------------------------------------------------- faithful mode
def hinge_vec_fun(sigma, rho, m, z):
    # individual_solver.py:11-13
    return sigma * np.maximum(np.sign(z + 1), 0) + rho * (z - m)


def prox_hinge_faithful(sigma, rho, m, max_iter=50, tol=1e-5):
    """individual_solver.py:15-42 (vec_bisect_method), including the early exit on
    the batch SUM of |f(mid)| < tol (:23) that makes small batches inexact."""
    sigma = np.asarray(sigma, dtype=np.float64).reshape(-1)
    m = np.asarray(m, dtype=np.float64).reshape(-1)
    lb = m - sigma / rho - 1
    ub = m + 1
    mid = (lb + ub) / 2
    i = 0
    while i < max_iter:
        f = hinge_vec_fun(sigma, rho, m, mid)
        if np.sum(np.abs(f)) < tol:
            return mid
        pos = f > 0
        lb = np.where(pos, lb, mid)
        ub = np.where(pos, mid, ub)
        mid = (lb + ub) / 2
        i += 1
    return mid


def prox_bce_faithful(sigma, rho, m, tol=1e-6, maxiter=50, shrink=0.7):
    """individual_solver.py:90-109 (newton_method) as called from :112-117:
    start x=m, full-vector Newton direction, ONE Armijo step size for the whole
    batch (c=1e-4) on the summed objective (:60-62), stop when the batch
    ||delta||_2 < tol.  The >1e5 fallback (:105-108) is dead in practice."""
    sigma = np.asarray(sigma, dtype=np.float64).reshape(-1)
    m = np.asarray(m, dtype=np.float64).reshape(-1)

    def fun(z):
        return (sigma * _log1exp_ref(z)).sum() + rho / 2 * np.dot(z - m, z - m)

    x = m.copy()
    for _ in range(maxiter):
        rtx = sigma * sigmoid(x) + rho * (x - m)
        invg = 1.0 / (sigma * dsigmoid(x) + rho)
        delta = -rtx * invg
        alpha = 1.0
        tempx = x + alpha * delta
        pre = fun(x)
        slope = np.dot(rtx, delta)
        while fun(tempx) > pre + 0.0001 * alpha * slope:
            alpha *= shrink
            tempx = x + alpha * delta
        x = tempx
        nd = np.linalg.norm(delta)
        if nd < tol:
            return x
        if nd > 1e5:
            break
    return x


def prox_faithful(loss, sigma, rho, m):
    # individual_solver.py:112-130
    if loss == "binary_cross_entropy":
        return prox_bce_faithful(sigma, rho, m)
    if loss == "hinge":
        return prox_hinge_faithful(sigma, rho, m)
    raise ValueError(
        f"Unrecognized loss '{loss}'! Options: ['binary_cross_entropy', 'multinomial_cross_entropy','hinge']"
    )
