"""w-sub-problem solvers.

With c = z + lambda/rho, q = D^T c, G = D^T D the three w-steps of the reference are
  l1 (ADMMmethod):        min 1/2||c - D w||^2 + reg/(2 rho) ||w||_1
                          (src/optim/algorithms.py:190-202 -> src/util/fast_lasso.py:22-69)
  l2:                     min rho/2||D w - c||^2 + reg/2 ||w||^2
                          (src/util/w_LBFGS.py:31-53)
  smoothed l1 (sADMM):    min rho/2||D w - c||^2 + sum_j h_t(w_j)
                          (src/util/w_LBFGS.py:11-28,54-62)
``*_exact`` solve them in Gram space (d-space) to machine precision - the form the
GPU path uses; ``*_faithful`` restate the reference's own n-space solvers.
Test infrastructure only - see oracle/__init__.py.
"""
import numpy as np


def lambda_max(G, iters=200, seed=0):
    """Largest eigenvalue of the (PSD) Gram matrix by power iteration."""
    rng = np.random.default_rng(seed)
    v = rng.standard_normal(G.shape[0])
    v /= np.linalg.norm(v)
    lam = 0.0
    for _ in range(iters):
        gv = G @ v
        lam = float(np.linalg.norm(gv))
        if lam == 0.0:
            return 0.0
        v = gv / lam
    return lam


def soft_threshold(x, k):
    # fast_lasso.py:15-19 (soft_thr)
    return np.sign(x) * np.maximum(np.abs(x) - k, 0.0)


def huber_prox(b, L, reg, t):
    """argmin_u L/2 (u-b)^2 + h_t(u),  h_t(u) = reg*u^2/(4t) if |u|<=t else
    reg/2 (|u| - t/2)   (w_LBFGS.py:11-19)."""
    u_in = b * L / (L + reg / (2.0 * t))
    u_out = b - np.sign(b) * reg / (2.0 * L)
    return np.where(np.abs(u_in) <= t, u_in, u_out)


def _fista_gram(G, q, w0, L, prox, tol=1e-14, max_iter=200000):
    """FISTA with gradient-based adaptive restart on 1/2 w'Gw - q'w + psi(w);
    ``prox(b)`` = argmin L/2||u-b||^2 + psi(u).  Fixed step 1/L, L >= lambda_max(G).
    Stops when ||w_k - w_{k-1}||_inf <= tol*max(1, ||w_k||_inf).  Same iteration the
    HIP w-step kernels run."""
    w = np.array(w0, dtype=np.float64).reshape(-1)
    yk = w.copy()
    t = 1.0
    it = 0
    for it in range(1, max_iter + 1):
        grad = G @ yk - q
        wn = prox(yk - grad / L)
        dw = wn - w
        if np.dot(yk - wn, dw) > 0:      # O'Donoghue-Candes gradient restart
            t = 1.0
            yk = wn.copy()
        else:
            tn = (1.0 + np.sqrt(1.0 + 4.0 * t * t)) / 2.0
            yk = wn + ((t - 1.0) / tn) * dw
            t = tn
        w = wn
        if np.max(np.abs(dw)) <= tol * max(1.0, np.max(np.abs(w))):
            break
    return w, it


def lasso_gram_exact(G, q, kappa, w0, L=None, tol=1e-14):
    """min 1/2 w'Gw - q'w + kappa||w||_1  (kappa = reg/(2 rho))."""
    if L is None:
        L = 1.0001 * lambda_max(G)
    return _fista_gram(G, q, w0, L, lambda b: soft_threshold(b, kappa / L), tol=tol)


def ridge_gram_exact(G, q, rho, reg):
    """(rho G + reg I) w = rho q   (normal equations of w_LBFGS.py:31-44)."""
    d = G.shape[0]
    return np.linalg.solve(rho * G + reg * np.eye(d), rho * q)


def smooth_l1_gram_exact(G, q, rho, reg, t, w0, L=None, tol=1e-14):
    """min rho/2 w'Gw - rho q'w + sum_j h_t(w_j)  (w_LBFGS.py:11-28)."""
    if L is None:
        L = 1.0001 * lambda_max(G)
    Ls = rho * L
    return _fista_gram(rho * G, rho * q, w0, Ls, lambda b: huber_prox(b, Ls, reg, t), tol=tol)


def lasso_kkt_residual(G, q, kappa, w):
    """Distance of 0 from the sub-differential of 1/2 w'Gw - q'w + kappa||w||_1."""
    g = G @ w - q
    r = np.where(w != 0, g + kappa * np.sign(w), np.sign(g) * np.maximum(np.abs(g) - kappa, 0))
    return float(np.max(np.abs(r)))


# ---------------------------------------------------------------- faithful mode
def fista_faithful(beta, X32, y, lam, L=17.0, eta=2.5, tol=7e-5, max_iter=5000):
    """fast_lasso.py:22-69 restated in fp32 NumPy: backtracking FISTA in n-space,
    three n x d sweeps per inner iteration (:41,:43,:55).  ``X32`` is the fp32 copy of
    D (the reference re-makes it on every call, :33)."""
    f32 = np.float32
    dbeta = np.asarray(beta, dtype=f32).reshape(-1).copy()
    dy = np.asarray(y, dtype=f32).reshape(-1)
    t = f32(1.0)
    dbeta_p = dbeta.copy()
    dbeta_prev = dbeta.copy()
    L_prev = f32(L)
    eta = f32(eta)
    lam = f32(lam)
    sweeps = 0
    for _ in range(max_iter):
        r = dy - X32 @ dbeta_p
        drbp = np.dot(r, r)
        g = X32.T @ r
        sweeps += 2
        i_k = -1
        while True:
            i_k += 1
            L_cur = f32(L_prev * (eta ** i_k))
            bstar = dbeta_p + g / L_cur
            dbeta = (np.maximum(np.abs(bstar) - lam / L_cur, f32(0)) * np.sign(bstar)).astype(f32)
            diff = dbeta - dbeta_p
            rhs = L_cur * np.dot(diff, diff) - f32(2.0) * np.dot(diff, g)
            r2 = dy - X32 @ dbeta
            sweeps += 1
            lhs = np.dot(r2, r2) - drbp
            if not lhs > rhs:
                break
        L_prev = L_cur
        tnext = f32((1.0 + np.sqrt(f32(1) + f32(4) * t * t)) / 2.0)
        diff = dbeta - dbeta_prev
        dbeta_p = (dbeta + ((t - f32(1.0)) / tnext) * diff).astype(f32)
        if np.linalg.norm(diff) < tol:
            break
        t = tnext
        dbeta_prev = dbeta
    return dbeta, sweeps


def ridge_lbfgs_faithful(w0, z, lam, rho, G, D, reg):
    """w_LBFGS.py:31-53: SciPy L-BFGS-B (maxiter 1000) on the n-space objective."""
    from scipy.optimize import minimize
    # column shapes and operation order as in the reference, so the same BLAS routines
    # (and the SVD-based matrix 2-norm of an (n,1) array, :35) produce the same roundings:
    # the hinge bisection's early exit amplifies last-bit differences
    b = z.reshape(-1, 1) + lam.reshape(-1, 1) / rho

    def f(w):                                                   # :31-37
        w = w.reshape(-1, 1)
        temp = D @ w - b
        return 0.5 * rho * (np.linalg.norm(temp, ord=2) ** 2) + 0.5 * reg * np.sum(w * w)

    def g(w):                                                   # :40-45 (D.T @ b every call)
        w = w.reshape(-1, 1)
        return rho * (G @ w - D.T @ b) + reg * w

    res = minimize(f, np.asarray(w0, dtype=np.float64).reshape(-1), jac=g, method="L-BFGS-B",
                   options={"disp": False, "maxiter": 1000})
    return res.x


def smooth_l1_lbfgs_faithful(w0, z, lam, rho, G, D, reg, t):
    """w_LBFGS.py:11-28,54-62: SciPy L-BFGS-B on the Huber-smoothed l1 objective."""
    from scipy.optimize import minimize
    b = (z + lam / rho).reshape(-1)
    DTb = D.T @ b

    def f(w):
        r = D @ w - b
        a = np.abs(w)
        inner = a <= t
        return (0.5 * rho * np.dot(r, r) + 0.25 * reg * np.sum(w[inner] ** 2) / t
                + 0.5 * reg * np.sum(a[~inner] - 0.5 * t))

    def g(w):
        a = np.abs(w)
        return rho * (G @ w - DTb) + np.where(a <= t, 0.5 * reg * w / t, 0.5 * reg * np.sign(w))

    res = minimize(f, np.asarray(w0, dtype=np.float64).reshape(-1), jac=g, method="L-BFGS-B",
                   options={"disp": False, "maxiter": 1000})
    return res.x
