"""The ADMM iteration of the reference, restated.

Follows src/optim/algorithms.py: state initialisation :20-75, one iteration
(Optimizer.main_loop) :119-164, the ADMMmethod / smoothADMMmethod loops :209-216 /
:248-260.  ``mode='exact'`` solves every sub-problem to machine precision and reuses
v = D w (two sweeps of D per iteration); ``mode='faithful'`` restates the reference's
inner solvers and its three GEMVs per iteration.  Test infrastructure only.
"""
import time
import numpy as np

from . import weights as _w
from . import pav as _pav
from . import wstep as _ws
from . import objective as _obj


class Trace(dict):
    __getattr__ = dict.__getitem__


def initial_rho(weight_function):
    # algorithms.py:47-52
    if weight_function == "ehrm":
        return 1e-4
    if weight_function in ("aorr", "aorr_dc"):
        return 2e-7
    return 1e-5


def next_rho(rho, primal, d):
    # algorithms.py:154-157: the only live branch (self.loss is never a weight name, :148-152)
    return min(rho * (1.02 if primal > 1e-2 else 1.07), 217.0 * d)


def z_step_exact(weight_function, loss, sigma_a, sigma_b, B, rho, m, use_c=True):
    """algorithms.py:88-106 with exact solves.  Stable (m, index) order; erm needs no
    sort or PAV because the prox is monotone in m when sigma is constant."""
    from .prox import prox_exact
    if weight_function == "erm":
        return prox_exact(loss, sigma_a, rho, m), None
    order = np.argsort(m, kind="stable")
    ms = m[order]
    if weight_function == "ehrm":
        zs, branch = _pav.ehrm_exact(sigma_a, sigma_b, B, rho, ms, use_c=use_c)
    else:
        f = _pav.pav_exact if use_c else _pav.pav_exact_py
        zs, _ = f(loss, sigma_a, rho, ms)
        branch = None
    z = np.empty_like(m)
    z[order] = zs
    return z, branch


def z_step_faithful(weight_function, loss, sigma_a, sigma_b, B, rho, m):
    # algorithms.py:92-104
    order = np.argsort(m)
    ms = np.sort(m)
    if weight_function == "ehrm":
        zs, _ = _pav.ehrm_faithful(sigma_a, sigma_b, B, rho, ms)
    else:
        zs, _ = _pav.pav_faithful(loss, sigma_a, rho, ms, maxiter=m.shape[0])
    z = np.zeros_like(m)
    z[order] = zs
    return z


def admm_solve(X, y, weight_function="erm", loss="binary_cross_entropy", l2_reg=None, l1_reg=None,
               B=None, args=None, w0=None, max_iter=200, tol=1e-4, mode="exact", smooth=False,
               t=1.0, store=True, use_c=True, w_tol=1e-14, timing=None, stamps=None):
    """Run the reference's solve loop.  Returns a Trace with per-iteration primal /
    dual residuals, rho (value used in the iteration), objective after the iteration,
    and the final state (w, z, lam, rho, iters, converged).  stamps: a list that receives time.perf_counter() at
    the start of the loop and at the end of every iteration (bench.py's cpu_baseline leg)."""
    if loss not in ("binary_cross_entropy", "hinge"):
        raise ValueError(
            f"Unrecognized loss '{loss}'! Options: ['binary_cross_entropy', 'multinomial_cross_entropy', 'hinge']")
    if B is not None and loss != "binary_cross_entropy":
        raise ValueError("erhm only can be with the binary_cross_entropy.")   # objective.py:57-58
    if B is not None and weight_function != "ehrm":
        raise ValueError(f"Unrecognized weight_function '{weight_function}'! Options: ['ehrm']")
    X = np.asarray(X, dtype=np.float64)
    n, d = X.shape
    sigma_a, sigma_b = _w.get_weights(weight_function, n, args)
    D = -np.asarray(y).reshape(-1, 1) * X                  # :23
    G = D.T @ D                                            # :24
    reg = l1_reg or l2_reg                                 # :30
    lam = 0.1 * reg / n * np.ones(n)                       # :32
    z = 0.1 * reg / n * np.ones(n)                         # :34
    w = (np.asarray(w0, dtype=np.float64).reshape(-1).copy() if w0 is not None
         else 0.001 * reg / d / n * np.ones(d))            # :39-42
    rho = initial_rho(weight_function)
    w_flag = 1 if l1_reg is not None else 2                # :57-60
    L = 1.0001 * _ws.lambda_max(G) if mode == "exact" else None

    def F(wv, v=None):
        v = D @ wv if v is None else v
        return _obj.objective_from_v(loss, sigma_a, v, wv, l2_reg, l1_reg)

    tr = Trace(primal=[], dual=[], rho=[], objective=[F(w)] if store else [], branch=[],
               z_time=0.0, w_time=0.0, inner=[])
    v = D @ w
    converged = False
    it = 0
    if stamps is not None:
        stamps.append(time.perf_counter())
    for it in range(max_iter):
        t0 = time.perf_counter()
        # ---- z-step (:88-106)
        if mode == "exact":
            m = v - lam / rho
            z, br = z_step_exact(weight_function, loss, sigma_a, sigma_b, B, rho, m, use_c)
            tr.branch.append(br)
        else:
            # (n,1)/(d,1) column shapes as in the reference: same BLAS calls, same roundings
            m = (D @ w.reshape(-1, 1) - lam.reshape(-1, 1) / rho).reshape(-1)
            z = z_step_faithful(weight_function, loss, sigma_a, sigma_b, B, rho, m)
        t1 = time.perf_counter()
        # ---- w-step (:109-116, :190-207, :238-246)
        pre_w = w.copy()
        c = z + lam / rho
        if mode == "exact":
            q = D.T @ c
            if w_flag == 1 and not smooth:
                w, k = _ws.lasso_gram_exact(G, q, reg / (2.0 * rho), w, L, tol=w_tol)
            elif w_flag == 1:
                w, k = _ws.smooth_l1_gram_exact(G, q, rho, reg, t, w, L, tol=w_tol)
            else:
                w, k = _ws.ridge_gram_exact(G, q, rho, reg), 1
            tr.inner.append(k)
        else:
            if w_flag == 1 and not smooth:
                if n <= 500 and d <= 60:                   # :194-197
                    from sklearn.linear_model import Lasso
                    mdl = Lasso(alpha=reg / (2 * rho * n), tol=1e-8, fit_intercept=False,
                                max_iter=50000, warm_start=True)
                    mdl.fit(X=D, y=c)
                    w = mdl.coef_.reshape(-1).astype(np.float64)
                else:                                      # :199-202
                    w32, _ = _ws.fista_faithful(w, D.astype(np.float32), c, reg / (2 * rho))
                    w = w32.astype(np.float64)
            elif w_flag == 1:
                w = _ws.smooth_l1_lbfgs_faithful(w, z, lam, rho, G, D, reg, t)
            else:
                w = _ws.ridge_lbfgs_faithful(w, z, lam, rho, G, D, reg)
        t2 = time.perf_counter()
        tr.z_time += t1 - t0
        tr.w_time += t2 - t1
        # ---- dual update and residuals (:132-136)
        if mode == "exact":
            v = D @ w
            lam = lam + rho * (z - v)
            primal = float(np.linalg.norm(z - v))
        else:
            w2 = w.reshape(-1, 1)
            lam = (lam.reshape(-1, 1) + rho * (z.reshape(-1, 1) - D @ w2)).reshape(-1)
            primal = float(np.linalg.norm(z.reshape(-1, 1) - D @ w2))
            v = None
        dual = float(np.linalg.norm(w - pre_w))
        tr.primal.append(primal)
        tr.dual.append(dual)
        tr.rho.append(rho)
        if primal < tol and dual < tol:                    # :137-141
            converged = True
            if stamps is not None:
                stamps.append(time.perf_counter())
            break
        rho = next_rho(rho, primal, d)                     # :154-157
        if store:
            tr.objective.append(F(w, v))                   # :159-161
        if stamps is not None:
            stamps.append(time.perf_counter())
        if smooth and it >= 17:                            # :254-255
            t = max(t * 0.9, 1e-9) % np.power(rho, -0.1) * np.power(float(it), -0.1)
    if smooth and w_flag == 1:                             # :257-258
        w = np.sign(w) * np.where((np.abs(w) - t) > 0, np.abs(w) - t, 0)
    tr.update(w=w, z=z, lam=lam, rho_final=rho, iters=it + 1, converged=converged,
              final_objective=F(w), t=t, n=n, d=d)
    if timing is not None:
        timing.update(z_time=tr.z_time, w_time=tr.w_time)
    return tr
