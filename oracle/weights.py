"""Rank-weight (sigma) generators, vectorised fp64 NumPy.

Restates src/optim/objective.py:97-187 of the reference (weights are listed in
ascending-rank order: sigma[0] multiplies the smallest loss).  Test
infrastructure only - see oracle/__init__.py.
"""
import math
import numpy as np

WEIGHT_FUNCTIONS = ("erm", "extremile", "superquantile", "esrm", "aorr", "aorr_dc", "ehrm")


def erm(n):
    # objective.py:97-98
    return np.ones(n, dtype=np.float64) / n


def extremile(n, r):
    # objective.py:101-105   ((i+1)^r - i^r) / n^r
    i = np.arange(n, dtype=np.float64)
    return ((i + 1.0) ** r - i ** r) / (n ** r)


def superquantile(n, q):
    # objective.py:108-117
    w = np.zeros(n, dtype=np.float64)
    idx = math.floor(n * q)
    frac = 1 - (n - idx - 1) / (n * (1 - q))
    if frac > 1e-12:
        w[idx] = frac
        w[idx + 1:] = 1 / (n * (1 - q))
    else:
        w[idx:] = 1 / (n - idx)
    return w


def esrm(n, rho):
    # objective.py:120-123
    i = np.arange(n, dtype=np.float64)
    upper = np.exp(rho * ((i + 1.0) / n))
    lower = np.exp(rho * (i / n))
    return math.exp(-rho) * (upper - lower) / (1 - math.exp(-rho))


def aorr(n, qlow, qup):
    # objective.py:126-136
    w = np.zeros(n, dtype=np.float64)
    lo = math.floor(n * qlow)
    up = math.floor(n * qup)
    frac = 1 - (up - lo - 1) / (n * (qup - qlow))
    if frac > 1e-12:
        w[lo] = frac
        w[lo + 1:up] = 1 / (n * (qup - qlow))
    else:
        w[lo:up] = 1 / (up - lo)
    return w


def aorr_dc(n, k, m):
    # objective.py:139-145
    if k <= m:
        raise ValueError("need args[0] > args[1]!")
    w = np.zeros(n, dtype=np.float64)
    w[m + 1:k] = 1 / (k - m)
    w[k + 1] = 1 - (k - m - 1) / (k - m)
    return w


def distort(p, gamma):
    # objective.py:148-150
    pg = p ** gamma
    return pg / ((pg + (1 - p) ** gamma) ** (1 / gamma))


def cpt_a(n):
    # objective.py:153-157  a_i = distort((i+1)/n, .69) - distort(i/n, .69)
    i = np.arange(n, dtype=np.float64)
    return distort((i + 1.0) / n, 0.69) - distort(i / n, 0.69)


def cpt_b(n):
    # objective.py:160-164  b_i = distort((n-i)/n, .61) - distort((n-i-1)/n, .61)
    i = np.arange(n, dtype=np.float64)
    return distort((n - i) / n, 0.61) - distort((n - i - 1.0) / n, 0.61)


def get_weights(name, n, args=None):
    """Return (alphas, betas) as the reference's rankbasedObjective does
    (objective.py:46-54, 166-187): betas is alphas unless name == 'ehrm'."""
    if name == "erm":
        a = erm(n)
        return a, a
    if name == "ehrm":
        return cpt_a(n), cpt_b(n)
    if args is None:
        raise ValueError("args for framework is None!")
    if name == "extremile":
        a = extremile(n, args[0])
    elif name == "superquantile":
        a = superquantile(n, args[0])
    elif name == "esrm":
        a = esrm(n, args[0])
    elif name == "aorr":
        a = aorr(n, args[0], args[1])
    elif name == "aorr_dc":
        a = aorr_dc(n, args[0], args[1])
    else:
        raise ValueError(
            f"Unrecognized framework '{name}'! Options: ['erm','extremile','superquantile','esrm','aorr','aorr_dc','ehrm']"
        )
    return a, a
