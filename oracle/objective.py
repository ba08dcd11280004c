"""Rank-based objective  F(w) = sum_i sigma_i * loss_(i)(w) + regulariser.

Restates src/optim/objective.py:11-24 (per-sample losses) and :71-87
(get_arrogate_loss) of the reference.  With D = -y*X and v = D w:
  binary_cross_entropy_with_logits(Xw, y01) == softplus(v)   (objective.py:11-16)
  hinge: max(1 - y*Xw, 0) == max(1 + v, 0)                    (objective.py:23-24)
The logged objective uses betas = alphas (objective.py:76), so the EHRM split at
lossB (:77-79) sums to the same dot product.  Regulariser: 0.5*l2*||w||^2 and
0.5*l1*||w||_1 (:83-86).  Test infrastructure only - see oracle/__init__.py.
"""
import numpy as np
from .prox import softplus


def sample_losses(loss, v):
    if loss == "binary_cross_entropy":
        return softplus(v)
    if loss == "hinge":
        return np.maximum(1.0 + v, 0.0)
    raise ValueError(
        f"Unrecognized loss '{loss}'! Options: ['binary_cross_entropy', 'multinomial_cross_entropy', 'hinge']"
    )


def objective_from_v(loss, alphas, v, w, l2_reg=None, l1_reg=None, include_reg=True):
    losses = np.sort(sample_losses(loss, np.asarray(v, dtype=np.float64).reshape(-1)))
    risk = float(np.dot(alphas, losses))
    w = np.asarray(w, dtype=np.float64).reshape(-1)
    if l2_reg and include_reg:
        risk += 0.5 * l2_reg * float(np.sum(w ** 2))
    if l1_reg and include_reg:
        risk += 0.5 * l1_reg * float(np.sum(np.abs(w)))
    return risk


def objective(loss, alphas, X, y, w, l2_reg=None, l1_reg=None, include_reg=True):
    D = -np.asarray(y, dtype=np.float64).reshape(-1, 1) * np.asarray(X, dtype=np.float64)
    return objective_from_v(loss, alphas, D @ np.asarray(w, dtype=np.float64).reshape(-1), w,
                            l2_reg, l1_reg, include_reg)
