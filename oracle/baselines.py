"""The reference's competitor baselines, restated.

``SGD_solver.py:9-96`` (mini-batch stochastic subgradient on the rank-based objective) and
``LSVRG_solver.py:9-98`` (loopless-style SVRG with a full-batch checkpoint per epoch), which drive
``existing_methods/lerm_main/src/optim/algorithms.py:54-98`` (StochasticSubgradientMethod) and ``:150-253``
(LSVRG) on ``existing_methods/lerm_main/src/optim/objective.py:41-112`` (ORMObjective).  The reference
differentiates ``dot(alphas, sort(losses))`` with autograd; this restatement uses the closed form
    d/dw sum_k a_k loss_(k) = sum_i a_{rank(i)} loss_i'(x_i . w) x_i        (stable ranks: ties by index)
and keeps the reference's behaviour where it is peculiar:
  * labels: ``if loss == 'logistic' or "binary_cross_entropy"`` is always true (SGD_solver.py:13,
    LSVRG_solver.py:14), so -1 becomes 0 for the hinge loss as well: hinge = max(1 - y01 * x.w, 0);
  * a mini-batch of b rows is weighted with the b-sample weights ``weight_function(b)`` (objective.py:72-75);
  * EHRM: sorted position k takes alpha_k if its loss is <= lossB, else beta_k (objective.py:84-86);
  * LSVRG, uniform sampling: the step uses ``alphas[i]`` with i the ROW index drawn, not its rank
    (algorithms.py:232-238);
  * the l1 subgradient at w_j = 0 is one ``torch.rand(1) * 2 - 1`` per step, shared by all zero coordinates, and it
    is drawn on every step (objective.py:100-106, algorithms.py:243-250).
Random streams are the reference's own: ``torch.manual_seed(25)`` + ``torch.randperm(n)`` per SGD epoch,
``numpy.random.RandomState(25).randint`` (uniform LSVRG), the GLOBAL ``numpy.random.choice(n, p=alphas)`` (non-uniform
LSVRG: unseeded in the reference - tests seed it) and the global ``torch.rand``.  They are drawn here with those
same calls and passed to the arithmetic as data, which is also how the device implementation receives them.
Test infrastructure only.
"""
import numpy as np

from . import weights as _w
from .prox import sigmoid, softplus


# ------------------------------------------------------------------ pieces shared with the device tests
def labels01(y):
    y = np.asarray(y, dtype=np.float64).reshape(-1).copy()
    y[y == -1] = 0                                   # SGD_solver.py:13-14 (always taken)
    return y


def sample_losses(loss, z, y01):
    if loss == "binary_cross_entropy":               # objective.py:11-15 (competitor's): BCE with logits
        return softplus(z) - y01 * z
    if loss == "hinge":                              # objective.py:36-37
        return np.maximum(1.0 - y01 * z, 0.0)
    raise ValueError(f"Unrecognized loss '{loss}'")


def loss_derivs(loss, z, y01):
    if loss == "binary_cross_entropy":
        return sigmoid(z) - y01
    t = 1.0 - y01 * z
    return np.where(t > 0, -y01, np.where(t == 0, -0.5 * y01, 0.0))     # torch.maximum splits the gradient at a tie


def rank_coefficients(losses, alphas, betas=None, lossB=None):
    """coefficient of loss_i in d risk / d loss_i: the weight of its stable rank (EHRM: alpha or beta by the
    sorted loss against lossB, objective.py:84-88)"""
    order = np.argsort(losses, kind="stable")
    srt = losses[order]
    wk = alphas if betas is None else np.where(srt <= lossB, alphas, betas)
    c = np.empty_like(losses)
    c[order] = wk
    return c, order


def reg_direction(w, l2_reg, l1_reg, n, rand01):
    g = np.zeros_like(w)
    if l2_reg:
        g += l2_reg * w / n
    if l1_reg:
        # the reference builds `res` as a float32 tensor (torch.zeros default dtype), so res * l1_reg / (2 n) is
        # float32 arithmetic before it is added to the float64 direction (objective.py:102-106)
        res = np.sign(w).astype(np.float32)
        res[w == 0] = np.float32(rand01) * np.float32(2.0) - np.float32(1.0)
        g += ((res * np.float32(l1_reg)) / np.float32(2 * n)).astype(np.float64)
    return g


def family_weights(weight_function, n, args):
    """the weight functions SGD_solver.py:16-60 builds (competitor's objective.py:115-199); aorr_dc differs from
    the ADMM side's generator: weights[m+1 : k+1] = 1/(k-m) (objective.py:169-172 of the competitor)"""
    if weight_function == "aorr_dc":
        k, m = int(args[0]), int(args[1])
        a = np.zeros(n)
        a[m + 1:k + 1] = 1.0 / (k - m)
        return a, None
    if weight_function == "ehrm":
        return _w.get_weights("ehrm", n, None)
    a, _ = _w.get_weights(weight_function, n, args)
    return a, None


def step_size(lr, n, d):
    return 1.0 / n if lr == 1 else (1.0 / (n * d) if lr == 2 else lr)      # SGD_solver.py:62-66


# ------------------------------------------------------------------ SGD
def sgd_epoch(X, y01, w, loss, lr, order, batch_size, steps, alphas_b, betas_b, lossB, l2_reg, l1_reg, rands):
    """StochasticSubgradientMethod.step x steps (algorithms.py:84-93) on the permutation `order` of one epoch"""
    n = X.shape[0]
    for s in range(steps):
        idx = order[s * batch_size: min(n, (s + 1) * batch_size)]
        Xb, yb = X[idx], y01[idx]
        z = Xb @ w
        losses = sample_losses(loss, z, yb)
        c, _ = rank_coefficients(losses, alphas_b, betas_b, lossB)
        g = Xb.T @ (c * loss_derivs(loss, z, yb))
        g += reg_direction(w, l2_reg, l1_reg, n, rands[s] if l1_reg else 0.0)
        w = w - lr * g
    return w


def sgd_solve(X, y, weight_function, loss, l2_reg=None, l1_reg=None, lossB=None, max_iter=20, batch_size=64, lr=0.01,
              args=None, log=None):
    """SGDmethod (SGD_solver.py:9-96).  `log(w)` (optional) is called at w0 and after every epoch."""
    import torch
    X = np.asarray(X, dtype=np.float64)
    y01 = labels01(y)
    n, d = X.shape
    lr = step_size(lr, n, d)
    steps = min(100, n // batch_size)                                  # algorithms.py:75-78 with epoch_len=100
    ab, bb = family_weights(weight_function, batch_size, args)
    if weight_function != "ehrm":
        bb, lossB = None, None
    w = np.zeros(d)
    torch.manual_seed(25)                                              # algorithms.py:73
    hist = [log(w)] if log else []
    for _ in range(max_iter):
        order = torch.randperm(n).numpy()                              # algorithms.py:81
        rands = [float(torch.rand(1)) for _ in range(steps)] if l1_reg else None
        # (the reference interleaves the rand draws with the steps; they do not touch the permutation of this
        # epoch, and the generator state after the epoch is the same)
        w = sgd_epoch(X, y01, w, loss, lr, order, batch_size, steps, ab, bb, lossB, l2_reg, l1_reg, rands)
        if log:
            hist.append(log(w))
    return w, hist


# ------------------------------------------------------------------ LSVRG
def lsvrg_checkpoint(X, y01, w, loss, alphas, betas, lossB):
    """LSVRG.start_epoch (algorithms.py:183-196): full-batch subgradient and the stable order of the losses"""
    z = X @ w
    losses = sample_losses(loss, z, y01)
    c, order = rank_coefficients(losses, alphas, betas, lossB)
    return X.T @ (c * loss_derivs(loss, z, y01)), order


def lsvrg_epoch(X, y01, w, loss, lr, samples, uniform, alphas, betas, lossB, l2_reg, l1_reg, rands):
    n = X.shape[0]
    g_chk, order = lsvrg_checkpoint(X, y01, w, loss, alphas, betas, lossB)
    w_chk = w.copy()
    for s, i in enumerate(samples):
        row = int(i) if uniform else int(order[int(i)])                # algorithms.py:201-208
        x, yy = X[row], y01[row]
        z, zc = float(x @ w), float(x @ w_chk)
        diff = (loss_derivs(loss, np.array([z]), np.array([yy]))[0]
                - loss_derivs(loss, np.array([zc]), np.array([yy]))[0]) * x
        if uniform:
            if betas is not None:
                lcur = sample_losses(loss, np.array([z]), np.array([yy]))[0]
                scale = n * (alphas[int(i)] if lcur <= lossB else betas[int(i)])      # algorithms.py:232-236
            else:
                scale = n * alphas[int(i)]                                            # :238
            direction = scale * diff + g_chk
        else:
            direction = diff + g_chk                                                  # :240
        direction = direction + reg_direction(w, l2_reg, l1_reg, n, rands[s] if l1_reg else 0.0)
        w = w - lr * direction
    return w


def lsvrg_solve(X, y, weight_function, loss, l2_reg=None, l1_reg=None, lossB=None, max_iter=20, lr=0.01, uniform=None,
                args=None, log=None):
    """LSVRGmethod (LSVRG_solver.py:9-98)"""
    import torch
    X = np.asarray(X, dtype=np.float64)
    y01 = labels01(y)
    n, d = X.shape
    lr = step_size(lr, n, d)
    alphas, betas = family_weights(weight_function, n, args)
    if weight_function != "ehrm":
        betas, lossB = None, None
    rng = np.random.RandomState(25)                                    # algorithms.py:175
    w = np.zeros(d)
    hist = [log(w)] if log else []
    for _ in range(max_iter):
        samples, rands = [], []
        for _s in range(100):                                          # epoch_len=100, LSVRG_solver.py:68
            samples.append(rng.randint(0, n) if uniform else np.random.choice(n, p=alphas))
            if l1_reg:
                rands.append(float(torch.rand(1)))
        w = lsvrg_epoch(X, y01, w, loss, lr, samples, bool(uniform), alphas, betas, lossB, l2_reg, l1_reg,
                        rands if l1_reg else None)
        if log:
            hist.append(log(w))
    return w, hist
