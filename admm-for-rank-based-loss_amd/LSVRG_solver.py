"""Drop-in mirror of the reference's ``LSVRG_solver.py``: ``LSVRGmethod`` with the reference's
signature, return values and printed fields (LSVRG_solver.py:9-98), the LSVRG method of
``existing_methods/lerm_main/src/optim/algorithms.py:150-253`` running on the GPU (include/rbl.h:
rbl_bl_lsvrg_epoch: the checkpoint - full-batch losses, stable sort, X^T c - with the library's sweep and sort
kernels, the 100 single-sample steps of an epoch in one launch).  The sample indices come from the reference's own
generators, drawn the same way: ``numpy.random.RandomState(25).randint`` for ``uniform``, the global
``numpy.random.choice(n, p=alphas)`` otherwise (unseeded in the reference; seed numpy to reproduce a run), and
``torch.rand(1)`` per step for the l1 subgradient at 0."""
import time

import numpy as np
import torch

try:
    from . import _baselines
except ImportError:      # package directory on sys.path: imported as ``LSVRG_solver``
    import _baselines


def LSVRGmethod(X, y, weight_function, loss, l2_reg=None, l1_reg=None, lossB=None,
                max_iter=20, lr=0.01, train_loss=None, test_loss=None, uniform=None, verbose=True, args=None):
    X = np.asarray(X.detach().cpu().numpy() if hasattr(X, "detach") else X, dtype=np.float64)
    n, d = X.shape
    if weight_function not in ("erm", "ehrm") and args is None:
        raise ValueError("args for framework is None")                            # LSVRG_solver.py:31-32
    alphas, betas = _baselines.competitor_weights(weight_function, n, args)
    if weight_function != "ehrm":
        betas, lossB = None, None
    lr = _baselines.step_size(lr, n, d)                                           # :62-66
    opt = _baselines.Baseline(X, y, loss, l2_reg=l2_reg, l1_reg=l1_reg, lossB=lossB)
    rng = np.random.RandomState(25)                                               # algorithms.py:175 (seed=25)
    epoch_len = 100                                                               # LSVRG_solver.py:68
    p = np.asarray(alphas, dtype=np.float64)

    def wt():
        return torch.from_numpy(opt.w.reshape(-1, 1))

    if test_loss is not None:
        train_losses = [train_loss(wt())]                                         # :70-71 (as the reference writes it)
    test_losses = [test_loss(wt())]
    t_array = [0]
    t_start = time.time()
    for it in range(max_iter):
        samples = np.empty(epoch_len, dtype=np.int32)
        rands = np.empty(epoch_len, dtype=np.float32) if l1_reg else None
        for s in range(epoch_len):                                                # the draws of LSVRG.step, :201-206, :247
            samples[s] = rng.randint(0, n) if uniform else np.random.choice(n, p=p)
            if l1_reg:
                rands[s] = float(torch.rand(1))
        opt.lsvrg_epoch(alphas, betas, samples, bool(uniform), lr, rands)         # start_epoch + epoch_len x step
        if test_loss is not None:
            train_losses.append(train_loss(wt()))
        test_losses.append(test_loss(wt()))
        t_array.append(time.time() - t_start)
        if verbose:
            if it % 10 == 0:
                print("iter:", it, "train loss:", train_losses[-1], "test loss:", test_losses[-1], "time:", t_array[-1])
    w = opt.w.reshape(-1, 1)
    opt.close()
    if train_loss is not None:
        return w, train_losses, test_losses, t_array
    return w
