"""Drop-in mirror of the reference's ``SGD_solver.py``: ``SGDmethod`` with the reference's
signature, return values and printed fields (SGD_solver.py:9-96), the mini-batch stochastic subgradient method of
``existing_methods/lerm_main/src/optim/algorithms.py:54-98`` running on the GPU (include/rbl.h: rbl_bl_sgd_epoch,
one launch per epoch).  The permutation of every epoch and the random sign of the l1 subgradient come from the same
``torch`` calls, in the same order, as in the reference (``torch.manual_seed(25)``, ``torch.randperm(n)``,
``torch.rand(1)``), so a run is reproducible against it.  With this package directory in front of the reference on
``PYTHONPATH``, ``from SGD_solver import SGDmethod`` resolves here."""
import time

import numpy as np
import torch

try:
    from . import _baselines
except ImportError:      # package directory on sys.path: imported as ``SGD_solver``
    import _baselines


def SGDmethod(X, y, weight_function, loss, l2_reg=None, l1_reg=None, lossB=None,
              max_iter=20, batch_size=64, lr=0.01, train_loss=None, test_loss=None, verbose=True, args=None):
    X = np.asarray(X.detach().cpu().numpy() if hasattr(X, "detach") else X, dtype=np.float64)
    n, d = X.shape
    ab, bb = _baselines.competitor_weights(weight_function, batch_size, args)     # objective.py:72-75: b-sample weights
    if weight_function != "ehrm":
        bb, lossB = None, None
    lr = _baselines.step_size(lr, n, d)                                           # :62-66
    opt = _baselines.Baseline(X, y, loss, l2_reg=l2_reg, l1_reg=l1_reg, lossB=lossB)
    torch.manual_seed(25)                                                         # algorithms.py:73 (seed=25)
    steps = min(100, n // batch_size)                                             # algorithms.py:75-78, epoch_len=100

    def wt():
        return torch.from_numpy(opt.w.reshape(-1, 1))

    if train_loss is not None:
        train_losses = [train_loss(wt())]                                         # :70-71
    test_losses = [test_loss(wt())]
    t_array = [0]
    t_start = time.time()
    for it in range(max_iter):
        order = torch.randperm(n).numpy()                                         # start_epoch, algorithms.py:80-82
        rands = np.array([float(torch.rand(1)) for _ in range(steps)], dtype=np.float32) if l1_reg else None
        opt.sgd_epoch(order[: min(n, steps * batch_size)], steps, batch_size, ab, bb, lr, rands)   # :84-93 x epoch_len
        if train_loss is not None:
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
