"""MI355X-native ADMM for rank-based loss minimisation (hot path of
RufengXiao/ADMM-for-rank-based-loss behind the reference's own class API).

    from admm_for_rank_based_loss_amd import ADMMmethod, smoothADMMmethod

or, drop-in for the reference's drivers, put this directory on PYTHONPATH and keep
``from src.optim.algorithms import ADMMmethod, smoothADMMmethod``.

Everything numerical runs in ``csrc/librbl.so`` (hand-written HIP for gfx950, C ABI in
``include/rbl.h``); importing this package never falls back to a CPU path.
"""
from . import _lib                                   # noqa: F401
from ._solver import Solver                          # noqa: F401
from .src.optim.algorithms import Optimizer, ADMMmethod, smoothADMMmethod   # noqa: F401
from .src.optim.objective import rankbasedObjective, get_weights            # noqa: F401
from .SGD_solver import SGDmethod                    # noqa: F401  (competitor baselines)
from .LSVRG_solver import LSVRGmethod                # noqa: F401

__all__ = ["ADMMmethod", "smoothADMMmethod", "Optimizer", "rankbasedObjective", "get_weights", "Solver", "SGDmethod", "LSVRGmethod"]
