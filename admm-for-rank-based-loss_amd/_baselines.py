"""Host side of the competitor baselines (include/rbl.h: rbl_bl_*): argument handling, the reference's own
random streams, one C-ABI call per epoch.  Every number of the iteration is computed by librbl.so."""
import ctypes as C

import numpy as np

try:
    from . import _lib, _solver
except ImportError:      # package directory itself on sys.path
    import _lib
    import _solver


def competitor_weights(weight_function, n, args):
    """the n-sample weights SGD_solver.py:16-60 / LSVRG_solver.py:17-58 build from the competitor's generators
    (existing_methods/lerm_main/src/optim/objective.py:115-199).  All but aorr_dc are the formulas of the ADMM side
    (src/optim/objective.py:97-164: computed by rbl_k_weights); the competitor's aorr_dc is a plain ranked range."""
    if weight_function == "ehrm":
        return _lib.k_weights("ehrm", n, None)
    if weight_function == "erm":
        return _lib.k_weights("erm", n, None)[0], None
    if args is None:
        raise ValueError("args for framework is None!")                         # SGD_solver.py:33-34
    if weight_function == "aorr_dc":
        k, m = int(args[0]), int(args[1])
        a = np.zeros(n)
        a[m + 1:k + 1] = 1.0 / (k - m)                                           # objective.py:169-172 (competitor's)
        return a, None
    if weight_function in ("superquantile", "extremile", "esrm", "aorr"):
        return _lib.k_weights(weight_function, n, args)[0], None
    raise ValueError(f"weight_function '{weight_function}' is not supported! Options: "
                     "['erm','extremile','superquantile','esrm','aorr','aorr_dc','ehrm']")   # SGD_solver.py:49-51


class Baseline:
    """One rbl_baseline handle: X (n, d) float64 and labels; w starts at 0 (algorithms.py:64-71)."""

    def __init__(self, X, y, loss, l2_reg=None, l1_reg=None, lossB=None, device=0):
        self._h = None
        self.lib = _lib.load()
        X = _solver._as_matrix(X)
        y = np.asarray(y, dtype=np.float64).reshape(-1).copy()
        y[y == -1] = 0          # `if loss == 'logistic' or "binary_cross_entropy"` is always true (SGD_solver.py:13-14)
        if loss not in ("binary_cross_entropy", "hinge"):
            raise ValueError(f"Unrecognized loss '{loss}'! Options: ['binary_cross_entropy', 'hinge']")
        if y.shape[0] != X.shape[0]:
            raise ValueError(f"y has {y.shape[0]} labels for {X.shape[0]} rows")
        self.n, self.d = X.shape
        h = C.c_void_p()
        _lib.check(self.lib.rbl_bl_create(self.n, self.d, _lib.ptr(X), _lib.ptr(y), _lib.LOSS[loss],
                                          0 if lossB is None else 1, 0.0 if lossB is None else float(lossB),
                                          float(l2_reg or 0.0), float(l1_reg or 0.0), int(device), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self.lib.rbl_bl_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def w(self):
        out = np.empty(self.d)
        _lib.check(self.lib.rbl_bl_get_w(self._h, _lib.ptr(out)))
        return out

    @w.setter
    def w(self, value):
        v = _lib.f64(value).reshape(-1)
        if v.size != self.d:
            raise ValueError(f"w has {v.size} entries, expected {self.d}")
        _lib.check(self.lib.rbl_bl_set_w(self._h, _lib.ptr(v)))

    def sgd_epoch(self, order, steps, batch, alphas_b, betas_b, lr, rands=None):
        order = np.ascontiguousarray(order, dtype=np.int32)
        ab = _lib.f64(alphas_b)
        bb = _lib.f64(betas_b) if betas_b is not None else None
        r = np.ascontiguousarray(rands, dtype=np.float32) if rands is not None else None
        _lib.check(self.lib.rbl_bl_sgd_epoch(self._h, _lib.ptr(order), int(steps), int(batch), _lib.ptr(ab), _lib.ptr(bb),
                                             float(lr), _lib.ptr(r)))

    def lsvrg_epoch(self, alphas, betas, samples, uniform, lr, rands=None):
        a = _lib.f64(alphas)
        b = _lib.f64(betas) if betas is not None else None
        smp = np.ascontiguousarray(samples, dtype=np.int32)
        r = np.ascontiguousarray(rands, dtype=np.float32) if rands is not None else None
        _lib.check(self.lib.rbl_bl_lsvrg_epoch(self._h, _lib.ptr(a), _lib.ptr(b), _lib.ptr(smp), int(smp.size),
                                               1 if uniform else 0, float(lr), _lib.ptr(r)))


def step_size(lr, n, d):
    return 1.0 / n if lr == 1 else (1.0 / (n * d) if lr == 2 else lr)              # SGD_solver.py:62-66
