"""Drop-in mirror of the reference's ``src/optim/objective.py`` public surface
(rankbasedObjective, get_weights and the weight generators), evaluated on the GPU.

``rankbasedObjective.get_arrogate_loss(w)`` = sum_i alphas_i * loss_(i)(w) + regulariser
(reference objective.py:71-87): one HBM sweep v = D w, the per-sample loss, a radix sort
when the weights are not constant, and a dot product - all in librbl.so.  ``alphas`` /
``betas`` are torch float64 (n,1) tensors like the reference's (objective.py:49-54).
"""
import numpy as np

try:
    from ... import _lib, _solver
except ImportError:      # package directory on sys.path: imported as ``src.optim.objective``
    import _lib
    import _solver


def get_weights(name, args=None):
    """objective.py:166-187: returns a generator ``n -> torch weights`` (a pair for ehrm)."""
    import torch
    if name not in _lib.WEIGHT:
        if name not in ("erm", "ehrm") and args is None:
            raise ValueError("args for framework is None!")
        raise ValueError(
            f"Unrecognized framework '{name}'! Options: ['erm','extremile','superquantile','esrm','aorr','aorr_dc','ehrm']")
    if name not in ("erm", "ehrm") and args is None:
        raise ValueError("args for framework is None!")
    if name == "ehrm":
        return (lambda n: torch.from_numpy(_lib.k_weights("ehrm", n)[0]),
                lambda n: torch.from_numpy(_lib.k_weights("ehrm", n)[1]))
    return lambda n: torch.from_numpy(_lib.k_weights(name, n, args)[0])


def get_erm_weights(n):
    return get_weights("erm")(n)


def get_extremile_weights(n, r):
    return get_weights("extremile", [r])(n)


def get_superquantile_weights(n, q):
    return get_weights("superquantile", [q])(n)


def get_esrm_weights(n, rho):
    return get_weights("esrm", [rho])(n)


def get_aorr_weights(n, qlow, qup):
    return get_weights("aorr", [qlow, qup])(n)


def get_aorr_dc_weights(n, k, m):
    if k <= m:
        raise ValueError("need args[0] > args[1]!")
    return get_weights("aorr_dc", [k, m])(n)


def get_cpt_weights_a(n):
    return get_weights("ehrm")[0](n)


def get_cpt_weights_b(n):
    return get_weights("ehrm")[1](n)


class rankbasedObjective:
    """objective.py:39-94.  Holds D = -y*X on the device (own handle, or the solver's)."""

    def __init__(self, X, y, weight_function="erm", loss="binary_cross_entropy", l2_reg=None, l1_reg=None,
                 B=None, n_class=None, args=None, storage="f32", device=0, _shared_solver=None):
        _solver.check_problem(weight_function, loss, B, args, need_prox=False)
        if loss == "multinomial_cross_entropy":
            raise ValueError("multinomial_cross_entropy is outside the ADMM hot path (binary losses only)")
        self.weight_function_name = weight_function
        self.loss_name = loss
        self.l2_reg, self.l1_reg = l2_reg, l1_reg
        self.B = B
        self.lossB = None if B is None else float(np.logaddexp(0.0, B))     # objective.py:59-63
        self.n_class = n_class
        if _shared_solver is not None:
            self._s = _shared_solver
            self.n, self.d = self._s.n_total, self._s.d
        else:
            Xm = _solver._as_matrix(X)
            self.n, self.d = Xm.shape
            self._s = _solver.Solver(self.n, self.d, weight_function, loss, args=args, B=B, storage=storage,
                                     device=device, objective_only=True)
            self._s.set_data(Xm, y)
        self._alphas = self._betas = None

    def _sig(self):
        if self._alphas is None:
            import torch
            a, b = self._s.sigma()
            self._alphas = torch.from_numpy(a).reshape(-1, 1)
            self._betas = self._alphas if self.weight_function_name != "ehrm" else torch.from_numpy(b).reshape(-1, 1)
        return self._alphas, self._betas

    @property
    def alphas(self):
        return self._sig()[0]

    @property
    def betas(self):
        return self._sig()[1]

    def get_arrogate_loss(self, w, include_reg=True):
        """objective.py:71-87 (betas = alphas there, :76, so the EHRM split sums to the same dot)."""
        wv = w.detach().cpu().numpy() if hasattr(w, "detach") else np.asarray(w)
        wv = np.asarray(wv, dtype=np.float64).reshape(-1)
        risk = self._s.risk(wv)
        if include_reg:
            risk += _solver.reg_terms(wv, self.l2_reg, self.l1_reg)
        return risk
