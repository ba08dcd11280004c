"""Drop-in mirror of the reference's ``src/optim/algorithms.py`` class API
(Optimizer / ADMMmethod / smoothADMMmethod), running the ADMM iteration on MI355X.

Same constructor arguments, methods, attributes, printed fields and error messages as
the reference (citations: reference ``src/optim/algorithms.py``), so ``run_SRM.py`` /
``run_EHRM.py`` / ``run_AoRR_*.py``-style callers work unchanged with this package
directory on ``PYTHONPATH``.  State (w, z, lambda) lives on the GPU; the ``w`` / ``z`` /
``lagrangian`` attributes download it on access.  All arithmetic is done by librbl.so
(include/rbl.h); this file is host glue.  Extra keyword arguments (after the reference's
own): ``storage`` ("f32" default | "f64" strict), ``device``.
"""
import time

import numpy as np

try:
    from ... import _lib, _solver
    from .objective import rankbasedObjective
except ImportError:      # package directory on sys.path: imported as ``src.optim.algorithms``
    import _lib
    import _solver
    from src.optim.objective import rankbasedObjective


class _OnDevice:
    """Marker returned by the default sub-problem hooks: the result already sits on the GPU."""
    __slots__ = ()


_ON_DEVICE = _OnDevice()


class Optimizer:
    def __init__(self, X, y, weight_function="erm", loss="binary_cross_entropy", l2_reg=None, l1_reg=None,
                 B=None, n_class=None, args=None, w0=None, max_iter=200, tol=1e-4, storage="f32", device=0,
                 _wstep=None, _smooth_t=1.0):
        # argument checks in the reference's order (objective first :22, then :55-68)
        _solver.check_problem(weight_function, loss, B, args)
        if l1_reg is None and l2_reg is None:
            raise ValueError("More arguments: l1_reg or l2_reg not l1_reg and l2_reg!")       # :62
        if B is not None and weight_function != "ehrm":
            raise ValueError(f"Unrecognized weight_function '{weight_function}'! Options: ['ehrm']")  # :65-68
        if weight_function == "ehrm" and B is None:
            raise ValueError("ehrm needs the reference point B")
        Xm = _solver._as_matrix(X)
        self.num_row, self.num_feature = Xm.shape                                             # :26-27
        self.reg = l1_reg or l2_reg                                                           # :30
        self.loss = loss                                                                      # :36
        self.tol, self.max_iter = tol, max_iter                                               # :44-45
        self.w_flag = 1 if l1_reg is not None else 2                                          # :57-60
        self.B = B
        self.w_tol = 7e-5                       # :69 (kept for callers; the GPU w-step is exact)
        self.z_maxiter = self.num_row           # :70 (the GPU PAV needs no sweep cap)
        self.store = False                                                                    # :71
        self.weight_function = weight_function                                                # :73
        self.l1_reg, self.l2_reg = l1_reg, l2_reg
        wstep = _wstep if _wstep is not None else (_lib.WSTEP_L1 if self.w_flag == 1 else _lib.WSTEP_L2)
        self._s = _solver.Solver(self.num_row, self.num_feature, weight_function, loss, reg=self.reg, wstep=wstep,
                                 B=B, args=args, smooth_t=_smooth_t, tol=tol, max_iter=max_iter, storage=storage,
                                 device=device)
        self._s.set_data(Xm, y)                                                               # :23 D = -y*X
        self._s.gram()                                                                        # :24 DTD
        if w0 is not None:                                                                    # :39-40
            self._s.set_state(w=np.asarray(w0, dtype=np.float64).reshape(-1))
        self.objective = rankbasedObjective(None, None, weight_function, loss, l2_reg, l1_reg, B, n_class, args,
                                            _shared_solver=self._s)                           # :22
        self._storage, self._device = storage, device
        self._last = None

    # ---- state views (reference attributes :32-52, :74-75) ---------------------------------
    @property
    def w(self):
        return self._s.get_state(want_z=False, want_lam=False)["w"].reshape(-1, 1)

    @w.setter
    def w(self, value):
        self._s.set_state(w=np.asarray(value, dtype=np.float64).reshape(-1))

    @property
    def z(self):
        return self._s.get_state(want_lam=False)["z"].reshape(-1, 1)

    @z.setter
    def z(self, value):
        if not isinstance(value, _OnDevice):
            self._s.set_state(z=np.asarray(value, dtype=np.float64).reshape(-1))

    @property
    def lagrangian(self):
        return self._s.get_state(want_z=False)["lam"].reshape(-1, 1)

    @lagrangian.setter
    def lagrangian(self, value):
        self._s.set_state(lam=np.asarray(value, dtype=np.float64).reshape(-1))

    @property
    def rho(self):
        return self._s.get_state(want_z=False, want_lam=False)["rho"]

    @rho.setter
    def rho(self, value):
        self._s.set_state(rho=float(value))

    @property
    def sigma_a(self):
        return self.objective.alphas.numpy().reshape(-1)

    @property
    def sigma_b(self):
        return self.objective.betas.numpy().reshape(-1)

    @property
    def D(self):
        return self._s.get_D()

    @property
    def DTD(self):
        D = self._s.get_D()
        return D.T @ D

    # ---- logging (:77-86) ------------------------------------------------------------------
    def start_store(self, X, y, weight_function="erm", loss="binary_cross_entropy", B=None, l2_reg=None,
                    l1_reg=None, n_class=None, args=None):
        self.test_objective = rankbasedObjective(X, y, weight_function, loss, l2_reg, l1_reg, B, n_class, args,
                                                 storage=self._storage, device=self._device)
        w = self.w
        self._s.profile_kernels(2)     # z_time / w_time below come from HIP events around the phases
        self.w_time = [0]
        self.z_time = [0]
        self.train_losses = [self.objective.get_arrogate_loss(w)]
        self.test_losses = [self.test_objective.get_arrogate_loss(w)]
        self.time_array = [0]
        self.store = True

    # ---- sub-problem hooks (:88-116, :186-207).  A subclass may override ``z_subproblem`` / ``w_subproblem`` (or
    # the ``_z_subproblem`` / ``_w_subproblem`` wrappers the reference's ADMMmethod defines, :186-207) and return
    # its own (n,1) / (d,1) array: main_loop then hands it to the library, which rebuilds what the later
    # phases derive from it (c = z + lambda/rho, ||z||^2; w_prev, the dual residual) - see
    # include/rbl.h: rbl_phase_z_external / rbl_phase_w_external.  The defaults leave the result on the GPU.
    def z_subproblem(self):
        self._s.phase_m()          # m = D w - lambda/rho                          :89
        self._s.phase_z()          # sort + PAV (prox only for erm) + scatter      :92-104
        return _ON_DEVICE

    def w_subproblem(self):
        self._s.phase_q()          # q = D^T (z + lambda/rho)
        self._s.phase_w()          # Gram-space lasso / ridge / smoothed-l1       :109-116, :190-207
        return _ON_DEVICE

    def _z_subproblem(self):
        return self.z_subproblem()

    def _w_subproblem(self):
        return self.w_subproblem()

    # ---- one iteration (:119-164) --------------------------------------------------------------
    def main_loop(self, i, t_start, verbose):
        z = self._z_subproblem()
        if not isinstance(z, _OnDevice):
            self._s.phase_z_external(np.asarray(z, dtype=np.float64).reshape(-1))
        w = self._w_subproblem()
        if not isinstance(w, _OnDevice):
            self._s.phase_w_external(np.asarray(w, dtype=np.float64).reshape(-1))
        need_obj = self.store or verbose
        self._s.phase_dual(want_objective=need_obj)    # v = D w, lambda += rho (z - v)    :132
        st = self._s.phase_finish()                    # residuals, stop test, rho rule    :135-157
        self._last = st
        if self.store:
            self.z_time.append(st.ms_z / 1e3 + self.z_time[i])
            self.w_time.append((st.ms_q + st.ms_w) / 1e3 + self.w_time[i])
        if st.converged:                                                                 # :137-141
            print('algorithm converges within tolerance')
            print('iter_num=', i, 'primal_feasibility: ', st.primal, 'dual_feasibility: ', st.dual)
            print('loss=', st.objective if need_obj else self.objective.get_arrogate_loss(self.w))
            return True
        if verbose and i % 10 == 0:                                                      # :142-145
            print('iter_num=', i, 'primal_feasibility: ', st.primal, 'dual_feasibility: ', st.dual)
            print('loss=', st.objective)
        if self.store:                                                                   # :159-162
            self.train_losses.append(st.objective)
            self.test_losses.append(self.test_objective.get_arrogate_loss(self.w))
            self.time_array.append(time.time() - t_start)
        return False

    def final_res(self):
        if self.store:
            return self.w, self.time_array, self.train_losses, self.test_losses          # :166-168
        raise ValueError("Data was not saved.")                                           # :170


class ADMMmethod(Optimizer):
    def __init__(self, X, y, weight_function="erm", loss="binary_cross_entropy", l2_reg=None, l1_reg=None, B=None,
                 n_class=None, args=None, w0=None, max_iter=200, tol=1e-4, storage="f32", device=0):
        super().__init__(X, y, weight_function, loss, l2_reg, l1_reg, B, n_class, args, w0, max_iter, tol,
                         storage=storage, device=device)

    def start_store(self, X, y, weight_function="erm", loss="binary_cross_entropy", B=None, l2_reg=None,
                    l1_reg=None, n_class=None, args=None):
        super().start_store(X, y, weight_function, loss, B, l2_reg, l1_reg, n_class, args)

    def main_loop(self, verbose=True):                                                    # :209-216
        t_start = time.time()
        for i in range(self.max_iter):
            if Optimizer.main_loop(self, i, t_start, verbose):
                break
        return self.w

    def final_res(self):
        return super().final_res()


class smoothADMMmethod(Optimizer):
    def __init__(self, X, y, weight_function="erm", loss="binary_cross_entropy", B=None, l2_reg=None, l1_reg=None,
                 n_class=None, args=None, w0=None, t=1, max_iter=200, tol=1e-4, storage="f32", device=0):
        wstep = _lib.WSTEP_SMOOTH_L1 if l1_reg is not None else None
        super().__init__(X, y, weight_function, loss, l2_reg, l1_reg, B, n_class, args, w0, max_iter, tol,
                         storage=storage, device=device, _wstep=wstep, _smooth_t=float(t))

    @property
    def t(self):
        return self._s.get_state(want_z=False, want_lam=False)["smooth_t"]

    @t.setter
    def t(self, value):
        self._s.set_state(smooth_t=float(value))

    def start_store(self, X, y, weight_function="erm", loss="binary_cross_entropy", B=None, l2_reg=None,
                    l1_reg=None, n_class=None, args=None):
        super().start_store(X, y, weight_function, loss, B, l2_reg, l1_reg, n_class, args)

    def main_loop(self, verbose=True):                                                    # :248-260
        t_start = time.time()
        for i in range(self.max_iter):
            # the t schedule of :254-255 is applied inside the library's phase_finish
            if Optimizer.main_loop(self, i, t_start, verbose):
                break
        if self.w_flag == 1:
            self._s.finalize_smooth()                                                      # :257-258
            print('final true loss=', self.objective.get_arrogate_loss(self.w))
        return self.w

    def final_res(self):
        return super().final_res()
