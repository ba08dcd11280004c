"""Thin object wrapper over a librbl solver handle (include/rbl.h).  Host logic only:
argument checking with the reference's error messages, buffer marshalling, and the
per-phase calls.  Every number is computed by the HIP library."""
import ctypes as C
import math

import numpy as np

try:
    from . import _lib
except ImportError:  # package directory itself on sys.path (drop-in `src.optim` layout)
    import _lib

_FAMILIES = "['erm','extremile','superquantile','esrm','aorr','aorr_dc','ehrm']"
_LOSSES = "['binary_cross_entropy', 'multinomial_cross_entropy', 'hinge']"


def check_problem(weight_function, loss, B, args, need_prox=True):
    """Argument validation in the order the reference performs it
    (rankbasedObjective.__init__ src/optim/objective.py:46-58, then
    Optimizer.__init__ src/optim/algorithms.py:64-68)."""
    if weight_function not in _lib.WEIGHT:
        if weight_function not in ("erm", "ehrm") and args is None:
            raise ValueError("args for framework is None!")                      # objective.py:171-172
        raise ValueError(f"Unrecognized framework '{weight_function}'! Options: {_FAMILIES}")  # :185-187
    if weight_function not in ("erm", "ehrm") and args is None:
        raise ValueError("args for framework is None!")
    if weight_function == "aorr_dc" and args[0] <= args[1]:
        raise ValueError("need args[0] > args[1]!")                              # objective.py:140-141
    if loss not in ("binary_cross_entropy", "multinomial_cross_entropy", "hinge"):
        raise ValueError(f"Unrecognized loss '{loss}'! Options: {_LOSSES}")      # objective.py:35-37
    if B is not None and loss != "binary_cross_entropy":
        raise ValueError("erhm only can be with the binary_cross_entropy.")      # objective.py:57-58
    if loss == "multinomial_cross_entropy" and need_prox:
        # the reference accepts it in the objective but its z-step has no prox for it
        # (src/util/individual_solver.py:124-125 is `pass`): unsupported there too
        raise ValueError(f"Unrecognized loss '{loss}'! Options: ['binary_cross_entropy', 'hinge'] for the ADMM z-step")


def _as_labels(y, n):
    y = np.asarray(y.detach().cpu().numpy() if hasattr(y, "detach") else y)
    y = np.ascontiguousarray(y.reshape(-1), dtype=np.float64)
    if y.shape[0] != n:
        raise ValueError(f"y has {y.shape[0]} labels for {n} rows")
    if np.all((y == 0) | (y == 1)):     # objective.py:12 turns -1 into 0 in place; accept both
        y = 2.0 * y - 1.0
    return y


def _as_matrix(X):
    X = X.detach().cpu().numpy() if hasattr(X, "detach") else np.asarray(X)
    if X.ndim != 2:
        raise ValueError("X must be a 2-D array")
    return np.ascontiguousarray(X, dtype=np.float64)


class Solver:
    """One librbl handle.  ``n`` local rows of an ``n_total``-row problem."""

    def __init__(self, n, d, weight_function="erm", loss="binary_cross_entropy", reg=0.0, wstep=_lib.WSTEP_L2,
                 B=None, args=None, smooth_t=1.0, rho0=0.0, tol=1e-4, w_tol=0.0, max_iter=200, storage="f32",
                 device=0, objective_only=False, n_total=None, row_offset=0):
        self._h = None
        self.lib = _lib.load()
        if storage not in _lib.STORAGE:
            raise ValueError(f"storage must be one of {sorted(_lib.STORAGE)}")
        cfg = _lib.RblConfig()
        cfg.n, cfg.d = int(n), int(d)
        cfg.n_total = int(n if n_total is None else n_total)
        cfg.row_offset = int(row_offset)
        cfg.loss = _lib.LOSS[loss]
        cfg.weight_function = _lib.WEIGHT[weight_function]
        a = list(args) if args is not None else []
        cfg.n_weight_args = min(len(a), 2)
        for k in range(cfg.n_weight_args):
            cfg.weight_args[k] = float(a[k])
        cfg.has_B = 0 if B is None else 1
        cfg.B = 0.0 if B is None else float(B)
        cfg.wstep = int(wstep)
        cfg.reg = float(reg)
        cfg.smooth_t = float(smooth_t)
        cfg.rho0 = float(rho0)
        cfg.tol = float(tol)
        cfg.w_tol = float(w_tol)
        cfg.max_iter = int(max_iter)
        cfg.storage = _lib.STORAGE[storage]
        cfg.device = int(device)
        cfg.objective_only = 1 if objective_only else 0
        self.cfg = cfg
        self.n, self.d, self.n_total = cfg.n, cfg.d, cfg.n_total
        h = C.c_void_p()
        _lib.check(self.lib.rbl_create(C.byref(cfg), C.byref(h)))
        self._h = h

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if getattr(self, "_h", None):
            self.lib.rbl_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---------------------------------------------------------------------- data
    def set_data(self, X, y):
        X = _as_matrix(X)
        y = _as_labels(y, X.shape[0])
        if X.shape != (self.n, self.d):
            raise ValueError(f"X is {X.shape}, expected {(self.n, self.d)}")
        _lib.check(self.lib.rbl_set_data(self._h, _lib.ptr(X), _lib.ptr(y), X.shape[1]))

    def generate_synthetic(self, seed=17, class_sep=1.0, flip_y=0.01):
        _lib.check(self.lib.rbl_generate_synthetic(self._h, int(seed), float(class_sep), float(flip_y)))

    def synth_local(self, seed=17, class_sep=1.0, flip_y=0.01):
        _lib.check(self.lib.rbl_synth_local(self._h, int(seed), float(class_sep), float(flip_y)))

    def synth_finish(self):
        _lib.check(self.lib.rbl_synth_finish(self._h))

    def labels(self):
        y = np.empty(self.n)
        _lib.check(self.lib.rbl_get_labels(self._h, _lib.ptr(y)))
        return y

    def gram(self):
        _lib.check(self.lib.rbl_gram_local(self._h))
        _lib.check(self.lib.rbl_gram_finish(self._h))

    def gram_local(self):
        _lib.check(self.lib.rbl_gram_local(self._h))

    def gram_finish(self):
        _lib.check(self.lib.rbl_gram_finish(self._h))

    def get_D(self):
        out = np.empty((self.n, self.d))
        _lib.check(self.lib.rbl_get_D(self._h, _lib.ptr(out)))
        return out

    def set_stream(self, stream_ptr):
        """hipStream_t as an integer (0 = the default stream); None = the handle's own stream."""
        p = C.c_void_p(-1) if stream_ptr is None else C.c_void_p(int(stream_ptr))
        _lib.check(self.lib.rbl_set_stream(self._h, p))

    # --------------------------------------------------------------------- state
    def get_state(self, want_z=True, want_lam=True):
        w = np.empty(self.d)
        z = np.empty(self.n) if want_z and not self.cfg.objective_only else None
        lam = np.empty(self.n) if want_lam and not self.cfg.objective_only else None
        rho, it, t = C.c_double(0), C.c_int64(0), C.c_double(0)
        _lib.check(self.lib.rbl_get_state(self._h, _lib.ptr(w), _lib.ptr(z), _lib.ptr(lam), C.byref(rho),
                                          C.byref(it), C.byref(t)))
        return dict(w=w, z=z, lam=lam, rho=rho.value, iter=it.value, smooth_t=t.value)

    def set_state(self, w=None, z=None, lam=None, rho=None, iter=None, smooth_t=None):
        w = _lib.f64(w).reshape(-1) if w is not None else None
        z = _lib.f64(z).reshape(-1) if z is not None else None
        lam = _lib.f64(lam).reshape(-1) if lam is not None else None
        for a, k, name in ((w, self.d, "w"), (z, self.n, "z"), (lam, self.n, "lam")):
            if a is not None and a.size != k:
                raise ValueError(f"{name} has {a.size} entries, expected {k}")
        r = C.byref(C.c_double(rho)) if rho is not None else None
        i = C.byref(C.c_int64(iter)) if iter is not None else None
        t = C.byref(C.c_double(smooth_t)) if smooth_t is not None else None
        _lib.check(self.lib.rbl_set_state(self._h, _lib.ptr(w), _lib.ptr(z), _lib.ptr(lam), r, i, t))

    def sigma(self):
        a, b = np.empty(self.n_total), np.empty(self.n_total)
        _lib.check(self.lib.rbl_get_sigma(self._h, _lib.ptr(a), _lib.ptr(b)))
        return a, b

    def info(self):
        ld, cu, L = C.c_int64(0), C.c_int(0), C.c_double(0)
        _lib.check(self.lib.rbl_info(self._h, C.byref(ld), C.byref(cu), C.byref(L)))
        return dict(ld=ld.value, num_cu=cu.value, lipschitz=L.value)

    # ------------------------------------------------------------------ hot path
    def step(self, want_objective=False):
        st = _lib.RblStats()
        _lib.check(self.lib.rbl_step(self._h, 1 if want_objective else 0, C.byref(st)))
        return st

    def solve(self, max_iter=0, want_objective=False):
        cap = int(max_iter if max_iter > 0 else self.cfg.max_iter)
        hist = {k: np.full(cap, np.nan) for k in ("objective", "primal", "dual", "rho", "time")}
        st = _lib.RblStats()
        _lib.check(self.lib.rbl_solve(self._h, cap, 1 if want_objective else 0, C.byref(st),
                                      _lib.ptr(hist["objective"]), _lib.ptr(hist["primal"]), _lib.ptr(hist["dual"]),
                                      _lib.ptr(hist["rho"]), _lib.ptr(hist["time"]), cap))
        k = int(st.iter)
        return st, {name: a[:k] for name, a in hist.items()}

    def finalize_smooth(self):
        _lib.check(self.lib.rbl_finalize_smooth(self._h))

    def risk(self, w):
        """sum_i sigma_i loss_(i)(w) without the regulariser (objective.py:73-82)."""
        w = _lib.f64(w).reshape(-1)
        if w.size != self.d:
            raise ValueError(f"w has {w.size} entries, expected {self.d}")
        out = C.c_double(0)
        _lib.check(self.lib.rbl_objective(self._h, _lib.ptr(w), 0, C.byref(out)))
        return out.value

    def accuracy(self, w, threshold=0.5):
        """calculate_accuracy of src/util/calculate_acc.py:3-19 on this handle's rows."""
        w = _lib.f64(w).reshape(-1)
        if w.size != self.d:
            raise ValueError(f"w has {w.size} entries, expected {self.d}")
        out = C.c_double(0)
        _lib.check(self.lib.rbl_accuracy(self._h, _lib.ptr(w), float(threshold), C.byref(out)))
        return out.value

    def fair_statistics(self, w, group, threshold=0.5):
        """(SPD, DI, EOD, AOD, TI, FNRD) of src/util/fair_metric.py:3-41 on this handle's rows."""
        w = _lib.f64(w).reshape(-1)
        g = _lib.f64(group).reshape(-1)
        if w.size != self.d or g.size != self.n:
            raise ValueError("fair_statistics: w / group have the wrong size")
        out = np.empty(6)
        _lib.check(self.lib.rbl_fair_statistics(self._h, _lib.ptr(w), _lib.ptr(g), float(threshold), _lib.ptr(out)))
        return tuple(float(x) for x in out)

    # ---------------------------------------------------------------- phase API
    def phase_m(self):
        _lib.check(self.lib.rbl_phase_m(self._h))

    def phase_z(self, m_all_ptr=None):
        _lib.check(self.lib.rbl_phase_z(self._h, C.c_void_p(m_all_ptr) if m_all_ptr else None))

    def phase_z_external(self, z):
        """a z-step computed by the caller (overridden hook): n values for this handle's rows"""
        z = _lib.f64(z).reshape(-1)
        if z.size != self.n:
            raise ValueError(f"z has {z.size} entries, expected {self.n}")
        _lib.check(self.lib.rbl_phase_z_external(self._h, _lib.ptr(z)))

    def phase_w_external(self, w):
        """a w-step computed by the caller (overridden hook): d values"""
        w = _lib.f64(w).reshape(-1)
        if w.size != self.d:
            raise ValueError(f"w has {w.size} entries, expected {self.d}")
        _lib.check(self.lib.rbl_phase_w_external(self._h, _lib.ptr(w)))

    def phase_q(self):
        _lib.check(self.lib.rbl_phase_q(self._h))

    def phase_w(self):
        _lib.check(self.lib.rbl_phase_w(self._h))

    def phase_dual(self, want_objective=False):
        _lib.check(self.lib.rbl_phase_dual(self._h, 1 if want_objective else 0))

    def phase_finish(self):
        st = _lib.RblStats()
        _lib.check(self.lib.rbl_phase_finish(self._h, C.byref(st)))
        return st

    def buffer(self, which):
        p, cnt = C.c_void_p(), C.c_int64(0)
        _lib.check(self.lib.rbl_buffer(self._h, int(which), C.byref(p), C.byref(cnt)))
        return p.value, cnt.value

    # ---- distributed z-step (include/rbl.h: rbl_zd_*; driver: dist.py:_z_distributed)
    def zd_sort_local(self, nsamples):
        _lib.check(self.lib.rbl_zd_sort_local(self._h, int(nsamples)))

    def zd_partition(self, splitters_ptr, nparts, to_host=False):
        """counts per destination stay on the device (BUF_ZD_COUNTS); to_host=True also downloads them"""
        out = (C.c_int64 * int(nparts))() if to_host else None
        _lib.check(self.lib.rbl_zd_partition(self._h, C.c_void_p(splitters_ptr), int(nparts), out))
        return [int(x) for x in out] if to_host else None

    def zd_sort_losses(self, nsamples):
        _lib.check(self.lib.rbl_zd_sort_losses(self._h, int(nsamples)))

    def zd_risk(self, n_recv, sigma_off):
        _lib.check(self.lib.rbl_zd_risk(self._h, int(n_recv), int(sigma_off)))

    def zd_prepare(self, n_recv, sigma_off):
        _lib.check(self.lib.rbl_zd_prepare(self._h, int(n_recv), int(sigma_off)))

    def zd_pav(self, fvals_ptr):
        _lib.check(self.lib.rbl_zd_pav(self._h, C.c_void_p(fvals_ptr)))

    def zd_bounds(self):
        _lib.check(self.lib.rbl_zd_bounds(self._h))

    def zd_seam_setup(self, rank, world, level, bounds_all_ptr):
        _lib.check(self.lib.rbl_zd_seam_setup(self._h, int(rank), int(world), int(level), C.c_void_p(bounds_all_ptr)))

    def zd_seam_propose(self, K, cand_prev_ptr, part_prev_ptr):
        _lib.check(self.lib.rbl_zd_seam_propose(self._h, int(K), C.c_void_p(cand_prev_ptr), C.c_void_p(part_prev_ptr)))

    def zd_seam_eval(self, K, cand_all_ptr):
        _lib.check(self.lib.rbl_zd_seam_eval(self._h, int(K), C.c_void_p(cand_all_ptr)))

    def zd_seam_sums(self, K, cand_prev_ptr, part_prev_ptr, nseams):
        _lib.check(self.lib.rbl_zd_seam_sums(self._h, int(K), C.c_void_p(cand_prev_ptr), C.c_void_p(part_prev_ptr),
                                             int(nseams)))

    def zd_seam_fill(self, sums_ptr):
        _lib.check(self.lib.rbl_zd_seam_fill(self._h, C.c_void_p(sums_ptr)))

    def zd_return_partition(self, nmax, world, to_host=False):
        out = (C.c_int64 * int(world))() if to_host else None
        _lib.check(self.lib.rbl_zd_return_partition(self._h, int(nmax), int(world), out))
        return [int(x) for x in out] if to_host else None

    def zd_scatter(self, n_back):
        _lib.check(self.lib.rbl_zd_scatter(self._h, int(n_back)))

    # sort-free distributed z-step for banded rank weights (include/rbl.h: rbl_zbd_*)
    def zbd_begin(self):
        """-> (applicable, [clusters that can pool])"""
        a, mask = C.c_int(0), C.c_int(0)
        _lib.check(self.lib.rbl_zbd_begin(self._h, C.byref(a), C.byref(mask)))
        return bool(a.value), [k for k in range(8) if mask.value >> k & 1]

    def zbd_hist(self, p):
        _lib.check(self.lib.rbl_zbd_hist(self._h, int(p)))

    def zbd_scan(self, p):
        _lib.check(self.lib.rbl_zbd_scan(self._h, int(p)))

    def zbd_eval(self, k):
        _lib.check(self.lib.rbl_zbd_eval(self._h, int(k)))

    def zbd_decide(self, k, last, want_settled=True):
        """-> settled (bool; the same on every rank) when want_settled, else None (no host wait)"""
        if not want_settled:
            _lib.check(self.lib.rbl_zbd_decide(self._h, int(k), int(bool(last)), None))
            return None
        st = C.c_int(0)
        _lib.check(self.lib.rbl_zbd_decide(self._h, int(k), int(bool(last)), C.byref(st)))
        return bool(st.value)

    def zbd_root_passes(self):
        return int(self.lib.rbl_zbd_root_passes())

    def zbd_gather(self, k):
        _lib.check(self.lib.rbl_zbd_gather(self._h, int(k)))

    def zbd_finish(self, k, packs_all_ptr, world):
        _lib.check(self.lib.rbl_zbd_finish(self._h, int(k), C.c_void_p(int(packs_all_ptr)), int(world)))

    def zbd_apply(self):
        st = C.c_int(0)
        _lib.check(self.lib.rbl_zbd_apply(self._h, C.byref(st)))
        return st.value

    def pending_reduce(self):
        m = C.c_int(0)
        _lib.check(self.lib.rbl_pending_reduce(self._h, C.byref(m)))
        return m.value

    def risk_from_v(self, v_all_ptr):
        out = C.c_double(0)
        _lib.check(self.lib.rbl_risk_from_v(self._h, C.c_void_p(v_all_ptr), C.byref(out)))
        return out.value

    # -------------------------------------------------------------- measurement
    def profile_kernels(self, enable=True):
        """0/False: no events in the iteration; 1/True: HIP events around the sweep kernels
        (kernel_time()); 2: also around the phases (the ms_* fields of the step statistics)."""
        _lib.check(self.lib.rbl_profile_kernels(self._h, int(enable)))

    def profile_sampling(self, every=1):
        """kernel events on every `every`-th iteration only (include/rbl.h: rbl_profile_sampling)."""
        _lib.check(self.lib.rbl_profile_sampling(self._h, int(every)))

    def reset_kernel_times(self):
        _lib.check(self.lib.rbl_reset_kernel_times(self._h))

    def kernel_samples(self, which):
        """the timed launches of one kernel since the last reset, in launch order (ms)"""
        cnt = C.c_int64(0)
        _lib.check(self.lib.rbl_kernel_samples(self._h, int(which), None, 0, C.byref(cnt)))
        out = np.empty(cnt.value)
        _lib.check(self.lib.rbl_kernel_samples(self._h, int(which), _lib.ptr(out), cnt.value, C.byref(cnt)))
        return out

    def kernel_time(self, which):
        ms, cnt = C.c_double(0), C.c_int64(0)
        _lib.check(self.lib.rbl_kernel_time(self._h, int(which), C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value


def reg_terms(w, l2_reg, l1_reg):
    """Regulariser of get_arrogate_loss (objective.py:83-86): both terms when both are set."""
    w = np.asarray(w, dtype=np.float64).reshape(-1)
    r = 0.0
    if l2_reg:
        r += 0.5 * l2_reg * float(np.sum(w ** 2))
    if l1_reg:
        r += 0.5 * l1_reg * float(np.sum(np.abs(w)))
    return r


def isnan(x):
    return isinstance(x, float) and math.isnan(x)
