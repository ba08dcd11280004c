"""ctypes binding of librbl.so (include/rbl.h).

The HIP library is the product: when it is missing or no GPU is present every compute
call fails loudly (there is no CPU fallback and nothing here imports the test oracle).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "librbl.so")

RBL_OK, RBL_ERR_INVALID, RBL_ERR_NO_DEVICE, RBL_ERR_HIP, RBL_ERR_STATE, RBL_ERR_NOMEM = 0, -1, -2, -3, -4, -5
LOSS = {"binary_cross_entropy": 0, "hinge": 1}
WEIGHT = {"erm": 0, "extremile": 1, "superquantile": 2, "esrm": 3, "aorr": 4, "aorr_dc": 5, "ehrm": 6}
WSTEP_L1, WSTEP_L2, WSTEP_SMOOTH_L1 = 1, 2, 3
STORAGE = {"f32": 0, "float32": 0, "f64": 1, "float64": 1}
BUF_M, BUF_Q, BUF_RED, BUF_G, BUF_V, BUF_Z, BUF_LAM, BUF_W, BUF_COLSTATS = range(9)
(BUF_ZD_SKEYS, BUF_ZD_SIDS, BUF_ZD_RKEYS, BUF_ZD_RIDS, BUF_ZD_SMALL, BUF_ZD_BIDS, BUF_ZD_BU, BUF_ZD_ZIDS,
 BUF_ZD_ZU, BUF_ZD_COUNTS) = range(16, 26)
BUF_ZB_HIST, BUF_ZB_TOT, BUF_ZB_PACK = 26, 27, 28
KERNEL_GEMV, KERNEL_GEMVT, KERNEL_SWEEP_ERM = 0, 1, 2


class RblConfig(C.Structure):
    _fields_ = [
        ("n", C.c_int64), ("d", C.c_int64), ("n_total", C.c_int64), ("row_offset", C.c_int64),
        ("loss", C.c_int32), ("weight_function", C.c_int32),
        ("weight_args", C.c_double * 2),
        ("n_weight_args", C.c_int32), ("has_B", C.c_int32),
        ("B", C.c_double),
        ("wstep", C.c_int32),
        ("reg", C.c_double), ("smooth_t", C.c_double), ("rho0", C.c_double), ("tol", C.c_double),
        ("w_tol", C.c_double),
        ("max_iter", C.c_int32), ("storage", C.c_int32), ("device", C.c_int32), ("objective_only", C.c_int32),
    ]


class RblStats(C.Structure):
    _fields_ = [
        ("iter", C.c_int64),
        ("primal", C.c_double), ("dual", C.c_double), ("rho", C.c_double), ("rho_next", C.c_double),
        ("objective", C.c_double),
        ("converged", C.c_int32), ("inner_iters", C.c_int32), ("ehrm_branch", C.c_int32), ("pav_merges", C.c_int32),
        ("ms_z", C.c_float), ("ms_q", C.c_float), ("ms_w", C.c_float), ("ms_v", C.c_float), ("ms_total", C.c_float),
        ("fused", C.c_int32), ("mispredicted", C.c_int32),
        ("fused_v", C.c_int32), ("host_syncs", C.c_int32), ("sort_passes", C.c_int32), ("zband", C.c_int32),
        ("wstep_form", C.c_int32),
    ]


class RblError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"librbl error {code}: {msg}")
        self.code = code
        self.msg = msg


_P = C.c_void_p
_D = C.POINTER(C.c_double)
_I64 = C.POINTER(C.c_int64)

# name -> (restype, argtypes): every symbol include/rbl.h declares
SIGNATURES = {
    "rbl_version": (C.c_int, []),
    "rbl_sizeof": (C.c_int, [C.c_int]),
    "rbl_last_error": (C.c_char_p, []),
    "rbl_device_count": (C.c_int, []),
    "rbl_create": (C.c_int, [C.POINTER(RblConfig), C.POINTER(_P)]),
    "rbl_destroy": (C.c_int, [_P]),
    "rbl_set_stream": (C.c_int, [_P, _P]),
    "rbl_set_data": (C.c_int, [_P, _P, _P, C.c_int64]),
    "rbl_generate_synthetic": (C.c_int, [_P, C.c_uint64, C.c_double, C.c_double]),
    "rbl_synth_local": (C.c_int, [_P, C.c_uint64, C.c_double, C.c_double]),
    "rbl_synth_finish": (C.c_int, [_P]),
    "rbl_get_labels": (C.c_int, [_P, _P]),
    "rbl_gram_local": (C.c_int, [_P]),
    "rbl_gram_finish": (C.c_int, [_P]),
    "rbl_get_D": (C.c_int, [_P, _P]),
    "rbl_get_state": (C.c_int, [_P, _P, _P, _P, _D, _I64, _D]),
    "rbl_set_state": (C.c_int, [_P, _P, _P, _P, _D, _I64, _D]),
    "rbl_get_sigma": (C.c_int, [_P, _P, _P]),
    "rbl_step": (C.c_int, [_P, C.c_int, C.POINTER(RblStats)]),
    "rbl_solve": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(RblStats), _P, _P, _P, _P, _P, C.c_int64]),
    "rbl_finalize_smooth": (C.c_int, [_P]),
    "rbl_objective": (C.c_int, [_P, _P, C.c_int, _D]),
    "rbl_accuracy": (C.c_int, [_P, _P, C.c_double, _D]),
    "rbl_fair_statistics": (C.c_int, [_P, _P, _P, C.c_double, _P]),
    "rbl_phase_m": (C.c_int, [_P]),
    "rbl_phase_z": (C.c_int, [_P, _P]),
    "rbl_phase_z_external": (C.c_int, [_P, _P]),
    "rbl_phase_w_external": (C.c_int, [_P, _P]),
    "rbl_zd_sort_local": (C.c_int, [_P, C.c_int]),
    "rbl_zd_partition": (C.c_int, [_P, C.c_void_p, C.c_int, C.POINTER(C.c_int64)]),
    "rbl_zd_prepare": (C.c_int, [_P, C.c_int64, C.c_int64]),
    "rbl_zd_sort_losses": (C.c_int, [_P, C.c_int]),
    "rbl_zd_risk": (C.c_int, [_P, C.c_int64, C.c_int64]),
    "rbl_zd_pav": (C.c_int, [_P, C.c_void_p]),
    "rbl_zd_bounds": (C.c_int, [_P]),
    "rbl_zd_seam_setup": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "rbl_zd_seam_propose": (C.c_int, [_P, C.c_int, C.c_void_p, C.c_void_p]),
    "rbl_zd_seam_eval": (C.c_int, [_P, C.c_int, C.c_void_p]),
    "rbl_zd_seam_sums": (C.c_int, [_P, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "rbl_zd_seam_fill": (C.c_int, [_P, C.c_void_p]),
    "rbl_zd_return_partition": (C.c_int, [_P, C.c_int64, C.c_int, C.POINTER(C.c_int64)]),
    "rbl_zd_scatter": (C.c_int, [_P, C.c_int64]),
    "rbl_zbd_begin": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "rbl_zbd_hist": (C.c_int, [_P, C.c_int]),
    "rbl_zbd_scan": (C.c_int, [_P, C.c_int]),
    "rbl_zbd_eval": (C.c_int, [_P, C.c_int]),
    "rbl_zbd_decide": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "rbl_zbd_root_passes": (C.c_int, []),
    "rbl_zbd_gather": (C.c_int, [_P, C.c_int]),
    "rbl_zbd_finish": (C.c_int, [_P, C.c_int, C.c_void_p, C.c_int]),
    "rbl_zbd_apply": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "rbl_phase_q": (C.c_int, [_P]),
    "rbl_phase_w": (C.c_int, [_P]),
    "rbl_phase_dual": (C.c_int, [_P, C.c_int]),
    "rbl_phase_finish": (C.c_int, [_P, C.POINTER(RblStats)]),
    "rbl_buffer": (C.c_int, [_P, C.c_int, C.POINTER(_P), _I64]),
    "rbl_pending_reduce": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "rbl_risk_from_v": (C.c_int, [_P, _P, _D]),
    "rbl_info": (C.c_int, [_P, _I64, C.POINTER(C.c_int), _D]),
    "rbl_kernel_time": (C.c_int, [_P, C.c_int, _D, _I64]),
    "rbl_reset_kernel_times": (C.c_int, [_P]),
    "rbl_kernel_samples": (C.c_int, [_P, C.c_int, _P, C.c_int64, _I64]),
    "rbl_profile_kernels": (C.c_int, [_P, C.c_int]),
    "rbl_profile_sampling": (C.c_int, [_P, C.c_int]),
    "rbl_k_prox": (C.c_int, [C.c_int, C.c_int64, _P, C.c_double, _P, _P]),
    "rbl_k_sort": (C.c_int, [C.c_int64, _P, _P, _P]),
    "rbl_k_pav": (C.c_int, [C.c_int, C.c_int64, _P, C.c_double, _P, _P, _I64]),
    "rbl_k_pav_ehrm": (C.c_int, [C.c_int64, _P, _P, C.c_double, C.c_double, _P, C.c_int, _P, C.POINTER(C.c_int)]),
    "rbl_k_gemv": (C.c_int, [C.c_int, C.c_int64, C.c_int64, _P, _P, _P]),
    "rbl_k_gemvt": (C.c_int, [C.c_int, C.c_int64, C.c_int64, _P, _P, _P]),
    "rbl_k_gram": (C.c_int, [C.c_int, C.c_int64, C.c_int64, _P, _P]),
    "rbl_k_wstep": (C.c_int, [C.c_int, C.c_int64, _P, _P, C.c_double, C.c_double, C.c_double, _P, C.c_double, _P,
                              C.POINTER(C.c_int)]),
    "rbl_k_weights": (C.c_int, [C.c_int, C.c_int64, _P, C.c_int, _P, _P]),
    "rbl_bl_create": (C.c_int, [C.c_int64, C.c_int64, _P, _P, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int,
                                C.POINTER(_P)]),
    "rbl_bl_destroy": (C.c_int, [_P]),
    "rbl_bl_set_w": (C.c_int, [_P, _P]),
    "rbl_bl_get_w": (C.c_int, [_P, _P]),
    "rbl_bl_sgd_epoch": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, C.c_double, _P]),
    "rbl_bl_lsvrg_epoch": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_double, _P]),
}

_lib = None


def _preload_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so
    (same SONAME as /opt/rocm's).  If librbl.so pulls in the system copy first and torch
    initialises its bundled copy later, the process holds two HIP runtimes and torch then
    reports "No HIP GPUs are available" (measured on the MI355X box, tools/diag_hip_runtime.py).
    Loading torch's copy first makes librbl.so resolve to it as well, whichever module is
    imported first.  Set RBL_NO_TORCH_HIP_PRELOAD=1 to use the system runtime."""
    if os.environ.get("RBL_NO_TORCH_HIP_PRELOAD") == "1":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(path):
            C.CDLL(path, mode=C.RTLD_GLOBAL)
    except Exception:
        pass    # without torch the system runtime is the only one: nothing to reconcile


def load():
    """Load librbl.so; raises (never falls back) when the HIP library is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        _preload_torch_hip_runtime()
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        for which, cls in ((0, RblConfig), (1, RblStats)):
            if lib.rbl_sizeof(which) != C.sizeof(cls):
                raise ImportError(f"{LIB_PATH}: sizeof({cls.__name__}) is {lib.rbl_sizeof(which)} in the library and "
                                  f"{C.sizeof(cls)} in this binding (include/rbl.h and _lib.py out of step: rebuild)")
        _lib = lib
    return _lib


def last_error():
    return load().rbl_last_error().decode("utf-8", "replace")


def check(code):
    if code != RBL_OK:
        msg = last_error()
        if code == RBL_ERR_INVALID:
            raise ValueError(msg)
        raise RblError(code, msg)


def ptr(a):
    """void* of a C-contiguous numpy array (None -> NULL)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def device_count():
    return load().rbl_device_count()


# ----------------------------------------------------------- kernel-level wrappers (tests)
def k_prox(loss, sigma, rho, m):
    sigma, m = f64(sigma).reshape(-1), f64(m).reshape(-1)
    out = np.empty_like(m)
    check(load().rbl_k_prox(LOSS[loss], m.size, ptr(sigma), float(rho), ptr(m), ptr(out)))
    return out


def k_sort(keys):
    keys = f64(keys).reshape(-1)
    out = np.empty_like(keys)
    perm = np.empty(keys.size, dtype=np.uint32)
    check(load().rbl_k_sort(keys.size, ptr(keys), ptr(out), ptr(perm)))
    return out, perm


def k_pav(loss, sigma, rho, m_sorted):
    sigma, m = f64(sigma).reshape(-1), f64(m_sorted).reshape(-1)
    out = np.empty_like(m)
    nm = C.c_int64(0)
    check(load().rbl_k_pav(LOSS[loss], m.size, ptr(sigma), float(rho), ptr(m), ptr(out), C.byref(nm)))
    return out, nm.value


def k_pav_ehrm(sigma_a, sigma_b, B, rho, m_sorted, branch=-1):
    sa, sb, m = f64(sigma_a).reshape(-1), f64(sigma_b).reshape(-1), f64(m_sorted).reshape(-1)
    out = np.empty_like(m)
    br = C.c_int(-1)
    check(load().rbl_k_pav_ehrm(m.size, ptr(sa), ptr(sb), float(B), float(rho), ptr(m), int(branch), ptr(out),
                                C.byref(br)))
    return out, br.value


def k_gemv(D, w, storage="f32"):
    D, w = f64(D), f64(w).reshape(-1)
    v = np.empty(D.shape[0])
    check(load().rbl_k_gemv(STORAGE[storage], D.shape[0], D.shape[1], ptr(D), ptr(w), ptr(v)))
    return v


def k_gemvt(D, c, storage="f32"):
    D, c = f64(D), f64(c).reshape(-1)
    q = np.empty(D.shape[1])
    check(load().rbl_k_gemvt(STORAGE[storage], D.shape[0], D.shape[1], ptr(D), ptr(c), ptr(q)))
    return q


def k_gram(D, storage="f32"):
    D = f64(D)
    G = np.empty((D.shape[1], D.shape[1]))
    check(load().rbl_k_gram(STORAGE[storage], D.shape[0], D.shape[1], ptr(D), ptr(G)))
    return G


def k_wstep(wstep, G, q, rho, reg, w0=None, smooth_t=1.0, tol=1e-13):
    G, q = f64(G), f64(q).reshape(-1)
    d = q.size
    w0 = f64(w0).reshape(-1) if w0 is not None else np.zeros(d)
    out = np.empty(d)
    it = C.c_int(0)
    check(load().rbl_k_wstep(int(wstep), d, ptr(G), ptr(q), float(rho), float(reg), float(smooth_t), ptr(w0),
                             float(tol), ptr(out), C.byref(it)))
    return out, it.value


def k_weights(weight_function, n, args=None):
    if weight_function not in WEIGHT:
        raise ValueError(
            f"Unrecognized framework '{weight_function}'! Options: ['erm','extremile','superquantile','esrm','aorr','aorr_dc','ehrm']")
    a = np.empty(n)
    b = np.empty(n)
    arr = f64(list(args)) if args is not None else None
    check(load().rbl_k_weights(WEIGHT[weight_function], n, ptr(arr), 0 if arr is None else arr.size, ptr(a), ptr(b)))
    return a, b
