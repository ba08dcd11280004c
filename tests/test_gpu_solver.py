"""Solver-level parity on the GPU: the reference-mirroring class API
(ADMMmethod / smoothADMMmethod) against (1) the CPU oracle's exact mode on the same
inputs, iteration by iteration, and (2) the golden trajectories generated from the real
reference (tests/golden/g9_*.npz), with the per-configuration contracts of SURVEY 8c."""
import io
import json
import contextlib

import numpy as np
import pytest

from conftest import load_golden, golden_json

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R():
    import admm_for_rank_based_loss_amd as rbl
    if rbl._lib.device_count() < 1:
        pytest.fail("no HIP device: the GPU tests must run the HIP library (no fallback)")
    return rbl


def _quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


CASES = [
    ("erm_bce_l1", dict(weight_function="erm", loss="binary_cross_entropy", l1_reg=0.01)),
    ("erm_hinge_l2", dict(weight_function="erm", loss="hinge", l2_reg=0.01)),
    ("superq_bce_l2", dict(weight_function="superquantile", loss="binary_cross_entropy", l2_reg=0.01, args=[0.5])),
    ("extremile_bce_l1", dict(weight_function="extremile", loss="binary_cross_entropy", l1_reg=0.01, args=[2.0])),
    ("esrm_hinge_l2", dict(weight_function="esrm", loss="hinge", l2_reg=0.01, args=[1.0])),
    ("aorr_hinge_l2", dict(weight_function="aorr", loss="hinge", l2_reg=1e-4, args=[0.2, 0.8])),
    ("aorr_bce_l2", dict(weight_function="aorr", loss="binary_cross_entropy", l2_reg=1e-4, args=[0.2, 0.8])),
    ("aorr_dc_bce_l2", dict(weight_function="aorr_dc", loss="binary_cross_entropy", l2_reg=1e-4, args=[300, 40])),
    ("ehrm_bce_l2", dict(weight_function="ehrm", loss="binary_cross_entropy", l2_reg=0.01, B=-5)),
]


@pytest.mark.parametrize("name,kw", CASES, ids=[c[0] for c in CASES])
def test_iterates_match_oracle_exact(R, name, kw):
    """fp64 storage: state after every one of 25 iterations vs the oracle's exact mode.
    Both solve each convex sub-problem to ~1e-13, so iterates agree to ~1e-9 (hinge:
    the kinks amplify rounding a little - 1e-7)."""
    from oracle import problems, admm
    X, y = problems.make_problem(1500, 24, seed=77)
    nit = 25
    ref = admm.admm_solve(X, y, max_iter=nit, mode="exact", tol=0.0, **kw)
    s = R.ADMMmethod(X, y, max_iter=nit, tol=0.0, storage="f64", **kw)
    tol = 1e-9 if kw["loss"] == "binary_cross_entropy" else 1e-7
    for i in range(nit):
        st = s._s.step(want_objective=True)
        assert abs(st.rho - ref.rho[i]) <= 1e-15 * ref.rho[i]
        assert abs(st.primal - ref.primal[i]) <= tol * max(1.0, ref.primal[i]), (name, i, st.primal, ref.primal[i])
        assert abs(st.dual - ref.dual[i]) <= tol * max(1.0, ref.dual[i]), (name, i)
        assert abs(st.objective - ref.objective[i + 1]) <= tol * max(1.0, abs(ref.objective[i + 1])), (name, i)
        if kw["weight_function"] == "ehrm":
            assert st.ehrm_branch == (0 if ref.branch[i] == "a" else 1)
        if "l2_reg" in kw:
            assert st.wstep_form == 1, (name, i, st.wstep_form)     # the CG as ONE persistent launch (k_cg_persist)
    state = s._s.get_state()
    scale = max(1.0, np.max(np.abs(ref.z)))
    assert np.max(np.abs(state["w"] - ref.w)) <= tol * max(1.0, np.max(np.abs(ref.w)))
    assert np.max(np.abs(state["z"] - ref.z)) <= 10 * tol * scale
    assert np.max(np.abs(state["lam"] - ref.lam)) <= 10 * tol * max(1e-3, np.max(np.abs(ref.lam)))


@pytest.mark.parametrize("n,d,t0", [(1500, 24, 1.0), (3000, 160, 1.0), (2000, 60, 0.05)],
                         ids=["1500x24", "3000x160_single_sweep", "2000x60_t0.05"])
def test_sadmm_iterates_match_oracle_exact(R, n, d, t0):
    """smoothADMMmethod (algorithms.py:224-260) iteration by iteration against the oracle's exact mode with
    smooth=True: the Huber-smoothed w-step (w_LBFGS.py:11-28) solved to ~1e-13 on both sides (device:
    preconditioned nonlinear CG with exact line search, oracle: FISTA to 1e-14), the t schedule of
    :254-255 (from iteration 17 on) and the final soft-threshold of :257-258.  1e-8 on every logged
    quantity, t to rounding."""
    from oracle import problems, admm
    X, y = problems.make_problem(n, d, seed=31 + d)
    nit = 45
    kw = dict(weight_function="erm", loss="binary_cross_entropy", l1_reg=0.01)
    ref = admm.admm_solve(X, y, max_iter=nit, mode="exact", tol=0.0, smooth=True, t=t0, **kw)
    s = R.smoothADMMmethod(X, y, max_iter=nit, tol=0.0, storage="f64", t=t0, **kw)
    inner = []
    for i in range(nit):
        st = s._s.step(want_objective=True)
        inner.append(st.inner_iters)
        assert st.wstep_form == 1, (i, st.wstep_form)               # the nonlinear CG as ONE persistent launch (k_ncg_persist)
        assert abs(st.rho - ref.rho[i]) <= 1e-15 * ref.rho[i]
        assert abs(st.primal - ref.primal[i]) <= 1e-8 * max(1.0, ref.primal[i]), (i, st.primal, ref.primal[i])
        assert abs(st.dual - ref.dual[i]) <= 1e-8 * max(1.0, ref.dual[i]), (i, st.dual, ref.dual[i])
        assert abs(st.objective - ref.objective[i + 1]) <= 1e-8 * max(1.0, abs(ref.objective[i + 1])), i
    assert abs(s.t - ref.t) <= 1e-12 * ref.t                        # the t schedule, :254-255
    assert max(inner) <= 200, inner                                 # the nonlinear CG, not its FISTA fallback
    state = s._s.get_state()
    s._s.finalize_smooth()                                          # :257-258
    w_final = s._s.get_state()["w"]
    assert np.max(np.abs(w_final - ref.w)) <= 1e-8 * max(1.0, np.max(np.abs(ref.w)))
    assert np.count_nonzero(w_final) == np.count_nonzero(ref.w)
    assert np.max(np.abs(state["z"] - ref.z)) <= 1e-7 * max(1.0, np.max(np.abs(ref.z)))
    assert np.max(np.abs(state["lam"] - ref.lam)) <= 1e-7 * max(1e-3, np.max(np.abs(ref.lam)))


@pytest.mark.parametrize("smooth", [False, True], ids=["cg_ridge", "ncg_smoothed_l1"])
def test_wstep_persistent_and_batched_forms_agree(R, smooth):
    """The d-space w-steps run as ONE persistent launch while the process holds a single solver handle on the device,
    and as batches of launches as soon as a second handle is alive (two persistent kernels side by side could each hold
    CUs the other waits for; csrc/wstep.hip wp_plan).  rbl_stats.wstep_form says which form ran; both solve the same
    sub-problem (w_LBFGS.py:11-62 / algorithms.py:108-120) to ~1e-13, so the iterates agree to 1e-9."""
    from oracle import problems
    X, y = problems.make_problem(3000, 160, seed=5)
    nit = 30
    if smooth:
        make = lambda: R.smoothADMMmethod(X, y, max_iter=nit, tol=0.0, storage="f64", weight_function="erm",
                                          loss="binary_cross_entropy", l1_reg=0.01)
    else:
        make = lambda: R.ADMMmethod(X, y, max_iter=nit, tol=0.0, storage="f64", weight_function="superquantile",
                                    loss="binary_cross_entropy", l2_reg=0.01, args=[0.5])

    def run(s):
        forms, rows = set(), []
        for _ in range(nit):
            st = s._s.step(want_objective=True)
            forms.add(st.wstep_form)
            rows.append((st.primal, st.dual, st.objective))
        return forms, np.array(rows), s._s.get_state()["w"]

    a = make()
    forms_a, rows_a, w_a = run(a)
    a._s.close()
    assert forms_a == {1}, forms_a
    other = make()                      # a second live handle of this process on the device
    b = make()
    forms_b, rows_b, w_b = run(b)
    other._s.close()
    b._s.close()
    assert forms_b == {0}, forms_b
    assert np.max(np.abs(rows_a - rows_b) / np.maximum(1.0, np.abs(rows_a))) <= 1e-9
    assert np.max(np.abs(w_a - w_b)) <= 1e-9 * max(1.0, np.max(np.abs(w_a)))


def test_f32_storage_close_to_f64(R):
    """fp32 storage of D perturbs every entry by <= 6e-8 relative: the final objective
    moves by far less than the 1e-6 contract."""
    from oracle import problems
    X, y = problems.make_problem(4000, 50, seed=5)
    kw = dict(weight_function="superquantile", loss="binary_cross_entropy", l2_reg=0.01, args=[0.5])
    out = {}
    for st in ("f32", "f64"):
        s = R.ADMMmethod(X, y, storage=st, **kw)
        w = _quiet(s.main_loop, verbose=False)
        out[st] = (s.objective.get_arrogate_loss(w), w)
    assert abs(out["f32"][0] - out["f64"][0]) <= 1e-7 * abs(out["f64"][0])
    assert np.max(np.abs(out["f32"][1] - out["f64"][1])) <= 1e-5


G9 = ["erm_bce_l1", "erm_bce_l2", "superq_bce_l2", "extremile_bce_l1", "esrm_hinge_l2", "superq_hinge_l2",
      "aorr_hinge_l2", "aorr_bce_l2", "ehrm_bce_l2", "sadmm_erm_bce_l1"]


@pytest.mark.parametrize("storage", ["f64", "f32"])
@pytest.mark.parametrize("name", G9)
def test_reference_trajectory_goldens(R, name, storage):
    """Whole solves vs the REAL reference (goldens).  Contract: runs that
    reach the stop rule agree on the final objective to 1e-6 relative (ehrm 2e-6); the
    initial objective and the rho schedule are identical; AoRR/hinge cannot be matched
    two-sidedly (reference bisection artifact) - there the GPU objective must not be
    worse than the reference's."""
    from oracle import problems
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", f"g9_{name}.npz")
    if not os.path.exists(path):
        pytest.skip(f"golden {name} not generated")
    g = np.load(path, allow_pickle=False)
    cfg = json.loads(str(g["config"]))
    X, y = problems.make_problem(cfg["n"], cfg["d"] - 1 if cfg["intercept"] else cfg["d"], cfg["seed"],
                                 intercept=cfg["intercept"])
    assert problems.sha256_of(X) == str(g["x_sha256"]) and problems.sha256_of(y) == str(g["y_sha256"])
    kw = cfg["kw"]
    cls = R.smoothADMMmethod if cfg["cls"] == "smoothADMMmethod" else R.ADMMmethod
    s = cls(X, y, max_iter=cfg["max_iter"], storage=storage, **kw)
    skw = {k: v for k, v in kw.items() if k != "B"}
    s.start_store(X, y, **skw)
    w = _quiet(s.main_loop, verbose=False)
    f_ref = float(g["final_objective"])
    f_gpu = s.objective.get_arrogate_loss(w)
    ref_obj = g["objective"]
    # same initial point and objective definition: F(w0) identical
    assert abs(s.train_losses[0] - ref_obj[0]) <= (1e-12 if storage == "f64" else 1e-7) * abs(ref_obj[0])
    rel = (f_gpu - f_ref) / abs(f_ref)
    converged_ref = bool(g["converged"])
    print(f"{name} [{storage}]: F_gpu={f_gpu:.12g} F_ref={f_ref:.12g} rel={rel:+.2e} iters gpu={len(s.train_losses)-1} ref={int(g['iters'])}")
    if name == "sadmm_erm_bce_l1":
        # the reference's L-BFGS w-step is loose: the exact smoothed w-step ends within 5e-6 of it
        # (the CPU oracle's exact mode does the same, tests/test_oracle_trajectories.py) and never above it
        assert abs(rel) <= 5e-6 and rel <= 1e-9
    elif name.startswith("aorr_hinge"):
        assert f_gpu <= f_ref * (1 + 1e-6) + 1e-12     # non-convex + reference artifact: one-sided
    elif converged_ref:
        # convex problems run to the stop rule: 1e-6.  ehrm / aorr are non-convex, both
        # runs stop at slightly different iterations near a flat stationary region
        # (reference 308 / 531 iterations): 1e-5 / 5e-4 relative (|dF| ~ 1e-6 absolute).
        tol = 1e-5 if name.startswith("ehrm") else (5e-4 if name.startswith("aorr") else 1e-6)
        assert abs(rel) <= tol, (name, rel)
    else:
        # the reference did not reach its stop rule in max_iter: compare one-sidedly
        assert f_gpu <= f_ref * (1 + 1e-5) + 1e-12


@pytest.mark.parametrize("name", ["aorr_bce_l2", "aorr_hinge_l2", "ehrm_bce_l2", "superq_bce_l2", "erm_bce_l1"])
def test_f32_storage_contract_whole_solves(R, name):
    """The fp32-storage contract (the shipped default, and what C2 / C3 / C4 are benched in): storing D = -y X in
    fp32 is EXACT fp64 arithmetic on a matrix whose entries moved by <= 6e-8 relative (iterates vs the oracle
    on the rounded matrix: tests/test_gpu_widths.py, 1e-9).  Whole solves of the golden configurations -
    including the non-convex AoRR / EHRM families, whose trajectories amplify the perturbation along the way
    (tests/stress_vs_oracle.py RAW=1: up to 5e-2 on intermediate residuals) - end within 1e-6 relative of
    the fp64-storage solve (measured round 2: 5e-9 aorr, 2e-7 ehrm, 1e-9 convex) after the same number of
    iterations +- 1."""
    from oracle import problems
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", f"g9_{name}.npz"), allow_pickle=False)
    cfg = json.loads(str(g["config"]))
    X, y = problems.make_problem(cfg["n"], cfg["d"] - 1 if cfg["intercept"] else cfg["d"], cfg["seed"],
                                 intercept=cfg["intercept"])
    out = {}
    for storage in ("f64", "f32"):
        s = R.ADMMmethod(X, y, max_iter=cfg["max_iter"], storage=storage, **cfg["kw"])
        s.start_store(X, y, **{k: v for k, v in cfg["kw"].items() if k != "B"})
        w = _quiet(s.main_loop, verbose=False)
        out[storage] = (s.objective.get_arrogate_loss(w), len(s.train_losses), w)
    (f64, n64, w64), (f32, n32, w32) = out["f64"], out["f32"]
    assert abs(f32 - f64) <= 1e-6 * abs(f64), (name, f32, f64)
    assert abs(n32 - n64) <= 1, (n32, n64)
    assert np.max(np.abs(w32 - w64)) <= 1e-4 * max(1.0, np.max(np.abs(w64)))


def test_c1_published_config(R):
    """BASELINE config C1 (run_SRM.py defaults): same data (sha256), final objective within
    1e-6 relative of the reference run here AND of the published table, not worse than the
    independent optimum F* + 1.5e-6 |F*|, support size 26."""
    from oracle import problems
    g = load_golden("g9_c1_srm_erm_bce_l1.npz")
    Xtr, Xte, ytr, yte = problems.c1_data()
    assert problems.sha256_of(Xtr) == str(g["x_sha256"])
    kw = dict(weight_function="erm", loss="binary_cross_entropy", l1_reg=0.01)
    s = R.ADMMmethod(Xtr, ytr, **kw)          # default fp32 storage: the shipped configuration
    s.start_store(Xte, yte, **kw)
    w = _quiet(s.main_loop, verbose=False)
    f = s.objective.get_arrogate_loss(w)
    table = golden_json("published_table_admm.json")
    f_pub = table["admm_train_losses"][-1]
    f_ref = float(g["final_objective"])
    fstar = 0.1475229955800
    assert abs(s.train_losses[0] - table["admm_train_losses"][0]) <= 1e-7 * table["admm_train_losses"][0]
    assert abs(f - f_ref) <= 1e-6 * f_ref and abs(f - f_pub) <= 1e-6 * f_pub
    assert f <= fstar * (1 + 1.5e-6)
    assert int(np.count_nonzero(w)) == 26
    assert np.max(np.abs(w.reshape(-1) - g["w"])) <= 1e-2
    wt, times, tr, te = s.final_res()
    assert len(times) == len(tr) == len(te)


def test_api_surface_and_errors(R):
    from oracle import problems
    X, y = problems.make_problem(300, 8, seed=1)
    s = R.ADMMmethod(X, y, "erm", "binary_cross_entropy", l2_reg=0.01)
    with pytest.raises(ValueError, match="Data was not saved."):
        s.final_res()
    assert s.num_row == 300 and s.num_feature == 8 and s.reg == 0.01
    assert s.w.shape == (8, 1) and s.z.shape == (300, 1) and s.lagrangian.shape == (300, 1)
    assert abs(s.rho - 1e-5) < 1e-20
    np.testing.assert_allclose(s.z, 0.1 * 0.01 / 300)                        # algorithms.py:34
    np.testing.assert_allclose(s.w, 0.001 * 0.01 / 8 / 300)                  # algorithms.py:42
    assert s.objective.alphas.shape == (300, 1)
    assert abs(R.ADMMmethod(X, y, "aorr", "hinge", l2_reg=1e-4, args=[0.2, 0.8]).rho - 2e-7) < 1e-22
    assert abs(R.ADMMmethod(X, y, "ehrm", l2_reg=0.01, B=-5).rho - 1e-4) < 1e-18
    w0 = np.linspace(-1, 1, 8)
    s2 = R.ADMMmethod(X, y, "erm", "hinge", l1_reg=0.02, w0=w0)
    np.testing.assert_allclose(s2.w.reshape(-1), w0)
    # the train objective equals an independent objective handle on the same data
    o = R.rankbasedObjective(X, y, "erm", "hinge", l1_reg=0.02)
    assert abs(o.get_arrogate_loss(w0) - s2.objective.get_arrogate_loss(w0)) <= 1e-14
    import torch
    assert abs(o.get_arrogate_loss(torch.from_numpy(w0.reshape(-1, 1)).double()) - o.get_arrogate_loss(w0)) == 0
    with pytest.raises(ValueError, match="Unrecognized framework"):
        R.ADMMmethod(X, y, "nope", l2_reg=0.1, args=[1])
    with pytest.raises(ValueError, match="args for framework is None"):
        R.ADMMmethod(X, y, "superquantile", l2_reg=0.1)
    with pytest.raises(ValueError, match="Unrecognized loss"):
        R.ADMMmethod(X, y, "erm", "square", l2_reg=0.1)
    with pytest.raises(ValueError, match="erhm only can be with the binary_cross_entropy"):
        R.ADMMmethod(X, y, "ehrm", "hinge", l2_reg=0.1, B=-5)
    with pytest.raises(ValueError, match=r"Unrecognized weight_function 'erm'! Options: \['ehrm'\]"):
        R.ADMMmethod(X, y, "erm", l2_reg=0.1, B=-5)


@pytest.mark.parametrize("kw,d", [
    (dict(weight_function="erm", loss="binary_cross_entropy", l1_reg=0.01), 160),      # single-sweep path
    (dict(weight_function="erm", loss="hinge", l2_reg=0.01), 24),                        # two-sweep erm path
    (dict(weight_function="superquantile", loss="binary_cross_entropy", l2_reg=0.01, args=[0.5]), 40),
    (dict(weight_function="ehrm", loss="binary_cross_entropy", l2_reg=0.01, B=-5), 140),
], ids=["erm_l1_fused", "erm_hinge_l2", "superq_l2", "ehrm_l2_vsweep"])
def test_overridden_subproblem_hooks(R, kw, d):
    """The reference's sub-problem hooks (algorithms.py:88-116, :186-207) can be overridden: a subclass whose
    z_subproblem / w_subproblem return their OWN arrays - here the CPU oracle's exact solves on the state read
    from the solver - must drive the same ADMM run as the built-in device steps (1e-9; hinge 1e-7), with the
    residuals, the rho schedule and the logged objective computed by the library from the injected z and w.
    Three subclasses: only z overridden, only w overridden, both."""
    from oracle import problems, admm, wstep, weights
    X, y = problems.make_problem(1200, d, seed=91)
    nit = 12
    ref = admm.admm_solve(X, y, max_iter=nit, mode="exact", tol=0.0, **kw)
    n = X.shape[0]
    wf, loss = kw["weight_function"], kw["loss"]
    sa, sb = weights.get_weights(wf, n, kw.get("args"))
    D = -y.reshape(-1, 1) * X
    G = D.T @ D
    Lmax = 1.0001 * wstep.lambda_max(G)
    reg = kw.get("l1_reg") or kw.get("l2_reg")

    def z_by_oracle(self):
        m = (D @ self.w - self.lagrangian / self.rho).reshape(-1)
        z, _ = admm.z_step_exact(wf, loss, sa, sb, kw.get("B"), self.rho, m)
        return z.reshape(-1, 1)

    def w_by_oracle(self):
        rho = self.rho
        q = D.T @ (self.z + self.lagrangian / rho).reshape(-1)
        if "l1_reg" in kw:
            w, _ = wstep.lasso_gram_exact(G, q, reg / (2 * rho), self.w.reshape(-1), Lmax)
        else:
            w = wstep.ridge_gram_exact(G, q, rho, reg)
        return w.reshape(-1, 1)

    class ZHook(R.ADMMmethod):
        z_subproblem = z_by_oracle

    class WHook(R.ADMMmethod):
        _w_subproblem = w_by_oracle          # the wrapper name the reference's ADMMmethod uses (:190)

    class BothHooks(R.ADMMmethod):
        z_subproblem = z_by_oracle
        w_subproblem = w_by_oracle

    # a z that is NOT what the library would have computed: nothing prepared ahead of time for the library's own z
    # (the next z-step and q of the single-sweep pass, the w-step enqueued before the host read the statistics)
    # may survive - the run with only z overridden must equal the run in which the oracle also does the w-step
    def z_perturbed(self):
        return z_by_oracle(self) + 0.01 * np.cos(np.arange(n)).reshape(-1, 1)

    class ZPert(R.ADMMmethod):
        z_subproblem = z_perturbed

    class BothPert(R.ADMMmethod):
        z_subproblem = z_perturbed
        w_subproblem = w_by_oracle

    pert = []
    for cls in (ZPert, BothPert):
        s = cls(X, y, max_iter=nit, tol=0.0, storage="f64", **kw)
        _quiet(s.main_loop, verbose=False)
        pert.append((s.w.reshape(-1), s.lagrangian.reshape(-1), s._last.primal, s._last.dual))
    ptol = 1e-8 if loss == "binary_cross_entropy" else 1e-6
    assert np.max(np.abs(pert[0][0] - pert[1][0])) <= ptol * max(1.0, np.max(np.abs(pert[1][0])))
    assert np.max(np.abs(pert[0][1] - pert[1][1])) <= ptol * max(1e-3, np.max(np.abs(pert[1][1])))
    assert abs(pert[0][2] - pert[1][2]) <= ptol * max(1.0, pert[1][2]) and abs(pert[0][3] - pert[1][3]) <= ptol * max(1.0, pert[1][3])
    assert np.max(np.abs(pert[0][0] - ref.w)) > 1e-6 * max(1.0, np.max(np.abs(ref.w)))      # the perturbation mattered

    tol = 1e-9 if loss == "binary_cross_entropy" else 1e-7
    for cls in (ZHook, WHook, BothHooks):
        s = cls(X, y, max_iter=nit, tol=0.0, storage="f64", **kw)
        s.start_store(X, y, **{k: v for k, v in kw.items() if k != "B"})
        _quiet(s.main_loop, verbose=False)
        assert len(s.train_losses) == nit + 1
        assert np.allclose(s.train_losses, ref.objective[: nit + 1], rtol=tol, atol=tol), (cls.__name__, s.train_losses[-1])
        st = s._last
        assert abs(st.primal - ref.primal[-1]) <= tol * max(1.0, ref.primal[-1]), cls.__name__
        assert abs(st.dual - ref.dual[-1]) <= tol * max(1.0, ref.dual[-1]), cls.__name__
        assert abs(st.rho - ref.rho[-1]) <= 1e-15 * ref.rho[-1]
        assert np.max(np.abs(s.w.reshape(-1) - ref.w)) <= tol * max(1.0, np.max(np.abs(ref.w))), cls.__name__
        assert np.max(np.abs(s.lagrangian.reshape(-1) - ref.lam)) <= 10 * tol * max(1e-3, np.max(np.abs(ref.lam)))


def test_w_step_ahead_of_the_host_is_invisible(R):
    """Single-sweep lasso iterations enqueue the NEXT w-step before the host has read the
    current statistics (api.hip: rbl_phase_finish).  Whatever is called between two iterations
    must see w_k and must not change the trajectory: the same solve (a) uninterrupted, (b) with
    state reads / objective / accuracy calls between the iterations, (c) with
    the look-ahead disabled, gives bit-identical iterates."""
    import subprocess
    import sys
    import os
    from oracle import problems
    X, y = problems.make_problem(4000, 140, seed=9)

    def run(poke):
        s = R.ADMMmethod(X, y, "erm", "binary_cross_entropy", l1_reg=0.01, storage="f64", tol=0.0, max_iter=30)._s
        ws, hist = [], []
        for i in range(30):
            st = s.step(i % 3 == 0)
            hist.append((st.primal, st.dual, st.rho))
            if poke:
                w = s.get_state(want_z=False, want_lam=False)["w"]       # reads w_prev while w_{k+1} is in flight
                ws.append(w.copy())
                if i % 4 == 1:
                    s.risk(w)                                            # cancels the look-ahead
                if i % 4 == 2:
                    s.accuracy(w)
        return np.array(hist), s.get_state(), ws

    h0, st0, _ = run(False)
    h1, st1, ws = run(True)
    assert np.array_equal(h0, h1)
    for key in ("w", "z", "lam"):
        assert np.array_equal(st0[key], st1[key]), key
    # the w read after iteration k is what the dual residual of iteration k+1 is measured against
    for i in range(1, 29):
        assert abs(np.linalg.norm(ws[i] - ws[i - 1]) - h1[i, 1]) <= 1e-12 * max(1.0, h1[i, 1])
    code = ("import sys, json, numpy as np; sys.path.insert(0, %r); import admm_for_rank_based_loss_amd as R;"
            "from oracle import problems; X, y = problems.make_problem(4000, 140, seed=9);"
            "s = R.ADMMmethod(X, y, 'erm', 'binary_cross_entropy', l1_reg=0.01, storage='f64', tol=0.0, max_iter=30)._s;"
            "h = [(lambda st: (st.primal, st.dual, st.rho))(s.step(i %% 3 == 0)) for i in range(30)];"
            "print(json.dumps(dict(h=h, w=s.get_state()['w'].tolist())))") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, RBL_NO_SPECULATE="1")
    out = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert np.array_equal(np.array(r["h"]), h0) and np.array_equal(np.array(r["w"]), st0["w"])


def test_ridge_by_eigendecomposition_variant():
    """RBL_RIDGE_EIG=1: the l2 w-step through a one-time Jacobi eigendecomposition of G (eig.hip) - opt-in
    (its setup does not pay back within a solve), kept exact: the same iterates as the CG default and as the
    oracle, in a process of its own (the variable is read once)."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, json, numpy as np; sys.path.insert(0, %r)\n"
        "import admm_for_rank_based_loss_amd as R\n"
        "from oracle import problems, admm\n"
        "for n, d, kw in ((1500, 24, dict(weight_function='superquantile', loss='binary_cross_entropy', l2_reg=0.01, args=[0.5])),\n"
        "                 (2000, 301, dict(weight_function='erm', loss='hinge', l2_reg=0.01)),\n"
        "                 (900, 1001, dict(weight_function='aorr', loss='binary_cross_entropy', l2_reg=1e-4, args=[0.2, 0.8]))):\n"
        "    X, y = problems.make_problem(n, d, seed=5)\n"
        "    ref = admm.admm_solve(X, y, max_iter=12, mode='exact', tol=0.0, **kw)\n"
        "    s = R.ADMMmethod(X, y, max_iter=12, tol=0.0, storage='f64', **kw)._s\n"
        "    tol = 1e-9 if kw['loss'] == 'binary_cross_entropy' else 1e-7\n"
        "    for i in range(12):\n"
        "        st = s.step(True)\n"
        "        assert st.inner_iters == 1, st.inner_iters\n"
        "        assert abs(st.primal - ref.primal[i]) <= tol * max(1.0, ref.primal[i]), (d, i)\n"
        "        assert abs(st.dual - ref.dual[i]) <= tol * max(1.0, ref.dual[i]), (d, i)\n"
        "    assert np.max(np.abs(s.get_state()['w'] - ref.w)) <= tol * max(1.0, np.max(np.abs(ref.w)))\n"
        "print('OK')\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RBL_RIDGE_EIG="1"), capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), (out.stdout[-500:], out.stderr[-2000:])


def test_stop_tolerance_is_literal(R):
    """algorithms.py:137 stops when both residuals are below tol.  The library takes tol literally:
    with the reference default it reports convergence at the iteration the oracle stops at; with
    tol = 0 it never does, and a fixed-length run stays in the single-sweep steady state past that
    point (a library-side default for tol <= 0 once made such runs fall out of it)."""
    from oracle import problems, admm
    # d = 140: fp64 rows of more than 32 16-byte packets, the narrowest the single-sweep kernel takes
    # (sweep_erm.hip: sweep_erm_supported); narrower problems run the two-sweep path and never set "fused"
    X, y = problems.make_problem(3000, 140, seed=21)
    kw = dict(weight_function="erm", loss="binary_cross_entropy", l1_reg=0.01)
    ref = admm.admm_solve(X, y, max_iter=400, mode="exact", tol=1e-4, **kw)
    assert ref.converged and 20 < ref.iters < 400
    s = R.ADMMmethod(X, y, max_iter=400, tol=1e-4, storage="f64", **kw)._s
    stop = None
    for k in range(400):
        if s.step(False).converged:
            stop = k + 1
            break
    assert stop == ref.iters
    s0 = R.ADMMmethod(X, y, max_iter=400, tol=0.0, storage="f64", **kw)._s
    total = ref.iters + 40
    flags = [(st.converged, st.fused, st.mispredicted) for st in (s0.step(False) for _ in range(total))]
    assert not any(f[0] for f in flags)
    assert all(f[1] for f in flags) and sum(f[2] for f in flags) <= 2
    w_ref = admm.admm_solve(X, y, max_iter=total, mode="exact", tol=0.0, **kw).w
    assert np.max(np.abs(s0.get_state()["w"] - w_ref)) <= 1e-10 * max(1.0, np.max(np.abs(w_ref)))


def test_objective_golden_g7(R):
    g = load_golden("g7_objective.npz")
    X, y, w = g["X"], g["y"], g["w"]
    for k in range(int(g["ncases"])):
        cfg = json.loads(str(g[f"c{k}_name"]))
        o = R.rankbasedObjective(X, y, cfg["weight_function"], cfg["loss"], l2_reg=cfg.get("l2_reg"),
                                 l1_reg=cfg.get("l1_reg"), B=cfg.get("B"), args=cfg["args"], storage="f64")
        val, ref = o.get_arrogate_loss(w), float(g[f"c{k}_val"])
        assert abs(val - ref) <= 1e-12 * max(1.0, abs(ref)), (cfg, val, ref)


def test_accuracy_mirror(R):
    """calculate_accuracy (reference src/util/calculate_acc.py:3-19), incl. its hinge quirk."""
    import importlib
    acc_mod = importlib.import_module("admm_for_rank_based_loss_amd.src.util.calculate_acc")
    rng = np.random.default_rng(8)
    X = rng.standard_normal((5000, 13))
    w = rng.standard_normal((13, 1))
    y = np.where(X @ w + 0.7 * rng.standard_normal((5000, 1)) >= 0, 1, -1)
    for thr in (0.5, 0.3, 0.8):
        p = 1 / (1 + np.exp(-(X @ w)))
        ref = np.mean(np.where(p >= thr, 1, -1) == y)
        got = acc_mod.calculate_accuracy(w, X, y, threshold=thr, loss="binary_cross_entropy")
        assert abs(got - ref) <= 2.0 / 5000           # fp32 storage may flip rows sitting on the threshold
    assert acc_mod.calculate_accuracy(w, X, y, loss="hinge") == np.mean(y == 1)
    with pytest.raises(ValueError, match="is not supported"):
        acc_mod.calculate_accuracy(w, X, y, loss="square")


def test_fair_statistics_mirror(R):
    """calculate_statistics (reference src/util/fair_metric.py:3-41) against a NumPy restatement
    of the same definitions on the same inputs (fp64 storage: bit-level agreement of counts)."""
    import importlib
    fm = importlib.import_module("admm_for_rank_based_loss_amd.src.util.fair_metric")
    rng = np.random.default_rng(12)
    n, d = 4000, 9
    X = rng.standard_normal((n, d))
    w = rng.standard_normal((d, 1))
    group = (rng.random(n) < 0.35).astype(int)
    y = np.where(X @ w + 0.8 * rng.standard_normal((n, 1)) + 0.3 * group[:, None] >= 0, 1, -1)
    for thr in (0.5, 0.4):
        p = (1 / (1 + np.exp(-(X @ w)))).reshape(-1)
        pred = (p >= thr).astype(int)
        y01 = (y.reshape(-1) + 1) // 2
        st = {}
        for g in (0, 1):
            m = group == g
            st[g] = dict(P=pred[m].mean(), TP=np.sum(m & (pred == 1) & (y01 == 1)), FN=np.sum(m & (pred == 0) & (y01 == 1)),
                         TN=np.sum(m & (pred == 0) & (y01 == 0)), FP=np.sum(m & (pred == 1) & (y01 == 0)))
        tpr = {g: st[g]["TP"] / (st[g]["TP"] + st[g]["FN"]) for g in st}
        fpr = {g: st[g]["FP"] / (st[g]["FP"] + st[g]["TN"]) for g in st}
        fnr = {g: st[g]["FN"] / (st[g]["TP"] + st[g]["FN"]) for g in st}
        b = p - y01 + 1
        mu = b.mean()
        ref = (st[1]["P"] - st[0]["P"], st[1]["P"] / st[0]["P"], tpr[1] - tpr[0],
               0.5 * (fpr[1] - fpr[0] + tpr[1] - tpr[0]), np.mean((b / mu) * np.log(b / mu)), fnr[1] - fnr[0])
        got = fm.calculate_statistics(w, X, y, group, threshold=thr)
        assert np.allclose(got, ref, rtol=2e-3, atol=2e-3), (thr, got, ref)   # fp32 storage may move rows at the threshold


def test_z_step_golden_g4(R):
    g = load_golden("g4_zstep.npz")
    for k in range(int(g["ncases"])):
        cfg = json.loads(str(g[f"c{k}_name"]))
        X, y, w, lam, zref = g[f"c{k}_X"], g[f"c{k}_y"], g[f"c{k}_w"], g[f"c{k}_lam"], g[f"c{k}_z"]
        kw = {a: cfg[a] for a in ("l2_reg", "l1_reg", "B", "args") if a in cfg}
        s = R.ADMMmethod(X, y, cfg["weight_function"], cfg["loss"], storage="f64", **kw)
        s.w, s.lagrangian, s.rho = w, lam, cfg["rho"]
        s._z_subproblem()
        tol = 1e-7 if cfg["loss"] == "binary_cross_entropy" else 1e-2      # hinge: reference bisection noise
        assert np.max(np.abs(s.z - zref)) <= tol, cfg


def test_synthetic_generator_and_state_roundtrip(R):
    s = R.Solver(20000, 37, "erm", "binary_cross_entropy", reg=0.01, wstep=1)
    s.generate_synthetic(seed=17)
    D = s.get_D()
    yv = s.labels()
    assert set(np.unique(yv)) == {-1.0, 1.0}
    Xs = -yv[:, None] * D
    assert np.max(np.abs(Xs.mean(axis=0))) < 1e-5 and np.max(np.abs(Xs.std(axis=0) - 1)) < 1e-4
    acc = np.mean(np.sign(Xs @ np.linalg.lstsq(Xs, yv, rcond=None)[0]) == yv)
    assert acc > 0.7                                                        # informative features exist
    st1 = s.step(True)
    state = s.get_state()
    s2 = R.Solver(20000, 37, "erm", "binary_cross_entropy", reg=0.01, wstep=1)
    s2.generate_synthetic(seed=17)
    assert np.array_equal(s2.get_D(), D)                                    # counter-based: reproducible
    s2.set_state(w=state["w"], z=state["z"], lam=state["lam"], rho=state["rho"], iter=state["iter"])
    a, b = s.step(True), s2.step(True)
    assert a.primal == b.primal and a.dual == b.dual and a.objective == b.objective   # deterministic kernels


def test_sharded_driver_on_gpu_world1_and_buffer_views(R):
    """The multi-GPU driver with one rank: the torch views of the library's exchange buffers
    alias device memory (zero copy) and the phase-by-phase path equals rbl_step."""
    import torch
    from admm_for_rank_based_loss_amd.dist import ShardedADMM, GpuEngine
    outs = []
    for use_driver in (False, True):
        s = R.Solver(30000, 64, "superquantile", "binary_cross_entropy", reg=0.01, wstep=2, args=[0.5], tol=0.0)
        s.generate_synthetic(seed=5)
        hist = []
        if use_driver:
            eng = GpuEngine(s, 0)
            drv = ShardedADMM(eng)
            drv.setup_gram()
            q = eng.buf("q")
            # exchange buffer = [q (ld) | D^T lambda seed (ld) | ||z||^2 | primal^2 | sum loss]
            assert q.is_cuda and q.dtype == torch.float64 and q.numel() == 2 * 64 + 3
            assert eng.buf("red").data_ptr() == q.data_ptr() + 8 * (2 * 64 + 1)
            for _ in range(6):
                st = drv.step(True)
                hist.append((st.primal, st.dual, st.objective))
            # the view sees what the library wrote (q = D^T c of the last iteration)
            assert float(q.abs().sum()) > 0
        else:
            s.gram()
            for _ in range(6):
                st = s.step(True)
                hist.append((st.primal, st.dual, st.objective))
        outs.append((np.array(hist), s.get_state()["w"]))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("loss,reg_kind", [("binary_cross_entropy", "l1_reg"), ("hinge", "l2_reg")])
def test_single_sweep_paths(loss, reg_kind):
    """erm runs one sweep of D per iteration (sweep_erm.hip).  The same solve (a) fused,
    (b) with every 3rd rho prediction deliberately corrupted (verification + unfused redo),
    (c) with the fusion disabled - must give the same iterates, and match the oracle."""
    import subprocess
    import sys
    import os
    from oracle import problems, admm
    here = os.path.dirname(os.path.abspath(__file__))
    runs = {}
    for name, env in (("fused", {}), ("mispredict", {"RBL_DEBUG_MISPREDICT_EVERY": "3"}), ("unfused", {"RBL_NO_FUSE": "1"})):
        e = dict(os.environ)
        e.update(env)
        out = subprocess.run([sys.executable, os.path.join(here, "_fused_probe.py"), loss, reg_kind], env=e,
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        runs[name] = json.loads(out.stdout.strip().splitlines()[-1])
    assert runs["fused"]["fused"] >= 38 and runs["fused"]["mispredicted"] == 0
    assert runs["mispredict"]["mispredicted"] >= 10
    assert runs["unfused"]["fused"] == 0
    X, y = problems.make_problem(3000, 160, seed=21)
    ref = admm.admm_solve(X, y, "erm", loss, max_iter=40, mode="exact", tol=0.0, **{reg_kind: 0.01})
    tol = 1e-9 if loss == "binary_cross_entropy" else 1e-7
    for name, r in runs.items():
        hist = np.array(r["hist"])
        assert np.allclose(hist[:, 2], ref.rho, rtol=1e-15), name
        assert np.allclose(hist[:, 0], ref.primal, rtol=tol, atol=tol), name
        assert np.allclose(hist[:, 3], ref.objective[1:], rtol=tol), name
        assert np.max(np.abs(np.array(r["w"]) - ref.w)) <= tol * max(1.0, np.max(np.abs(ref.w))), name
        assert np.max(np.abs(np.array(r["z"]) - ref.z)) <= 10 * tol * max(1.0, np.max(np.abs(ref.z))), name
    # the three paths agree with each other far below the oracle tolerance
    for name in ("mispredict", "unfused"):
        assert np.max(np.abs(np.array(runs[name]["w"]) - np.array(runs["fused"]["w"]))) <= 1e-12
        assert np.max(np.abs(np.array(runs[name]["lam"]) - np.array(runs["fused"]["lam"]))) <= 1e-12 * max(
            1.0, np.max(np.abs(runs["fused"]["lam"])))


@pytest.mark.parametrize("rows,cols,storage,loss,reg_kind", [
    (700, 1100, "f32", "binary_cross_entropy", "l1_reg"),    # workgroup-per-row kernel, 2 packets per thread
    (517, 2300, "f32", "hinge", "l2_reg"),                   # 5 packets per thread, ragged last super-batch
    (300, 3100, "f64", "binary_cross_entropy", "l1_reg"),    # fp64 storage, 4 packets per thread
    (260, 7000, "f32", "binary_cross_entropy", "l2_reg"),    # 4 packets per thread, d >> n
    (1030, 900, "f64", "hinge", "l1_reg"),                   # wave-per-row kernel, 8 passes (fp64 only)
    (901, 333, "f32", "binary_cross_entropy", "l1_reg"),     # d not a multiple of 4 (padded columns), 2 passes
    (645, 1001, "f32", "hinge", "l2_reg"),                   # d = 1001 (the AoRR driver's intercept column), 4 passes
    (1203, 131, "f64", "binary_cross_entropy", "l2_reg"),    # 2 passes in fp64, odd everything
    (5, 1100, "f32", "binary_cross_entropy", "l2_reg"),      # fewer rows than one super-batch, wide rows
    (17, 2300, "f64", "hinge", "l2_reg"),                    # one full super-batch + 1 row
    (3, 260, "f32", "binary_cross_entropy", "l1_reg"),       # 3 rows, wave-per-row kernel
    (33, 520, "f64", "binary_cross_entropy", "l2_reg"),      # 2 super-batches + 1 row
    (300, 2048, "f32", "hinge", "l1_reg"),                   # the widest row of the wave-per-row kernel (fp32: 8 packets per lane)
    (7, 9000, "f32", "binary_cross_entropy", "l1_reg"),      # 5 packets per thread (C5's kernel): fewer rows than one sub-batch row set
    (49, 10000, "f32", "hinge", "l2_reg"),                   # C5's width: one super-batch of 48 rows + 1 row
    (130, 11000, "f32", "binary_cross_entropy", "l2_reg"),   # 6 packets per thread, two rows per sub-batch (round 3)
    (70, 15000, "f32", "hinge", "l1_reg"),                   # 8 packets per thread: the widest shape (w = 117 KB of LDS)
])
def test_single_sweep_wide_rows(rows, cols, storage, loss, reg_kind):
    """The single-sweep path for every row width (sweep_erm.hip: wave-per-row up to 4 / 8 passes,
    workgroup-per-row above): same iterates as the two-sweep path on the same data and storage."""
    import subprocess
    import sys
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    runs = {}
    for name, env in (("fused", {}), ("unfused", {"RBL_NO_FUSE": "1"})):
        e = dict(os.environ)
        e.update(env)
        out = subprocess.run([sys.executable, os.path.join(here, "_fused_probe.py"), loss, reg_kind, str(rows), str(cols),
                              storage], env=e, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        runs[name] = json.loads(out.stdout.strip().splitlines()[-1])
    assert runs["fused"]["fused"] >= 38 and runs["fused"]["mispredicted"] == 0
    assert runs["unfused"]["fused"] == 0
    hf, hu = np.array(runs["fused"]["hist"]), np.array(runs["unfused"]["hist"])
    assert np.array_equal(hf[:, 2], hu[:, 2])                       # same rho schedule
    assert np.allclose(hf[:, 0], hu[:, 0], rtol=1e-9, atol=1e-12)   # primal residual
    assert np.allclose(hf[:, 3], hu[:, 3], rtol=1e-10)              # objective
    for key in ("w", "z", "lam"):
        a, b = np.array(runs["fused"][key]), np.array(runs["unfused"][key])
        assert np.max(np.abs(a - b)) <= 1e-10 * max(1.0, np.max(np.abs(b))), key


@pytest.mark.parametrize("n,d", [(2, 1), (3, 2), (17, 3), (64, 5), (100, 1), (257, 33), (40, 90), (1000, 7)])
def test_small_and_odd_shapes(R, n, d):
    """Edge shapes through the whole iteration (single rows, d = 1, d > n, sizes that are not
    multiples of any tile): 6 iterations against the oracle for an erm and a rank-weighted
    configuration, both losses."""
    from oracle import problems, admm
    rng = np.random.default_rng(n * 131 + d)
    X = rng.standard_normal((n, d))
    y = np.where(rng.random((n, 1)) < 0.5, -1, 1).astype(np.int64)
    cfgs = [dict(weight_function="erm", loss="binary_cross_entropy", l2_reg=0.05),
            dict(weight_function="erm", loss="hinge", l1_reg=0.05),
            dict(weight_function="extremile", loss="binary_cross_entropy", l1_reg=0.05, args=[2.0]),
            dict(weight_function="superquantile", loss="hinge", l2_reg=0.05, args=[0.5])]
    for kw in cfgs:
        ref = admm.admm_solve(X, y, max_iter=6, mode="exact", tol=0.0, use_c=True, **kw)
        s = R.ADMMmethod(X, y, max_iter=6, tol=0.0, storage="f64", **kw)
        for i in range(6):
            st = s._s.step(True)
            assert abs(st.primal - ref.primal[i]) <= 1e-7 * max(1.0, ref.primal[i]), (kw, i)
            assert abs(st.objective - ref.objective[i + 1]) <= 1e-7 * max(1.0, abs(ref.objective[i + 1])), (kw, i)
        assert np.max(np.abs(s._s.get_state()["w"] - ref.w)) <= 1e-7 * max(1.0, np.max(np.abs(ref.w))), kw
