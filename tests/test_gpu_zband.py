"""The sort-free z-step for piecewise-constant rank weights (csrc/zband.hip; reference z_subproblem:
src/optim/algorithms.py:96-104 + src/util/pav.py:84-161).  The fast path is taken from 4 096 rows on; here it is
forced onto small problems (RBL_ZBAND_MIN_N) and checked
* against the CPU oracle's exact mode, iteration by iteration (the same bars as test_iterates_match_oracle_exact),
* against the library's own sort + merge-tree PAV path on device-generated problems of 200 000 - 400 000 rows,
  iterate by iterate, for every banded weight family, both losses, rank fractions that do and do not fall on a row,
* for what it reports: stats.zband = 1 (sort-free), 2 (not certified: redone with the sort), 0 (sort)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R():
    import admm_for_rank_based_loss_amd as rbl
    if rbl._lib.device_count() < 1:
        pytest.fail("no HIP device: the GPU tests must run the HIP library (no fallback)")
    return rbl


ORACLE_CASES = [
    ("superq_bce_l2", dict(weight_function="superquantile", loss="binary_cross_entropy", l2_reg=0.01, args=[0.5])),
    ("superq_0.37_bce_l1", dict(weight_function="superquantile", loss="binary_cross_entropy", l1_reg=0.01, args=[0.37])),
    ("superq_hinge_l2", dict(weight_function="superquantile", loss="hinge", l2_reg=0.01, args=[0.5])),
    ("aorr_hinge_l2", dict(weight_function="aorr", loss="hinge", l2_reg=1e-4, args=[0.2, 0.8])),
    ("aorr_bce_l2", dict(weight_function="aorr", loss="binary_cross_entropy", l2_reg=1e-4, args=[0.2, 0.8])),
    ("aorr_0.13_0.71_bce_l2", dict(weight_function="aorr", loss="binary_cross_entropy", l2_reg=1e-4, args=[0.13, 0.71])),
    ("aorr_dc_bce_l2", dict(weight_function="aorr_dc", loss="binary_cross_entropy", l2_reg=1e-4, args=[300, 40])),
]


@pytest.mark.parametrize("name,kw", ORACLE_CASES, ids=[c[0] for c in ORACLE_CASES])
def test_banded_z_step_iterates_match_oracle(R, name, kw, monkeypatch):
    from oracle import problems, admm
    monkeypatch.setenv("RBL_ZBAND_MIN_N", "16")
    X, y = problems.make_problem(1500, 24, seed=77)
    nit = 30
    ref = admm.admm_solve(X, y, max_iter=nit, mode="exact", tol=0.0, **kw)
    s = R.ADMMmethod(X, y, max_iter=nit, tol=0.0, storage="f64", **kw)
    tol = 1e-9 if kw["loss"] == "binary_cross_entropy" else 1e-7
    modes = []
    for i in range(nit):
        st = s._s.step(want_objective=True)
        modes.append(st.zband)
        assert abs(st.rho - ref.rho[i]) <= 1e-15 * ref.rho[i]
        assert abs(st.primal - ref.primal[i]) <= tol * max(1.0, ref.primal[i]), (name, i, st.primal, ref.primal[i], modes)
        assert abs(st.dual - ref.dual[i]) <= tol * max(1.0, ref.dual[i]), (name, i, modes)
        assert abs(st.objective - ref.objective[i + 1]) <= tol * max(1.0, abs(ref.objective[i + 1])), (name, i, modes)
    state = s._s.get_state()
    assert np.max(np.abs(state["w"] - ref.w)) <= tol * max(1.0, np.max(np.abs(ref.w))), modes
    assert np.max(np.abs(state["z"] - ref.z)) <= 10 * tol * max(1.0, np.max(np.abs(ref.z))), modes
    assert modes[0] == 0                       # iteration 0: all m equal, the sort path is taken outright
    assert set(modes) <= {0, 1, 2}
    assert modes.count(1) >= nit // 2, modes


DEVICE_CASES = [
    ("superq_bce", 400_000, 40, dict(weight_function="superquantile", loss="binary_cross_entropy", args=[0.5], reg=0.01, wstep=2)),
    ("superq_0.9_bce_l1", 300_007, 33, dict(weight_function="superquantile", loss="binary_cross_entropy", args=[0.9], reg=0.01, wstep=1)),
    ("superq_hinge", 250_000, 24, dict(weight_function="superquantile", loss="hinge", args=[0.5], reg=0.01, wstep=2)),
    ("aorr_hinge", 300_000, 21, dict(weight_function="aorr", loss="hinge", args=[0.2, 0.8], reg=1e-4, wstep=2)),
    ("aorr_bce", 300_011, 21, dict(weight_function="aorr", loss="binary_cross_entropy", args=[0.2, 0.8], reg=1e-4, wstep=2)),
    ("aorr_dc_bce", 200_000, 16, dict(weight_function="aorr_dc", loss="binary_cross_entropy", args=[150_000, 30_000], reg=1e-4, wstep=2)),
]


@pytest.mark.parametrize("name,n,d,kw", DEVICE_CASES, ids=[c[0] for c in DEVICE_CASES])
def test_banded_z_step_equals_sorted_path(R, name, n, d, kw, monkeypatch):
    """two handles on the same generated problem, one with the fast path switched off: same iterates.  The two paths
    sum the pooled block in different orders (~1e-16 relative), nothing else differs."""
    nit = 40

    def run(no_zband):
        monkeypatch.setenv("RBL_NO_ZBAND", "1" if no_zband else "0")
        s = R.Solver(n, d, kw["weight_function"], kw["loss"], reg=kw["reg"], wstep=kw["wstep"], args=kw["args"], tol=0.0,
                     storage="f64")
        s.generate_synthetic(seed=5)
        hist, modes = [], []
        for _ in range(nit):
            st = s.step(True)
            hist.append((st.primal, st.dual, st.rho, st.objective))
            modes.append(st.zband)
        return s.get_state(), np.array(hist), modes

    a, ha, ma = run(True)
    b, hb, mb = run(False)
    assert set(ma) == {0}
    assert mb[0] == 0 and set(mb) <= {0, 1, 2}
    tol = 1e-9 if kw["loss"] == "binary_cross_entropy" else 1e-7
    assert np.allclose(ha, hb, rtol=tol, atol=tol * 1e-3), (name, mb, np.max(np.abs(ha - hb)))
    assert np.max(np.abs(a["w"] - b["w"])) <= tol * max(1.0, np.max(np.abs(a["w"]))), mb
    assert np.max(np.abs(a["z"] - b["z"])) <= 10 * tol * max(1.0, np.max(np.abs(a["z"]))), mb
    # (aorr_dc: ... c | 0 (one rank) | frac (one rank) | 0 ...: what pools at its upper edge is the two single ranks alone)
    assert mb.count(1) >= nit - 10, (name, mb)


def test_banded_z_step_is_reproducible(R, monkeypatch):
    """fixed-order sums: two runs give the same bits"""
    outs = []
    for _ in range(2):
        s = R.Solver(200_000, 24, "superquantile", "binary_cross_entropy", reg=0.01, wstep=2, args=[0.5], tol=0.0, storage="f32")
        s.generate_synthetic(seed=9)
        modes = [s.step(False).zband for _ in range(12)]
        assert modes.count(1) >= 10, modes
        outs.append(s.get_state())
    assert np.array_equal(outs[0]["w"], outs[1]["w"]) and np.array_equal(outs[0]["z"], outs[1]["z"])


@pytest.mark.parametrize("loss,args,expect", [("hinge", [0.2, 0.8], 1), ("binary_cross_entropy", [0.45, 0.55], 2)],
                         ids=["certified", "redone_with_the_sort"])
def test_reading_z_mid_iteration_settles_the_fast_path(R, monkeypatch, loss, args, expect):
    """hook contract (algorithms.py:88-116 mirrored in src/optim/algorithms.py): an overridden w_subproblem that looks at
    z before the library's w-step must see a certified z - the read settles the sort-free z-step, and where that was
    not certified the sort path's z is what it gets.  Same trajectory, bit for bit, as the run that never looks."""
    from oracle import problems
    from admm_for_rank_based_loss_amd.src.optim.algorithms import Optimizer
    monkeypatch.setenv("RBL_ZBAND_MIN_N", "16")
    X, y = problems.make_problem(3000, 16, seed=5)
    kw = dict(weight_function="aorr", loss=loss, l2_reg=1e-4, args=args, max_iter=14, tol=0.0, storage="f64")

    class Peek(R.ADMMmethod):
        peeked = None

        def w_subproblem(self):
            self.peeked.append(self._s.get_state(want_lam=False)["z"].copy())
            return super().w_subproblem()

    runs = []
    for cls in (R.ADMMmethod, Peek):
        s = cls(X, y, **kw)
        s.peeked = []
        modes = []
        for i in range(kw["max_iter"]):
            Optimizer.main_loop(s, i, 0.0, False)          # one iteration (ADMMmethod.main_loop is the loop over them)
            modes.append(s._last.zband)
        runs.append((s._s.get_state(), modes, s.peeked))
    (a, ma, _), (b, mb, zs) = runs
    assert ma == mb and expect in mb, (ma, mb)               # the branch this case is about was exercised
    assert np.array_equal(a["w"], b["w"]) and np.array_equal(a["z"], b["z"]) and np.array_equal(a["lam"], b["lam"])
    assert np.array_equal(zs[-1], b["z"])                   # what the hook saw is the z of that iteration


def test_banded_z_step_against_the_reference_goldens(R, monkeypatch):
    """g4_zstep.npz holds z_subproblem() outputs of the REAL reference; its superquantile / aorr cases through the
    sort-free z-step (forced onto 500 rows, iteration counter past 0 so that the fast path is taken)."""
    import json
    from conftest import load_golden
    monkeypatch.setenv("RBL_ZBAND_MIN_N", "16")
    g = load_golden("g4_zstep.npz")
    seen = 0
    for k in range(int(g["ncases"])):
        cfg = json.loads(str(g[f"c{k}_name"]))
        if cfg["weight_function"] not in ("superquantile", "aorr"):
            continue
        X, y, w, lam, zref = g[f"c{k}_X"], g[f"c{k}_y"], g[f"c{k}_w"], g[f"c{k}_lam"], g[f"c{k}_z"]
        kw = {a: cfg[a] for a in ("l2_reg", "l1_reg", "B", "args") if a in cfg}
        s = R.ADMMmethod(X, y, cfg["weight_function"], cfg["loss"], storage="f64", **kw)
        s.w, s.lagrangian, s.rho = w, lam, cfg["rho"]
        s._s.set_state(iter=3)
        s._z_subproblem()
        z = s.z                                  # reading z settles the fast path (sort path if it was not certified)
        st = s._s.step(False)                    # (the step's report tells which path this handle is on)
        assert np.max(np.abs(z - zref)) <= (1e-7 if cfg["loss"] == "binary_cross_entropy" else 1e-2), cfg
        assert st.zband in (0, 1, 2)
        seen += 1
    assert seen >= 2


def test_uncertified_z_step_redo_is_bit_identical_with_the_lasso(R, monkeypatch):
    """rbl_phase_w runs the w-step on a q formed from a z that is not certified yet; when the verdict says
    "redo", z and q are rebuilt with the sort and the w-step runs a second time from w_k.  With the l1 w-step (the
    active-set lasso keeps no state between calls but its warm start, which rbl_phase_w puts back) the iterates up to the
    first CERTIFIED sort-free step must equal a run with the fast path switched off BIT FOR BIT - the redone iterations
    (mode 2) and the pauses between them are the sort path on identical inputs.  aorr[0.45, 0.55] / BCE: the narrow
    middle band is swallowed by the pooled block in the first iterations, which the fast path reports instead of
    answering."""
    from oracle import problems
    X, y = problems.make_problem(3000, 16, seed=5)
    nit = 14

    def run(no_zband):
        monkeypatch.setenv("RBL_NO_ZBAND", "1" if no_zband else "0")
        monkeypatch.setenv("RBL_ZBAND_MIN_N", "16")
        s = R.Solver(3000, 16, "aorr", "binary_cross_entropy", reg=1e-4, wstep=1, args=[0.45, 0.55], tol=0.0, storage="f64")
        s.set_data(X, y)
        out = []
        for _ in range(nit):
            st = s.step(True)
            state = s.get_state()
            out.append((st.zband, st.primal, st.dual, st.objective, state["w"].copy(), state["z"].copy()))
        return out

    a, b = run(True), run(False)
    modes = [o[0] for o in b]
    first_certified = modes.index(1) if 1 in modes else nit
    assert 2 in modes[:first_certified], modes          # at least one redone iteration was compared
    for k in range(first_certified):
        assert a[k][1:3] == b[k][1:3], (k, modes)                # primal and dual residual: the same bits
        assert np.array_equal(a[k][4], b[k][4]) and np.array_equal(a[k][5], b[k][5]), (k, modes)
        # the LOGGED objective of banded weights is summed per band by the sort-free risk kernel when the fast path is
        # on and along the sorted losses when it is off: the same number in a different order of additions
        assert abs(a[k][3] - b[k][3]) <= 1e-12 * max(1.0, abs(a[k][3])), (k, modes)
