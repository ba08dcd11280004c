"""Kernel-level parity: every HIP kernel of the hot path, called through the C ABI
(rbl_k_* entry points of include/rbl.h), against the CPU oracle on the same seeded
inputs and against the golden vectors generated from the reference.  Tolerances are
written next to each check; index/permutation work is bit-exact."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

LOSS = {0: "binary_cross_entropy", 1: "hinge"}
FAMILIES = [("superquantile", [0.5]), ("extremile", [2.0]), ("esrm", [1.0]), ("aorr", [0.2, 0.8]),
            ("aorr_dc", [80, 3]), ("erm", None)]


@pytest.fixture(scope="module")
def L():
    import admm_for_rank_based_loss_amd as rbl
    if rbl._lib.device_count() < 1:
        pytest.fail("no HIP device: the GPU tests must run the HIP library (no fallback)")
    return rbl._lib


# ------------------------------------------------------------------------------ prox
def test_prox_vs_oracle(L):
    from oracle import prox
    rng = np.random.default_rng(1)
    for n in (1, 2, 63, 64, 65, 1000, 100003):
        sigma = rng.random(n) * 1e-2
        sigma[rng.integers(0, n, size=max(1, n // 7))] = 0.0
        m = 6 * rng.standard_normal(n)
        for rho in (2e-7, 1e-5, 1e-3, 1.0, 40.0):
            for loss in ("binary_cross_entropy", "hinge"):
                x = L.k_prox(loss, sigma, rho, m)
                ref = prox.prox_exact(loss, sigma, rho, m)
                # same bracketed-Newton root to a few ulps of the bracket scale
                tol = 1e-12 * max(1.0, np.max(np.abs(ref)))
                assert np.max(np.abs(x - ref)) <= tol, (loss, n, rho)


def test_prox_golden_g1(L):
    g = load_golden("g1_prox.npz")
    from oracle import prox
    for k in range(int(g["ncases"])):
        rho, lid = g[f"c{k}_meta"]
        loss = LOSS[int(lid)]
        sigma, m, xref = g[f"c{k}_sigma"], g[f"c{k}_m"], g[f"c{k}_x"].reshape(-1)
        x = L.k_prox(loss, sigma, rho, m)
        lf = prox.softplus if loss == "binary_cross_entropy" else (lambda t: np.maximum(1 + t, 0))
        f_gpu = np.sum(sigma * lf(x) + rho / 2 * (x - m) ** 2)
        f_ref = np.sum(sigma * lf(xref) + rho / 2 * (xref - m) ** 2)
        assert f_gpu <= f_ref + 1e-12 * max(1.0, abs(f_ref))          # never worse than the reference
        if loss == "binary_cross_entropy":
            gref = sigma * prox.sigmoid(xref) + rho * (xref - m)
            if np.max(np.abs(gref / (sigma * prox.dsigmoid(xref) + rho))) < 1e-10:
                assert np.max(np.abs(x - xref)) <= 1e-9               # where the reference converged


def test_prox_empty(L):
    assert L.k_prox("hinge", np.zeros(0), 1.0, np.zeros(0)).shape == (0,)


# ------------------------------------------------------------------------------ sort
def test_sort_bit_exact(L):
    rng = np.random.default_rng(2)
    for n in (1, 2, 255, 256, 4095, 4096, 4097, 70001, 1 << 20):
        keys = rng.standard_normal(n) * 10.0 ** rng.integers(-3, 4, size=n)
        if n > 10:
            keys[rng.integers(0, n, size=n // 5)] = keys[rng.integers(0, n, size=n // 5)]   # ties
            keys[:3] = [0.0, np.inf, -np.inf]
        out, perm = L.k_sort(keys)
        ref = np.argsort(keys, kind="stable")
        # stable (key, index) order: the permutation and the keys are bit-exact
        assert np.array_equal(perm, ref.astype(np.uint32))
        assert np.array_equal(out.view(np.uint64), keys[ref].view(np.uint64))
    # the key transform is a total order on the bits: -0.0 sorts before +0.0
    out, perm = L.k_sort(np.array([0.0, -0.0, 1.0, -1.0]))
    assert list(perm) == [3, 1, 0, 2]


def test_sort_sorted_and_reversed(L):
    for n in (5000, 123457):
        a = np.linspace(-3, 3, n)
        for keys in (a, a[::-1].copy(), np.zeros(n), np.full(n, -1.5)):
            out, perm = L.k_sort(keys)
            assert np.array_equal(out, np.sort(keys))
            assert np.array_equal(perm, np.argsort(keys, kind="stable").astype(np.uint32))


# ------------------------------------------------------------------------------- PAV
def test_pav_vs_oracle(L):
    from oracle import pav, weights
    rng = np.random.default_rng(3)
    for fam, args in FAMILIES + [("ehrm", None)]:
        for loss in ("binary_cross_entropy", "hinge"):
            for rho in (2e-7, 1e-5, 1e-3, 1.0):
                n = int(rng.integers(900, 5000))
                sa, sb = weights.get_weights(fam, n, args)
                m = np.sort(2 * rng.standard_normal(n) - 0.5)
                u, merges = L.k_pav(loss, sb, rho, m)
                ref, _ = pav.pav_exact(loss, sb, rho, m)
                assert np.all(np.diff(u) >= 0), (fam, loss, rho)
                # the isotonic solution is unique: two exact algorithms agree to rounding
                assert np.max(np.abs(u - ref)) <= 1e-10 * max(1.0, np.max(np.abs(ref))), (fam, loss, rho, n)


def test_pav_wave_search_regression(L):
    """Inputs captured from iteration 19 of a superquantile solve (n = 1500): the wave-cooperative
    64-ary search once returned lo - 1 when exactly 64 positions were left and none satisfied the
    predicate, which pooled 3 positions too many at the top of the tile.  Expected values: the
    oracle's exact PAV on the same inputs."""
    g = load_golden("g10_pav_wave_search_regression.npz")
    u, _ = L.k_pav("binary_cross_entropy", g["sigma"], float(g["rho"]), g["m"])
    assert np.max(np.abs(u - g["u"])) <= 1e-12


def test_pav_many_sizes_vs_oracle(L):
    """many (n, family, rho) combinations - partial tiles, sizes around multiples of 64 and of the
    2048-position tile, several upper levels - against the oracle's exact PAV"""
    from oracle import pav, weights
    rng = np.random.default_rng(11)
    sizes = [65, 127, 128, 129, 1023, 1025, 1088, 1500, 1984, 2047, 2048, 2049, 2112, 4097, 6200, 8191, 8256, 20000, 70001]
    fams = FAMILIES + [("ehrm", None)]
    for n in sizes:
        for rep in range(3):
            fam, args = fams[int(rng.integers(len(fams)))]
            loss = ("binary_cross_entropy", "hinge")[int(rng.integers(2))]
            rho = float(10.0 ** rng.uniform(-6.5, 0.5))
            sa, sb = weights.get_weights(fam, n, args)
            sg = sb if rep % 2 else sa
            m = np.sort(rng.standard_normal(n) * float(10.0 ** rng.uniform(-1, 1)) + rng.uniform(-2, 2))
            if rep == 2:
                m = np.round(m, 2)          # many ties
            u, _ = L.k_pav(loss, sg, rho, m)
            ref, _ = pav.pav_exact(loss, sg, rho, m)
            assert np.max(np.abs(u - ref)) <= 1e-9 * max(1.0, np.max(np.abs(ref))), (n, fam, loss, rho, rep)


def test_pav_golden_g2(L):
    g = load_golden("g2_pav.npz")
    tight = 0
    for k in range(int(g["ncases"])):
        rho, lid = g[f"c{k}_meta"]
        loss = LOSS[int(lid)]
        sigma, m, uref = g[f"c{k}_sigma"], g[f"c{k}_m"], g[f"c{k}_u"]
        u, _ = L.k_pav(loss, sigma, rho, m)
        err = np.max(np.abs(u - uref))
        if loss == "binary_cross_entropy":
            assert err <= 1e-6, (str(g[f"c{k}_name"]), err)   # reference Newton stop: ||delta||_2 < 1e-6
        tight += err <= 1e-9
    assert tight >= 30


def test_pav_large_and_pathological(L):
    from oracle import pav, weights
    rng = np.random.default_rng(4)
    n = 300000
    for fam, args, loss, rho in [("superquantile", [0.5], "binary_cross_entropy", 1e-5),
                                 ("aorr", [0.2, 0.8], "hinge", 2e-7),
                                 ("extremile", [2.0], "binary_cross_entropy", 1e-3),
                                 ("esrm", [1.0], "hinge", 1e-4)]:
        sa, _ = weights.get_weights(fam, n, args)
        m = np.sort(2 * rng.standard_normal(n))
        u, merges = L.k_pav(loss, sa, rho, m)
        ref, nb = pav.pav_exact(loss, sa, rho, m)
        assert np.max(np.abs(u - ref)) <= 1e-9 * max(1.0, np.max(np.abs(ref))), (fam, loss, rho)
    # EHRM at the same size: the CPT weights pool short blocks all over the upper part of the order (the sequential
    # bottom step), long ones at small rho (cooperative fills of the one-launch upper levels), both branches
    sa, sb = weights.get_weights("ehrm", n)
    for rho, shift in ((1e-4, 0.0), (3e-4, -6.0), (2e-6, 0.0)):
        m = np.sort(2 * rng.standard_normal(n) + shift)
        z, br = L.k_pav_ehrm(sa, sb, -5.0, rho, m)
        zo, bo = pav.ehrm_exact(sa, sb, -5.0, rho, m)
        assert br == (0 if bo == "a" else 1), (rho, shift)
        assert np.max(np.abs(z - zo)) <= 1e-9 * max(1.0, np.max(np.abs(zo))), (rho, shift)
    # all-equal m with increasing sigma pools everything into one block
    n = 5000
    u, _ = L.k_pav("binary_cross_entropy", np.linspace(0, 1, n), 1e-2, np.zeros(n))
    assert np.ptp(u) <= 1e-12
    # idempotence property: PAV of (sigma, m) twice through the same kernel is stable
    s = rng.random(n)
    m = np.sort(rng.standard_normal(n))
    u1, _ = L.k_pav("hinge", s, 0.5, m)
    u2, _ = L.k_pav("hinge", s, 0.5, m)
    assert np.array_equal(u1, u2)
    for nn in (1, 2, 3):
        u, _ = L.k_pav("binary_cross_entropy", np.full(nn, 0.1), 1e-3, np.linspace(-1, 1, nn))
        assert u.shape == (nn,) and np.all(np.diff(u) >= 0)


@pytest.mark.parametrize("spec", [None, "0", "-1"], ids=["speculate_b", "speculate_a", "separate_test"])
def test_pav_ehrm_golden_g3(L, spec, monkeypatch):
    """EHRM z-step on sorted m against the reference's own outputs and the oracle; the automatic branch choice runs with
    branch b speculated (the default: the tree of b and the singleton-stage sums in one kernel, the other tree only when
    the exact test contradicts), with branch a speculated (so that the cases whose answer is b take the fall-back), and
    through round 2's separate test kernel (RBL_EHRM_SPEC=-1)."""
    from oracle import pav
    if spec is not None:
        monkeypatch.setenv("RBL_EHRM_SPEC", spec)
    g = load_golden("g3_pav_cpt.npz")
    for k in range(int(g["ncases"])):
        rho, B, shift = g[f"c{k}_meta"]
        sa, sb, m, uref = g[f"c{k}_sa"], g[f"c{k}_sb"], g[f"c{k}_m"], g[f"c{k}_u"]
        z, br = L.k_pav_ehrm(sa, sb, B, rho, m)
        zo, bo = pav.ehrm_exact(sa, sb, B, rho, m)
        assert br == (0 if bo == "a" else 1)
        assert np.max(np.abs(z - zo)) <= 1e-10
        assert np.max(np.abs(z - uref)) <= 1e-7, (k, rho, shift)     # reference Newton stop 1e-4
        for forced in (0, 1):
            zf, brf = L.k_pav_ehrm(sa, sb, B, rho, m, branch=forced)
            zof, _ = pav.ehrm_exact(sa, sb, B, rho, m, branch="ab"[forced])
            assert brf == forced and np.max(np.abs(zf - zof)) <= 1e-10


# ---------------------------------------------------------------------------- sweeps
@pytest.mark.parametrize("storage", ["f32", "f64"])
def test_gemv_gemvt(L, storage):
    rng = np.random.default_rng(5)
    for n, d in [(1, 1), (3, 2), (17, 5), (64, 20), (257, 33), (1000, 100), (513, 255), (2000, 1000),
                 (700, 1001), (300, 2500), (150, 9000)]:
        D = rng.standard_normal((n, d))
        if storage == "f32":
            D = D.astype(np.float32).astype(np.float64)      # compare on exactly representable data
        w = rng.standard_normal(d)
        c = rng.standard_normal(n)
        v = L.k_gemv(D, w, storage)
        q = L.k_gemvt(D, c, storage)
        # fp64 accumulation of d (resp. n) products: a few ulps of the absolute sum
        assert np.max(np.abs(v - D @ w)) <= 1e-13 * np.max(np.abs(D) @ np.abs(w) + 1), (n, d)
        assert np.max(np.abs(q - D.T @ c)) <= 1e-13 * np.max(np.abs(D.T) @ np.abs(c) + 1), (n, d)


def test_gemv_linearity_large(L):
    # size-independent property at a bench-like row length: D(a w1 + b w2) = a D w1 + b D w2
    rng = np.random.default_rng(6)
    n, d = 20000, 1000
    D = rng.standard_normal((n, d)).astype(np.float32).astype(np.float64)
    w1, w2 = rng.standard_normal(d), rng.standard_normal(d)
    lhs = L.k_gemv(D, 2.0 * w1 - 0.5 * w2)
    rhs = 2.0 * L.k_gemv(D, w1) - 0.5 * L.k_gemv(D, w2)
    assert np.max(np.abs(lhs - rhs)) <= 1e-11
    # <D w, c> == <w, D^T c>
    c = rng.standard_normal(n)
    assert abs(np.dot(L.k_gemv(D, w1), c) - np.dot(w1, L.k_gemvt(D, c))) <= 1e-9 * n


@pytest.mark.parametrize("storage", ["f32", "f64"])
def test_gram_mfma(L, storage):
    rng = np.random.default_rng(7)
    for n, d in [(5, 3), (100, 17), (1000, 100), (3000, 129), (4001, 300), (900, 1000)]:
        D = rng.standard_normal((n, d))
        if storage == "f32":
            D = D.astype(np.float32).astype(np.float64)
        G = L.k_gram(D, storage)
        ref = D.T @ D
        assert np.array_equal(G, G.T)
        assert np.max(np.abs(G - ref)) <= 1e-12 * n, (n, d)


# ---------------------------------------------------------------------------- w-step
def test_wstep_vs_oracle_and_golden(L):
    from oracle import wstep
    g = load_golden("g56_wstep.npz")
    X, y, z, lam, w0 = g["X"], g["y"], g["z"], g["lam"], g["w0"].reshape(-1)
    rho, reg, t = g["meta"]
    D = -y * X
    G = D.T @ D
    c = (z + lam / rho).reshape(-1)
    q = D.T @ c
    kappa = reg / (2 * rho)
    w1, it1 = L.k_wstep(1, G, q, rho, reg, w0)
    ref1, _ = wstep.lasso_gram_exact(G, q, kappa, w0)
    assert wstep.lasso_kkt_residual(G, q, kappa, w1) <= 1e-8 * np.max(np.abs(q))
    assert np.max(np.abs(w1 - ref1)) <= 1e-10 * max(1.0, np.max(np.abs(ref1)))
    obj = lambda w: 0.5 * np.sum((c - D @ w) ** 2) + kappa * np.sum(np.abs(w))
    assert obj(w1) <= obj(g["w_fista"]) + 1e-9 * abs(obj(w1))          # one-sided vs the reference's FISTA
    w2, it2 = L.k_wstep(2, G, q, rho, reg, w0)
    ref2 = wstep.ridge_gram_exact(G, q, rho, reg)
    assert np.max(np.abs(w2 - ref2)) <= 1e-10 * max(1.0, np.max(np.abs(ref2)))
    w3, it3 = L.k_wstep(3, G, q, rho, reg, w0, smooth_t=t)
    ref3, _ = wstep.smooth_l1_gram_exact(G, q, rho, reg, t, w0)
    assert np.max(np.abs(w3 - ref3)) <= 1e-10 * max(1.0, np.max(np.abs(ref3)))


@pytest.mark.parametrize("d", [700, 1500], ids=["d700_G_rows_in_LDS", "d1500_8_per_thread_G_from_L2"])
def test_persistent_d_space_wsteps_are_optimal(L, d):
    """The one-launch forms of the ridge CG and of the smoothed-l1 w-step (csrc/wstep.hip: k_cg_persist, k_ncg_persist with
    its linear first phase) at widths the iterate tests do not reach, judged by their optimality conditions
    (w_LBFGS.py:11-62): cold start, warm start from the solution of a slightly different right-hand side (the case the
    linear phase is made for), a start far from the solution's Huber pattern, and a Gram matrix with two exactly collinear
    columns (a wrong pattern then makes the linear system singular)."""
    rng = np.random.default_rng(d)
    n = 3000
    D = rng.standard_normal((n, d))
    D[:, 5] = 2.0 * D[:, 3] - D[:, 4]                       # exactly collinear, as the generator's redundant columns
    D[:, :8] *= 3.0
    G = D.T @ D
    c = rng.standard_normal(n)
    rho, reg = 1e-3, 0.05
    hub = lambda w, t: np.where(np.abs(w) <= t, reg * w / (2 * t), 0.5 * reg * np.sign(w))
    for trial, t in enumerate((1.0, 1e-2, 1e-4)):
        q = D.T @ (c + 0.1 * trial)
        w_r, _ = L.k_wstep(2, G, q, rho, reg, np.zeros(d))
        assert np.max(np.abs(rho * (G @ w_r - q) + reg * w_r)) <= 1e-10 * np.max(np.abs(rho * q)), ("ridge", d, t)
        scale = max(np.max(np.abs(rho * q)), 0.5 * reg)
        w_c, it_c = L.k_wstep(3, G, q, rho, reg, np.zeros(d), smooth_t=t)                       # cold
        assert np.max(np.abs(rho * (G @ w_c - q) + hub(w_c, t))) <= 1e-10 * scale, ("cold", d, t, it_c)
        q2 = q + 1e-4 * np.max(np.abs(q)) * rng.standard_normal(d)
        w_w, it_w = L.k_wstep(3, G, q2, rho, reg, w_c, smooth_t=t)                               # warm: pattern (nearly) stable
        assert np.max(np.abs(rho * (G @ w_w - q2) + hub(w_w, t))) <= 1e-10 * scale, ("warm", d, t, it_w)
        w_f, it_f = L.k_wstep(3, G, q, rho, reg, 50.0 * rng.standard_normal(d), smooth_t=t)      # far: every sign assumed wrong
        assert np.max(np.abs(rho * (G @ w_f - q) + hub(w_f, t))) <= 1e-10 * scale, ("far", d, t, it_f)
        # (with collinear columns the minimiser need not be unique outside the quadratic zone: compare the objective)
        phi = lambda w: (0.5 * rho * w @ (G @ w) - rho * q @ w
                         + np.sum(np.where(np.abs(w) <= t, reg * w * w / (4 * t), 0.5 * reg * (np.abs(w) - 0.5 * t))))
        assert abs(phi(w_f) - phi(w_c)) <= 1e-10 * max(1.0, abs(phi(w_c))), ("same minimum", d, t, phi(w_f), phi(w_c))


def test_lasso_active_set_solver(L):
    """The l1 w-step's exact active-set (feature-sign) kernel: cold start, warm start, exactly
    collinear columns, supports beyond its capacity (falls back to FISTA) - always the exact
    lasso minimiser (KKT residual and agreement with the oracle's Gram-space solve)."""
    from oracle import wstep, problems
    rng = np.random.default_rng(11)
    for trial in range(5):
        n, d = 3000, int(rng.integers(40, 400))
        X, y = problems.make_problem(n, d, seed=100 + trial)       # has 2 exactly collinear columns
        D = -y * X
        G = D.T @ D
        q = D.T @ rng.standard_normal(n)
        qm = np.max(np.abs(q))
        for frac in (0.9, 0.5, 0.2, 0.02):
            kappa = frac * qm
            ref, _ = wstep.lasso_gram_exact(G, q, kappa, np.zeros(d))
            f = lambda w: 0.5 * w @ G @ w - q @ w + kappa * np.sum(np.abs(w))
            for w0 in (np.zeros(d), ref + 1e-3 * (ref != 0), np.where(rng.random(d) < 0.05, 0.01, 0.0)):
                w, it = L.k_wstep(1, G, q, 0.5, 2 * 0.5 * kappa, w0)       # reg/(2 rho) = kappa
                assert wstep.lasso_kkt_residual(G, q, kappa, w) <= 1e-9 * qm, (trial, frac)
                assert f(w) <= f(ref) + 1e-10 * abs(f(ref)) + 1e-12
                if np.count_nonzero(ref) <= 60:    # unique, well inside the active-set capacity
                    assert np.max(np.abs(w - ref)) <= 1e-9 * max(1.0, np.max(np.abs(ref))), (trial, frac)
    # all-zero solution and a single coordinate
    G = np.eye(8) * 3.0
    q = np.arange(8.0)
    w, _ = L.k_wstep(1, G, q, 0.5, 2 * 0.5 * 10.0, np.zeros(8))
    assert np.array_equal(w, np.zeros(8))
    w, _ = L.k_wstep(1, G, q, 0.5, 2 * 0.5 * 6.5, np.zeros(8))
    assert np.allclose(w, [0, 0, 0, 0, 0, 0, 0, (7 - 6.5) / 3.0], atol=1e-14)


# --------------------------------------------------------------------------- weights
def test_weights_golden_g8(L):
    import json
    g = load_golden("g8_weights.npz")
    for k in range(int(g["ncases"])):
        cfg = json.loads(str(g[f"c{k}_name"]))
        a, b = L.k_weights(cfg["weight_function"], cfg["n"], cfg["args"])
        assert np.max(np.abs(a - g[f"c{k}_a"])) <= 1e-15, cfg
        assert np.max(np.abs(b - g[f"c{k}_b"])) <= 1e-15, cfg
    with pytest.raises(ValueError, match="args for framework is None"):
        L.k_weights("superquantile", 10, None)
    with pytest.raises(ValueError, match="Unrecognized framework"):
        L.k_weights("nope", 10, [1])
    with pytest.raises(ValueError, match="need args"):
        L.k_weights("aorr_dc", 10, [2, 5])


# ------------------------------------------------------------------- synthetic generator
@pytest.mark.parametrize("n,d,storage", [(1000, 37, "f64"), (1000, 1000, "f32"), (3001, 6, "f64"), (257, 3, "f32")])
def test_synthetic_generator_vs_numpy_restatement(n, d, storage):
    """k_synth + column statistics + standardisation (synth.hip, sweep.hip, api.hip: rbl_synth_*) against
    oracle/synth.py: labels, flips, cluster draws and special-column positions bit for bit (integer Philox
    work); matrix values to 2e-5 (the device's __logf / __sincosf fast intrinsics against NumPy's float32
    log / sin / cos), through the whole pipeline D = -y * (x - mean) / std."""
    import admm_for_rank_based_loss_amd as rbl
    from oracle import synth
    s = rbl.Solver(n, d, "erm", "binary_cross_entropy", reg=0.01, wstep=2, storage=storage)
    s.generate_synthetic(seed=17)
    D, y = s.get_D(), s.labels()
    Dn, yn = synth.standardized_D(17, d, n, storage=storage)
    assert np.array_equal(y, yn)
    assert np.max(np.abs(D - Dn)) <= 2e-5 * max(1.0, np.max(np.abs(Dn))), float(np.max(np.abs(D - Dn)))
    # a shard of a larger problem regenerates the same raw rows: labels of rows [off, off + n) of a 4n-row problem
    s2 = rbl.Solver(n, d, "erm", "binary_cross_entropy", reg=0.01, wstep=2, storage=storage, n_total=4 * n,
                    row_offset=n + 3)
    s2.synth_local(seed=17)
    _, y2 = synth.raw_rows(17, d, n + 3, 2 * n + 3)
    assert np.array_equal(s2.labels(), y2)
