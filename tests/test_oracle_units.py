"""CPU oracle vs the golden vectors generated from the real reference
(tests/golden/make_goldens.py).  G1 element prox, G2 PAV, G3 EHRM PAV, G4 z-step,
G5/G6 w-step (one-sided: the reference's inner solvers are loose), G7 objective,
G8 weights.  Tolerances are the ones SURVEY.md section 8c states."""
import json

import numpy as np
import pytest

from conftest import load_golden
from oracle import prox, pav, weights, wstep, objective, admm

LOSS = {0: "binary_cross_entropy", 1: "hinge"}


def _elem_obj(loss, sigma, rho, m, x):
    return sigma * objective.sample_losses(loss, x - (1.0 if loss == "hinge" else 0.0) * 0) + rho / 2 * (x - m) ** 2


def _prox_obj(loss, sigma, rho, m, x):
    l = prox.softplus(x) if loss == "binary_cross_entropy" else np.maximum(1.0 + x, 0.0)
    return float(np.sum(sigma * l + rho / 2 * (x - m) ** 2))


def test_g1_element_prox():
    g = load_golden("g1_prox.npz")
    n_tight = 0
    for k in range(int(g["ncases"])):
        rho, lid = g[f"c{k}_meta"]
        loss = LOSS[int(lid)]
        sigma, m, xref = g[f"c{k}_sigma"], g[f"c{k}_m"], g[f"c{k}_x"].reshape(-1)
        xf = prox.prox_faithful(loss, sigma, rho, m)
        # the faithful restatement reproduces the reference's own (inexact) output
        assert np.max(np.abs(xf - xref)) <= 1e-6 * max(1.0, np.max(np.abs(xref))), (k, loss, rho)
        xe = prox.prox_exact(loss, sigma, rho, m)
        # exact is never worse than the reference on the element objective (one-sided)
        fe, fr = _prox_obj(loss, sigma, rho, m, xe), _prox_obj(loss, sigma, rho, m, xref)
        assert fe <= fr + 1e-12 * max(1.0, abs(fr)), (k, loss, rho, fe, fr)
        # where the reference itself converged, both agree to 1e-9
        if loss == "binary_cross_entropy":
            gref = sigma * prox.sigmoid(xref) + rho * (xref - m)
            hess = sigma * prox.dsigmoid(xref) + rho
            if np.max(np.abs(gref / hess)) < 1e-10:
                assert np.max(np.abs(xe - xref)) <= 1e-9
                n_tight += 1
        else:
            conv = np.abs(xref - prox.prox_hinge_exact(sigma, rho, m)) < 1e-9
            n_tight += int(conv.all())
    assert n_tight >= 20


def test_prox_bce_exact_is_a_root():
    rng = np.random.default_rng(5)
    sigma = rng.random(20000) * 1e-2
    sigma[::7] = 0.0
    m = 6 * rng.standard_normal(20000)
    for rho in (2e-7, 1e-5, 1e-3, 1.0, 50.0):
        x = prox.prox_bce_exact(sigma, rho, m)
        gval = sigma * prox.sigmoid(x) + rho * (x - m)
        assert np.max(np.abs(gval) / (sigma + rho * np.abs(m) + 1e-300)) < 5e-16
        assert np.all(x <= m) and np.all(x >= m - sigma / rho)


def _pav_cases():
    g = load_golden("g2_pav.npz")
    return g, int(g["ncases"])


def test_g2_pav_exact_vs_reference():
    g, nc = _pav_cases()
    tight = 0
    for k in range(nc):
        rho, lid = g[f"c{k}_meta"]
        loss = LOSS[int(lid)]
        sigma, m, uref = g[f"c{k}_sigma"], g[f"c{k}_m"], g[f"c{k}_u"]
        name = str(g[f"c{k}_name"])
        u, _ = pav.pav_exact(loss, sigma, rho, m)
        assert np.all(np.diff(u) >= 0), name
        fe, fr = _prox_obj(loss, sigma, rho, m, u), _prox_obj(loss, sigma, rho, m, uref)
        # exact isotonic solution is optimal: never above the reference's objective
        if np.all(np.diff(uref) >= -1e-12):
            assert fe <= fr + 1e-10 * max(1.0, abs(fr)), (name, fe, fr)
        err = np.max(np.abs(u - uref))
        if loss == "binary_cross_entropy":
            # the reference's Newton stops on a batch ||delta||_2 < 1e-6
            # (individual_solver.py:103): its own output is only that accurate
            assert err <= 1e-6, (name, err)
        # hinge: the bisection early exit leaves up to ~sigma/(2 rho) error on small
        # batches - only counted, not bounded
        tight += err <= 1e-9
    assert tight >= 30, tight


def test_g2_pav_faithful_vs_reference():
    g, nc = _pav_cases()
    ok = 0
    for k in range(nc):
        rho, lid = g[f"c{k}_meta"]
        loss = LOSS[int(lid)]
        sigma, m, uref = g[f"c{k}_sigma"], g[f"c{k}_m"], g[f"c{k}_u"]
        u, _ = pav.pav_faithful(loss, sigma, rho, m, maxiter=m.shape[0])
        err = np.max(np.abs(u - uref)) / max(1.0, np.max(np.abs(uref)))
        ok += err <= 1e-8
    # the sweep emulation reproduces the reference's output, artifacts included
    assert ok == nc, (ok, nc)


def test_pav_three_forms_agree():
    rng = np.random.default_rng(9)
    for fam, args in [("superquantile", [0.5]), ("extremile", [2.0]), ("esrm", [1.0]),
                      ("aorr", [0.2, 0.8]), ("aorr_dc", [80, 3]), ("ehrm", None)]:
        for loss in ("binary_cross_entropy", "hinge"):
            for rho in (2e-7, 1e-5, 1e-3, 1.0):
                n = int(rng.integers(150, 400))
                sa, sb = weights.get_weights(fam, n, args)
                m = np.sort(2 * rng.standard_normal(n) - 0.5)
                a, _ = pav.pav_exact_py(loss, sb, rho, m)
                b, _ = pav.pav_exact(loss, sb, rho, m)
                c, _ = pav.pav_tree_exact(loss, sb, rho, m)
                assert np.max(np.abs(a - b)) <= 1e-12
                assert np.max(np.abs(a - c)) <= 1e-11


def test_pav_edge_cases():
    for loss in ("binary_cross_entropy", "hinge"):
        for n in (1, 2, 3):
            s = np.full(n, 0.1)
            m = np.linspace(-1, 1, n)
            for f in (pav.pav_exact_py, pav.pav_exact, pav.pav_tree_exact):
                u, _ = f(loss, s, 1e-3, m)
                assert u.shape == (n,) and np.all(np.diff(u) >= 0)
        # all-equal m with increasing sigma pools everything
        n = 257
        s = np.linspace(0, 1, n)
        m = np.zeros(n)
        for f in (pav.pav_exact_py, pav.pav_exact, pav.pav_tree_exact):
            u, nb = f(loss, s, 1e-2, m)
            assert np.ptp(u) <= 1e-12
        # idempotence: PAV of an isotonic solution's own m/sigma is itself
        u1, _ = pav.pav_exact(loss, s, 1e-2, np.sort(np.random.default_rng(1).standard_normal(n)))
        assert np.all(np.diff(u1) >= 0)


def test_g3_ehrm_pav():
    g = load_golden("g3_pav_cpt.npz")
    exact_ok = 0
    nc = int(g["ncases"])
    for k in range(nc):
        rho, B, shift = g[f"c{k}_meta"]
        sa, sb, m, uref = g[f"c{k}_sa"], g[f"c{k}_sb"], g[f"c{k}_m"], g[f"c{k}_u"]
        za, _ = pav.ehrm_exact(sa, sb, B, rho, m, branch="a")
        zb, _ = pav.ehrm_exact(sa, sb, B, rho, m, branch="b")
        ea, eb = np.max(np.abs(za - uref)), np.max(np.abs(zb - uref))
        # the reference output is one of the two clean candidates ...
        # (to the reference's own Newton stop ||delta|| < 1e-4, PAV_cpt.py:72,85)
        assert min(ea, eb) <= 1e-7, (k, rho, shift, ea, eb)
        # ... and the singleton-stage scalar test picks it in the large majority of cases
        z, br = pav.ehrm_exact(sa, sb, B, rho, m)
        exact_ok += np.max(np.abs(z - uref)) <= 1e-7
        zf, picks = pav.ehrm_faithful(sa, sb, B, rho, m)
        assert np.max(np.abs(zf - uref)) <= 1e-9, (k, rho, shift)
    assert exact_ok >= int(0.85 * nc), (exact_ok, nc)


def test_g4_z_step():
    g = load_golden("g4_zstep.npz")
    for k in range(int(g["ncases"])):
        cfg = json.loads(str(g[f"c{k}_name"]))
        X, y, w, lam, zref = g[f"c{k}_X"], g[f"c{k}_y"], g[f"c{k}_w"], g[f"c{k}_lam"], g[f"c{k}_z"]
        n = X.shape[0]
        D = -y * X
        rho = cfg["rho"]
        m = (D @ w - lam / rho).reshape(-1)
        sa, sb = weights.get_weights(cfg["weight_function"], n, cfg["args"])
        z, _ = admm.z_step_exact(cfg["weight_function"], cfg["loss"], sa, sb, cfg.get("B"), rho, m)
        tol = 1e-8 if cfg["loss"] == "binary_cross_entropy" else 1e-2
        assert np.max(np.abs(z - zref.reshape(-1))) <= tol, cfg
        zf = admm.z_step_faithful(cfg["weight_function"], cfg["loss"], sa, sb, cfg.get("B"), rho, m)
        assert np.max(np.abs(zf - zref.reshape(-1))) <= 1e-7, cfg


def test_g56_w_step_one_sided():
    g = load_golden("g56_wstep.npz")
    X, y, z, lam, w0 = g["X"], g["y"], g["z"], g["lam"], g["w0"]
    rho, reg, t = g["meta"]
    D = -y * X
    G = D.T @ D
    c = (z + lam / rho).reshape(-1)
    q = D.T @ c
    kappa = reg / (2 * rho)

    def lasso_obj(w):
        return 0.5 * np.sum((c - D @ w) ** 2) + kappa * np.sum(np.abs(w))

    w, it = wstep.lasso_gram_exact(G, q, kappa, w0.reshape(-1))
    assert wstep.lasso_kkt_residual(G, q, kappa, w) <= 1e-8 * max(1.0, np.max(np.abs(q)))
    assert lasso_obj(w) <= lasso_obj(g["w_fista"]) + 1e-9 * abs(lasso_obj(w))

    def ridge_obj(w):
        return 0.5 * rho * np.sum((D @ w - c) ** 2) + 0.5 * reg * np.sum(w ** 2)

    w2 = wstep.ridge_gram_exact(G, q, rho, reg)
    assert ridge_obj(w2) <= ridge_obj(g["w_l2"].reshape(-1)) + 1e-12 * abs(ridge_obj(w2))
    assert np.max(np.abs(w2 - g["w_l2"].reshape(-1))) <= 5e-3 * np.max(np.abs(w2))
    wl = wstep.ridge_lbfgs_faithful(w0, z.reshape(-1), lam.reshape(-1), rho, G, D, reg)
    assert np.max(np.abs(wl - g["w_l2"].reshape(-1))) <= 1e-9

    def sm_obj(w):
        a = np.abs(w)
        return (0.5 * rho * np.sum((D @ w - c) ** 2)
                + np.sum(np.where(a <= t, 0.25 * reg * w ** 2 / t, 0.5 * reg * (a - 0.5 * t))))

    w3, _ = wstep.smooth_l1_gram_exact(G, q, rho, reg, t, w0.reshape(-1))
    assert sm_obj(w3) <= sm_obj(g["w_smooth"].reshape(-1)) + 1e-12 * abs(sm_obj(w3))
    wf, _ = wstep.fista_faithful(w0.reshape(-1), D.astype(np.float32), c, kappa)
    assert np.max(np.abs(wf - g["w_fista"])) <= 5e-3 * max(1.0, np.max(np.abs(g["w_fista"])))


def test_g7_objective():
    g = load_golden("g7_objective.npz")
    X, y, w = g["X"], g["y"], g["w"]
    for k in range(int(g["ncases"])):
        cfg = json.loads(str(g[f"c{k}_name"]))
        sa, _ = weights.get_weights(cfg["weight_function"], X.shape[0], cfg["args"])
        val = objective.objective(cfg["loss"], sa, X, y, w, cfg.get("l2_reg"), cfg.get("l1_reg"))
        ref = float(g[f"c{k}_val"])
        assert abs(val - ref) <= 1e-12 * max(1.0, abs(ref)), (cfg, val, ref)


def test_g8_weights():
    g = load_golden("g8_weights.npz")
    for k in range(int(g["ncases"])):
        cfg = json.loads(str(g[f"c{k}_name"]))
        a, b = weights.get_weights(cfg["weight_function"], cfg["n"], cfg["args"])
        assert np.max(np.abs(a - g[f"c{k}_a"])) <= 1e-15, cfg
        assert np.max(np.abs(b - g[f"c{k}_b"])) <= 1e-15, cfg


def test_weight_errors():
    with pytest.raises(ValueError, match="args for framework is None"):
        weights.get_weights("superquantile", 10, None)
    with pytest.raises(ValueError, match="Unrecognized framework"):
        weights.get_weights("nope", 10, [1])
    with pytest.raises(ValueError, match="need args"):
        weights.get_weights("aorr_dc", 10, [2, 5])


def test_zdist_rank_tree_and_rounds():
    """oracle/zdist.py: the merge tree over ranks pairs every rank exactly once per level where it
    has a partner, the round count decides every position, and a single-process simulation of the
    distributed z-step (P chunks, K-ary seam search on summaries) equals the exact PAV."""
    from oracle import zdist, pav, weights
    for world in (1, 2, 3, 5, 8, 13):
        for level in range(1, zdist.num_levels(world) + 1):
            seen = {}
            for r in range(world):
                info = zdist.seam_of(r, world, level)
                if info is None:
                    continue
                k, side, a0, b0, b1 = info
                assert a0 <= r < b1 and (side == 0) == (r < b0)
                seen.setdefault(k, set()).add(r)
            for k, rs in seen.items():
                assert rs == set(range(min(rs), max(rs) + 1))      # a seam's ranks are contiguous
    for K in (1, 3, 15, 63):
        for n in (0, 1, K, K + 1, 1000, 10 ** 6):
            s, r = n, 0
            while s > 0:                      # what a round leaves undecided in the worst case
                s = 0 if s <= K else s // (K + 1)
                r += 1
            assert zdist.num_rounds(n, K) >= r
    rng = np.random.default_rng(5)
    for trial in range(30):
        n, P, K = int(rng.integers(1, 3000)), int(rng.choice([2, 3, 4, 7, 8])), int(rng.choice([2, 5, 15, 63]))
        loss = ("binary_cross_entropy", "hinge")[trial % 2]
        fam, args = (("superquantile", [0.5]), ("extremile", [2.0]), ("aorr", [0.2, 0.8]), ("esrm", [1.0]))[trial % 4]
        sigma, _ = weights.get_weights(fam, n, args)
        rho = float(10.0 ** rng.uniform(-6, 0))
        m = np.sort(rng.standard_normal(n) * 2)
        if trial % 5 == 0:
            m = np.round(m, 1)
        cuts = np.sort(rng.integers(0, n + 1, size=P - 1))           # uneven chunks, some possibly empty
        b = np.concatenate([[0], cuts, [n]])
        chunks = [zdist.RankChunk(loss, rho, m[b[r]:b[r + 1]], sigma[b[r]:b[r + 1]]) for r in range(P)]
        rounds = zdist.num_rounds(max(c.n for c in chunks), K)
        for level in range(1, zdist.num_levels(P) + 1):
            bounds = np.array([c.bounds() for c in chunks])
            for r, c in enumerate(chunks):
                c.seam_setup(r, P, level, bounds)
            cand = part = None
            for _ in range(rounds):
                if part is not None:
                    for c in chunks:
                        c.update(cand, part, K, P, level)
                cand = np.concatenate([c.propose(K) for c in chunks])
                part = sum(c.evaluate(cand, K, P, level) for c in chunks)
            for c in chunks:
                c.update(cand, part, K, P, level)
            nseams = (P + 1) // 2
            tot = sum(c.pooled_sums(nseams) for c in chunks)
            for c in chunks:
                c.fill(tot)
        u = np.concatenate([c.u for c in chunks])
        ref = pav.pav_exact(loss, sigma, rho, m)[0] if n else np.zeros(0)
        assert np.max(np.abs(u - ref), initial=0.0) <= 1e-10 * max(1.0, np.max(np.abs(ref), initial=0.0)), (trial, n, P, K)


# --------------------------------------------------------------- synthetic generator (oracle/synth.py)
def test_philox_known_answers_and_generator_restatement():
    """The NumPy restatement of the device generator: Philox4x32-10 against the published Random123
    known-answer vectors (kat_vectors: philox4x32 10), slices regenerate independently of the range
    asked for (counter based), labels are balanced with ~1 % flips, columns are standardised with D = -y X
    (load_data.py:101-116 statistics, algorithms.py:23)."""
    from oracle import synth
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = synth.philox4x32_10(*[np.array([c], dtype=np.uint64) for c in ctr], *key)
        assert tuple(int(g[0]) for g in got) == want
    X, y = synth.raw_rows(17, 37, 0, 5000)
    Xs, ys = synth.raw_rows(17, 37, 1234, 1300)
    assert np.array_equal(X[1234:1300], Xs) and np.array_equal(y[1234:1300], ys)
    assert X.dtype == np.float32 and set(np.unique(y)) == {-1.0, 1.0} and abs(np.mean(y)) < 0.05
    special, mix, A, vertex = synth.special_columns(17, 37)
    assert len(set(special)) == 4 and all(0 <= c < 37 for c in special)
    assert sorted(vertex) == [0, 1, 2, 3] and A.shape == (4, 4) and np.all(np.abs(A) < 1)
    noise = [j for j in range(37) if j not in special]
    assert np.max(np.abs(X[:, noise].mean(axis=0))) < 0.06 and np.max(np.abs(X[:, noise].std(axis=0) - 1)) < 0.05
    # redundant columns are exact linear combinations of the informative ones (make_classification's n_redundant)
    f0, f1 = X[:, special[0]].astype(np.float64), X[:, special[1]].astype(np.float64)
    assert np.max(np.abs(X[:, special[2]] - (f0 * mix[0] + f1 * mix[2]))) < 1e-5
    D, yd = synth.standardized_D(17, 37, 5000)
    Xstd = -yd[:, None] * D
    assert np.max(np.abs(Xstd.mean(axis=0))) < 1e-12 and np.max(np.abs(Xstd.std(axis=0) - 1)) < 1e-12
    # the informative plane carries the label: the best of the linear classifiers along 180 directions of it beats 70 %
    # (seed 17 draws two classes that a line separates; one seed in three draws the XOR arrangement of the four
    # clusters, as make_classification does)
    P = Xstd[:, special[:2]]
    acc = max(np.mean(np.sign(P @ np.array([np.cos(t), np.sin(t)])) == yd) for t in np.linspace(0, np.pi, 180, endpoint=False))
    assert max(acc, 1 - acc) > 0.7, acc


def test_bounded_kkt_checker_of_the_full_size_tests():
    """tests/test_gpu_fullsize.py checks EHRM at 6.25 M rows through the KKT conditions of the order- and
    bound-constrained problem (z = max(B, PAV(sigma_b, m)) / min(B, PAV(sigma_a, m)), PAV_cpt.py:205-226).  The checker
    itself is pinned here on the oracle's exact EHRM z-step: it accepts both branches and rejects the wrong branch's
    weights and a block moved by 1e-6."""
    from test_gpu_fullsize import check_isotonic_kkt
    from oracle import weights, pav
    rng = np.random.default_rng(0)
    n = 3000
    sa, sb = weights.get_weights("ehrm", n)
    for B, shift, rho in ((-5.0, 0.0, 1e-4), (-5.0, -6.0, 1e-2), (0.5, 0.0, 1e-4)):
        m = np.sort(rng.normal(size=n) * 2 + shift)
        for br in ("a", "b"):
            z, _ = pav.ehrm_exact(sa, sb, B, rho, m, branch=br)
            good, bad = (sb, sa) if br == "b" else (sa, sb)
            kw = dict(lower=B) if br == "b" else dict(upper=B)
            assert check_isotonic_kkt("binary_cross_entropy", good, rho, m, z, **kw) >= 1
            if np.mean(z == B) < 0.6:     # (with nearly everything on the bound little is left to tell the weights apart)
                with pytest.raises(AssertionError):
                    check_isotonic_kkt("binary_cross_entropy", bad, rho, m, z, **kw)
            z2 = z.copy()
            k = n // 2
            z2[z2 == z2[k]] += 1e-6 * (1 + abs(z2[k]))
            with pytest.raises(AssertionError):
                check_isotonic_kkt("binary_cross_entropy", good, rho, m, np.maximum.accumulate(z2), **kw)
