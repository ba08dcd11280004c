"""The structure the sort-free z-step (csrc/zband.hip) relies on, checked on the CPU: oracle/zband.py (bands, one block
per band edge as the root of the pooled derivative, clamps in row order) against the exact PAV of oracle/pav.py on the
sorted vector (src/optim/algorithms.py:96-104 + src/util/pav.py:84-161), for the reference's banded weight families."""
import numpy as np
import pytest

from oracle import pav, weights, zband


def _exact(loss, sigma, rho, m):
    order = np.argsort(m, kind="stable")
    zs, _ = pav.pav_exact(loss, sigma, rho, m[order])
    z = np.empty_like(m)
    z[order] = zs
    return z


CASES = [("superquantile", [0.5]), ("superquantile", [0.37]), ("superquantile", [0.9]), ("aorr", [0.2, 0.8]), ("aorr", [0.13, 0.71]),
         ("aorr", [0.45, 0.55]), ("aorr_dc", [0.7, 0.1]), ("aorr_dc", [0.35, 0.3])]   # aorr_dc: (k, m) as fractions of n here


@pytest.mark.parametrize("loss", ["binary_cross_entropy", "hinge"])
@pytest.mark.parametrize("wf,args", CASES, ids=[f"{w}{a}" for w, a in CASES])
def test_banded_z_step_equals_exact_pav(loss, wf, args):
    rng = np.random.default_rng(hash((wf, tuple(args), loss)) % 2**32)
    seen = {zband.OK: 0}
    for trial in range(40):
        n = int(rng.choice([50, 333, 2000, 9001]))
        sigma, _ = weights.get_weights(wf, n, [int(args[0] * n), int(args[1] * n)] if wf == "aorr_dc" else args)
        # the regimes of an ADMM run: the shift sigma/rho of the prox from a few spreads of m ("one block holds most
        # rows") down to a thousandth of it ("a handful of rows pool")
        spread = float(10.0 ** rng.uniform(-3, 1))
        rho = float(np.max(sigma) / (spread * 10.0 ** rng.uniform(-3, 0.5)))
        m = rng.normal(0.0, spread, n) + rng.choice([-1.0, 0.0, 1.5])
        z, status = zband.z_step(loss, sigma, rho, m)
        seen[status] = seen.get(status, 0) + 1
        if status != zband.OK:
            assert z is None          # reported, never a wrong answer
            continue
        ref = _exact(loss, sigma, rho, m)
        tol = 1e-10 * max(1.0, np.max(np.abs(ref)))
        assert np.max(np.abs(z - ref)) <= tol, (wf, args, loss, n, rho, spread, float(np.max(np.abs(z - ref))))
    assert seen[zband.OK] >= 15, seen


def test_banded_z_step_reports_what_it_cannot_certify():
    n = 1000
    sigma, _ = weights.get_weights("superquantile", n, [0.5])
    z, status = zband.z_step("binary_cross_entropy", sigma, 1e-5, np.zeros(n))        # iteration 0: every m equal
    assert z is None and status == zband.TIE
    three = np.zeros(n)                                                                  # three single-rank bands in a row
    three[400:500] = 1e-3
    three[501], three[503] = 5e-4, 7e-4
    assert zband.z_step("binary_cross_entropy", three, 1e-3, np.random.default_rng(0).normal(size=n))[1] == zband.UNSUPPORTED
    smooth, _ = weights.get_weights("extremile", n, [2.0])                              # every rank its own weight
    assert zband.z_step("binary_cross_entropy", smooth, 1e-3, np.random.default_rng(0).normal(size=n))[1] == zband.UNSUPPORTED
    # AoRR with a narrow middle band and a tiny rho: the block at the lower edge swallows the whole middle band
    sigma_n, _ = weights.get_weights("aorr", n, [0.45, 0.55])
    st = zband.z_step("binary_cross_entropy", sigma_n, 1e-9, np.random.default_rng(1).normal(0, 0.01, n))[1]
    assert st in (zband.SWALLOW_R, zband.OK)


def test_bands_and_clusters():
    s, v = zband.bands_of(weights.superquantile(1000, 0.5))
    assert list(s) in ([0, 500, 501, 1000], [0, 500, 1000]) and v[0] == 0.0
    cl = zband.clusters_of(s, v)
    assert cl == [(0, len(v) - 1, True)]
    s, v = zband.bands_of(weights.aorr(1000, 0.2, 0.8))
    cl = zband.clusters_of(s, v)
    assert cl[0][2] is True and cl[-1][2] is False and cl[-1][1] == len(v) - 1      # the upper edge cannot pool


def test_banded_z_step_against_the_reference_goldens():
    """g4_zstep.npz: z = z_subproblem() of the REAL reference (tests/golden/make_goldens.py) - the banded families in it"""
    import json
    from conftest import load_golden
    g = load_golden("g4_zstep.npz")
    seen = 0
    for k in range(int(g["ncases"])):
        cfg = json.loads(str(g[f"c{k}_name"]))
        if cfg["weight_function"] not in ("superquantile", "aorr"):
            continue
        X, y, w, lam, zref = g[f"c{k}_X"], g[f"c{k}_y"], g[f"c{k}_w"], g[f"c{k}_lam"], g[f"c{k}_z"]
        rho = cfg["rho"]
        m = ((-y * X) @ w - lam / rho).reshape(-1)
        sigma, _ = weights.get_weights(cfg["weight_function"], X.shape[0], cfg["args"])
        z, status = zband.z_step(cfg["loss"], sigma, rho, m)
        assert status == zband.OK, cfg
        assert np.max(np.abs(z - zref.reshape(-1))) <= (1e-8 if cfg["loss"] == "binary_cross_entropy" else 1e-2), cfg
        seen += 1
    assert seen >= 2


@pytest.mark.parametrize("world", [1, 3])
def test_pass_based_restatement_equals_exact_pav(world):
    """oracle/zband.py:Passes (the steps of rbl_zbd_* / csrc/zband.hip: radix select on summed histograms, 16-candidate
    root passes on summed block sums, the undecided elements gathered) on rows split over `world` ranks"""
    rng = np.random.default_rng(5 + world)
    certified = 0
    for trial in range(60):
        loss = ("binary_cross_entropy", "hinge")[trial % 2]
        n = int(rng.choice([300, 2000, 9001]))
        wf, args = (("superquantile", [0.5]), ("superquantile", [0.37]), ("aorr", [0.2, 0.8]),
                    ("aorr_dc", [int(0.7 * n), int(0.1 * n)]))[trial % 4]
        sigma, _ = weights.get_weights(wf, n, args)
        spread = float(10.0 ** rng.uniform(-3, 1))
        rho = float(np.max(sigma) / (spread * 10.0 ** rng.uniform(-3, 0.5)))
        m = rng.normal(0.0, spread, n) + rng.choice([-1.0, 0.0, 1.5])
        cuts = np.linspace(0, n, world + 1).astype(int)
        ranks = [zband.Passes(loss, sigma) for _ in range(world)]
        for r, P in enumerate(ranks):
            P.begin(m[cuts[r]:cuts[r + 1]], rho)
        for p in range(6):
            h = sum(P.hist(p).astype(np.int64) for P in ranks)
            for P in ranks:
                P.scan(p, h)
        for k in ranks[0].root_clusters():
            for it in range(zband.ROOT_PASSES):
                tot = sum(P.eval(k) for P in ranks)
                for P in ranks:
                    P.decide(k, tot, it == zband.ROOT_PASSES - 1)
            packs = np.concatenate([P.gather(k) for P in ranks])
            for P in ranks:
                P.finish(k, packs, world)
        outs = [P.apply() for P in ranks]
        assert len({st for _, st in outs}) == 1                   # every rank reads the same verdict
        if outs[0][1] != zband.OK:
            continue
        certified += 1
        z = np.concatenate([zz for zz, _ in outs])
        ref = _exact(loss, sigma, rho, m)
        assert np.max(np.abs(z - ref)) <= 1e-10 * max(1.0, np.max(np.abs(ref))), (wf, loss, n, world)
    assert certified >= 45, certified
