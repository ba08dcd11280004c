"""Full-size property checks of the single-GPU bench configurations (nothing in the suite ran
C2 / C3 row counts).  The oracle cannot run 6-10 M rows x 1000 columns in test time, so at full size the
iteration is checked through size-independent properties, all evaluated in NumPy on vectors pulled from
the device:

* z-step (src/optim/algorithms.py:88-106): z is nondecreasing in the stable (m, row) order, and satisfies the
  KKT conditions of the isotonic problem  min sum_i sigma_i loss(u_i) + rho/2 (u_i - m_(i))^2, u_1 <= ... <= u_n
  on EVERY block: one-sided derivative sums over each block bracket 0, no prefix of a block wants to move
  down, no suffix up (erm: element-wise stationarity).  Together with monotonicity these are necessary and
  sufficient for the (unique) minimiser.
* sweeps: v = D w and the lambda update (:132) agree with NumPy on a 50 000-row slice of D, obtained by
  regenerating those rows with the counter-based generator in a small handle and pulling them with get_D().
* the logged objective (src/optim/objective.py:71-87) equals the oracle's objective_from_v on the device's
  full v; primal / dual residuals equal their NumPy restatement from the pulled vectors.
* shard-count invariance at full size: a 2-handle row-sharded run (ShardedADMM, ranks as threads with hub
  collectives) reproduces primal / dual / rho / objective of the single handle.

Which z-step was checked is part of the test: `rbl_stats.zband` is recorded per iteration and
asserted - C2sq and C3 must have had BOTH a certified sort-free z (mode 1, csrc/zband.hip) and a sort + merge-tree
PAV z (mode 0 / 2, csrc/sort.hip + pav.hip) through the KKT check; C2sq_sort pins the sorted path at 6 M positions
with RBL_NO_ZBAND=1.  C4shard (EHRM, 6.25 M rows: sort + both-branch prox + merge-tree PAV where the CPT weights pool
everywhere, src/util/PAV_cpt.py:169-293) and C5shard (1.25 M x 10 000 through the workgroup-per-row kernel
k_sweep_erm_wide) are one GPU's share of BASELINE configs[3] / configs[4] - the kernels those bench lines time.
"""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CFG = {
    # BASELINE configs[1]
    "C2": dict(n=6_000_000, d=1000, wf="erm", loss="binary_cross_entropy", wstep=1, reg=0.01, args=None),
    # its rank-weighted variant (sort + PAV at 6 M positions, BCE blocks)
    "C2sq": dict(n=6_000_000, d=1000, wf="superquantile", loss="binary_cross_entropy", wstep=2, reg=0.01, args=[0.5]),
    # BASELINE configs[2]: AoRR, intercept column -> d = 1001, hinge
    "C3": dict(n=10_000_000, d=1001, wf="aorr", loss="hinge", wstep=2, reg=1e-4, args=[0.2, 0.8]),
    # C2sq with the sort-free z-step switched off: radix sort + merge-tree PAV + scatter at 6 M positions
    "C2sq_sort": dict(n=6_000_000, d=1000, wf="superquantile", loss="binary_cross_entropy", wstep=2, reg=0.01, args=[0.5],
                      env={"RBL_NO_ZBAND": "1"}, nit=4),
    # one GPU's share of BASELINE configs[3]: EHRM (CPT weights: smooth, so the z-step keeps the sort)
    "C4shard": dict(n=6_250_000, d=1000, wf="ehrm", loss="binary_cross_entropy", wstep=2, reg=0.01, args=None, B=-5.0, nit=4),
    # one GPU's share of BASELINE configs[4]: d = 10 000 -> the workgroup-per-row single-sweep kernel
    "C5shard": dict(n=1_250_000, d=10_000, wf="erm", loss="binary_cross_entropy", wstep=1, reg=0.01, args=None, nit=4,
                    sub_rows=20_000),
}
NIT = {"C2": 3, "C2sq": 10, "C3": 10}
SEED = 17
SUB_ROWS, SUB_OFF = 50_000, 1_234_567


@pytest.fixture(scope="module")
def R():
    import admm_for_rank_based_loss_amd as rbl
    if rbl._lib.device_count() < 1:
        pytest.fail("no HIP device: the GPU tests must run the HIP library (no fallback)")
    return rbl


def _dloss(loss, x):
    """(left, right) derivative of the per-sample loss at x"""
    from oracle.prox import sigmoid
    if loss == "binary_cross_entropy":
        s = sigmoid(x)
        return s, s
    return (x > -1.0).astype(np.float64), (x >= -1.0).astype(np.float64)


def check_isotonic_kkt(loss, sigma, rho, m_sorted, z_sorted, rtol=1e-9, lower=None, upper=None):
    """KKT of the generalised isotonic problem on every block of equal z (see the module docstring).
    lower / upper: a one-sided bound u_i >= lower (u_i <= upper) on top of the order constraints - EHRM's
    z = max(B, PAV(sigma_b, m)) / min(B, PAV(sigma_a, m)) (PAV_cpt.py:205-226; clipping commutes with PAV for a
    one-sided bound).  The block sitting ON the bound cannot move through it, so only the directions
    that stay feasible are tested there: at a lower bound no suffix may want to move up, at an upper bound no prefix
    may want to move down."""
    n = z_sorted.shape[0]
    if lower is not None:
        assert np.all(z_sorted >= lower)
    if upper is not None:
        assert np.all(z_sorted <= upper)
    assert np.all(np.diff(z_sorted) >= 0), "z is not monotone in the sorted order of m"
    starts = np.flatnonzero(np.concatenate(([True], z_sorted[1:] != z_sorted[:-1])))
    dl, dr = _dloss(loss, z_sorted)
    lin = rho * (z_sorted - m_sorted)
    gl, gr = sigma * dl + lin, sigma * dr + lin            # one-sided derivatives of f_i at the block value
    scale = np.add.reduceat(sigma + np.abs(lin), starts)   # magnitude of the terms that cancel in a block
    bid = np.cumsum(np.concatenate(([0], (z_sorted[1:] != z_sorted[:-1]).astype(np.int64))))
    tol_i = rtol * scale[bid] + 1e-300
    # whole block: sum of left derivatives <= 0 <= sum of right derivatives
    tl, tr_ = np.add.reduceat(gl, starts), np.add.reduceat(gr, starts)
    zb = z_sorted[starts]
    on_lo = (zb == lower) if lower is not None else np.zeros(zb.shape, dtype=bool)
    on_hi = (zb == upper) if upper is not None else np.zeros(zb.shape, dtype=bool)
    assert np.all((tl <= rtol * scale + 1e-300) | on_lo) and np.all((tr_ >= -rtol * scale - 1e-300) | on_hi), \
        (float(np.max(tl / scale)), float(np.min(tr_ / scale)))
    # prefixes must not want to move down: sum_{i <= k in block} gl_i <= 0
    cl = np.cumsum(gl)
    base = np.concatenate(([0.0], cl))[starts][bid]
    # cumulative sums over n terms carry ~n eps of the largest prefix: compare on a per-block restart
    pre = cl - base
    err_guard = 64 * np.finfo(float).eps * np.maximum.accumulate(np.abs(cl))
    assert np.all((pre <= tol_i + err_guard) | on_lo[bid]), float(np.max((pre - err_guard) / scale[bid]))
    # suffixes must not want to move up: sum_{i >= k in block} gr_i >= 0  <=>  total_r - prefix_r(before k) >= 0
    cr = np.cumsum(gr)
    base_r = np.concatenate(([0.0], cr))[starts][bid]
    before = np.concatenate(([0.0], cr[:-1])) - base_r
    before[starts] = 0.0
    suf = tr_[bid] - before
    err_guard_r = 64 * np.finfo(float).eps * np.maximum.accumulate(np.abs(cr))
    assert np.all((suf >= -tol_i - err_guard_r) | on_hi[bid]), float(np.min((suf + err_guard_r) / scale[bid]))
    return starts.shape[0]


def _hub_driver(ShardedADMM, hub):
    import torch

    class Hub2(ShardedADMM):
        def _x(self, item):
            hub["slot"][self.rank] = item
            hub["bar"].wait()
            items = list(hub["slot"])
            hub["bar"].wait()
            return items

        def _allreduce(self, t):
            if t.numel() == 0:
                return
            torch.cuda.synchronize()
            items = self._x(t)
            if self.rank == 0:
                tot = items[0].clone()
                for x in items[1:]:
                    tot += x
                hub["total"] = tot
                torch.cuda.synchronize()
            hub["bar"].wait()
            t.copy_(hub["total"])
            torch.cuda.synchronize()
            hub["bar"].wait()

        def _gather_small(self, t):
            torch.cuda.synchronize()
            out = torch.cat([x.reshape(-1) for x in self._x(t.clone())])
            torch.cuda.synchronize()
            hub["bar"].wait()
            return out

        def _gather_counts(self, counts_dev):
            torch.cuda.synchronize()
            m = np.array([x.cpu().numpy() for x in self._x(counts_dev.clone())], dtype=np.int64)
            hub["bar"].wait()
            return m.reshape(self.world, self.world)

        def _alltoall(self, send, send_counts, recv, recv_counts):
            torch.cuda.synchronize()
            items = self._x((send, [int(c) for c in send_counts]))
            pos = 0
            for src, (buf, cnts) in enumerate(items):
                off, c = sum(cnts[: self.rank]), cnts[self.rank]
                assert c == int(recv_counts[src])
                recv[pos:pos + c].copy_(buf[off:off + c])
                pos += c
            torch.cuda.synchronize()
            hub["bar"].wait()

    return Hub2


def _sharded_rank(R, rank, world, cfg, nit, hub, out, errs):
    try:
        import torch
        from admm_for_rank_based_loss_amd.dist import ShardedADMM, GpuEngine, shard_rows
        torch.cuda.set_device(0)
        lo, cnt, _ = shard_rows(cfg["n"], world, rank)
        s = R.Solver(cnt, cfg["d"], cfg["wf"], cfg["loss"], reg=cfg["reg"], wstep=cfg["wstep"], args=cfg["args"],
                     B=cfg.get("B"), n_total=cfg["n"], row_offset=lo, tol=0.0, storage="f32")
        drv = _hub_driver(ShardedADMM, hub)(GpuEngine(s, 0), world=world, rank=rank)
        drv.setup_synthetic(SEED)
        drv.setup_gram()
        hist = []
        for _ in range(nit):
            st = drv.step(True)
            hist.append((st.primal, st.dual, st.rho, st.objective))
        out[rank] = np.array(hist)
        s.close()
    except BaseException as e:      # a dead rank must not leave the other one in a barrier
        errs.append((rank, repr(e)))
        hub["bar"].abort()


@pytest.mark.parametrize("name", ["C2", "C2sq", "C3", "C2sq_sort", "C4shard", "C5shard"])
def test_full_size_properties(R, name, monkeypatch):
    import torch
    from admm_for_rank_based_loss_amd.dist import GpuEngine
    from oracle import objective as oobj, weights, pav as opav
    cfg = CFG[name]
    for k, val in cfg.get("env", {}).items():
        monkeypatch.setenv(k, val)          # read by the handle at its first rank-weighted z-step
    n, d = cfg["n"], cfg["d"]
    loss, rw, ehrm = cfg["loss"], cfg["wf"] != "erm", cfg["wf"] == "ehrm"
    nit = cfg.get("nit", NIT.get(name, 3))
    sub_rows = cfg.get("sub_rows", SUB_ROWS)
    B = cfg.get("B")
    s = R.Solver(n, d, cfg["wf"], loss, reg=cfg["reg"], wstep=cfg["wstep"], args=cfg["args"], B=B, tol=0.0, storage="f32")
    eng = GpuEngine(s, 0)          # library kernels on torch's stream: the views below are ordered with them
    s.generate_synthetic(SEED)
    s.gram()

    def pull(which):
        return eng.buf(which).cpu().numpy().copy()

    sigma, sigma_b = weights.get_weights(cfg["wf"], n, cfg["args"])
    dev_sigma = s.sigma()
    if ehrm:
        # CPT weights (objective.py:148-164) are DIFFERENCES of distortion values of size ~1 that differ by ~1/n: one
        # ulp of pow() (device libm against NumPy's) is 1e-16 absolute = 1e-9 of a weight at n = 6.25 M - in the
        # reference's own Python floats as much as here.  Absolute bar; the z-step below is checked against the
        # weights the device holds.
        assert np.max(np.abs(dev_sigma[0] - sigma)) <= 4e-15 and np.max(np.abs(dev_sigma[1] - sigma_b)) <= 4e-15
        assert abs(dev_sigma[0].sum() - 1.0) <= 1e-12 and abs(dev_sigma[1].sum() - 1.0) <= 1e-12
        sigma_z, sigma_zb = dev_sigma
    else:
        assert np.allclose(dev_sigma[0], sigma, rtol=1e-13, atol=1e-300)  # sigma generator at full size (objective.py:97-136)
        sigma_z, sigma_zb = sigma, sigma_b

    hist, modes, kkt_modes, branches = [], [], [], []
    for it in range(nit):
        st0 = s.get_state()
        lam0, rho = st0["lam"], st0["rho"]
        s.phase_m()
        s.phase_z()
        z = s.get_state(want_lam=False)["z"]     # (reading z settles a sort-free z-step: certified, or redone with the sort)
        # ---- z-step properties.  m = v - lambda/rho with the v the iteration used
        v_prev = pull("v") if (it > 0 or rw) else None
        if v_prev is None:
            m = pull("m")                 # iteration 0 of erm: the unfused kernel wrote m
        else:
            m = v_prev - lam0 / rho
            if rw:
                assert np.array_equal(m, pull("m"))      # k_make_m: bit-exact restatement
        if rw:
            order = np.argsort(m, kind="stable")
            if ehrm:
                # z = max(B, PAV(sigma_b, m)) or min(B, PAV(sigma_a, m)): the branch is whichever side of B the
                # whole vector sits on; KKT with that branch's weights and the one-sided bound
                zs = z[order]
                br = 1 if np.all(zs >= B) else 0
                assert br == 1 or np.all(zs <= B)
                nblocks = check_isotonic_kkt(loss, sigma_zb if br else sigma_z, rho, m[order], zs,
                                             lower=B if br else None, upper=None if br else B)
                branches.append(br)
                if it in (0, nit - 1):
                    # the choice itself = the reference's singleton-stage scalar test (PAV_cpt.py:205-226)
                    # recomputed in NumPy with exact element solves
                    want = opav.ehrm_branch_exact(sigma_z, sigma_zb, B, rho, m[order])
                    assert br == (0 if want == "a" else 1), (it, br, want)
            else:
                nblocks = check_isotonic_kkt(loss, sigma, rho, m[order], z[order])
            assert 1 <= nblocks <= n
        else:
            # erm: every row its own block (prox is monotone in m for a constant sigma)
            dl, dr = _dloss(loss, z)
            lin = rho * (z - m)
            sc = sigma[0] + np.abs(lin)
            assert np.all(sigma[0] * dl + lin <= 1e-9 * sc) and np.all(sigma[0] * dr + lin >= -1e-9 * sc)
        s.phase_q()
        s.phase_w()
        s.phase_dual(True)
        st = s.phase_finish()
        hist.append((st.primal, st.dual, st.rho, st.objective))
        modes.append(int(st.zband))
        if ehrm:
            assert int(st.ehrm_branch) == branches[-1]
        # ---- dual update, residuals, objective from the pulled vectors
        new = s.get_state()
        w, v = new["w"], pull("v")
        lam_expect = lam0 + rho * (z - v)
        assert np.max(np.abs(new["lam"] - lam_expect)) <= 1e-12 * max(1e-6, np.max(np.abs(lam_expect)))
        assert abs(st.primal - np.linalg.norm(z - v)) <= 1e-10 * max(1.0, st.primal)
        assert abs(st.dual - np.linalg.norm(w - st0["w"])) <= 1e-10 * max(1.0, st.dual)
        kwreg = dict(l1_reg=cfg["reg"]) if cfg["wstep"] == 1 else dict(l2_reg=cfg["reg"])
        f_ref = oobj.objective_from_v(loss, sigma_z, v, w, **kwreg)     # (EHRM too: betas = alphas, objective.py:76)
        assert abs(st.objective - f_ref) <= 1e-10 * max(1.0, abs(f_ref)), (st.objective, f_ref)
        if it == nit - 1:
            w_last, v_last = w, v
    print("full-size %s: rbl_stats.zband per iteration %s" % (name, modes))      # (pytest -s shows it)
    # ---- which z-step the KKT check above has seen (rbl_stats.zband: 0 sort + PAV, 1 sort-free and certified,
    # 2 sort-free, not certified, redone with the sort; -1 erm)
    if name in ("C2sq", "C3"):
        assert modes[0] == 0, modes                       # iteration 0: every m equal, the fast path is skipped outright
        # both paths went through the KKT check above: certified sort-free z-steps (the first few attempts pool most
        # rows into one block and are redone with the sort, then the fast path pauses for 2, 4, ... iterations)
        assert modes.count(1) >= 3 and (0 in modes or 2 in modes), modes
        assert modes[-1] == 1, modes                      # ... and the steady state is the sort-free one
    elif name == "C2sq_sort":
        assert modes == [0] * nit, modes                  # RBL_NO_ZBAND=1: radix sort + merge-tree PAV every time
    elif ehrm:
        assert modes == [0] * nit, modes                  # CPT weights are smooth: EHRM keeps the sort
    else:
        assert modes == [-1] * nit, modes
    hist = np.array(hist)

    # ---- v = D w on two 50 000-row slices (one near the start, one at the END of the matrix: a launch that
    # wraps at 2^32 threads leaves the tail rows wrong) regenerated by the counter-based generator with the
    # GLOBAL column statistics of the full matrix (the generator gives the same rows under any sharding)
    y_full = s.labels()
    for sub_off in (min(SUB_OFF, n // 5), n - sub_rows - 321):
        sub = R.Solver(sub_rows, d, cfg["wf"], loss, reg=cfg["reg"], wstep=cfg["wstep"], args=cfg["args"], B=B, n_total=n,
                       row_offset=sub_off, tol=0.0, storage="f32")
        sub_eng = GpuEngine(sub, 0)
        sub.synth_local(SEED)
        sub_eng.buf("colstats").copy_(eng.buf("colstats"))
        torch.cuda.synchronize()
        sub.synth_finish()
        Dsub = sub.get_D()
        ysub = sub.labels()
        assert np.array_equal(ysub, y_full[sub_off:sub_off + sub_rows])
        v_sub = Dsub @ w_last
        got = v_last[sub_off:sub_off + sub_rows]
        # fp64 accumulation of d products, different summation order: d * eps * sum |a||b|
        bound = 4 * d * np.finfo(float).eps * (np.abs(Dsub) @ np.abs(w_last)) + 1e-300
        assert np.all(np.abs(got - v_sub) <= bound), (sub_off, float(np.max(np.abs(got - v_sub) / bound)))
        # standardised columns (load_data.py:115) and D = -y X (algorithms.py:23) in EVERY slice: unit second
        # moment, and X = -y D has centred columns
        assert np.max(np.abs(Dsub)) < 20 and abs(float(np.mean(Dsub * Dsub)) - 1.0) < 0.05
        Xsub = -ysub[:, None] * Dsub
        cm = np.abs(Xsub.mean(axis=0))
        assert np.max(cm) < 0.05, (sub_off, float(np.max(cm)))         # X columns are centred in every slice
        sub.close()
        del sub_eng
    s.close()
    del eng
    torch.cuda.empty_cache()

    # ---- shard-count invariance at full size: two row shards of the same problem
    hub = dict(bar=threading.Barrier(2), slot=[None, None], total=None)
    out, errs = [None, None], []
    ts = [threading.Thread(target=_sharded_rank, args=(R, r, 2, cfg, nit, hub, out, errs)) for r in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(900)
    assert not errs, errs
    assert np.array_equal(out[0], out[1])
    assert np.array_equal(out[0][:, 2], hist[:, 2])                                  # rho schedule: bit-identical
    assert np.allclose(out[0][:, [0, 1, 3]], hist[:, [0, 1, 3]], rtol=1e-8, atol=1e-11), (out[0], hist)
