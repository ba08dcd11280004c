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
}
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


def check_isotonic_kkt(loss, sigma, rho, m_sorted, z_sorted, rtol=1e-9):
    """KKT of the generalised isotonic problem on every block of equal z (see the module docstring)."""
    n = z_sorted.shape[0]
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
    assert np.all(tl <= rtol * scale + 1e-300) and np.all(tr_ >= -rtol * scale - 1e-300), \
        (float(np.max(tl / scale)), float(np.min(tr_ / scale)))
    # prefixes must not want to move down: sum_{i <= k in block} gl_i <= 0
    cl = np.cumsum(gl)
    base = np.concatenate(([0.0], cl))[starts][bid]
    # cumulative sums over n terms carry ~n eps of the largest prefix: compare on a per-block restart
    pre = cl - base
    err_guard = 64 * np.finfo(float).eps * np.maximum.accumulate(np.abs(cl))
    assert np.all(pre <= tol_i + err_guard), float(np.max((pre - err_guard) / scale[bid]))
    # suffixes must not want to move up: sum_{i >= k in block} gr_i >= 0  <=>  total_r - prefix_r(before k) >= 0
    cr = np.cumsum(gr)
    base_r = np.concatenate(([0.0], cr))[starts][bid]
    before = np.concatenate(([0.0], cr[:-1])) - base_r
    before[starts] = 0.0
    suf = tr_[bid] - before
    err_guard_r = 64 * np.finfo(float).eps * np.maximum.accumulate(np.abs(cr))
    assert np.all(suf >= -tol_i - err_guard_r), float(np.min((suf + err_guard_r) / scale[bid]))
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
                     n_total=cfg["n"], row_offset=lo, tol=0.0, storage="f32")
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


@pytest.mark.parametrize("name", ["C2", "C2sq", "C3"])
def test_full_size_properties(R, name):
    import torch
    from admm_for_rank_based_loss_amd.dist import GpuEngine
    from oracle import objective as oobj, weights
    cfg = CFG[name]
    n, d = cfg["n"], cfg["d"]
    loss, rw = cfg["loss"], cfg["wf"] != "erm"
    nit = 3
    s = R.Solver(n, d, cfg["wf"], loss, reg=cfg["reg"], wstep=cfg["wstep"], args=cfg["args"], tol=0.0, storage="f32")
    eng = GpuEngine(s, 0)          # library kernels on torch's stream: the views below are ordered with them
    s.generate_synthetic(SEED)
    s.gram()

    def pull(which):
        return eng.buf(which).cpu().numpy().copy()

    sigma = weights.get_weights(cfg["wf"], n, cfg["args"])[0]
    dev_sigma = s.sigma()[0]
    assert np.allclose(dev_sigma, sigma, rtol=1e-13, atol=1e-300)     # sigma generator at full size (objective.py:97-136)

    hist = []
    for it in range(nit):
        st0 = s.get_state()
        lam0, rho = st0["lam"], st0["rho"]
        s.phase_m()
        s.phase_z()
        z = s.get_state(want_lam=False)["z"]
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
        # ---- dual update, residuals, objective from the pulled vectors
        new = s.get_state()
        w, v = new["w"], pull("v")
        lam_expect = lam0 + rho * (z - v)
        assert np.max(np.abs(new["lam"] - lam_expect)) <= 1e-12 * max(1e-6, np.max(np.abs(lam_expect)))
        assert abs(st.primal - np.linalg.norm(z - v)) <= 1e-10 * max(1.0, st.primal)
        assert abs(st.dual - np.linalg.norm(w - st0["w"])) <= 1e-10 * max(1.0, st.dual)
        kwreg = dict(l1_reg=cfg["reg"]) if cfg["wstep"] == 1 else dict(l2_reg=cfg["reg"])
        f_ref = oobj.objective_from_v(loss, sigma, v, w, **kwreg)
        assert abs(st.objective - f_ref) <= 1e-10 * max(1.0, abs(f_ref)), (st.objective, f_ref)
        if it == nit - 1:
            w_last, v_last = w, v
    hist = np.array(hist)

    # ---- v = D w on two 50 000-row slices (one near the start, one at the END of the matrix: a launch that
    # wraps at 2^32 threads leaves the tail rows wrong) regenerated by the counter-based generator with the
    # GLOBAL column statistics of the full matrix (the generator gives the same rows under any sharding)
    y_full = s.labels()
    for sub_off in (SUB_OFF, n - SUB_ROWS - 321):
        sub = R.Solver(SUB_ROWS, d, cfg["wf"], loss, reg=cfg["reg"], wstep=cfg["wstep"], args=cfg["args"], n_total=n,
                       row_offset=sub_off, tol=0.0, storage="f32")
        sub_eng = GpuEngine(sub, 0)
        sub.synth_local(SEED)
        sub_eng.buf("colstats").copy_(eng.buf("colstats"))
        torch.cuda.synchronize()
        sub.synth_finish()
        Dsub = sub.get_D()
        ysub = sub.labels()
        assert np.array_equal(ysub, y_full[sub_off:sub_off + SUB_ROWS])
        v_sub = Dsub @ w_last
        got = v_last[sub_off:sub_off + SUB_ROWS]
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
