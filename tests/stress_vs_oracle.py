"""Randomized whole-solver parity run (TEST INFRASTRUCTURE, not collected by pytest): random
family / loss / regulariser / shape, 12 ADMM iterations on the GPU against the oracle's exact mode.
    python tests/stress_vs_oracle.py SEED TRIALS [MAX_ROWS]
fp32 storage is EXACT fp64 arithmetic on the fp32-rounded matrix: the oracle is fed the rounded X (what the
device stores), so one tolerance (1e-8 BCE, 1e-6 hinge) holds for both storage types and every family.
Against the UNROUNDED matrix (`RAW=1` in the environment) fp32 storage is a perturbed problem: round 1
measured 2e-4 on the convex families over 12 iterations and up to 7e-4 (EHRM) / 5e-2 (AoRR) on the
non-convex ones, whose trajectories amplify the 6e-8 relative perturbation of D.
Round 1: 660 fp64 trials (seeds 1-5 and 7, rows up to 80 000), worst relative deviation 8e-14.  Round 2 (seed 21,
150 trials with sADMM runs (*), rank-weighted widths up to 1030 and 40 % fp32 storage): no mismatch; seed 31, 150
trials with RBL_ZBAND_MIN_N=16 in the environment (the 54 superquantile / aorr draws take the sort-free z-step and
objective of csrc/zband.hip): no mismatch.  Round 3, final library (one-block-per-CU sweep shapes, 8-packet fp32 rows,
persistent w-steps, block values from the expansion around the block mean): seeds 41 (120) and 51 (500 trials): worst
primal 2.8e-10, worst w 1.8e-11 (75 EHRM and 51 sADMM draws among the 500); seeds 43 (80) and 53 (300) with
RBL_ZBAND_MIN_N=16: worst 1.9e-11; after the linear first phase of the smoothed-l1 w-step: seeds 61 (250) and 71 (700 trials, 69 sADMM
draws, worst of those 1.7e-12); no mismatch.  Seed 81 (600 trials, final library): 2 flagged, both aorr / hinge / sADMM draws whose
first w-step returns w ~ 1e-9 (dual residual 1.1e-9) and whose final w is exactly 0 on both sides: m = D w - lambda/rho is then
a handful of tied values split only by D w ~ 1e-9 - below what either w-step resolves (gradient tolerance 1e-13 absolute) -
so the ORDER across aorr's band edges differs between the two solvers from iteration 1 on and the non-convex trajectory
with it (primal residuals 2e-4 apart, same final w).  The same two trials are flagged with round 2's final library (commit
ce57536) and with either w-step form: not something round 3 changed.  STRESS_VERBOSE=1 prints the trajectories of a flagged trial."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import admm_for_rank_based_loss_amd as R
from oracle import problems, admm
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
fams = [("erm", None), ("superquantile", [0.5]), ("superquantile", [0.9]), ("extremile", [2.0]), ("esrm", [1.0]),
        ("aorr", [0.2, 0.8]), ("aorr_dc", [80, 3]), ("ehrm", None)]
bad = 0
t0 = time.time()
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    fam, args = fams[int(rng.integers(len(fams)))]
    loss = "binary_cross_entropy" if fam == "ehrm" else ("binary_cross_entropy", "hinge")[int(rng.integers(2))]
    n = int(rng.integers(200, int(sys.argv[3]) if len(sys.argv) > 3 else 6000)); d = int(rng.integers(3, 60))
    if fam == "erm" and rng.random() < 0.5:
        d = int(rng.integers(140, 400))          # single-sweep kernel territory (fp64: PK > 32)
        n = int(rng.integers(200, 3000))
        if rng.random() < 0.3:
            d = int(rng.integers(1030, 2600))    # workgroup-per-row kernel
            n = int(rng.integers(20, 600))
    if fam != "erm" and rng.random() < 0.35:
        d = int(rng.integers(130, 1030))         # v-only single-sweep kernel territory (all pass counts)
        n = int(rng.integers(200, 2500))
    storage = "f32" if rng.random() < 0.4 else "f64"
    kw = dict(weight_function=fam, loss=loss, args=args)
    if fam == "ehrm": kw["B"] = -5
    if fam == "aorr_dc": kw["args"] = [min(80, n // 3), 3]
    regk = "l1_reg" if rng.random() < 0.4 and fam != "ehrm" else "l2_reg"
    kw[regk] = float(10.0 ** rng.uniform(-4, -1))
    smooth = regk == "l1_reg" and rng.random() < 0.3        # smoothADMMmethod: Huber-smoothed w-step + t schedule
    t_init = float(10.0 ** rng.uniform(-3, 0)) if smooth else 1.0
    X, y = problems.make_problem(n, d, seed=int(rng.integers(1 << 30)))
    raw = os.environ.get("RAW") == "1"
    if storage == "f32" and not raw:
        X = X.astype(np.float32).astype(np.float64)      # the matrix the device stores
    nit = 24 if smooth else 12
    try:
        ref = admm.admm_solve(X, y, max_iter=nit, mode="exact", tol=0.0, smooth=smooth, t=t_init, **kw)
        s = (R.smoothADMMmethod(X, y, max_iter=nit, tol=0.0, storage=storage, t=t_init, **kw) if smooth
             else R.ADMMmethod(X, y, max_iter=nit, tol=0.0, storage=storage, **kw))
        worst = 0.0
        trace = []
        for i in range(nit):
            st = s._s.step(want_objective=True)
            worst = max(worst, abs(st.primal - ref.primal[i]) / max(1.0, ref.primal[i]))
            trace.append((i, st.primal, ref.primal[i], st.dual, ref.dual[i], st.objective, ref.objective[i + 1], st.zband, st.inner_iters))
        if smooth:
            s._s.finalize_smooth()
        w = s._s.get_state()["w"]
        werr = np.max(np.abs(w - ref.w)) / max(1.0, np.max(np.abs(ref.w)))
        tol = 1e-8 if loss == "binary_cross_entropy" else 1e-6
        if storage == "f32" and raw:
            # D rounded to fp32 is a perturbed problem; the non-convex families amplify the perturbation
            tol = 5e-2 if fam in ("aorr", "aorr_dc") else (5e-3 if fam == "ehrm" else 2e-4)
        flag = "" if (worst <= tol and werr <= tol) else "  <<<<<< MISMATCH"
        if flag: bad += 1
        if flag and os.environ.get("STRESS_VERBOSE") == "1":
            for row in trace:
                print("      it %2d primal %.12e / %.12e  dual %.6e / %.6e  obj %.12e / %.12e  zband %d inner %d" % row, flush=True)
            print("      t_init", t_init, "w (gpu) max", float(np.max(np.abs(w))), "w (oracle) max", float(np.max(np.abs(ref.w))), flush=True)
        print(f"{trial:3d} {fam:13s}{'*' if smooth else ' '}{loss[:5]} {storage} n={n:5d} d={d:4d} {regk}={kw[regk]:.1e} primal_err={worst:.1e} w_err={werr:.1e}{flag}", flush=True)
    except Exception as e:
        print(trial, fam, loss, n, d, "EXC", repr(e)[:200], flush=True); bad += 1
print("bad =", bad, "time", round(time.time() - t0, 1))
