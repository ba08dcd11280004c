#!/usr/bin/env python3
"""Inner iterations of the smoothed-l1 w-step over 45 sADMM iterations of a small problem (the oracle's generator), for
comparing RBL_NCG_ACTIVE=0 (nonlinear CG alone) with the default (linear first phase, csrc/wstep.hip):
    python tests/diag_sadmm_inner.py ROWS COLS T0"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import admm_for_rank_based_loss_amd as R
from oracle import problems
n, d, t0 = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
X, y = problems.make_problem(n, d, seed=31 + d)
kw = dict(weight_function="erm", loss="binary_cross_entropy", l1_reg=0.01)
s = R.smoothADMMmethod(X, y, max_iter=45, tol=0.0, storage="f64", t=t0, **kw)
tot = 0; inn = []
for i in range(45):
    st = s._s.step(want_objective=True)
    tot += st.inner_iters; inn.append(st.inner_iters)
print("n", n, "d", d, "t0", t0, "RBL_NCG_ACTIVE", os.environ.get("RBL_NCG_ACTIVE"), "total inner", tot, inn)
