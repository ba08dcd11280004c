#!/usr/bin/env python3
"""Which switch changes the bits of the first iterations?  (diagnostic for the redo bit-identity test)"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import admm_for_rank_based_loss_amd as R   # noqa: E402
from oracle import problems                 # noqa: E402

X, y = problems.make_problem(3000, 16, seed=5)


def run(env, wstep=1, nit=4):
    for k in ("RBL_NO_ZBAND", "RBL_ZBAND_MIN_N", "RBL_SORT32", "RBL_PAV_UPPER_PERSIST", "RBL_PAV_NO_SEQ"):
        os.environ.pop(k, None)
    os.environ.update(env)
    s = R.Solver(3000, 16, "aorr", "binary_cross_entropy", reg=1e-4, wstep=wstep, args=[0.45, 0.55], tol=0.0, storage="f64")
    s.set_data(X, y)
    out = []
    for _ in range(nit):
        st = s.step(True)
        state = s.get_state()
        h = hashlib.sha1(state["z"].tobytes()).hexdigest()[:8] + "/" + hashlib.sha1(state["w"].tobytes()).hexdigest()[:8]
        out.append("%d:%s:%.17g" % (st.zband, h, st.primal))
    s.close()
    return out


for name, env in [("nozband s32", {"RBL_NO_ZBAND": "1"}), ("nozband s32 again", {"RBL_NO_ZBAND": "1"}),
                  ("nozband s64", {"RBL_NO_ZBAND": "1", "RBL_SORT32": "0"}),
                  ("nozband s64 oldpav", {"RBL_NO_ZBAND": "1", "RBL_SORT32": "0", "RBL_PAV_UPPER_PERSIST": "0", "RBL_PAV_NO_SEQ": "1"}),
                  ("zband s32", {"RBL_ZBAND_MIN_N": "16"}), ("zband s64", {"RBL_ZBAND_MIN_N": "16", "RBL_SORT32": "0"})]:
    for wstep in (1, 2):
        print("%-20s wstep %d  %s" % (name, wstep, "  ".join(run(env, wstep))))
