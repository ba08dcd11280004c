"""Helper for tests/test_gpu_solver.py::test_single_sweep_paths: runs an erm solve on the
GPU under the environment it is started with and prints the iterates as JSON."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import admm_for_rank_based_loss_amd as rbl  # noqa: E402
from oracle import problems  # noqa: E402

loss, reg_kind = sys.argv[1], sys.argv[2]
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
cols = int(sys.argv[4]) if len(sys.argv) > 4 else 160
storage = sys.argv[5] if len(sys.argv) > 5 else "f64"
X, y = problems.make_problem(rows, cols, seed=21)
kw = {reg_kind: 0.01}
s = rbl.ADMMmethod(X, y, "erm", loss, storage=storage, tol=0.0, max_iter=40, **kw)
hist, fused, mis = [], 0, 0
for i in range(40):
    st = s._s.step(True)
    hist.append([st.primal, st.dual, st.rho, st.objective])
    fused += st.fused
    mis += st.mispredicted
state = s._s.get_state()
print(json.dumps(dict(hist=hist, w=state["w"].tolist(), z=state["z"].tolist(), lam=state["lam"].tolist(),
                      fused=fused, mispredicted=mis)))
