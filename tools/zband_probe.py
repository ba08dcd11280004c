"""Diagnostic: how the pooled block of a banded rank-weight z-step evolves over the ADMM iterations
(block value, element count, movement per iteration).  python tools/zband_probe.py C2sq 40"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import admm_for_rank_based_loss_amd as rbl

name, iters = sys.argv[1], int(sys.argv[2])
rows = int(sys.argv[3]) if len(sys.argv) > 3 else None
cfg = dict(bench.CONFIGS[name])
if rows:
    cfg["rows"] = rows
s = rbl.Solver(cfg["rows"], cfg["cols"], cfg["weight_function"], cfg["loss"], reg=cfg["reg"], wstep=cfg["wstep"], B=cfg["B"],
               args=cfg["args"], tol=0.0, storage="f32")
s.generate_synthetic(seed=17)
prev = None
for it in range(iters):
    st = s.step(False)
    z = s.get_state(want_lam=False)["z"]
    vals, cnt = np.unique(z, return_counts=True)
    order = np.argsort(-cnt)[:3]
    top = [(float(vals[k]), int(cnt[k])) for k in order if cnt[k] > 1]
    x = top[0][0] if top else float("nan")
    print(f"it {it:3d} rho {st.rho:.3e} primal {st.primal:.3e} blocks>1: {int((cnt > 1).sum()):5d} top {top} "
          f"dx {x - prev if prev is not None else float('nan'):+.3e} z range [{z.min():.4f}, {z.max():.4f}] "
          f"std {z.std():.4e} merges {st.pav_merges}", flush=True)
    prev = x
