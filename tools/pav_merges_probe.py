#!/usr/bin/env python3
"""How much pooling does the merge-tree PAV see per iteration?  rbl_stats.pav_merges (seam merges + sequential-stage
merges) and the z-step's device time for a rank-weighted configuration at full size.
    python tools/pav_merges_probe.py [C4shard|C2sq] [iterations]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (CONFIGS)
import admm_for_rank_based_loss_amd as rbl  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C4shard"
nit = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cfg = bench.CONFIGS[name]
if name == "C2sq":
    os.environ["RBL_NO_ZBAND"] = "1"
s = rbl.Solver(cfg["rows"], cfg["cols"], cfg["weight_function"], cfg["loss"], reg=cfg["reg"], wstep=cfg["wstep"], B=cfg["B"],
               args=cfg["args"], storage="f32", tol=0.0)
s.generate_synthetic(17)
s.gram()
s.profile_kernels(2)
for i in range(nit):
    st = s.step(False)
    print("iter %2d  rho %.3e  merges %9d  branch %2d  sort passes %d  ms_z %.3f  ms_total %.3f" % (
        i, st.rho, st.pav_merges, st.ehrm_branch, st.sort_passes, st.ms_z, st.ms_total))
s.close()
