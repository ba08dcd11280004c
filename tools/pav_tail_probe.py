#!/usr/bin/env python3
"""Where does the bottom PAV kernel spend its time along the rank axis?  Takes the m of an EHRM run at C4shard's size after
30 iterations, sorts it, and runs the kernel-level PAV entry (rbl_k_pav_ehrm) on slices of the ranks - under
`rocprofv3 --kernel-trace` the durations of k_pav_bottom per slice tell which part of the order holds the slow tiles.
    rocprofv3 --kernel-trace --output-format csv -d OUT -o x -- python3 tools/pav_tail_probe.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import admm_for_rank_based_loss_amd as rbl
from admm_for_rank_based_loss_amd import _lib
from admm_for_rank_based_loss_amd.dist import _DevArray

n = 6_250_000
s = rbl.Solver(n, 1000, "ehrm", "binary_cross_entropy", reg=0.01, wstep=2, B=-5.0, storage="f32", tol=0.0)
s.generate_synthetic(17)
s.gram()
for _ in range(30):
    st = s.step(False)
s.phase_m()
torch.cuda.synchronize()
p, c = s.buffer(_lib.BUF_M)
m = np.sort(torch.as_tensor(_DevArray(p, c), device="cuda:0").cpu().numpy())
rho = st.rho_next
sa, sb = s.sigma()
s.close()
print("rho", rho, "branch", st.ehrm_branch, "m range", m[0], m[-1], flush=True)
if "--dump" in sys.argv:      # input of tools/pav_lab.hip: the 16 highest tiles and 16 from the middle (branch b weights)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    for tag, lo, hi in (("top", n - 16 * 2048, n), ("mid", n // 2, n // 2 + 16 * 2048)):
        m[lo:hi].tofile(os.path.join(out, "pav_lab_ms_%s.bin" % tag))
        sa[lo:hi].tofile(os.path.join(out, "pav_lab_sa_%s.bin" % tag))
        sb[lo:hi].tofile(os.path.join(out, "pav_lab_sb_%s.bin" % tag))
    print("dumped; rho =", repr(rho), flush=True)
    sys.exit(0)
T = 2048
for name, lo, hi in (("all", 0, n), ("first half", 0, n // 2), ("second half", n // 2, n), ("last 10%", n - n // 10, n),
                     ("last 1%", n - n // 100, n), ("last 16 tiles", n - 16 * T, n), ("last tile", n - T, n),
                     ("first 16 tiles", 0, 16 * T), ("middle 16 tiles", n // 2, n // 2 + 16 * T)):
    out, br = _lib.k_pav_ehrm(sa[lo:hi], sb[lo:hi], -5.0, rho, m[lo:hi], branch=st.ehrm_branch)
    print("slice", name, hi - lo, "branch", br, flush=True)
