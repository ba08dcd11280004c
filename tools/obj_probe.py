#!/usr/bin/env python3
"""What does objective logging cost per iteration at C2?  40 iterations with and without want_objective on one handle
(round 3: 3.573 against 3.622 ms - k_loss_sum 35 us, k_sum_partials 5 us, the 8 B/row of v the pass then stores).
    python tools/obj_probe.py            (under rocprofv3 --kernel-trace for the kernel list)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import admm_for_rank_based_loss_amd as rbl  # noqa: E402

s = rbl.Solver(6_000_000, 1000, "erm", "binary_cross_entropy", reg=0.01, wstep=1, storage="f32", tol=0.0)
s.generate_synthetic(17)
s.gram()
for w in (False, True):
    for _ in range(5):
        s.step(w)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(40):
        st = s.step(w)
    torch.cuda.synchronize()
    print("want_objective", w, "ms/iter %.4f" % ((time.perf_counter() - t) / 40 * 1e3), "host_syncs", st.host_syncs)
s.close()
