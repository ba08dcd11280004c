"""Long runs of the sort-free z-step against the sort + PAV path on generated problems of 1-2 M rows (hundreds of
iterations, every logged quantity compared): python tools/zband_soak.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import admm_for_rank_based_loss_amd as rbl

for (n, d, wf, loss, args, reg, wstep, nit) in [(2_000_000, 64, "superquantile", "binary_cross_entropy", [0.5], 0.01, 2, 300),
                                                (1_500_000, 48, "aorr", "hinge", [0.2, 0.8], 1e-4, 2, 300),
                                                (1_000_000, 32, "superquantile", "hinge", [0.8], 0.01, 1, 200),
                                                (1_200_000, 40, "aorr", "binary_cross_entropy", [0.1, 0.6], 1e-4, 2, 250)]:
    runs = []
    for off in ("1", "0"):
        os.environ["RBL_NO_ZBAND"] = off
        s = rbl.Solver(n, d, wf, loss, reg=reg, wstep=wstep, args=args, tol=0.0, storage="f32")
        s.generate_synthetic(seed=3)
        hist, modes = [], []
        t0 = time.time()
        for _ in range(nit):
            st = s.step(True)
            hist.append((st.primal, st.dual, st.objective))
            modes.append(st.zband)
        runs.append((np.array(hist), s.get_state(), modes, time.time() - t0))
    (ha, a, ma, ta), (hb, b, mb, tb) = runs
    eh = float(np.max(np.abs(ha - hb) / np.maximum(1.0, np.abs(ha))))
    ew = float(np.max(np.abs(a["w"] - b["w"])) / max(1.0, np.max(np.abs(a["w"]))))
    ez = float(np.max(np.abs(a["z"] - b["z"])) / max(1.0, np.max(np.abs(a["z"]))))
    print(f"{wf}{args} {loss[:5]} n={n} d={d} its={nit}: sort {ta:.2f} s, sort-free {tb:.2f} s; modes 1:{mb.count(1)} 2:{mb.count(2)} 0:{mb.count(0)}; "
          f"max rel diff hist {eh:.1e} w {ew:.1e} z {ez:.1e}", flush=True)
    tol = 1e-8 if loss == "binary_cross_entropy" else 1e-6
    assert eh <= tol and ew <= tol and ez <= 10 * tol
print("OK")
