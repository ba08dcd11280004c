"""Which z-step path every iteration took (stats.zband) on the banded test problems and, optionally, bench configs.
python tools/zband_modes.py [C2sq|C3] [iterations]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import admm_for_rank_based_loss_amd as rbl

if len(sys.argv) > 1:
    import bench
    cfg = bench.CONFIGS[sys.argv[1]]
    cases = [(sys.argv[1], cfg["rows"], cfg["cols"], dict(weight_function=cfg["weight_function"], loss=cfg["loss"], args=cfg["args"],
                                                         reg=cfg["reg"], wstep=cfg["wstep"]), "f32")]
    nit = int(sys.argv[2]) if len(sys.argv) > 2 else 60
else:
    import test_gpu_zband as t
    cases = [c + ("f64",) for c in t.DEVICE_CASES]
    nit = 60
for name, n, d, kw, storage in cases:
    s = rbl.Solver(n, d, kw["weight_function"], kw["loss"], reg=kw["reg"], wstep=kw["wstep"], args=kw["args"], tol=0.0, storage=storage)
    s.generate_synthetic(seed=5)
    modes = "".join(str(s.step(False).zband) for _ in range(nit))
    print(f"{name:22s} {modes}", flush=True)
