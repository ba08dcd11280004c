"""Randomized parity run of the sharded driver on the device path (TEST INFRASTRUCTURE, not
collected by pytest): random family / shape / number of ranks (2..8, ranks as threads of this
process with hub collectives, tests/test_gpu_dist.py) against the single-handle run.
    python tests/stress_dist.py SEED TRIALS
Round 1: 220 trials (seeds 1-4), worst relative deviation 2e-12, ranks bit-identical in all.  Round 2 (seed 7, 60
trials; the 28 superquantile / aorr draws of 4096 rows or more take the sort-free distributed z-step rbl_zbd_*): clean.
Round 3 (seeds 9 and 11, 40 + 150 trials, 2-8 ranks; early-out of the root passes, staged collectives under gloo): worst
relative deviation of w 3e-12, ranks bit-identical in all."""
import os
import sys
import threading

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import torch  # noqa: E402,F401
import admm_for_rank_based_loss_amd as rbl  # noqa: E402
from admm_for_rank_based_loss_amd import dist as _d  # noqa: E402,F401
import test_gpu_dist as T  # noqa: E402

rbl._lib.load()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
fams = [("erm", None), ("superquantile", [0.5]), ("superquantile", [0.95]), ("extremile", [2.0]), ("esrm", [1.0]),
        ("aorr", [0.2, 0.8]), ("ehrm", None)]


def run(world, cfg):
    hub, out, errs = T._Hub(world), [None] * world, []
    ts = [threading.Thread(target=T._thread_rank, args=(r, world, cfg, hub, out, errs)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(600)
    if errs:
        raise RuntimeError(errs)
    return out


bad = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 20):
    fam, args = fams[int(rng.integers(len(fams)))]
    loss = "binary_cross_entropy" if fam == "ehrm" else ("binary_cross_entropy", "hinge")[int(rng.integers(2))]
    n = int(rng.integers(50, 70000))
    d = int(rng.integers(3, 50))
    if rng.random() < 0.4:
        d = int(rng.integers(66, 400))      # single-sweep erm / v-only kernels (fp64 storage: more than 32 packets)
        n = int(rng.integers(50, 20000))
    world = int(rng.choice([2, 3, 5, 7, 8]))
    cfg = dict(n=n, d=d, wf=fam, loss=loss, args=args, reg=float(10.0 ** rng.uniform(-4, -1)),
               wstep=1 if (fam != "ehrm" and rng.random() < 0.4) else 2, iters=5)
    if fam == "ehrm":
        cfg["B"] = -5.0
    one = run(1, cfg)[0]
    out = run(world, cfg)
    same = all(np.array_equal(out[0]["w"], r["w"]) and np.array_equal(out[0]["hist"], r["hist"]) for r in out[1:])
    zz = np.concatenate([r["z"] for r in out])
    ew = np.max(np.abs(out[0]["w"] - one["w"])) / max(1.0, np.max(np.abs(one["w"])))
    ez = np.max(np.abs(zz - one["z"])) / max(1.0, np.max(np.abs(one["z"])))
    eh = np.max(np.abs(out[0]["hist"] - one["hist"]) / np.maximum(1e-3, np.abs(one["hist"])))
    ok = same and ew <= 1e-9 and ez <= 1e-8 and eh <= 1e-7
    bad += not ok
    print(f"{trial:3d} world={world} {fam:13s} {loss[:5]} n={n:6d} d={d:2d} wstep={cfg['wstep']} ranks_identical={same} "
          f"w_err={ew:.1e} z_err={ez:.1e} hist_err={eh:.1e}{'' if ok else '   <<<<<< MISMATCH'}", flush=True)
print("bad =", bad)
