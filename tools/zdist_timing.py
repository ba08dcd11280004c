#!/usr/bin/env python3
"""What one distributed z-step costs on ONE rank of an 8-GPU run (750 000 of 6 000 000 rows), through RCCL with a
1-rank group: the sort-based protocol (sample sort, all-to-all, chunk PAV, merge tree over ranks, return trip) against
the sort-free one for banded rank weights (histograms / sums all-reduced, undecided elements gathered).  Every
collective is issued (always_allreduce) but degenerates to a copy: this measures the local kernels, the launch path
of the collectives and the host waits, not the wire.   python tools/zdist_timing.py [rows] [cols]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29741")
import numpy as np
import torch
import torch.distributed as dist
import admm_for_rank_based_loss_amd as rbl
from admm_for_rank_based_loss_amd.dist import ShardedADMM, GpuEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 750_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
for wf, loss, args, reg in (("superquantile", "binary_cross_entropy", [0.5], 0.01), ("aorr", "hinge", [0.2, 0.8], 1e-4)):
    s = rbl.Solver(n, d, wf, loss, reg=reg, wstep=2, args=args, n_total=n, row_offset=0, tol=0.0, storage="f32")
    e = GpuEngine(s, 0)
    drv = ShardedADMM(e)
    drv.always_allreduce = True
    drv.setup_synthetic(seed=3)
    drv.setup_gram()
    for _ in range(8):
        drv.step(False)
    res = {}
    for name, fn in (("sort-based", drv._z_distributed), ("sort-free", drv._z_banded)):
        ts = []
        for rep in range(12):
            e.phase_m()
            torch.cuda.synchronize()
            drv.n_coll = drv.n_sync = 0
            t0 = time.perf_counter()
            ok = fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
            assert ok is not False, "not certified"
        res[name] = (1e3 * float(np.median(ts[2:])), drv.n_coll, drv.n_sync)
    print(f"{wf}/{loss[:5]} {n}x{d}: " + "; ".join(f"{k} {v[0]:.3f} ms ({v[1]} collectives, {v[2]} host waits)" for k, v in res.items()), flush=True)
dist.destroy_process_group()
