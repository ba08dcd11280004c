#!/usr/bin/env python3
"""One-rank RCCL smoke test of the zero-copy views the multi-GPU driver hands to torch.distributed:
all_reduce / all_gather_into_tensor / all_to_all_single on librbl-owned device buffers of every
dtype the driver uses (float64 exchange buffer, int64 keys, int32 row ids), on the stream the
library runs on.  A 1-GPU box cannot run 2 RCCL ranks; this checks that RCCL accepts the foreign
allocations and dtypes, and that a sharded step runs end to end on the nccl backend.
    python tools/rccl_view_smoke.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29731")

import torch
import torch.distributed as dist
import admm_for_rank_based_loss_amd as rbl
from admm_for_rank_based_loss_amd import _lib
from admm_for_rank_based_loss_amd.dist import ShardedADMM, GpuEngine

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
s = rbl.Solver(20000, 40, "superquantile", "binary_cross_entropy", reg=0.01, wstep=2, args=[0.5], n_total=20000,
               row_offset=0, tol=0.0, storage="f32")
e = GpuEngine(s, 0)
drv = ShardedADMM(e)
drv.setup_synthetic(seed=3)
drv.setup_gram()
st = drv.step(True)
print("step ok: primal %.3e objective %.6f" % (st.primal, st.objective))
x = e.buf("q")
before = x.clone()
dist.all_reduce(x)
assert torch.equal(x, before)
g = torch.empty_like(x)
dist.all_gather_into_tensor(g, x)
assert torch.equal(g, x)
e.phase_m()
e.zd_sort_local(16)
sk, si = e.zd_send_buffers()
rk, ri = e.zd_recv_buffers(sk.numel())
dist.all_to_all_single(rk, sk, [sk.numel()], [sk.numel()])
dist.all_to_all_single(ri, si, [si.numel()], [si.numel()])
torch.cuda.synchronize()
assert torch.equal(rk, sk) and torch.equal(ri, si)
small = e._small(0, 16)
dist.all_reduce(small)
print("views ok: float64 %d, int64 %d, int32 %d elements through RCCL" % (x.numel(), sk.numel(), si.numel()))
# the distributed z-step protocol on one rank (every collective degenerates to a copy, but goes
# through RCCL): same z as the single-handle z-step
import numpy as np
e.phase_m()
e.phase_z(None)
z_ref = s.get_state()["z"].copy()
s.set_state(z=np.zeros_like(z_ref))
e.phase_m()
drv._z_distributed()
z_got = s.get_state()["z"]
assert np.array_equal(z_ref, z_got), float(np.max(np.abs(z_ref - z_got)))
print("distributed z-step through RCCL (1 rank): identical z")
# the sort-free distributed z-step (rbl_zbd_*): its int32 histogram view and its float64 views through RCCL, same z as
# the sort-based one to rounding (the pooled block is summed in another order)
os.environ["RBL_ZBAND_MIN_N"] = "16"
s2 = rbl.Solver(20000, 40, "superquantile", "binary_cross_entropy", reg=0.01, wstep=2, args=[0.5], n_total=20000,
                row_offset=0, tol=0.0, storage="f32")
e2 = GpuEngine(s2, 0)
drv2 = ShardedADMM(e2)
drv2.always_allreduce = True          # a 1-rank group normally skips its identity all-reduces: issue them
drv2.setup_synthetic(seed=3)
drv2.setup_gram()
for _ in range(3):
    drv2.step(False)
e2.phase_m()
assert drv2._z_banded(), "the sort-free z-step was not certified"
z_fast = s2.get_state()["z"].copy()
s2.set_state(z=np.zeros_like(z_fast))
e2.phase_m()
drv2._z_distributed()
z_sort = s2.get_state()["z"]
assert np.max(np.abs(z_fast - z_sort)) <= 1e-12 * max(1.0, np.max(np.abs(z_sort))), float(np.max(np.abs(z_fast - z_sort)))
print("sort-free distributed z-step through RCCL (1 rank): int32 / float64 views accepted, same z")
dist.destroy_process_group()
print("OK")
