"""The one-process-per-GPU path on real device memory: two ranks (both on cuda:0 - the test
box has one GPU - with the gloo backend, because RCCL refuses two ranks on one device) run
the sharded driver on their row shards of an on-device generated problem; the iterates must
equal the single-handle run (shard-count invariance, including the counter-based
generator and the sharded Gram / column statistics)."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

CFGS = [
    dict(n=40000, d=48, wf="erm", loss="binary_cross_entropy", reg=0.01, wstep=1, iters=8),
    dict(n=30001, d=33, wf="superquantile", args=[0.5], loss="binary_cross_entropy", reg=0.01, wstep=2, iters=6),
    dict(n=20000, d=21, wf="ehrm", B=-5.0, loss="binary_cross_entropy", reg=0.01, wstep=2, iters=5),
]


def _run(rank, world, port, cfg, out):
    sys.path.insert(0, ROOT)
    if world > 1:
        os.environ.update(cfg.get("env", {}))     # only the sharded run: the single-handle run stays the plain path
    import torch
    import torch.distributed as dist
    import admm_for_rank_based_loss_amd as rbl
    from admm_for_rank_based_loss_amd.dist import ShardedADMM, GpuEngine, shard_rows
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, cnt, _ = shard_rows(cfg["n"], world, rank)
        s = rbl.Solver(cnt, cfg["d"], cfg["wf"], cfg["loss"], reg=cfg["reg"], wstep=cfg["wstep"], B=cfg.get("B"),
                       args=cfg.get("args"), n_total=cfg["n"], row_offset=lo, tol=0.0, storage="f64")
        drv = ShardedADMM(GpuEngine(s, 0), dist_z=cfg.get("dist_z", True))
        drv.setup_synthetic(seed=11)
        drv.setup_gram()
        hist = []
        for _ in range(cfg["iters"]):
            st = drv.step(True)
            hist.append((st.primal, st.dual, st.rho, st.objective))
        state = s.get_state()
        np.savez(out % rank, w=state["w"], z=state["z"], hist=np.array(hist), lo=lo)
    finally:
        if world > 1:
            dist.destroy_process_group()


@pytest.mark.parametrize("cfg", CFGS, ids=[c["wf"] for c in CFGS])
def test_two_ranks_one_gpu_match_single_handle(cfg, tmp_path):
    _check(cfg, 2, tmp_path)


MORE = [
    (4, dict(n=30001, d=33, wf="superquantile", args=[0.5], loss="binary_cross_entropy", reg=0.01, wstep=2, iters=6)),
    (3, dict(n=25000, d=21, wf="aorr", args=[0.2, 0.8], loss="hinge", reg=1e-4, wstep=2, iters=6)),
    (4, dict(n=20000, d=21, wf="ehrm", B=-5.0, loss="binary_cross_entropy", reg=0.01, wstep=2, iters=5)),
    (2, dict(n=30001, d=33, wf="extremile", args=[2.0], loss="binary_cross_entropy", reg=0.01, wstep=2, iters=5,
             dist_z=False)),
]


MORE += [
    # every 3rd rho prediction corrupted on both ranks: the verification + two-sweep redo (and its two
    # partial all-reduces) inside the sharded driver
    (2, dict(n=40000, d=48, wf="erm", loss="binary_cross_entropy", reg=0.01, wstep=1, iters=9,
             env={"RBL_DEBUG_MISPREDICT_EVERY": "3"})),
    (4, dict(n=5, d=3, wf="superquantile", args=[0.5], loss="binary_cross_entropy", reg=0.01, wstep=2, iters=4)),   # rank 3 owns no row
    (3, dict(n=4099, d=5, wf="esrm", args=[1.0], loss="hinge", reg=0.01, wstep=1, iters=6)),
]


@pytest.mark.parametrize("world,cfg", MORE, ids=["superq_w4", "aorr_hinge_w3", "ehrm_w4", "extremile_replicated_z",
                                                  "erm_mispredictions_w2", "tiny_with_an_empty_rank", "esrm_hinge_l1_w3"])
def test_distributed_z_step_on_device(world, cfg, tmp_path):
    """rank-weighted problems with the sorted order partitioned over 3 / 4 ranks on the device path
    (rbl_zd_*: sample sort, chunk PAV, merge tree over ranks), and the replicated all-gather form."""
    _check(cfg, world, tmp_path)


def _check(cfg, world, tmp_path):
    import torch.multiprocessing as mp
    out1 = str(tmp_path / "w1_r%d.npz")
    out2 = str(tmp_path / "w2_r%d.npz")
    port = 29600 + (os.getpid() + 13 * world) % 1000
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_run, args=(0, 1, port, cfg, out1))
    p.start(); p.join(300)
    assert p.exitcode == 0
    mp.spawn(_run, args=(world, port, cfg, out2), nprocs=world, join=True)
    one = np.load(out1 % 0)
    rs = [np.load(out2 % r) for r in range(world)]
    r0, r1 = rs[0], rs[1]
    # replicated quantities agree between the ranks bit-for-bit
    for r in rs[1:]:
        assert np.array_equal(r0["w"], r["w"]) and np.array_equal(r0["hist"], r["hist"])
    z2 = np.concatenate([r["z"] for r in rs])
    # sharding changes only the order of the fp64 partial sums (slabs per rank): ~1e-12
    assert np.max(np.abs(r0["w"] - one["w"])) <= 1e-9 * max(1.0, np.max(np.abs(one["w"])))
    assert np.max(np.abs(z2 - one["z"])) <= 1e-8 * max(1.0, np.max(np.abs(one["z"])))
    assert np.allclose(r0["hist"], one["hist"], rtol=1e-8, atol=1e-12)


def test_rccl_accepts_the_library_views_one_rank():
    """The box has one GPU and RCCL refuses two ranks per device, so the nccl backend can only be
    exercised with world_size 1: a sharded step, every collective of the driver on the library's
    zero-copy views (float64 / int64 / int32) and the whole distributed z-step protocol run through
    RCCL and give the single-handle z bit for bit (tools/rccl_view_smoke.py)."""
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_view_smoke.py")], capture_output=True, text=True,
                         timeout=600, env=dict(os.environ, MASTER_PORT=str(29700 + os.getpid() % 200)))
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    assert out.stdout.strip().splitlines()[-1] == "OK"
