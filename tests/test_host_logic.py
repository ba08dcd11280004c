"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol that
include/rbl.h declares (no compute without a GPU), compute calls fail loudly instead of
falling back, argument validation raises the reference's messages, the row sharding is
consistent, and the multi-GPU driver (one process per GPU) gives shard-count-invariant
iterates - exercised with world_size 2 over gloo and a NumPy engine built on the oracle."""
import os
import re
import ctypes as C
import sys

import numpy as np
import pytest

from conftest import ROOT


def _pkg():
    import admm_for_rank_based_loss_amd as rbl
    return rbl


def test_library_exports_every_declared_symbol():
    rbl = _pkg()
    header = open(os.path.join(ROOT, "include", "rbl.h")).read()
    declared = set(re.findall(r"\b(rbl_[A-Za-z0-9_]+)\s*\(", header))
    declared -= {"rbl_config", "rbl_stats", "rbl_solver"}
    assert len(declared) >= 40
    lib = rbl._lib.load()
    missing = [name for name in sorted(declared) if not hasattr(lib, name)]
    assert not missing, missing
    # the Python binding covers the same set
    assert declared == set(rbl._lib.SIGNATURES), declared ^ set(rbl._lib.SIGNATURES)
    assert lib.rbl_version() == 106


def test_structure_layouts_agree_with_the_library():
    """rbl_step writes sizeof(rbl_stats) bytes into the caller's buffer: the ctypes mirrors - the package's and the
    stub INTEGRATION.md shows a maintainer - must have the library's sizes (rbl_sizeof) and the header's field names."""
    rbl = _pkg()
    lib = rbl._lib.load()
    assert lib.rbl_sizeof(0) == C.sizeof(rbl._lib.RblConfig) and lib.rbl_sizeof(1) == C.sizeof(rbl._lib.RblStats)
    assert lib.rbl_sizeof(7) == -1
    header = open(os.path.join(ROOT, "include", "rbl.h")).read()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for struct, cls in (("rbl_config", rbl._lib.RblConfig), ("rbl_stats", rbl._lib.RblStats)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), header, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = [n.strip().split("[")[0] for decl in re.findall(r"(?:int64_t|int32_t|double|float)\s+([^;]+);", body)
                 for n in decl.split(",")]
        assert names == [f[0] for f in cls._fields_], (struct, names)
        stub = re.search(r"class %s\(C\.Structure\):.*?_fields_ = \[(.*?)\]\n" % struct, doc, re.S).group(1)
        assert re.findall(r'\("([a-z_A-Z0-9]+)",', stub) == names, struct


def test_no_cpu_fallback_without_device():
    rbl = _pkg()
    if rbl._lib.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(rbl._lib.RblError, match="no HIP device"):
        rbl._lib.k_prox("hinge", np.ones(3), 1.0, np.zeros(3))
    X = np.random.default_rng(0).standard_normal((20, 3))
    y = np.sign(X[:, :1])
    with pytest.raises(rbl._lib.RblError, match="no HIP device"):
        rbl.ADMMmethod(X, y, "erm", "hinge", l2_reg=0.1)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "admm-for-rank-based-loss_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), (dirpath, fn)


def test_argument_validation_messages():
    rbl = _pkg()
    X = np.zeros((10, 2))
    y = np.ones((10, 1))
    cases = [
        (dict(weight_function="nope", l2_reg=0.1, args=[1]), "Unrecognized framework 'nope'"),
        (dict(weight_function="superquantile", l2_reg=0.1), "args for framework is None!"),
        (dict(weight_function="aorr_dc", l2_reg=0.1, args=[2, 5]), "need args"),
        (dict(loss="square", l2_reg=0.1), "Unrecognized loss 'square'"),
        (dict(weight_function="ehrm", loss="hinge", l2_reg=0.1, B=-5), "erhm only can be with the binary_cross_entropy."),
        (dict(weight_function="erm", l2_reg=0.1, B=-5), r"Unrecognized weight_function 'erm'! Options: \['ehrm'\]"),
        (dict(), "l1_reg or l2_reg"),
    ]
    for kw, msg in cases:
        with pytest.raises(ValueError, match=msg):
            rbl.ADMMmethod(X, y, **kw)
    with pytest.raises(ValueError, match="Unrecognized framework"):
        rbl.get_weights("nope", [1])


def test_shard_rows():
    sys.path.insert(0, os.path.join(ROOT, "admm-for-rank-based-loss_amd"))
    from admm_for_rank_based_loss_amd.dist import shard_rows
    for n, world in [(10, 1), (10, 3), (6_000_000, 8), (7, 8), (1000, 7)]:
        got = [shard_rows(n, world, r) for r in range(world)]
        assert sum(g[1] for g in got) == n
        nmax = got[0][2]
        pos = 0
        for r, (lo, cnt, nm) in enumerate(got):
            assert lo == min(r * nmax, n) and nm == nmax and 0 <= cnt <= nmax
            assert lo == pos or cnt == 0
            pos += cnt


def _worker(rank, world, port, cfg, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from _numpy_engine import NumpyEngine
        from admm_for_rank_based_loss_amd.dist import ShardedADMM, shard_rows
        from oracle import problems
        X, y = problems.make_problem(cfg["n"], cfg["d"], cfg["seed"])
        lo, cnt, _ = shard_rows(cfg["n"], world, rank)
        if cfg.get("fused"):
            from _numpy_engine import FusedNumpyEngine
            eng = FusedNumpyEngine(X[lo:lo + cnt], y[lo:lo + cnt], cfg["n"], lo, cfg["wf"], cfg["loss"], cfg["reg"],
                                   cfg["l1"], mispredict_every=cfg.get("mispredict_every", 0))
        else:
            eng = NumpyEngine(X[lo:lo + cnt], y[lo:lo + cnt], cfg["n"], lo, cfg["wf"], cfg["loss"], cfg["reg"], cfg["l1"],
                              B=cfg.get("B"), args=cfg.get("args"))
        drv = ShardedADMM(eng, dist_z=cfg.get("dist_z", True))
        drv.banded_z = cfg.get("banded", False)      # the sort-free z-step for banded weights only where a case asks for it
        drv.setup_gram()
        hist, zcoll = [], []
        for _ in range(cfg["iters"]):
            nb = getattr(eng, "n_banded", 0)
            st = drv.step(want_objective=True)
            hist.append((st.primal, st.dual, st.rho, st.objective))
            zcoll.append((st.z_collectives, getattr(eng, "n_banded", 0) - nb, drv.zb_clusters))
        if rank == 0:
            np.savez(out, w=eng.w, hist=np.array(hist), zcoll=np.array(zcoll),
                     fused=getattr(eng, "n_fused", 0), mispred=getattr(eng, "n_mispred", 0), banded=getattr(eng, "n_banded", 0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("cfg", [
    dict(n=600, d=12, seed=3, wf="erm", loss="binary_cross_entropy", reg=0.01, l1=True, iters=12),
    dict(n=501, d=9, seed=4, wf="superquantile", args=[0.5], loss="binary_cross_entropy", reg=0.01, l1=False, iters=12),
    dict(n=400, d=7, seed=5, wf="aorr", args=[0.2, 0.8], loss="hinge", reg=1e-4, l1=False, iters=10),
    dict(n=300, d=6, seed=6, wf="ehrm", B=-5, loss="binary_cross_entropy", reg=0.01, l1=False, iters=10),
], ids=["erm_l1", "superq_l2_uneven", "aorr_hinge", "ehrm"])
def test_sharded_driver_world2_gloo_matches_single_process(cfg, tmp_path):
    _check_sharded(cfg, 2, tmp_path)


@pytest.mark.parametrize("world,cfg", [
    (2, dict(n=600, d=12, seed=3, wf="erm", loss="binary_cross_entropy", reg=0.01, l1=True, iters=14, fused=True)),
    (3, dict(n=701, d=10, seed=13, wf="erm", loss="hinge", reg=0.01, l1=False, iters=14, fused=True)),
    (2, dict(n=600, d=12, seed=3, wf="erm", loss="binary_cross_entropy", reg=0.01, l1=True, iters=14, fused=True,
             mispredict_every=3)),
], ids=["bce_l1_world2", "hinge_l2_world3", "mispredict_world2"])
def test_single_collective_erm_iteration_gloo(world, cfg, tmp_path):
    """dist.py's `pending_reduce` branches (mask 1 / 2 slices, mask 3 = ONE all-reduce of the whole exchange
    buffer after the pass, rho predicted from global sums) on the NumPy restatement of librbl's
    single-sweep protocol (tests/_numpy_engine.py: FusedNumpyEngine), world_size 2 / 3 over gloo, against
    the single-process oracle."""
    got = _check_sharded(cfg, world, tmp_path)
    assert int(got["fused"]) == cfg["iters"]
    if cfg.get("mispredict_every"):
        assert int(got["mispred"]) >= 3
    else:
        assert int(got["mispred"]) <= 1       # only a residual within rounding of the 1e-2 threshold can mispredict


@pytest.mark.parametrize("world,cfg", [
    (3, dict(n=501, d=9, seed=4, wf="superquantile", args=[0.5], loss="binary_cross_entropy", reg=0.01, l1=False, iters=8)),
    (4, dict(n=403, d=7, seed=5, wf="aorr", args=[0.2, 0.8], loss="hinge", reg=1e-4, l1=False, iters=8)),
    (4, dict(n=300, d=6, seed=6, wf="ehrm", B=-5, loss="binary_cross_entropy", reg=0.01, l1=False, iters=8)),
    (2, dict(n=501, d=9, seed=4, wf="extremile", args=[2.0], loss="binary_cross_entropy", reg=0.01, l1=False, iters=8,
             dist_z=False)),
    (4, dict(n=5, d=3, seed=8, wf="superquantile", args=[0.5], loss="binary_cross_entropy", reg=0.01, l1=False, iters=5)),
    (3, dict(n=64, d=4, seed=9, wf="esrm", args=[1.0], loss="hinge", reg=0.01, l1=True, iters=6)),
], ids=["superq_world3", "aorr_world4", "ehrm_world4", "extremile_replicated_z", "tiny_with_an_empty_rank", "esrm_hinge_l1_world3"])
def test_distributed_z_step_more_ranks(world, cfg, tmp_path):
    """rank-weighted problems with the sorted order partitioned over 3 / 4 ranks (sample sort,
    local PAV, merge tree over ranks: dist.py:_z_distributed on the NumPy engine), and the
    replicated all-gather form."""
    _check_sharded(cfg, world, tmp_path)


@pytest.mark.parametrize("world,cfg", [
    (3, dict(n=2001, d=9, seed=4, wf="superquantile", args=[0.5], loss="binary_cross_entropy", reg=0.01, l1=False, iters=12)),
    (4, dict(n=1603, d=7, seed=5, wf="aorr", args=[0.2, 0.8], loss="hinge", reg=1e-4, l1=False, iters=12)),
    (2, dict(n=1500, d=8, seed=7, wf="aorr_dc", args=[1100, 200], loss="binary_cross_entropy", reg=1e-4, l1=False, iters=12)),
    (4, dict(n=900, d=6, seed=9, wf="superquantile", args=[0.37], loss="hinge", reg=0.01, l1=True, iters=12)),
], ids=["superq_world3", "aorr_hinge_world4", "aorr_dc_world2", "superq_0.37_hinge_l1_world4"])
def test_sort_free_distributed_z_step(world, cfg, tmp_path):
    """rank weights that are constant on a few bands: dist.py:_z_banded (histograms and block sums summed over the
    ranks, the undecided elements gathered; oracle/zband.py:Passes restates rbl_zbd_* for the NumPy engine) over gloo
    against the single-process oracle.  The sort-free path must have been certified on all but the first iterations."""
    got = _check_sharded(dict(cfg, banded=True), world, tmp_path)
    assert int(got["banded"]) >= cfg["iters"] - 4, int(got["banded"])
    # collectives of a certified sort-free z-step: 6 histogram sums + per band edge that can pool ONE root-pass sum (the
    # ranks stop after the pass that settles: rbl_zbd_decide's verdict) and one gather - 8 for one edge, where round 2
    # always issued 4 root-pass sums (11).  The first certified steps may need a second pass (no history yet).
    zc = got["zcoll"]
    steady = [int(c) for c, banded, _ in zc[-4:] if banded]
    edges = int(zc[-1][2])
    assert len(steady) >= 3 and max(steady) <= 6 + 2 * edges, (zc.tolist(), edges)
    assert all(int(c) <= 6 + 5 * int(e) for c, banded, e in zc if banded), zc.tolist()


def _check_sharded(cfg, world, tmp_path):
    import torch.multiprocessing as mp
    from oracle import admm, problems
    out = str(tmp_path / "r0.npz")
    port = 29500 + (os.getpid() + hash(cfg["wf"]) + 7 * world) % 2000
    mp.spawn(_worker, args=(world, port, cfg, out), nprocs=world, join=True)
    got = np.load(out)
    X, y = problems.make_problem(cfg["n"], cfg["d"], cfg["seed"])
    kw = dict(weight_function=cfg["wf"], loss=cfg["loss"], args=cfg.get("args"), B=cfg.get("B"))
    kw["l1_reg" if cfg["l1"] else "l2_reg"] = cfg["reg"]
    ref = admm.admm_solve(X, y, max_iter=cfg["iters"], mode="exact", tol=0.0, **kw)
    # shard-count invariance: two ranks reproduce the single-process iterates
    assert np.max(np.abs(got["w"] - ref.w)) <= 1e-10 * max(1.0, np.max(np.abs(ref.w)))
    hist = got["hist"]
    assert np.allclose(hist[:, 0], ref.primal, rtol=1e-9, atol=1e-12)
    assert np.allclose(hist[:, 1], ref.dual, rtol=1e-9, atol=1e-12)
    assert np.allclose(hist[:, 2], ref.rho, rtol=1e-15)
    assert np.allclose(hist[:, 3], ref.objective[1:], rtol=1e-9)
    return got
