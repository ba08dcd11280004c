#!/usr/bin/env python3
"""Benchmark of the hot path: ADMM iterations/sec on synthetic n x d data.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--rows R] [--cols D] [--config NAME]

Workload at N=1 (BASELINE.json configs[1], "C2"): SRM, erm weights, binary cross entropy,
l1_reg = 0.01, synthetic 6 000 000 x 1 000 generated on the device, D stored fp32
(24 GB), fp64 accumulation.  One "step" = one full ADMM iteration (z-step, q = D^T c sweep,
d-space w-step, v = D w sweep, dual update, residuals, rho schedule; reference
src/optim/algorithms.py:119-157).  For N > 1 the SAME 6M-row problem is row-sharded over
the N GPUs (strong scaling): per iteration one all-reduce of d doubles and one of 2 doubles
(RCCL); launched by torch.distributed.run, one rank per GPU.

Prints ONE JSON line with the contract fields plus
  roofline:     algorithmic HBM bytes of the dominant sweep kernel / its average launch
                duration (HIP events on the library's stream, inside the timed region)
  cpu_baseline: the reference-faithful CPU restatement (oracle, "port") timed on this
                box's host cores on a bounded sample of the same workload (rank 0, N=1).
"""
import argparse
import json
import os
import sys
import time

# multi-process GPU work on this pool needs dmabuf IPC (RCCL / tensor sharing fail with the legacy mode)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable copy rate

CONFIGS = {
    # name: (rows, cols, weight_function, loss, reg kind, reg, args, B, intercept)
    "C2": dict(rows=6_000_000, cols=1000, weight_function="erm", loss="binary_cross_entropy", wstep=1, reg=0.01,
               args=None, B=None, label="SRM erm / BCE / l1=0.01, synthetic 6000000x1000 (BASELINE configs[1])"),
    "C3": dict(rows=10_000_000, cols=1001, weight_function="aorr", loss="hinge", wstep=2, reg=1e-4,
               args=[0.2, 0.8], B=None, label="AoRR aorr[0.2,0.8] / hinge / l2=1e-4, synthetic 10000000x1001"),
    # one GPU's share of the 8-GPU configurations (BASELINE configs[3], configs[4]) as standalone problems
    "C4shard": dict(rows=6_250_000, cols=1000, weight_function="ehrm", loss="binary_cross_entropy", wstep=2, reg=0.01,
                    args=None, B=-5.0, label="EHRM ehrm / BCE / l2=0.01 / B=-5, synthetic 6250000x1000"),
    "C5shard": dict(rows=1_250_000, cols=10000, weight_function="erm", loss="binary_cross_entropy", wstep=1, reg=0.01,
                    args=None, B=None, label="SRM erm / BCE / l1=0.01, synthetic 1250000x10000 dense"),
    "C2hinge": dict(rows=6_000_000, cols=1000, weight_function="erm", loss="hinge", wstep=1, reg=0.01,
                    args=None, B=None, label="SRM erm / hinge / l1=0.01, synthetic 6000000x1000"),
    "C2smooth": dict(rows=6_000_000, cols=1000, weight_function="erm", loss="binary_cross_entropy", wstep=3, reg=0.01,
                     args=None, B=None, label="sADMM erm / BCE / smoothed l1=0.01, synthetic 6000000x1000"),
    "C2l2": dict(rows=6_000_000, cols=1000, weight_function="erm", loss="binary_cross_entropy", wstep=2, reg=0.01,
                 args=None, B=None, label="SRM erm / BCE / l2=0.01, synthetic 6000000x1000"),
    "C2sq": dict(rows=6_000_000, cols=1000, weight_function="superquantile", loss="binary_cross_entropy", wstep=2,
                 reg=0.01, args=[0.5], B=None, label="SRM superquantile(0.5) / BCE / l2=0.01, synthetic 6000000x1000"),
}


def cpu_baseline(cfg, seconds_budget=20.0):
    """Reference-faithful CPU mode (oracle/admm.py mode='faithful': fp32 n-space FISTA with
    backtracking, batch Newton prox, sweep PAV - the reference's own structure) on a
    sample of the workload's rows (at most 60 000, fewer when a 2 000-row probe says that would exceed
    the time budget); per-iteration cost is linear in n, so it/s is scaled by
    sample_rows / rows."""
    import numpy as np
    from oracle import problems, admm
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        cores = os.cpu_count() or 1
    d = cfg["cols"]
    kw = dict(weight_function=cfg["weight_function"], loss=cfg["loss"], args=cfg["args"], B=cfg["B"])
    kw["l1_reg" if cfg["wstep"] == 1 else "l2_reg"] = cfg["reg"]
    iters = 6
    # size the sample for about seconds_budget of CPU work: a 2 000-row probe gives the cost per row and
    # iteration (the reference's EHRM z-step, Newton systems inside a Python PAV loop, is ~100x the others)
    n_p = min(2_000, cfg["rows"])
    X, y = problems.make_problem(n_p, d, seed=17)
    t0 = time.perf_counter()
    admm.admm_solve(X, y, max_iter=2, mode="faithful", store=False, tol=0.0, **kw)
    per_row_iter = (time.perf_counter() - t0) / (2 * n_p)
    n_s = int(min(60_000, cfg["rows"], max(n_p, seconds_budget / (iters * per_row_iter))))
    X, y = problems.make_problem(n_s, d, seed=17)
    t0 = time.perf_counter()
    tr = admm.admm_solve(X, y, max_iter=iters, mode="faithful", store=False, tol=0.0, **kw)
    dt = time.perf_counter() - t0
    its_sample = tr.iters / dt
    return dict(value=its_sample * n_s / cfg["rows"], unit="iterations/s", cores=int(cores), kind="port",
                sample=f"{tr.iters} reference-faithful iterations on a {n_s}x{d} sample "
                       f"({its_sample:.3f} it/s measured, scaled by {n_s}/{cfg['rows']}; includes the one-time "
                       f"D^T D and the first iteration's long FISTA)")


def time_to_gap(s, cfg, n_total, d, max_iter=400):
    """Second half of the metric: wall-clock until F(w_k) - F* <= 1e-6 where, as in the
    reference's driver (run_SRM.py:100-111), F* is the smallest objective of the logged run,
    i.e. of the run up to the reference's own stop rule (both residuals < 1e-4,
    algorithms.py:137).  The solver is reset to the reference's initial state
    (algorithms.py:32-52) and run with objective logging until that rule fires."""
    import numpy as np
    reg, n = cfg["reg"], n_total
    wf = cfg["weight_function"]
    rho0 = 1e-4 if wf == "ehrm" else (2e-7 if wf in ("aorr", "aorr_dc") else 1e-5)
    s.set_state(w=np.full(d, 0.001 * reg / d / n), z=np.full(s.n, 0.1 * reg / n), lam=np.full(s.n, 0.1 * reg / n),
                rho=rho0, iter=0)
    F, T = [], []
    stopped = False
    t0 = time.perf_counter()
    for k in range(max_iter):
        st = s.step(True)
        F.append(st.objective)
        T.append(time.perf_counter() - t0)
        if st.primal < 1e-4 and st.dual < 1e-4:
            stopped = True
            break
    F = np.array(F)
    fstar = float(F.min())

    def first(mask):
        idx = np.flatnonzero(mask)
        return {"iterations": int(idx[0]) + 1, "seconds": float(T[int(idx[0])])} if idx.size else None

    return {"stop_rule_reached": stopped, "iterations": int(F.size), "seconds_total": float(T[-1]),
            "final_objective": float(F[-1]), "F_star": fstar,
            "gap_abs_1e-6": first(F - fstar <= 1e-6), "gap_rel_1e-6": first(F - fstar <= 1e-6 * abs(fstar))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="C2", choices=sorted(CONFIGS))
    ap.add_argument("--rows", type=int, default=0)
    ap.add_argument("--cols", type=int, default=0)
    ap.add_argument("--storage", default="f32", choices=["f32", "f64"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gap", action="store_true", help="skip the wall-clock-to-1e-6-gap run")
    ap.add_argument("--phase-times", action="store_true",
                    help="HIP events around every phase (config.phase_ms_last); costs ~5 us of stream time per event")
    ap.add_argument("--seed", type=int, default=17)
    ap.add_argument("--sharded-driver", action="store_true",
                    help="N=1 only: run the multi-GPU driver (ShardedADMM over a 1-rank RCCL group) instead of the "
                         "single handle, to measure the driver's own per-iteration cost")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    import admm_for_rank_based_loss_amd as rbl
    from admm_for_rank_based_loss_amd import _lib

    cfg = dict(CONFIGS[a.config])
    if a.rows:
        cfg["rows"] = a.rows
    if a.cols:
        cfg["cols"] = a.cols
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
    sharded = world > 1 or a.sharded_driver
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29733")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    if _lib.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: librbl has no CPU fallback")

    n_total, d = cfg["rows"], cfg["cols"]
    from admm_for_rank_based_loss_amd.dist import ShardedADMM, GpuEngine, shard_rows
    off, n_local, _ = shard_rows(n_total, world, rank)
    t_setup = time.perf_counter()
    s = rbl.Solver(n_local, d, cfg["weight_function"], cfg["loss"], reg=cfg["reg"], wstep=cfg["wstep"], B=cfg["B"],
                   args=cfg["args"], storage=a.storage, device=local_rank, n_total=n_total, row_offset=off,
                   tol=0.0)   # tol 0: a fixed number of iterations, never "converged"
    if sharded:
        drv = ShardedADMM(GpuEngine(s, local_rank))
        drv.always_allreduce = a.sharded_driver
        drv.setup_synthetic(a.seed)
        drv.setup_gram()
        step = lambda: drv.step(False)
    else:
        s.generate_synthetic(a.seed)
        s.gram()
        step = lambda: s.step(False)
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup

    for _ in range(a.warmup):
        step()
    s.profile_kernels(2 if a.phase_times else 1)
    # the roofline timing brackets every 4th launch of the sweep kernels with HIP events (every launch in
    # short runs): two event records cost ~11 us of stream time, 2 % of a rank's pass at 8 GPUs
    prof_every = 4 if a.steps >= 16 else 1
    s.profile_sampling(prof_every)
    s.reset_kernel_times()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    last = None
    n_fused = n_mispred = 0
    for _ in range(a.steps):
        last = step()
        n_fused += int(last.fused)
        n_mispred += int(last.mispredicted)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    esz = 4 if a.storage == "f32" else 8
    kt = {}
    for name, kid in (("gemv", _lib.KERNEL_GEMV), ("gemvt", _lib.KERNEL_GEMVT), ("sweep_erm", _lib.KERNEL_SWEEP_ERM)):
        ms, cnt = s.kernel_time(kid)
        kt[name] = dict(avg_ms=ms / max(cnt, 1), launches=cnt, total_ms=ms)
    dom = max(kt, key=lambda k: kt[k]["total_ms"])     # the kernel the timed region spends most time in
    bytes_per_launch = n_local * d * esz            # algorithmic: every element of this rank's D read once
    achieved = bytes_per_launch / (kt[dom]["avg_ms"] * 1e-3) / 1e9 if kt[dom]["avg_ms"] > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tpath) and world == 1:
        try:
            tj = json.load(open(tpath))
            if tj.get("rows") == n_total and tj.get("cols") == d and tj.get("storage") == a.storage:
                traffic = tj.get("kernels", {}).get(dom, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    if rank == 0:
        out = {
            "metric": "ADMM iterations/sec + wall-clock to 1e-6 primal gap, n×d synthetic",
            "value": a.steps / dt, "unit": "iterations/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": cfg["label"], "rows": n_total, "cols": d, "storage": a.storage,
                       "sharding": f"rows/{world}", "driver": "ShardedADMM" if sharded else "single handle", "setup_s": round(t_setup, 3),
                       "inner_iters_last": int(last.inner_iters), "phase_ms_last": ({
                           "z": round(last.ms_z, 3), "q": round(last.ms_q, 3), "w": round(last.ms_w, 3),
                           "v": round(last.ms_v, 3), "total": round(last.ms_total, 3)} if a.phase_times else None),
                       "single_sweep_iterations": n_fused, "rho_mispredictions": n_mispred},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": "k_" + dom,
                         "bytes_per_launch": bytes_per_launch, "timed_every": prof_every,
                         "kernels": {k: {"avg_ms": round(v["avg_ms"], 4), "launches": v["launches"],
                                         "GBps": round(bytes_per_launch / (v["avg_ms"] * 1e-3) / 1e9, 1)
                                         if v["avg_ms"] > 0 else 0.0} for k, v in kt.items()}},
        }
        if world == 1 and not a.no_gap:
            try:
                out["config"]["time_to_gap"] = time_to_gap(s, cfg, n_total, d)
            except Exception as e:      # the second half of the metric is informative, never fatal
                out["config"]["time_to_gap"] = {"error": repr(e)[:200]}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(out))
    if sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
