#!/usr/bin/env python3
"""Benchmark of the hot path: ADMM iterations/sec on synthetic n x d data.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--rows R] [--cols D] [--config NAME] [--storage f32|f64]

Workload at N=1 (BASELINE.json configs[1], "C2"): SRM, erm weights, binary cross entropy,
l1_reg = 0.01, synthetic 6 000 000 x 1 000 generated on the device, D stored fp32
(24 GB), fp64 accumulation.  One "step" = one full ADMM iteration (reference
src/optim/algorithms.py:119-157: z-step, w-step, dual update, residuals, rho schedule).  For erm that
is ONE pass over D (k_sweep_erm: dual update of iteration k + z-step and D^T c of iteration k+1 while the row
is in registers) plus the d-space w-step; rank-weighted configurations (C2sq, C3, C4shard) take two passes
(v-only sweep + k_gemvt) around the sort + PAV z-step.  For N > 1 the SAME problem is row-sharded over the N
GPUs (strong scaling, one rank per GPU, launched by torch.distributed.run): erm iterations issue ONE RCCL
all-reduce of 2 ld + 3 doubles per iteration; rank-weighted ones add the distributed z-step's exchanges
(`config.collectives_per_iteration` / `config.host_syncs_per_iteration` in the N > 1 line say how many).

Prints ONE JSON line with the contract fields plus
  roofline:     algorithmic HBM bytes of the dominant sweep kernel / its average launch
                duration (HIP events on the library's stream, inside the timed region)
  cpu_baseline: the reference-faithful CPU restatement (oracle, "port") timed on this
                box's host cores on a bounded sample of the same workload (rank 0, N=1); one-time setup
                (D, D^T D) and the first iterations are timed separately from the per-iteration rate
  config.time_to_gap: the second half of the metric - wall-clock to F(w_k) - F* <= 1e-6 with F* from a
                SEPARATE tightened run (tol 1e-8, <= 2000 iterations, fp64 storage; SURVEY 8d, run_SRM.py:100,111)
  config.c1:    BASELINE configs[0] (6000 x 1000 through set_data) on the GPU next to the published 16.78 s /
                10.94 s of table/erm_synthetic_6000x1000_l1_binary_cross_entropy.xlsx
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
    "C3dc": dict(rows=10_000_000, cols=1001, weight_function="aorr_dc", loss="hinge", wstep=2, reg=1e-4,
                 args=[8_000_000, 2_000_000], B=None,
                 label="AoRR aorr_dc[k=8M,m=2M] / hinge / l2=1e-4, synthetic 10000000x1001 (the weights of run_AoRR_fixed.py)"),
    # one GPU's share of the 8-GPU configurations (BASELINE configs[3], configs[4]) as standalone problems
    "C4shard": dict(rows=6_250_000, cols=1000, weight_function="ehrm", loss="binary_cross_entropy", wstep=2, reg=0.01,
                    args=None, B=-5.0, label="EHRM ehrm / BCE / l2=0.01 / B=-5, synthetic 6250000x1000"),
    "C5shard": dict(rows=1_250_000, cols=10000, weight_function="erm", loss="binary_cross_entropy", wstep=1, reg=0.01,
                    args=None, B=None, label="SRM erm / BCE / l1=0.01, synthetic 1250000x10000 dense"),
    # the 8-GPU configurations themselves (BASELINE configs[3], configs[4]): launch with --gpus 8 (rows / 8 per rank)
    "C4": dict(rows=50_000_000, cols=1000, weight_function="ehrm", loss="binary_cross_entropy", wstep=2, reg=0.01,
               args=None, B=-5.0, label="EHRM ehrm / BCE / l2=0.01 / B=-5, synthetic 50000000x1000 (BASELINE configs[3])"),
    "C5": dict(rows=10_000_000, cols=10000, weight_function="erm", loss="binary_cross_entropy", wstep=1, reg=0.01,
               args=None, B=None, label="SRM erm / BCE / l1=0.01, synthetic 10000000x10000 dense (BASELINE configs[4])"),
    "C2hinge": dict(rows=6_000_000, cols=1000, weight_function="erm", loss="hinge", wstep=1, reg=0.01,
                    args=None, B=None, label="SRM erm / hinge / l1=0.01, synthetic 6000000x1000"),
    "C2smooth": dict(rows=6_000_000, cols=1000, weight_function="erm", loss="binary_cross_entropy", wstep=3, reg=0.01,
                     args=None, B=None, label="sADMM erm / BCE / smoothed l1=0.01, synthetic 6000000x1000"),
    "C2l2": dict(rows=6_000_000, cols=1000, weight_function="erm", loss="binary_cross_entropy", wstep=2, reg=0.01,
                 args=None, B=None, label="SRM erm / BCE / l2=0.01, synthetic 6000000x1000"),
    "C2sq": dict(rows=6_000_000, cols=1000, weight_function="superquantile", loss="binary_cross_entropy", wstep=2,
                 reg=0.01, args=[0.5], B=None, label="SRM superquantile(0.5) / BCE / l2=0.01, synthetic 6000000x1000"),
}


def device_copy_rate(torch, device, gib=8, repeats=3):
    """Measured device-to-device copy rate of this box ("also report against the measured device-copy
    bandwidth of the box"): a `gib` GiB buffer copied `repeats` times on torch's current stream, best repeat, bytes read +
    bytes written per second.  Runs after the timed region; ~20 ms."""
    n = (gib << 30) // 4
    src = torch.empty(n, dtype=torch.float32, device=device).fill_(1.0)
    dst = torch.empty_like(src)
    dst.copy_(src)
    best = None
    for _ in range(repeats):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dst.copy_(src)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1)
        best = ms if best is None or ms < best else best
    del src, dst
    torch.cuda.empty_cache()
    return 2.0 * n * 4 / (best * 1e-3) / 1e9


def _host_threads():
    try:
        from threadpoolctl import threadpool_info
        return int(max([p.get("num_threads", 1) for p in threadpool_info()] + [1]))
    except Exception:
        return os.cpu_count() or 1


def cpu_baseline(cfg, seconds_budget=20.0, repeats=2):
    """Reference-faithful CPU mode (oracle/admm.py mode='faithful': fp32 n-space FISTA with
    backtracking, batch Newton prox, sweep PAV - the reference's own structure) on a FIXED sample of the
    workload's rows: 60 000 rows for erm (capped at 6e7 matrix elements for wide problems) - the size SURVEY 8d
    names; the rank-weighted families, whose reference z-step is a Python loop (EHRM: Newton systems inside it,
    ~100x the others), keep a sample sized by a 2 000-row probe to about `seconds_budget` of CPU work.  Per-
    iteration cost is linear in n, so it/s is scaled by sample_rows / rows.  ONE solve of
    2 + K iterations per repeat with a time stamp after every iteration: the rate is K / (t[2+K] - t[2]), so the
    one-time D^T D and the first two iterations (the long first FISTA, fast_lasso.py:22-69 from a cold start)
    are out of `value` and listed beside it.  `value` is the BEST of the repeats (the host is shared); every
    repeat's rate and their spread are reported (round 2's single short run moved 4x between boxes)."""
    import numpy as np
    from oracle import problems, admm
    cores = _host_threads()
    d = cfg["cols"]
    kw = dict(weight_function=cfg["weight_function"], loss=cfg["loss"], args=cfg["args"], B=cfg["B"])
    kw["l1_reg" if cfg["wstep"] in (1, 3) else "l2_reg"] = cfg["reg"]
    smooth = cfg["wstep"] == 3
    K = 5
    if cfg["weight_function"] == "erm":
        n_s = int(min(60_000, cfg["rows"], max(2_000, 60_000_000 // d)))
        sizing = "fixed"
    else:
        n_p = min(2_000, cfg["rows"])
        X, y = problems.make_problem(n_p, d, seed=17)
        t0 = time.perf_counter()
        admm.admm_solve(X, y, max_iter=2, mode="faithful", store=False, tol=0.0, smooth=smooth, **kw)
        per_row_iter = (time.perf_counter() - t0) / (2 * n_p)
        n_s = int(min(60_000, cfg["rows"], max(n_p, seconds_budget / (repeats * (K + 4) * per_row_iter))))
        sizing = "2000-row probe"
    X, y = problems.make_problem(n_s, d, seed=17)
    rates, setups, firsts = [], [], []
    for _ in range(repeats):
        st = []
        t0 = time.perf_counter()
        tr = admm.admm_solve(X, y, max_iter=2 + K, mode="faithful", store=False, tol=0.0, smooth=smooth, stamps=st, **kw)
        setups.append(st[0] - t0)                    # D = -y X and D^T D (algorithms.py:23-24)
        firsts.append(st[2] - st[0])
        rates.append(K / max(st[2 + K] - st[2], 1e-9))
    best = max(rates)
    return dict(value=best * n_s / cfg["rows"], unit="iterations/s", cores=int(cores), kind="port",
                setup_s=round(min(setups), 3), first_two_iterations_s=round(min(firsts), 3),
                sample_rows=n_s, sample_sizing=sizing, sample_rate_its=[round(r, 4) for r in rates],
                spread=round((max(rates) - min(rates)) / max(rates), 3),
                sample=f"iterations 3..{tr.iters} of the reference-faithful mode on a {n_s}x{d} sample ({sizing}), best of "
                       f"{repeats} solves ({', '.join('%.3f' % r for r in rates)} it/s on the sample), scaled by "
                       f"{n_s}/{cfg['rows']}; the one-time D^T D ({min(setups):.2f} s) and the first two iterations "
                       f"(cold-start FISTA, {min(firsts):.2f} s) are excluded and listed beside it")


def c1_cpu_same_box(Xtr, ytr, f_star_gpu, repeats=2):
    """BASELINE configs[0] measured directly on THIS box's host cores: the
    oracle's reference-faithful mode solves the reference's own 6000 x 1000 problem to the reference's stop rule
    (tol 1e-4, max_iter 200, objective logged every iteration as run_SRM.py does through start_store), with a time
    stamp per iteration.  Time to the 1e-6 gap is given against the run's own smallest objective (run_SRM.py:100
    takes F* as the minimum over the logged runs) and against the GPU's tightened F*.  The port is vectorised
    NumPy where the reference builds n Python Block objects per iteration (pav.py:67-68): it is several times
    FASTER than the reference itself (2.3 s against the reference's 16.7 s in the build container, 8 cores), so the
    ratio to it under-states the speed-up over the reference."""
    import numpy as np
    from oracle import admm
    runs = []
    for _ in range(repeats):
        st = []
        t0 = time.perf_counter()
        tr = admm.admm_solve(Xtr, ytr, weight_function="erm", loss="binary_cross_entropy", l1_reg=0.01, mode="faithful",
                             store=True, stamps=st)
        total = time.perf_counter() - t0
        F = np.array(tr.objective[1:])               # objective after iteration k (k = 1..)
        T = np.array(st[1:1 + F.size]) - t0          # includes the one-time D^T D, as the reference's clock does not (:209-212)
        def first(gap_to):
            idx = np.flatnonzero(F - gap_to <= 1e-6)
            return {"iterations": int(idx[0]) + 1, "seconds": float(T[idx[0]])} if idx.size else None
        runs.append({"iterations_to_stop": int(tr.iters), "converged": bool(tr.converged), "seconds_to_stop": round(total, 3),
                     "setup_s": round(st[0] - t0, 3), "final_objective": float(tr.final_objective),
                     "gap_1e-6_vs_own_min": first(float(F.min())), "gap_1e-6_vs_gpu_F_star": first(float(f_star_gpu))})
    best = min(runs, key=lambda r: r["seconds_to_stop"])
    secs = [r["seconds_to_stop"] for r in runs]
    return dict(best, kind="port", cores=_host_threads(), repeats=secs,
                spread=round((max(secs) - min(secs)) / max(secs), 3))


def _initial_state(s, cfg, n_total, d):
    """the reference's initial point (algorithms.py:32-52)"""
    import numpy as np
    reg, wf = cfg["reg"], cfg["weight_function"]
    rho0 = 1e-4 if wf == "ehrm" else (2e-7 if wf in ("aorr", "aorr_dc") else 1e-5)
    s.set_state(w=np.full(d, 0.001 * reg / d / n_total), z=np.full(s.n, 0.1 * reg / n_total),
                lam=np.full(s.n, 0.1 * reg / n_total), rho=rho0, iter=0, smooth_t=1.0)   # t: smoothADMMmethod's default


def f_star_run(make_solver, cfg, n_total, d, tol=1e-8, max_iter=2000):
    """F* of the gap metric (mirrors run_SRM.py:100,111: the best objective over the logged
    runs): a SEPARATE, tightened solve of the same problem - stop tolerance 1e-8 on both residuals instead
    of the reference's 1e-4, at most 2000 iterations, fp64 storage of D when it fits - whose smallest logged
    objective is F*."""
    s, storage = make_solver(tol)
    t0 = time.perf_counter()
    best, k_best, k, st = float("inf"), 0, 0, None
    for k in range(1, max_iter + 1):
        st = s.step(True)
        if st.objective < best:
            best, k_best = st.objective, k
        if st.converged:
            break
    out = {"F_star": best, "at_iteration": k_best, "iterations": k, "tol": tol, "converged": bool(st.converged),
           "storage": storage, "seconds": round(time.perf_counter() - t0, 3),
           "final_primal": st.primal, "final_dual": st.dual}
    s.close()
    return out


def time_to_gap(s, cfg, n_total, d, fstar, max_iter=2000):
    """Second half of the metric: wall-clock from the reference's initial state (algorithms.py:32-52) until
    F(w_k) - F* <= 1e-6 (absolute, and relative 1e-6 |F*|), objective logged every iteration as the
    reference's drivers do (start_store).  F* comes from f_star_run.  The reference's own stop rule (both
    residuals < 1e-4, algorithms.py:137) is recorded on the way: when it fires ABOVE the gap - it does at
    large n, where the un-normalised residual norms fall below 1e-4 long before the objective has converged
    - `gap_*_at_stop_rule` is null and `stop_rule_gap` says how far away it stopped; the run then simply
    continues (the handle's own tolerance is 0) until the gap is reached or max_iter."""
    import numpy as np
    _initial_state(s, cfg, n_total, d)
    F, T = [], []
    stop_k = None
    t0 = time.perf_counter()
    for k in range(max_iter):
        st = s.step(True)
        F.append(st.objective)
        T.append(time.perf_counter() - t0)
        if stop_k is None and st.primal < 1e-4 and st.dual < 1e-4:
            stop_k = k
        if stop_k is not None and st.objective - fstar <= 1e-6 * min(1.0, abs(fstar)):
            break
    F = np.array(F)

    def first(mask, upto=None):
        idx = np.flatnonzero(mask if upto is None else mask[: upto + 1])
        return {"iterations": int(idx[0]) + 1, "seconds": float(T[int(idx[0])])} if idx.size else None

    ga, gr = F - fstar <= 1e-6, F - fstar <= 1e-6 * abs(fstar)
    out = {"F_star": fstar, "iterations_run": int(F.size), "seconds_run": float(T[-1]),
           "objective_last": float(F[-1]), "gap_last": float(F[-1] - fstar),
           "gap_abs_1e-6": first(ga), "gap_rel_1e-6": first(gr),
           "stop_rule": None}
    if stop_k is not None:
        out["stop_rule"] = {"iterations": stop_k + 1, "seconds": float(T[stop_k]), "objective": float(F[stop_k])}
        out["stop_rule_gap"] = float(F[stop_k] - fstar)
        out["gap_abs_1e-6_at_stop_rule"] = first(ga, stop_k)
        out["gap_rel_1e-6_at_stop_rule"] = first(gr, stop_k)
    return out


def c1_record(rbl, device):
    """BASELINE configs[0] on the GPU: SRM erm / BCE / l1 = 0.01 on the reference's own 6000 x 1000 data
    (run_SRM.py:21-36 via load_data.py:101-116: make_classification(10000, 1000, random_state=17), labels
    0 -> -1, preprocessing.scale, train_test_split(test_size=0.4, random_state=17)) uploaded with set_data,
    D stored fp64 as the reference holds it.  Time to the reference's stop rule and to the 1e-6 gap, next to the
    published CPU numbers of table/erm_synthetic_6000x1000_l1_binary_cross_entropy.xlsx (BASELINE.md section 1;
    hardware not stated by the reference)."""
    from sklearn.datasets import make_classification
    from sklearn import preprocessing
    from sklearn.model_selection import train_test_split
    X, label = make_classification(n_samples=10000, n_features=1000, n_classes=2, random_state=17)
    label[label == 0] = -1
    X = preprocessing.scale(X)
    Xtr, _, ytr, _ = train_test_split(X, label.reshape(-1, 1), test_size=0.4, random_state=17)
    n, d = Xtr.shape
    cfg = dict(weight_function="erm", loss="binary_cross_entropy", wstep=1, reg=0.01, args=None, B=None)

    def make(tol):
        s = rbl.Solver(n, d, "erm", "binary_cross_entropy", reg=0.01, wstep=1, storage="f64", device=device, tol=tol)
        s.set_data(Xtr, ytr)
        s.gram()
        return s

    t0 = time.perf_counter()
    s = make(0.0)
    setup = time.perf_counter() - t0
    fs = f_star_run(lambda tol: (make(tol), "f64"), cfg, n, d)
    s.step(False)                       # first-call costs (module load, lazy allocations) out of the timed solve
    gap = time_to_gap(s, cfg, n, d, fs["F_star"])
    s.close()
    try:
        cpu = c1_cpu_same_box(Xtr, ytr, fs["F_star"])
    except Exception as e:
        cpu = {"error": repr(e)[:200]}
    return {"workload": "SRM erm / BCE / l1=0.01, reference data 6000x1000 (BASELINE configs[0]), fp64 storage",
            "setup_s": round(setup, 3), "f_star_run": fs, "time_to_gap": gap, "cpu_same_box": cpu,
            "published_cpu": {"iterations_to_stop": 86, "seconds_to_stop": 16.78, "seconds_to_gap_1e-6": 10.94,
                              "final_objective": 0.1475231430518671, "hardware": "not stated by the reference",
                              "source": "table/erm_synthetic_6000x1000_l1_binary_cross_entropy.xlsx (BASELINE.md section 1)"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="C2", choices=sorted(CONFIGS))
    ap.add_argument("--rows", type=int, default=0)
    ap.add_argument("--cols", type=int, default=0)
    ap.add_argument("--storage", default="f32", choices=["f32", "f64"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gap", action="store_true", help="skip the wall-clock-to-1e-6-gap run (and its F* run)")
    ap.add_argument("--no-c1", action="store_true", help="skip the C1 (6000x1000 reference data) sub-record")
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
    # rehearsal of the N > 1 flow of this script on a ONE-GPU box (tools/rehearse_bench_n2.sh): every rank on device 0,
    # gloo instead of RCCL (which refuses two ranks on one device; the driver then stages its collectives through the
    # host).  The line it prints says so and is not a measurement of anything.
    rehearsal = os.environ.get("RBL_BENCH_REHEARSE_ONE_GPU") == "1"
    if rehearsal:
        local_rank = 0
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
    sharded = world > 1 or a.sharded_driver
    saved_stdout = None
    if sharded:
        # RCCL prints a version banner on stdout when its first communicator comes up: stdout carries exactly ONE
        # JSON line (rank 0), so everything until that line is written goes to stderr
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29733")
        torch.cuda.set_device(local_rank)
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    if _lib.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: librbl has no CPU fallback")

    n_total, d = cfg["rows"], cfg["cols"]
    from admm_for_rank_based_loss_amd.dist import ShardedADMM, GpuEngine, shard_rows
    off, n_local, _ = shard_rows(n_total, world, rank)
    if n_local * d * (4 if a.storage == "f32" else 8) > 230e9:
        raise SystemExit(f"--config {a.config}: {n_local} x {d} rows per GPU do not fit one MI355X (288 GB): "
                         f"launch with more GPUs (--gpus N through torch.distributed.run)")
    t_setup = time.perf_counter()
    s = rbl.Solver(n_local, d, cfg["weight_function"], cfg["loss"], reg=cfg["reg"], wstep=cfg["wstep"], B=cfg["B"],
                   args=cfg["args"], storage=a.storage, device=local_rank, n_total=n_total, row_offset=off,
                   tol=0.0)   # tol 0: a fixed number of iterations, never "converged"
    # Everything that makes the device wait for the host happens BEFORE the last piece of set-up work on the device
    # (the Gram product): the sweep runs ~5 % slower for its first ~25 launches after the device has sat idle for
    # 5 ms or more (shader-clock DPM, DESIGN section 5 "the ramp"; tools/ramp_probe.py) but not after compute
    # kernels.  Round 2 had a gc.collect() and the profiling set-up (~40 ms of idle device) between the warm-up and
    # the timed region, the first half of round 3 had them between the set-up and the warm-up; now the device goes
    # Gram product -> warm-up -> timed steps without a host-side pause.
    import gc

    def quiet_host():
        s.profile_kernels(2 if a.phase_times else 1)
        # HIP events around every launch of the sweep kernels at N = 1 (two event records cost ~11 us of stream
        # time, 0.3 % of a 4 ms pass); every 4th one with several GPUs, where a rank's pass is 0.55 ms
        s.profile_sampling(1 if world == 1 else (4 if a.steps >= 16 else 1))
        gc.collect()
        gc.disable()      # no collector pause inside the timed region (with N ranks the slowest one sets the pace)

    if sharded:
        drv = ShardedADMM(GpuEngine(s, local_rank))
        drv.always_allreduce = a.sharded_driver
        drv.setup_synthetic(a.seed)
        quiet_host()
        drv.setup_gram()
        step = lambda: drv.step(False)
    else:
        s.generate_synthetic(a.seed)
        quiet_host()
        s.gram()
        step = lambda: s.step(False)
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup
    prof_every = 1 if world == 1 else (4 if a.steps >= 16 else 1)
    for _ in range(a.warmup):
        step()
    s.reset_kernel_times()      # host bookkeeping only: the device goes straight from the warm-up into the timed steps
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    last = None
    n_fused = n_mispred = n_coll = n_drv_sync = n_lib_sync = n_zband = n_zredo = 0
    t_steps = []
    for _ in range(a.steps):
        last = step()
        t_steps.append(time.perf_counter())
        n_fused += int(last.fused)
        n_mispred += int(last.mispredicted)
        n_coll += int(getattr(last, "collectives", 0))          # set by the multi-GPU driver (dist.py)
        n_drv_sync += int(getattr(last, "driver_syncs", 0))
        n_lib_sync += int(last.host_syncs)
        n_zband += int(last.zband == 1)
        n_zredo += int(last.zband == 2)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # the box's own device-to-device copy rate, AFTER the timed region (measured before it - 16 GB allocated and freed
    # ahead of the data - the sweep ran 1-1.5 % slower for the whole run: interleaved on one box 277.9 / 278.3 against
    # 282.0 / 281.6 it/s; physical placement of the 24 GB matrix)
    copy_gbs = (device_copy_rate(torch, torch.device("cuda", local_rank))
                if world == 1 and os.environ.get("RBL_BENCH_NO_COPY") != "1" else None)
    esz = 4 if a.storage == "f32" else 8
    kt = {}
    for name, kid in (("gemv", _lib.KERNEL_GEMV), ("gemvt", _lib.KERNEL_GEMVT), ("sweep_erm", _lib.KERNEL_SWEEP_ERM)):
        ms, cnt = s.kernel_time(kid)
        kt[name] = dict(avg_ms=ms / max(cnt, 1), launches=cnt, total_ms=ms, samples=s.kernel_samples(kid))
    dom = max(kt, key=lambda k: kt[k]["total_ms"])     # the kernel the timed region spends most time in
    bytes_per_launch = n_local * d * esz            # algorithmic: every element of this rank's D read once
    achieved = bytes_per_launch / (kt[dom]["avg_ms"] * 1e-3) / 1e9 if kt[dom]["avg_ms"] > 0 else 0.0
    # One record shows both ends of the timed region: the first and the last timed launch of
    # the dominant kernel, and a steady-state sub-record = the median over the LAST THIRD of the timed launches
    # (kernel) and of the timed steps (wall clock between the returns of consecutive step() calls).  `value`,
    # `ms_per_step`, `achieved` and `frac` stay the means over all K timed steps.
    import statistics
    smp = [float(x) for x in kt[dom]["samples"]]
    third = smp[len(smp) - max(1, len(smp) // 3):] if smp else []
    k_med = statistics.median(third) if third else 0.0
    dts = [t_steps[i] - t_steps[i - 1] for i in range(1, len(t_steps))]
    dts_third = dts[len(dts) - max(1, len(dts) // 3):] if dts else []
    step_med = statistics.median(dts_third) if dts_third else 0.0
    steady = {"launches": len(third), "kernel_ms_median": round(k_med, 4),
              "achieved": (bytes_per_launch / (k_med * 1e-3) / 1e9) if k_med > 0 else 0.0,
              "frac": (bytes_per_launch / (k_med * 1e-3) / 1e9 / HBM_PEAK_GBS) if k_med > 0 else 0.0,
              "ms_per_step_median": round(step_med * 1e3, 4),
              "iterations_per_s": (1.0 / step_med) if step_med > 0 else 0.0,
              "window": "median over the last third of the timed launches / timed steps"}
    # HBM traffic cannot be counted from inside this process (PMC counters need rocprofv3 passes of their own,
    # MI355X_MICROARCH.md): the figure is READ from the committed summary of such passes over this same
    # command (tools/profile_round.sh -> profiles/traffic_latest.json) and labelled with its source; null
    # when no summary matches the workload
    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tpath) and world == 1:
        try:
            tj = json.load(open(tpath))
            if tj.get("rows") == n_total and tj.get("cols") == d and tj.get("storage") == a.storage:
                traffic = tj.get("kernels", {}).get(dom, {}).get("hbm_bytes_per_launch")
                if traffic is not None:
                    traffic_source = ("profiles/traffic_latest.json: rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of "
                                      "this command in an EARLIER run on another box (%s), not measured in this run"
                                      % tj.get("source", "see tools/profile_round.sh"))
        except Exception:
            traffic = None

    if rank == 0:
        out = {
            "metric": "ADMM iterations/sec + wall-clock to 1e-6 primal gap, n×d synthetic",
            "value": a.steps / dt, "unit": "iterations/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic" if not rehearsal else "synthetic - REHEARSAL: all ranks on one GPU over gloo, not a measurement",
            "config": {"workload": cfg["label"], "rows": n_total, "cols": d, "storage": a.storage,
                       "sharding": f"rows/{world}", "driver": "ShardedADMM" if sharded else "single handle", "setup_s": round(t_setup, 3),
                       "inner_iters_last": int(last.inner_iters), "phase_ms_last": ({
                           "z": round(last.ms_z, 3), "q": round(last.ms_q, 3), "w": round(last.ms_w, 3),
                           "v": round(last.ms_v, 3), "total": round(last.ms_total, 3)} if a.phase_times else None),
                       "single_sweep_iterations": n_fused, "rho_mispredictions": n_mispred,
                       "sort_free_z_steps": n_zband, "sort_free_z_steps_redone": n_zredo,
                       # how an iteration talks (rank 0): collectives issued by the multi-GPU driver, its host
                       # waits (device -> host reads of count / bound vectors) and the library's own (the
                       # pinned statistics block, the w-step's status word)
                       "collectives_per_iteration": round(n_coll / a.steps, 2),
                       "driver_syncs_per_iteration": round(n_drv_sync / a.steps, 2),
                       "host_syncs_per_iteration": round(n_lib_sync / a.steps, 2)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": "k_" + dom,
                         "device_copy_GBs": round(copy_gbs, 1) if copy_gbs else None,
                         "achieved_over_device_copy": round(achieved / copy_gbs, 3) if copy_gbs else None,
                         "bytes_per_launch": bytes_per_launch, "timed_every": prof_every,
                         "kernel_ms_first": round(smp[0], 4) if smp else None,
                         "kernel_ms_last": round(smp[-1], 4) if smp else None,
                         "kernel_ms_min": round(min(smp), 4) if smp else None,
                         "kernel_ms_max": round(max(smp), 4) if smp else None,
                         "steady_state": steady,
                         "kernels": {k: {"avg_ms": round(v["avg_ms"], 4), "launches": v["launches"],
                                         "GBps": round(bytes_per_launch / (v["avg_ms"] * 1e-3) / 1e9, 1)
                                         if v["avg_ms"] > 0 else 0.0} for k, v in kt.items()}},
        }
        if world == 1 and not a.no_gap:
            try:
                # F* from a separate tightened run: fp64 storage when a second copy of D fits beside this one
                fs_storage = "f64" if n_total * d * 8 <= 110e9 else a.storage

                def make(tol):
                    s2 = rbl.Solver(n_total, d, cfg["weight_function"], cfg["loss"], reg=cfg["reg"], wstep=cfg["wstep"],
                                    B=cfg["B"], args=cfg["args"], storage=fs_storage, device=local_rank, tol=tol)
                    s2.generate_synthetic(a.seed)
                    s2.gram()
                    return s2, fs_storage

                fs = f_star_run(make, cfg, n_total, d)
                out["config"]["f_star_run"] = fs
                out["config"]["time_to_gap"] = time_to_gap(s, cfg, n_total, d, fs["F_star"])
            except Exception as e:      # the second half of the metric is informative, never fatal
                out["config"]["time_to_gap"] = {"error": repr(e)[:200]}
        if world == 1 and not a.no_c1 and a.config == "C2" and not a.rows and not a.cols:
            try:
                s.close()
                out["config"]["c1"] = c1_record(rbl, local_rank)
            except Exception as e:
                out["config"]["c1"] = {"error": repr(e)[:200]}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
        sys.stdout.flush()
        if saved_stdout is not None:
            os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        if saved_stdout is not None:
            os.dup2(2, 1)          # whatever the teardown prints is not part of the record
    if sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
