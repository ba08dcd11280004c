#!/usr/bin/env python3
"""Epoch times of the competitor baselines (SGDmethod / LSVRGmethod) on the GPU next to the CPU
restatement (oracle/baselines.py) on the reference's C1 shape (6000 x 1000):  python tests/diag_baseline_timing.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import admm_for_rank_based_loss_amd as R      # noqa: E402
from oracle import problems, baselines         # noqa: E402

X, y = problems.make_problem(6000, 1000, seed=17)
kw = dict(weight_function="superquantile", loss="binary_cross_entropy", l2_reg=1.0, args=[0.5])
for name, gpu, cpu, extra, epochs in (
        ("SGD (batch 64, 93 steps/epoch)", R.SGDmethod, baselines.sgd_solve, dict(lr=0.01), 40),
        ("LSVRG uniform (100 steps/epoch)", R.LSVRGmethod, baselines.lsvrg_solve, dict(lr=0.0001, uniform=True), 40),
        ("LSVRG non-uniform", R.LSVRGmethod, baselines.lsvrg_solve, dict(lr=0.0001, uniform=None), 40)):
    np.random.seed(1); torch.manual_seed(1)
    gpu(X, y, max_iter=2, train_loss=lambda w: 0.0, test_loss=lambda w: 0.0, verbose=False, **kw, **extra)    # warm-up
    np.random.seed(1); torch.manual_seed(1)
    t0 = time.perf_counter()
    wg = gpu(X, y, max_iter=epochs, train_loss=lambda w: 0.0, test_loss=lambda w: 0.0, verbose=False, **kw, **extra)[0]
    tg = time.perf_counter() - t0
    np.random.seed(1); torch.manual_seed(1)
    t0 = time.perf_counter()
    wc, _ = cpu(X, y, max_iter=epochs, **kw, **extra)
    tc = time.perf_counter() - t0
    err = np.max(np.abs(wg.reshape(-1) - wc)) / max(1e-300, np.max(np.abs(wc)))
    print(f"{name:34s} GPU {tg / epochs * 1e3:8.2f} ms/epoch   CPU restatement {tc / epochs * 1e3:8.2f} ms/epoch   "
          f"max rel. deviation of w after {epochs} epochs {err:.1e}")
