#!/usr/bin/env python3
"""The ADMM part of the reference driver ``run_EHRM.py`` on the GPU package.

Same call sequence as the reference (run_EHRM.py:21-41): a 60/40 split that carries the group
attribute along (``train_test_split_group``, src/util/split_group.py:3-24: one seeded permutation),
``ADMMmethod`` with the ``ehrm`` weights, BCE, l2_reg = 0.01 and B = -5, ``start_store`` /
``main_loop`` / ``final_res``, the test accuracy and the six group-fairness statistics
(``calculate_statistics``).  The reference runs this on the UTKFace landmarks, which are not in
this repository: the data here are synthetic (make_classification + scale) with a binary group
attribute drawn to correlate with the label (--group-corr), so every statistic is exercised.
The rows run_EHRM.py:93-95 would write to the xlsx (train losses, cumulative times,
[acc, SPD, DI, EOD, AOD, TI, FNRD]) go to a CSV in the same order.  The competitor baselines
(SGD, LSVRG) are out of scope.

    python examples/run_ehrm.py [--rows 10000] [--cols 136] [--l2 0.01] [--B -5] [--max-iter 200] [--out rows.csv]
"""
import argparse
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def train_test_split_group(X, y, group, test_size, seed):
    """src/util/split_group.py:3-24: the first int(n * test_size) entries of one seeded permutation are the test set"""
    import numpy as np
    rng = np.random.RandomState(seed)
    idx = rng.permutation(X.shape[0])
    n_test = int(X.shape[0] * test_size)
    te, tr = idx[:n_test], idx[n_test:]
    return X[tr], X[te], y[tr], y[te], group[tr], group[te]


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10000)
    ap.add_argument("--cols", type=int, default=136)            # UTKFace: 68 landmarks x 2 coordinates
    ap.add_argument("--seed", type=int, default=17)             # run_EHRM.py:21
    ap.add_argument("--l2", type=float, default=0.01)           # :29
    ap.add_argument("--B", type=float, default=-5.0)            # :33
    ap.add_argument("--group-corr", type=float, default=0.3, help="P(group = 1 | y = +1) - P(group = 1 | y = -1)")
    ap.add_argument("--max-iter", type=int, default=200)
    ap.add_argument("--out", default=None)
    ap.add_argument("--quiet", action="store_true")
    a = ap.parse_args(argv)

    import numpy as np
    from sklearn.datasets import make_classification
    from sklearn import preprocessing
    from admm_for_rank_based_loss_amd import ADMMmethod
    from admm_for_rank_based_loss_amd.src.util.calculate_acc import calculate_accuracy
    from admm_for_rank_based_loss_amd.src.util.fair_metric import calculate_statistics

    X, label = make_classification(n_samples=a.rows, n_features=a.cols, n_classes=2, random_state=a.seed)
    label[label == 0] = -1
    rng = np.random.RandomState(a.seed + 1)
    p1 = 0.5 + 0.5 * a.group_corr * label                       # P(group = 1 | y)
    group = (rng.uniform(size=a.rows) < p1).astype(np.int64)
    label = label.reshape((-1, 1))
    X = preprocessing.scale(X)
    X_train, X_test, y_train, y_test, g_train, g_test = train_test_split_group(X, label, group, 0.4, a.seed)   # :24

    wf, loss = "ehrm", "binary_cross_entropy"                                                                  # :26-27
    admm = ADMMmethod(X_train, y_train, wf, loss, l2_reg=a.l2, l1_reg=None, B=a.B, args=None, max_iter=a.max_iter)   # :36
    admm.start_store(X_test, y_test, wf, loss, l2_reg=a.l2, l1_reg=None, args=None)                            # :37
    admm.main_loop(verbose=not a.quiet)                                                                        # :38
    w, times, train_losses, test_losses = admm.final_res()                                                     # :39
    acc = calculate_accuracy(w.reshape(-1, 1), X_test, y_test, threshold=0.5, loss=loss)                       # :40
    SPD, DI, EOD, AOD, TI, FNRD = calculate_statistics(w.reshape(-1, 1), X_test, y_test, g_test, threshold=0.5)   # :41
    print("admm train loss:", train_losses[-1])
    print("admm test loss:", test_losses[-1])
    print("admm time:", times[-1])
    print("admm test acc:", acc)
    print("admm SPD:", SPD, "admm DI:", DI, "admm EOD:", EOD, "admm AOD:", AOD, "admm TI:", TI, "admm FNRD:", FNRD)
    rows = [train_losses, times, [acc, SPD, DI, EOD, AOD, TI, FNRD]]                                           # :93-95
    if a.out:
        with open(a.out, "w", newline="") as f:
            csv.writer(f).writerows(rows)
        print("rows written to", a.out)
    return dict(rows=rows, w=w, X_test=X_test, y_test=y_test, g_test=g_test, X_train=X_train, y_train=y_train)


if __name__ == "__main__":
    main()
