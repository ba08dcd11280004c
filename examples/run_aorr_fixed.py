#!/usr/bin/env python3
"""The ADMM part of the reference driver ``run_AoRR_fixed.py`` on the GPU package.

Same call sequence as the reference (run_AoRR_fixed.py:82-156): a per-class random split into
train / validation / test (``sample_per_class`` with ``RandomState(seed)``, 50 % / 25 % / 25 %), labels
mapped to +-1, an intercept column appended, ``ADMMmethod`` with the ``aorr_dc`` weights and the FIXED
ranks ``args = [k, m]`` (sum of the losses ranked m+1 .. k, objective.py:139-145), ``start_store`` on the test
split with ``args = [1, 0]`` (:153), ``main_loop`` / ``final_res`` and the test accuracy.  What the driver hands
to its DCA baseline afterwards (:188-190: ``objective.alphas``, ``reg``, the two ``get_arrogate_loss``
callbacks) is returned and printed so that the hand-over is exercised; the baselines themselves (SGD, LSVRG,
DCA) are out of scope.

Data: the reference reads ``dataset/<name>.csv`` (features, label in the last column); ``--csv PATH`` does the
same for any such file.  Without it a synthetic problem of the same shape as ``australian`` (690 x 14) is
generated (make_classification + preprocessing.scale) - the reference's data files do not travel with this
package.

    python examples/run_aorr_fixed.py [--csv dataset/australian.csv] [--k 80] [--m 3] [--loss hinge] [--out rows.csv]
"""
import argparse
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def sample_per_class(random_state, labels, size_ratio, forbidden_indices=None):
    """run_AoRR_fixed.py:22-39: int(ratio * class size) indices of every class, drawn without replacement"""
    import numpy as np
    forbidden = set() if forbidden_indices is None else set(int(i) for i in forbidden_indices)
    out = []
    for c in range(len(np.unique(labels))):
        idx = [i for i in range(len(labels)) if labels[i] == c and i not in forbidden]
        out.append(random_state.choice(idx, int(len(idx) * size_ratio), replace=False))
    return np.concatenate(out)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--csv", default=None, help="features..., label (0/1 or -1/1) in the last column, no header")
    ap.add_argument("--rows", type=int, default=690)             # australian
    ap.add_argument("--cols", type=int, default=14)
    ap.add_argument("--seed", type=int, default=17)              # run_AoRR_fixed.py:43
    ap.add_argument("--loss", default="hinge")                   # :107
    ap.add_argument("--l2", type=float, default=0.0001)          # :108
    ap.add_argument("--k", type=int, default=80)                 # kvalue (australian, :131)
    ap.add_argument("--m", type=int, default=3)                  # kvalue2 (:132)
    ap.add_argument("--max-iter", type=int, default=200)
    ap.add_argument("--storage", default="f64")                  # non-convex family: the reference's fp64 D by default
    ap.add_argument("--out", default=None)
    ap.add_argument("--quiet", action="store_true")
    a = ap.parse_args(argv)

    import numpy as np
    import torch
    from sklearn import preprocessing
    from admm_for_rank_based_loss_amd import ADMMmethod
    from admm_for_rank_based_loss_amd.src.util.calculate_acc import calculate_accuracy

    if a.csv:
        raw = np.loadtxt(a.csv, delimiter=",")
        data, label = raw[:, :-1].astype(float), raw[:, -1].astype(np.int32)
        label[label == -1] = 0                                   # :83
    else:
        from sklearn.datasets import make_classification
        data, label = make_classification(n_samples=a.rows, n_features=a.cols, n_classes=2, random_state=a.seed)
        label = label.astype(np.int32)
    data = preprocessing.scale(data)                             # :81
    rs = np.random.RandomState(a.seed)                           # :85
    train_idx = sample_per_class(rs, label, 0.5)                                           # :88
    val_idx = sample_per_class(rs, label, 0.25 * 2, forbidden_indices=train_idx)           # :89
    test_idx = np.setdiff1d(np.arange(len(label)), np.concatenate((train_idx, val_idx)))   # :90-91
    X_train, X_val, X_test = data[train_idx], data[val_idx], data[test_idx]
    y_train, y_val, y_test = (np.where(label[i] == 0, -1, 1).reshape(-1, 1) for i in (train_idx, val_idx, test_idx))
    X_train_other = np.hstack((X_train, np.ones((X_train.shape[0], 1))))                   # :149
    X_test_other = np.hstack((X_test, np.ones((X_test.shape[0], 1))))                      # :150

    wf, args = "aorr_dc", [a.k, a.m]                                                       # :106, :146
    admm = ADMMmethod(X_train_other, y_train, wf, a.loss, l2_reg=a.l2, l1_reg=None, args=args, max_iter=a.max_iter,
                      storage=a.storage)                                                   # :152
    admm.start_store(X_test_other, y_test, wf, a.loss, l2_reg=a.l2, l1_reg=None, args=[1, 0])     # :153
    admm.main_loop(verbose=not a.quiet)                                                    # :154
    w, times, train_losses, test_losses = admm.final_res()                                 # :155
    acc = calculate_accuracy(w.reshape(-1, 1), X_test_other, y_test, threshold=0.5, loss=a.loss)   # :156

    sigma = admm.objective.alphas.numpy().reshape(-1)                                      # handed to DCAmethod, :188
    w_t = torch.as_tensor(w.reshape(-1, 1), dtype=torch.float64)
    print("admm train loss:", train_losses[-1])
    print("admm time:", times[-1])
    print("admm test acc:", acc)
    print("sigma: %d of %d ranks weighted (aorr_dc k=%d, m=%d), reg = %g" % (int((sigma > 0).sum()), sigma.size, a.k, a.m,
                                                                              admm.reg))
    print("callbacks: train %.12g  test %.12g" % (admm.objective.get_arrogate_loss(w_t),
                                                  admm.test_objective.get_arrogate_loss(w_t)))
    rows = [train_losses, times, [acc]]
    if a.out:
        with open(a.out, "w", newline="") as f:
            csv.writer(f).writerows(rows)
        print("rows written to", a.out)
    return dict(rows=rows, w=w, sigma=sigma, reg=admm.reg, X_train=X_train_other, y_train=y_train, X_val=X_val, y_val=y_val,
                X_test=X_test_other, y_test=y_test, train_cb=admm.objective.get_arrogate_loss(w_t),
                test_cb=admm.test_objective.get_arrogate_loss(w_t))


if __name__ == "__main__":
    main()
