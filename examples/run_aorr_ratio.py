#!/usr/bin/env python3
"""The ADMM part of the reference driver ``run_AoRR_ratio.py`` on the GPU package.

Same call sequence as the reference (run_AoRR_ratio.py:22-47): synthetic data
(make_classification + preprocessing.scale, load_data.py:101-116), a 50/25/25
train/validation/test split, an intercept column appended, ``ADMMmethod`` with the ``aorr``
weights and ``args = [k1/n, k2/n]``, ``start_store`` / ``main_loop`` / ``final_res`` and the test
accuracy.  The rows run_AoRR_ratio.py:106-108 would write to the xlsx (train losses, cumulative
times, [test accuracy]) go to a CSV in the same order.  The competitor baselines of that driver
(SGD, LSVRG, DCA) are out of scope; what they are handed from the solver (``objective.alphas``,
``reg``, the two ``get_arrogate_loss`` callbacks, run_AoRR_ratio.py:69-72) is printed so the
hand-over is exercised.

    python examples/run_aorr_ratio.py [--rows 1000] [--cols 1000] [--loss hinge] [--l2 1e-4]
                                      [--args 0.2 0.8] [--max-iter 200] [--out rows.csv]
"""
import argparse
import csv
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1000)           # run_AoRR_ratio.py:22
    ap.add_argument("--cols", type=int, default=1000)           # :23
    ap.add_argument("--seed", type=int, default=17)             # :25
    ap.add_argument("--loss", default="hinge")                  # :33
    ap.add_argument("--l2", type=float, default=0.0001)         # :34
    ap.add_argument("--args", type=float, nargs=2, default=[0.2, 0.8])   # :37
    ap.add_argument("--max-iter", type=int, default=200)
    ap.add_argument("--out", default=None)
    ap.add_argument("--quiet", action="store_true")
    a = ap.parse_args(argv)

    import numpy as np
    import torch
    from sklearn.datasets import make_classification
    from sklearn import preprocessing
    from sklearn.model_selection import train_test_split
    from admm_for_rank_based_loss_amd import ADMMmethod
    from admm_for_rank_based_loss_amd.src.util.calculate_acc import calculate_accuracy

    X, label = make_classification(n_samples=a.rows, n_features=a.cols, n_classes=2, random_state=a.seed)
    label[label == 0] = -1
    label = label.reshape((-1, 1))
    X = preprocessing.scale(X)
    X_train, X_test, y_train, y_test = train_test_split(X, label, test_size=0.5, random_state=a.seed)      # :28
    X_val, X_test, y_val, y_test = train_test_split(X_test, y_test, test_size=0.5, random_state=a.seed)    # :29
    X_train_other = np.hstack((X_train, np.ones((X_train.shape[0], 1))))                                   # :40
    X_test_other = np.hstack((X_test, np.ones((X_test.shape[0], 1))))                                      # :41

    wf = "aorr"
    kw = dict(l2_reg=a.l2, l1_reg=None, args=list(a.args))
    admm = ADMMmethod(X_train_other, y_train, wf, a.loss, max_iter=a.max_iter, **kw)                       # :43
    admm.start_store(X_test_other, y_test, wf, a.loss, **kw)                                               # :44
    admm.main_loop(verbose=not a.quiet)                                                                    # :45
    w, times, train_losses, test_losses = admm.final_res()                                                 # :46
    acc = calculate_accuracy(w.reshape(-1, 1), X_test_other, y_test, threshold=0.5, loss=a.loss)          # :47

    # what the DCA baseline is handed (run_AoRR_ratio.py:69-72)
    sigma = admm.objective.alphas.numpy().reshape(-1)
    k_hi, k_lo = math.floor(X_train.shape[0] * a.args[1]), math.ceil(X_train.shape[0] * a.args[0])
    w_t = torch.as_tensor(w.reshape(-1, 1), dtype=torch.float64)
    print("admm train loss:", train_losses[-1])
    print("admm test loss:", test_losses[-1])
    print("admm time:", times[-1])
    print("admm test acc:", acc)
    print("sigma: %d of %d ranks weighted (k in [%d, %d)), reg = %g" % (int((sigma > 0).sum()), sigma.size, k_lo, k_hi,
                                                                        admm.reg))
    print("callbacks: train %.12g  test %.12g" % (admm.objective.get_arrogate_loss(w_t),
                                                  admm.test_objective.get_arrogate_loss(w_t)))
    rows = [train_losses, times, [acc]]                                                                    # :106-108
    if a.out:
        with open(a.out, "w", newline="") as f:
            csv.writer(f).writerows(rows)
        print("rows written to", a.out)
    return dict(rows=rows, w=w, sigma=sigma, train_cb=admm.objective.get_arrogate_loss(w_t),
                test_cb=admm.test_objective.get_arrogate_loss(w_t))


if __name__ == "__main__":
    main()
