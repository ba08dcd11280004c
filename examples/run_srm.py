#!/usr/bin/env python3
"""The ADMM / sADMM part of the reference driver ``run_SRM.py`` on the GPU package.

Same call sequence as the reference (run_SRM.py:21-49): synthetic data from sklearn's
make_classification + preprocessing.scale + train_test_split, ``ADMMmethod`` and
``smoothADMMmethod`` with ``start_store`` / ``main_loop`` / ``final_res``, test accuracy by
``calculate_accuracy``.  Instead of the xlsx (openpyxl) the rows run_SRM.py:132-139 would write
(train losses, cumulative times, test accuracy - ADMM then sADMM) go to a CSV with the same
row order, so they can be laid beside ``table/erm_synthetic_6000x1000_l1_binary_cross_entropy.xlsx``.
``--baselines N`` also runs the driver's three competitor calls (run_SRM.py:52-69: SGD, LSVRG with non-uniform and
with uniform sampling, fed with the ADMM solver's two objective callbacks) for N epochs each on the device
mirrors ``SGD_solver.SGDmethod`` / ``LSVRG_solver.LSVRGmethod`` and appends their rows in the sheet's order (:143-153).

    python examples/run_srm.py [--rows 10000] [--cols 1000] [--weight erm] [--loss binary_cross_entropy]
                               [--l1 0.01 | --l2 0.01] [--args 0.5] [--baselines 0] [--out rows.csv]
"""
import argparse
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10000)          # run_SRM.py:21
    ap.add_argument("--cols", type=int, default=1000)           # run_SRM.py:22
    ap.add_argument("--seed", type=int, default=17)             # run_SRM.py:24
    ap.add_argument("--weight", default="erm")
    ap.add_argument("--loss", default="binary_cross_entropy")
    ap.add_argument("--l1", type=float, default=None)
    ap.add_argument("--l2", type=float, default=None)
    ap.add_argument("--args", type=float, nargs="*", default=[0.2, 0.8])
    ap.add_argument("--out", default=None)
    ap.add_argument("--baselines", type=int, default=0, help="epochs of SGD / LSVRG (the reference runs 1000); 0 = skip")
    ap.add_argument("--quiet", action="store_true")
    a = ap.parse_args()
    if a.l1 is None and a.l2 is None:
        a.l1 = 0.01                                             # run_SRM.py:33

    from sklearn.datasets import make_classification
    from sklearn import preprocessing
    from sklearn.model_selection import train_test_split
    from admm_for_rank_based_loss_amd import ADMMmethod, smoothADMMmethod, SGDmethod, LSVRGmethod
    from admm_for_rank_based_loss_amd.src.util.calculate_acc import calculate_accuracy

    X, label = make_classification(n_samples=a.rows, n_features=a.cols, n_classes=2, random_state=a.seed)
    label[label == 0] = -1
    label = label.reshape((-1, 1))
    X = preprocessing.scale(X)                                  # src/util/load_data.py:105-116
    X_train, X_test, y_train, y_test = train_test_split(X, label, test_size=0.4, random_state=a.seed)

    kw = dict(l2_reg=a.l2, l1_reg=a.l1, args=a.args)
    verbose = not a.quiet
    admm = ADMMmethod(X_train, y_train, a.weight, a.loss, **kw)                       # run_SRM.py:39
    admm.start_store(X_test, y_test, a.weight, a.loss, **kw)                         # :40
    admm.main_loop(verbose=verbose)                                                  # :41
    w, times, train_losses, test_losses = admm.final_res()                           # :42
    acc = calculate_accuracy(w.reshape(-1, 1), X_test, y_test, threshold=0.5, loss=a.loss)   # :43
    rows = [train_losses, times, [acc]]
    print("admm train loss:", train_losses[-1])
    print("admm test loss:", test_losses[-1])
    print("admm time:", times[-1])
    print("admm test acc:", acc)
    if a.l1 is not None:                                                             # :45-50
        sadmm = smoothADMMmethod(X_train, y_train, a.weight, a.loss, **kw)
        sadmm.start_store(X_test, y_test, a.weight, a.loss, **kw)
        sadmm.main_loop(verbose=verbose)
        sw, stimes, strain, stest = sadmm.final_res()
        sacc = calculate_accuracy(sw.reshape(-1, 1), X_test, y_test, threshold=0.5, loss=a.loss)
        rows += [strain, stimes, [sacc]]
        print("sadmm train loss:", strain[-1])
        print("sadmm test loss:", stest[-1])
        print("sadmm time:", stimes[-1])
        print("sadmm test acc:", sacc)
    if a.baselines > 0:
        # run_SRM.py:52-53: the baselines take l2_reg * n and no l1 term; their losses are logged through the ADMM
        # solver's objectives (:57, :63, :69)
        l2_b = a.l2 * X_train.shape[0] if a.l2 is not None else None
        cb = dict(train_loss=admm.objective.get_arrogate_loss, test_loss=admm.test_objective.get_arrogate_loss,
                  verbose=verbose, args=a.args)
        for name, fn, extra in (("sgd", SGDmethod, dict(batch_size=64, lr=1e-5)),                  # :55-58
                                ("lsvrg_nu", LSVRGmethod, dict(lr=1, uniform=None)),               # :61-64
                                ("lsvrg_u", LSVRGmethod, dict(lr=1, uniform=True))):               # :67-70
            bw, btrain, btest, btimes = fn(X_train, y_train, a.weight, a.loss, l2_reg=l2_b, l1_reg=None,
                                           max_iter=a.baselines, **extra, **cb)
            bacc = calculate_accuracy(bw.reshape(-1, 1), X_test, y_test, threshold=0.5, loss=a.loss)
            rows += [btrain, btimes, [bacc]]
            print(name, "train loss:", btrain[-1], "time:", btimes[-1], "test acc:", bacc)
    if a.out:
        with open(a.out, "w", newline="") as f:
            csv.writer(f).writerows(rows)
        print("rows written to", a.out)
    return rows


if __name__ == "__main__":
    main()
