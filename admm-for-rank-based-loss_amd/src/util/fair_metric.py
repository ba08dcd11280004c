"""Drop-in mirror of the reference's ``src/util/fair_metric.py``: the group
fairness statistics ``run_EHRM.py:41`` prints after a solve, evaluated on the GPU (one sweep
v = D w, group-wise confusion counts and the Theil-index sums in one reduction kernel)."""
import numpy as np

try:
    from ... import _solver
except ImportError:      # package directory on sys.path: imported as ``src.util.fair_metric``
    import _solver


def calculate_statistics(w, X_test, label_test, group_test, threshold=0.5):
    """Returns (SPD, DI, EOD, AOD, TI, FNRD) as reference fair_metric.py:3-41 (group 0 = G1,
    group 1 = G2; predictions from sigmoid(x.w) >= threshold)."""
    X = _solver._as_matrix(X_test)
    s = _solver.Solver(X.shape[0], X.shape[1], "erm", "binary_cross_entropy", objective_only=True)
    try:
        s.set_data(X, label_test)
        return s.fair_statistics(np.asarray(w, dtype=np.float64).reshape(-1),
                                 np.asarray(group_test, dtype=np.float64).reshape(-1), threshold)
    finally:
        s.close()
