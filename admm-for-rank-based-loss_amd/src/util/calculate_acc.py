"""Drop-in mirror of the reference's ``src/util/calculate_acc.py``: the
test accuracy the drivers print after a solve (``run_SRM.py:43``), evaluated on the GPU from
one sweep v = D w."""
import numpy as np

try:
    from ... import _solver
except ImportError:      # package directory on sys.path: imported as ``src.util.calculate_acc``
    import _solver


def calculate_accuracy(w, X_test, y_test, threshold=0.5, loss='binary_cross_entropy'):
    """Fraction of rows with prediction == label (reference calculate_acc.py:3-19).
    binary_cross_entropy: predict +1 iff sigmoid(x.w) >= threshold.  hinge: the reference sets
    every prediction to +1 (calculate_acc.py:13-15); mirrored as is."""
    if loss not in ('binary_cross_entropy', 'hinge'):
        raise ValueError(f"loss '{loss}' is not supported! Options: ['binary_cross_entropy','hinge']")
    X = _solver._as_matrix(X_test)
    s = _solver.Solver(X.shape[0], X.shape[1], "erm", loss, objective_only=True)
    try:
        s.set_data(X, y_test)
        return s.accuracy(np.asarray(w, dtype=np.float64).reshape(-1), threshold)
    finally:
        s.close()
