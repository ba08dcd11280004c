"""Synthetic problem generators owned by the build (NumPy only) plus the recipe of
the reference's one published configuration.  Test infrastructure only."""
import hashlib
import numpy as np


def make_problem(n, d, seed, flip=0.01, class_sep=1.0, intercept=False):
    """Column-standardised two-class problem with the statistics of the reference's
    synthetic branch (src/util/load_data.py:101-116: sklearn make_classification
    defaults = 2 informative + 2 redundant columns, the rest N(0,1) noise, 1 % label
    flips, then preprocessing.scale).  Not bit-identical to sklearn - it only has to
    be reproducible from (n, d, seed).  Returns X (n,d) float64, y (n,1) int64 +-1."""
    rng = np.random.default_rng(seed)
    ninf = min(2, d)
    y01 = rng.integers(0, 2, size=n)
    X = rng.standard_normal((n, d))
    centers = class_sep * (2.0 * rng.integers(0, 2, size=(4, ninf)) - 1.0)
    cluster = 2 * y01 + rng.integers(0, 2, size=n)
    X[:, :ninf] += centers[cluster]
    if d >= 4:
        mix = 2.0 * rng.random((ninf, 2)) - 1.0
        X[:, 2:4] = X[:, :ninf] @ mix
    flips = rng.random(n) < flip
    y01 = np.where(flips, 1 - y01, y01)
    X = X[:, rng.permutation(d)]
    X = (X - X.mean(axis=0)) / X.std(axis=0)
    if intercept:
        X = np.hstack([X, np.ones((n, 1))])          # run_AoRR_ratio.py:40
    y = (2 * y01 - 1).astype(np.int64).reshape(-1, 1)
    return np.ascontiguousarray(X), y


def sha256_of(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def c1_data():
    """BASELINE config C1 exactly as the reference builds it: run_SRM.py:21-28 with
    src/util/load_data.py:101-116 (make_classification(10000, 1000, random_state=17),
    labels 0 -> -1, preprocessing.scale, train_test_split(test_size=0.4,
    random_state=17)) -> X_train 6000x1000.  Needs scikit-learn (a third-party
    dependency of the reference, present in the image)."""
    from sklearn.datasets import make_classification
    from sklearn import preprocessing
    from sklearn.model_selection import train_test_split
    X, label = make_classification(n_samples=10000, n_features=1000, n_classes=2, random_state=17)
    label[label == 0] = -1
    label = label.reshape((-1, 1))
    X = preprocessing.scale(X)
    return train_test_split(X, label, test_size=0.4, random_state=17)
