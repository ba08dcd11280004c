"""The competitor baselines on the GPU (include/rbl.h: rbl_bl_*; mirrors SGD_solver.py /
LSVRG_solver.py of the package) through the reference's own function signatures:
* against the golden vectors of the REAL reference (tests/golden/g11_baselines.npz): w after every epoch;
* against the CPU oracle (oracle/baselines.py) on larger seeded problems (d not a multiple of 4, mini-batches that do
  not fill a wave, EHRM, l1 with its random signs), epoch by epoch.
Tolerance 1e-10 relative: the device sums a mini-batch's rows in a fixed order, BLAS in another."""
import json

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R():
    import admm_for_rank_based_loss_amd as rbl
    if rbl._lib.device_count() < 1:
        pytest.fail("no HIP device: the GPU tests must run the HIP library (no fallback)")
    return rbl


@pytest.mark.parametrize("k", range(13))
def test_baselines_vs_reference_goldens(R, k):
    import torch
    g = load_golden("g11_baselines.npz")
    cfg = json.loads(str(g[f"c{k}_cfg"]))
    algo, np_seed, torch_seed = cfg.pop("algo"), cfg.pop("np_seed"), cfg.pop("torch_seed")
    np.random.seed(np_seed)
    torch.manual_seed(torch_seed)
    ws = []
    fn = R.SGDmethod if algo == "sgd" else R.LSVRGmethod
    out = fn(g["X"], g["y"], train_loss=lambda w: ws.append(w.numpy().reshape(-1).copy()) or 0.0, test_loss=lambda w: 0.0,
             verbose=False, **cfg)
    ref = g[f"c{k}_w"]
    w, train_losses, test_losses, t_array = out
    assert len(train_losses) == len(test_losses) == len(t_array) == cfg["max_iter"] + 1 and t_array[0] == 0
    assert w.shape == (g["X"].shape[1], 1) and np.array_equal(w.reshape(-1), ws[-1])
    assert np.max(np.abs(np.array(ws) - ref)) <= 1e-10 * max(1.0, np.max(np.abs(ref))), (algo, cfg)


@pytest.mark.parametrize("algo,n,d,kw", [
    ("sgd", 5000, 37, dict(weight_function="superquantile", loss="binary_cross_entropy", l2_reg=1.0, lr=0.05, max_iter=4,
                           args=[0.5], batch_size=50)),
    ("sgd", 3001, 130, dict(weight_function="ehrm", loss="binary_cross_entropy", l2_reg=1.0, lr=0.02, max_iter=3, lossB=0.69)),
    ("sgd", 2000, 9, dict(weight_function="aorr", loss="hinge", l1_reg=1.0, lr=0.05, max_iter=3, args=[0.2, 0.8], batch_size=256)),
    ("sgd", 6000, 21, dict(weight_function="superquantile", loss="binary_cross_entropy", l2_reg=1.0, lr=0.05, max_iter=3,
                           args=[0.5], batch_size=1000)),      # (mini-batches up to 1024 rows: one thread of the workgroup per row)
    ("lsvrg", 20000, 64, dict(weight_function="extremile", loss="binary_cross_entropy", l2_reg=1.0, lr=0.01, max_iter=3,
                              args=[2.0], uniform=None)),
    ("lsvrg", 7001, 21, dict(weight_function="ehrm", loss="binary_cross_entropy", l1_reg=1.0, lr=0.01, max_iter=3, lossB=0.69,
                             uniform=True)),
    ("lsvrg", 4000, 1001, dict(weight_function="aorr_dc", loss="hinge", l2_reg=1.0, lr=0.005, max_iter=2, args=[900, 40],
                               uniform=None)),
], ids=["sgd_superq_b50", "sgd_ehrm_d130", "sgd_aorr_hinge_l1_b256", "sgd_superq_b1000", "lsvrg_extremile_20000", "lsvrg_ehrm_l1_uniform",
        "lsvrg_aorr_dc_hinge_d1001"])
def test_baselines_vs_oracle(R, algo, n, d, kw):
    import torch
    from oracle import problems, baselines
    X, y = problems.make_problem(n, d, seed=400 + d)
    runs = []
    for fn in ((baselines.sgd_solve if algo == "sgd" else baselines.lsvrg_solve), (R.SGDmethod if algo == "sgd" else R.LSVRGmethod)):
        np.random.seed(77)
        torch.manual_seed(78)
        ws = []
        if fn.__module__.startswith("oracle"):
            fn(X, y, log=lambda w: ws.append(np.array(w).copy()), **kw)
        else:
            fn(X, y, train_loss=lambda w: ws.append(w.numpy().reshape(-1).copy()) or 0.0, test_loss=lambda w: 0.0,
               verbose=False, **kw)
        runs.append(np.array(ws))
    ref, got = runs
    assert ref.shape == got.shape == (kw["max_iter"] + 1, d) and np.max(np.abs(ref[-1])) > 0
    assert np.max(np.abs(got - ref)) <= 1e-10 * max(1.0, np.max(np.abs(ref))), float(np.max(np.abs(got - ref)))


def test_baselines_api_errors(R):
    from oracle import problems
    X, y = problems.make_problem(200, 5, seed=1)
    with pytest.raises(ValueError, match="args for framework is None"):
        R.SGDmethod(X, y, "superquantile", "binary_cross_entropy", l2_reg=1.0, test_loss=lambda w: 0.0, verbose=False)
    with pytest.raises(ValueError, match="not supported"):
        R.SGDmethod(X, y, "nope", "binary_cross_entropy", l2_reg=1.0, test_loss=lambda w: 0.0, verbose=False, args=[1])
    with pytest.raises(ValueError, match="Unrecognized loss"):
        R.LSVRGmethod(X, y, "erm", "square", l2_reg=1.0, test_loss=lambda w: 0.0, verbose=False)
    # train_loss=None: only w comes back (SGD_solver.py:93-96)
    w = R.SGDmethod(X, y, "erm", "hinge", l2_reg=1.0, max_iter=1, test_loss=lambda w: 0.0, verbose=False)
    assert isinstance(w, np.ndarray) and w.shape == (5, 1)
