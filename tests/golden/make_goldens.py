#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by importing the REAL reference.

Run in the build container only (needs /root/reference; it never travels):
    python tests/golden/make_goldens.py [group ...]
Groups: g1 g2 g3 g4 g56 g7 g8 g9 c1 table g11   (default: all)

The reference snapshot has a return-shape bug (src/optim/algorithms.py:97,101 unpack
two values, src/util/pav.py:178 and src/util/PAV_cpt.py:293 return one); the two-line
wrapper below is the only deviation from the shipped code.  Fixtures hold DATA only
(inputs + the reference's outputs); inputs come from oracle/problems.py generators.
"""
import io
import json
import os
import sys
import time
import zipfile
import contextlib
import xml.etree.ElementTree as ET

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(1, REF)

import torch  # noqa: E402
import src.util.pav as ref_pav  # noqa: E402
import src.util.PAV_cpt as ref_cpt  # noqa: E402
import src.util.individual_solver as ref_ind  # noqa: E402
import src.util.fast_lasso as ref_fista  # noqa: E402
import src.util.w_LBFGS as ref_wl  # noqa: E402
import src.optim.objective as ref_obj  # noqa: E402


def _wrap(cls):
    orig = cls.get_opt
    cls.get_opt = lambda self, *a, **k: (orig(self, *a, **k), None)


_wrap(ref_pav.PAV_solver)
_wrap(ref_cpt.PAV_solver_CPT)
from src.optim.algorithms import ADMMmethod, smoothADMMmethod  # noqa: E402

from oracle import problems  # noqa: E402

FAMILIES = [("erm", None), ("superquantile", [0.5]), ("extremile", [2.0]), ("esrm", [1.0]),
            ("aorr", [0.2, 0.8]), ("aorr_dc", [80, 3])]


def ref_weights(name, n, args):
    wf = ref_obj.get_weights(name, args)
    if isinstance(wf, tuple):
        return wf[0](n).numpy().copy(), wf[1](n).numpy().copy()
    a = wf(n).numpy().copy()
    return a, a


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print(f"  wrote {name}  ({os.path.getsize(path)/1024:.1f} KiB)")


def sorted_m(rng, n, scale=2.0, shift=0.0):
    return np.sort(scale * rng.standard_normal(n) + shift)


# ------------------------------------------------------------------ G1 element prox
def g1():
    rng = np.random.default_rng(101)
    out, k = {}, 0
    for loss in ("binary_cross_entropy", "hinge"):
        for rho in (2e-7, 1e-5, 1e-3, 1.0):
            for n in (1, 2, 7, 64, 1000):
                sigma = rng.random(n) / n * (2.0 if n > 1 else 1.0)
                if n >= 7:
                    sigma[rng.integers(0, n, size=max(1, n // 5))] = 0.0
                m = 3.0 * rng.standard_normal(n)
                x = ref_ind.individual_solver(loss, sigma.copy(), rho, m.copy())
                out[f"c{k}_sigma"], out[f"c{k}_m"], out[f"c{k}_x"] = sigma, m, np.asarray(x)
                out[f"c{k}_meta"] = np.array([rho, 0 if loss == "binary_cross_entropy" else 1])
                k += 1
    out["ncases"] = np.array(k)
    save("g1_prox.npz", **out)


# ------------------------------------------------------------------------ G2 PAV
def g2():
    rng = np.random.default_rng(202)
    out, k = {}, 0
    for fam, args in FAMILIES:
        for loss in ("binary_cross_entropy", "hinge"):
            for rho in (2e-7, 1e-5, 1e-3, 1.0):
                n = 600 if rho < 1e-4 else 1500
                if fam == "aorr_dc":
                    n = max(n, 200)
                sa, sb = ref_weights(fam, n, args)
                m = sorted_m(rng, n, 2.0, -0.5)
                t0 = time.time()
                solver = ref_pav.PAV_solver(sa.copy(), m.copy(), rho, loss, None, sb.copy())
                res, _ = solver.get_opt(maxiter=n)
                out[f"c{k}_sigma"], out[f"c{k}_m"], out[f"c{k}_u"] = sa, m, np.asarray(res)
                out[f"c{k}_meta"] = np.array([rho, 0 if loss == "binary_cross_entropy" else 1])
                out[f"c{k}_name"] = np.array(f"{fam}/{loss}/rho={rho}")
                print(f"   g2 case {k}: {fam}/{loss}/rho={rho} n={n}  {time.time()-t0:.1f}s")
                k += 1
    out["ncases"] = np.array(k)
    save("g2_pav.npz", **out)


# -------------------------------------------------------------------- G3 PAV (CPT)
def g3():
    rng = np.random.default_rng(303)
    out, k = {}, 0
    B = -5.0
    for n in (50, 400, 1200):
        sa, sb = ref_weights("ehrm", n, None)
        for rho in (1e-4, 1e-3, 1e-1):
            for shift in (-9.0, -5.0, -2.0, 1.0):
                m = sorted_m(rng, n, 1.5, shift)
                solver = ref_cpt.PAV_solver_CPT(sa.copy(), sb.copy(), B, m.copy(), rho)
                first_pick = None
                res, _ = solver.get_opt()
                res = np.asarray(res)
                out[f"c{k}_sa"], out[f"c{k}_sb"], out[f"c{k}_m"], out[f"c{k}_u"] = sa, sb, m, res
                out[f"c{k}_meta"] = np.array([rho, B, shift])
                k += 1
    out["ncases"] = np.array(k)
    save("g3_pav_cpt.npz", **out)


# ----------------------------------------------------------------------- G4 z-step
def g4():
    out, k = {}, 0
    for fam, args, loss, kw in [("superquantile", [0.5], "binary_cross_entropy", dict(l2_reg=0.01)),
                                ("extremile", [2.0], "hinge", dict(l2_reg=0.01)),
                                ("aorr", [0.2, 0.8], "binary_cross_entropy", dict(l2_reg=1e-4)),
                                ("erm", None, "binary_cross_entropy", dict(l1_reg=0.01)),
                                ("ehrm", None, "binary_cross_entropy", dict(l2_reg=0.01, B=-5))]:
        X, y = problems.make_problem(500, 12, seed=40 + k)
        s = ADMMmethod(X, y, fam, loss, args=args, **kw)
        rng = np.random.default_rng(400 + k)
        s.w = 0.3 * rng.standard_normal((12, 1))
        s.lagrangian = 1e-4 * rng.standard_normal((500, 1))
        s.rho = 1e-3
        z = s._z_subproblem()
        out[f"c{k}_X"], out[f"c{k}_y"] = X, y
        out[f"c{k}_w"], out[f"c{k}_lam"], out[f"c{k}_z"] = s.w.copy(), s.lagrangian.copy(), z
        out[f"c{k}_name"] = np.array(json.dumps(dict(weight_function=fam, loss=loss, args=args, rho=1e-3, **kw)))
        k += 1
    out["ncases"] = np.array(k)
    save("g4_zstep.npz", **out)


# ---------------------------------------------------------------- G5/G6 w-step
def g56():
    out = {}
    X, y = problems.make_problem(2000, 100, seed=56)
    D = -y * X
    G = D.T @ D
    rng = np.random.default_rng(56)
    z = rng.standard_normal((2000, 1))
    lam = 1e-3 * rng.standard_normal((2000, 1))
    rho, reg = 1e-3, 0.01
    w0 = 1e-6 * np.ones((100, 1))
    c = z + lam / rho
    w_fista = ref_fista.FISTA(beta=w0.reshape(-1), X=D, y=c.reshape(-1), lam=reg / (2 * rho),
                              L=np.float32(17), eta=np.float32(2.5), tol=7e-5, max_iter=5000,
                              dtype=torch.float32)
    w_l2 = ref_wl.w_solver(2, w0, z, lam, rho, G, D, reg)
    w_sm = ref_wl.w_solver(1, w0, z, lam, rho, G, D, reg, 0.5)
    out.update(X=X, y=y, z=z, lam=lam, w0=w0, meta=np.array([rho, reg, 0.5]),
               w_fista=np.asarray(w_fista, dtype=np.float64), w_l2=w_l2, w_smooth=w_sm)
    save("g56_wstep.npz", **out)


# ------------------------------------------------------------------ G7 objective
def g7():
    out, k = {}, 0
    X, y = problems.make_problem(400, 9, seed=7)
    rng = np.random.default_rng(7)
    w = 0.5 * rng.standard_normal((9, 1))
    fams = FAMILIES + [("ehrm", None)]
    for fam, args in fams:
        for loss in ("binary_cross_entropy", "hinge"):
            if fam == "ehrm" and loss == "hinge":
                continue
            for reg in (dict(l2_reg=0.01), dict(l1_reg=0.01)):
                kw = dict(reg)
                if fam == "ehrm":
                    kw["B"] = -5
                o = ref_obj.rankbasedObjective(torch.from_numpy(X.copy()), torch.from_numpy(y.copy()),
                                               fam, loss, args=args, **kw)
                val = o.get_arrogate_loss(torch.from_numpy(w).double())
                out[f"c{k}_val"] = np.array(val)
                out[f"c{k}_name"] = np.array(json.dumps(dict(weight_function=fam, loss=loss, args=args, **kw)))
                k += 1
    out.update(X=X, y=y, w=w, ncases=np.array(k))
    save("g7_objective.npz", **out)


# -------------------------------------------------------------------- G8 weights
def g8():
    out, k = {}, 0
    for n in (10, 1000):
        for fam, args in FAMILIES + [("ehrm", None), ("superquantile", [0.3]), ("aorr", [0.1, 0.55]),
                                      ("extremile", [1.5]), ("esrm", [2.5])]:
            if fam == "aorr_dc" and n == 10:
                args = [7, 2]
            a, b = ref_weights(fam, n, args)
            out[f"c{k}_a"], out[f"c{k}_b"] = a, b
            out[f"c{k}_name"] = np.array(json.dumps(dict(weight_function=fam, n=n, args=args)))
            k += 1
    out["ncases"] = np.array(k)
    save("g8_weights.npz", **out)


# ---------------------------------------------------------------- G9 trajectories
# ------------------------------------------------------------------ G11 competitor baselines (SGD / LSVRG)
def g11():
    """SGDmethod (SGD_solver.py:9-96) and LSVRGmethod (LSVRG_solver.py:9-98) of the reference on small problems:
    w after every epoch.  The global numpy / torch generators the reference draws from without seeding them
    (np.random.choice in the non-uniform LSVRG, torch.rand for the l1 subgradient at 0) are seeded here, and the
    seeds are part of the fixture."""
    from SGD_solver import SGDmethod
    from LSVRG_solver import LSVRGmethod
    X, y = problems.make_problem(300, 10, seed=23)
    out = {"X": X, "y": y}
    cases = [
        ("sgd", dict(weight_function="erm", loss="binary_cross_entropy", l2_reg=0.5, lr=0.05, max_iter=3)),
        ("sgd", dict(weight_function="superquantile", loss="hinge", l2_reg=0.5, lr=0.05, max_iter=3, args=[0.5])),
        ("sgd", dict(weight_function="aorr", loss="binary_cross_entropy", l2_reg=0.5, lr=0.05, max_iter=3, args=[0.2, 0.8])),
        ("sgd", dict(weight_function="aorr_dc", loss="hinge", l2_reg=0.5, lr=0.05, max_iter=2, args=[40, 3])),
        ("sgd", dict(weight_function="ehrm", loss="binary_cross_entropy", l2_reg=0.5, lr=0.05, max_iter=3, lossB=0.65)),
        ("sgd", dict(weight_function="erm", loss="binary_cross_entropy", l1_reg=0.5, lr=0.05, max_iter=2)),
        ("sgd", dict(weight_function="extremile", loss="binary_cross_entropy", l2_reg=0.5, lr=1, max_iter=2, args=[2.0],
                     batch_size=32)),
        ("lsvrg", dict(weight_function="superquantile", loss="binary_cross_entropy", l2_reg=0.5, lr=0.01, max_iter=3,
                       args=[0.5], uniform=True)),
        ("lsvrg", dict(weight_function="superquantile", loss="binary_cross_entropy", l2_reg=0.5, lr=0.01, max_iter=3,
                       args=[0.5], uniform=None)),
        ("lsvrg", dict(weight_function="ehrm", loss="binary_cross_entropy", l2_reg=0.5, lr=0.01, max_iter=3, lossB=0.65,
                       uniform=True)),
        ("lsvrg", dict(weight_function="ehrm", loss="binary_cross_entropy", l2_reg=0.5, lr=0.01, max_iter=2, lossB=0.65,
                       uniform=None)),
        ("lsvrg", dict(weight_function="aorr", loss="hinge", l2_reg=0.5, lr=0.01, max_iter=2, args=[0.2, 0.8], uniform=None)),
        ("lsvrg", dict(weight_function="erm", loss="binary_cross_entropy", l1_reg=0.5, lr=0.01, max_iter=2, uniform=True)),
    ]
    for k, (algo, kw) in enumerate(cases):
        ws = []
        np.random.seed(1000 + k)          # the non-uniform LSVRG draws from the global numpy generator
        torch.manual_seed(2000 + k)       # torch.rand of the l1 subgradient (SGD re-seeds with 25 itself)
        fn = SGDmethod if algo == "sgd" else LSVRGmethod
        with contextlib.redirect_stdout(io.StringIO()):
            w, trl, tel, tarr = fn(X, y, train_loss=lambda w: ws.append(w.numpy().reshape(-1).copy()) or 0.0,
                                   test_loss=lambda w: 0.0, verbose=False, **kw)
        out[f"c{k}_cfg"] = json.dumps(dict(algo=algo, np_seed=1000 + k, torch_seed=2000 + k, **kw))
        out[f"c{k}_w"] = np.array(ws)                      # (max_iter + 1, d): w0 and w after every epoch
        print(f"  g11 case {k} {algo} {kw['weight_function']}/{kw['loss']}: |w_final| = {np.linalg.norm(w):.6f}")
    out["ncases"] = np.array(len(cases))
    save("g11_baselines.npz", **out)


def run_traj(X, y, cls, kw, max_iter=200):
    buf = io.StringIO()
    t0 = time.time()
    with contextlib.redirect_stdout(buf):
        s = cls(X, y, max_iter=max_iter, **kw)
        skw = {k: v for k, v in kw.items() if k != "B"}
        s.start_store(X, y, **skw)
        prim, dual, rhos = [], [], []
        # instrument one step at a time through the public per-iteration API
        import time as _t
        t_start = _t.time()
        conv = False
        from src.optim.algorithms import Optimizer
        for i in range(s.max_iter):
            rho_i = s.rho
            pre_w = s.w.copy()
            done = Optimizer.main_loop(s, i, t_start, False)
            prim.append(float(np.linalg.norm(s.z - s.D @ s.w)))
            dual.append(float(np.linalg.norm(s.w - pre_w)))
            rhos.append(float(rho_i))
            if done:
                conv = True
                break
            if cls is smoothADMMmethod and i >= 17:
                s.t = max(s.t * 0.9, 1e-9) % np.power(s.rho, -0.1) * np.power(i, -0.1)
        if cls is smoothADMMmethod and s.w_flag == 1:
            s.w = np.sign(s.w) * np.where((np.abs(s.w) - s.t) > 0, np.abs(s.w) - s.t, 0)
    final = s.objective.get_arrogate_loss(torch.from_numpy(np.asarray(s.w, dtype=np.float64)).double())
    return dict(primal=np.array(prim), dual=np.array(dual), rho=np.array(rhos),
                objective=np.array(s.train_losses), w=np.asarray(s.w, dtype=np.float64).reshape(-1),
                z=np.asarray(s.z).reshape(-1), lam=np.asarray(s.lagrangian).reshape(-1),
                iters=np.array(len(prim)), converged=np.array(conv), final_objective=np.array(final),
                wall=np.array(time.time() - t0), z_time=np.array(s.z_time[-1]), w_time=np.array(s.w_time[-1]))


TRAJ = [
    ("erm_bce_l1", 1500, 60, 11, ADMMmethod, dict(weight_function="erm", loss="binary_cross_entropy", l1_reg=0.01), 200),
    ("erm_bce_l2", 1500, 60, 12, ADMMmethod, dict(weight_function="erm", loss="binary_cross_entropy", l2_reg=0.01), 200),
    ("superq_bce_l2", 1500, 60, 13, ADMMmethod, dict(weight_function="superquantile", loss="binary_cross_entropy", l2_reg=0.01, args=[0.5]), 500),
    ("extremile_bce_l1", 1200, 40, 14, ADMMmethod, dict(weight_function="extremile", loss="binary_cross_entropy", l1_reg=0.01, args=[2.0]), 200),
    ("esrm_hinge_l2", 1000, 20, 15, ADMMmethod, dict(weight_function="esrm", loss="hinge", l2_reg=0.01, args=[1.0]), 400),
    ("superq_hinge_l2", 2000, 20, 16, ADMMmethod, dict(weight_function="superquantile", loss="hinge", l2_reg=0.01, args=[0.5]), 800),
    ("aorr_hinge_l2", 500, 21, 17, ADMMmethod, dict(weight_function="aorr", loss="hinge", l2_reg=1e-4, args=[0.2, 0.8]), 200),
    ("aorr_bce_l2", 800, 21, 18, ADMMmethod, dict(weight_function="aorr", loss="binary_cross_entropy", l2_reg=1e-4, args=[0.2, 0.8]), 1000),
    ("ehrm_bce_l2", 1000, 30, 19, ADMMmethod, dict(weight_function="ehrm", loss="binary_cross_entropy", l2_reg=0.01, B=-5), 700),
    ("sadmm_erm_bce_l1", 1500, 60, 11, smoothADMMmethod, dict(weight_function="erm", loss="binary_cross_entropy", l1_reg=0.01), 200),
]


def g9():
    only = os.environ.get("G9_ONLY")
    for name, n, d, seed, cls, kw, mi in TRAJ:
        if only and name not in only.split(","):
            continue
        intercept = name.startswith("aorr")
        X, y = problems.make_problem(n, d - 1 if intercept else d, seed, intercept=intercept)
        t0 = time.time()
        r = run_traj(X, y, cls, kw, max_iter=mi)
        print(f"   g9 {name}: iters={int(r['iters'])} conv={bool(r['converged'])} F={float(r['final_objective']):.12g} {time.time()-t0:.1f}s")
        save(f"g9_{name}.npz", config=np.array(json.dumps(dict(n=n, d=d, seed=seed, intercept=intercept,
             cls=cls.__name__, max_iter=mi, kw=kw))), x_sha256=np.array(problems.sha256_of(X)),
             y_sha256=np.array(problems.sha256_of(y)), **r)


def c1():
    Xtr, Xte, ytr, yte = problems.c1_data()
    print("   c1 sha256", problems.sha256_of(Xtr)[:16], problems.sha256_of(ytr)[:16])
    kw = dict(weight_function="erm", loss="binary_cross_entropy", l1_reg=0.01)
    r = run_traj(Xtr, ytr, ADMMmethod, kw)
    print(f"   c1: iters={int(r['iters'])} F={float(r['final_objective']):.14g} wall={float(r['wall']):.1f}s")
    save("g9_c1_srm_erm_bce_l1.npz", config=np.array(json.dumps(dict(recipe="oracle.problems.c1_data", kw=kw))),
         x_sha256=np.array(problems.sha256_of(Xtr)), y_sha256=np.array(problems.sha256_of(ytr)),
         x_first_row=Xtr[0, :3].copy(), y_first=ytr[:5, 0].copy(), **r)


def table():
    """Parse the ADMM rows of the reference's only published result file (an .xlsx is a
    zip of XML; openpyxl is not installed).  run_SRM.py:124-153 writes one list per row."""
    path = os.path.join(REF, "table", "erm_synthetic_6000x1000_l1_binary_cross_entropy.xlsx")
    ns = {"m": "http://schemas.openxmlformats.org/spreadsheetml/2006/main"}
    with zipfile.ZipFile(path) as zf:
        shared = []
        if "xl/sharedStrings.xml" in zf.namelist():
            for si in ET.fromstring(zf.read("xl/sharedStrings.xml")).findall("m:si", ns):
                shared.append("".join(t.text or "" for t in si.iter("{%s}t" % ns["m"])))
        sheet = ET.fromstring(zf.read("xl/worksheets/sheet1.xml"))
    rows = []
    for row in sheet.find("m:sheetData", ns).findall("m:row", ns):
        vals = []
        for cell in row.findall("m:c", ns):
            v = cell.find("m:v", ns)
            if v is None:
                continue
            vals.append(shared[int(v.text)] if cell.get("t") == "s" else float(v.text))
        rows.append(vals)
    out = dict(source="table/erm_synthetic_6000x1000_l1_binary_cross_entropy.xlsx sheet1",
               admm_train_losses=rows[0], admm_time=rows[1], admm_test_acc=rows[2],
               sadmm_train_losses=rows[3], sadmm_time=rows[4], sadmm_test_acc=rows[5])
    with open(os.path.join(HERE, "published_table_admm.json"), "w") as f:
        json.dump(out, f)
    print(f"   table: admm rows {len(rows[0])} losses / {len(rows[1])} times, final {rows[0][-1]!r}")


GROUPS = dict(g1=g1, g2=g2, g3=g3, g4=g4, g56=g56, g7=g7, g8=g8, g9=g9, c1=c1, table=table, g11=g11)

if __name__ == "__main__":
    which = sys.argv[1:] or list(GROUPS)
    versions = dict(numpy=np.__version__, torch=torch.__version__)
    import scipy, sklearn  # noqa: E401
    versions.update(scipy=scipy.__version__, sklearn=sklearn.__version__, python=sys.version.split()[0])
    for g in which:
        print(f"[{g}]")
        t0 = time.time()
        GROUPS[g]()
        print(f"  {g} done in {time.time()-t0:.1f}s")
    with open(os.path.join(HERE, "MANIFEST.json"), "w") as f:
        import hashlib
        files = {}
        for fn in sorted(os.listdir(HERE)):
            if fn.endswith((".npz", ".json")) and fn != "MANIFEST.json":
                files[fn] = hashlib.sha256(open(os.path.join(HERE, fn), "rb").read()).hexdigest()
        json.dump(dict(generated_by="tests/golden/make_goldens.py", reference="/root/reference @ v2",
                       shim="get_opt wrapped to return (res, None)", versions=versions, files=files), f, indent=1)
