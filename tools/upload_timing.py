import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import admm_for_rank_based_loss_amd as rbl
n, d = int(sys.argv[1]), 1000
rng = np.random.default_rng(0)
X = rng.standard_normal((n, d))
y = np.where(rng.random(n) < 0.5, 1.0, -1.0)
s = rbl.Solver(n, d, "erm", "binary_cross_entropy", reg=0.01, wstep=1, storage="f32")
for rep in range(3):
    t0 = time.perf_counter(); s.set_data(X, y); t = time.perf_counter() - t0
    print(f"set_data {n}x{d} fp64 host ({X.nbytes/1e9:.1f} GB): {t:.3f} s = {X.nbytes/1e9/t:.1f} GB/s")
D = s.get_D()[:5, :5]
print(np.max(np.abs(D - (-y[:5, None] * X[:5, :5]).astype(np.float32))))
