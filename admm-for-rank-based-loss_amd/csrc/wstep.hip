// wstep.hip - the w-sub-problem in Gram space (d-space), fp64.
//
// The reference solves the w-step in n-space: backtracking FISTA with >= 3 sweeps of D per
// inner iteration (src/util/fast_lasso.py:22-69, called from src/optim/algorithms.py:190-202)
// or SciPy L-BFGS-B with 2 sweeps per evaluation (src/util/w_LBFGS.py:31-62).  With
// G = D^T D (algorithms.py:24) and q = D^T (z + lambda/rho) the same convex problems are
//   lasso:        min 1/2 w'Gw - q'w + reg/(2 rho) ||w||_1
//   ridge:        (rho G + reg I) w = rho q
//   smoothed l1:  min rho/2 w'Gw - rho q'w + sum_j h_t(w_j)      (w_LBFGS.py:11-28)
// and cost one d x d mat-vec per inner iteration (G stays in L2 / Infinity Cache).
// Lasso: exact active-set kernel (lasso_fs.hip), FISTA with fixed step 1/L and gradient restart as its
// fallback; ridge: CG; smoothed l1: Jacobi-preconditioned nonlinear CG (Polak-Ribiere+) with an EXACT
// line search - the objective is a C1 piecewise quadratic, so along a direction p its derivative is
// piecewise linear in the step and G (w + a p) = G w + a G p needs no further mat-vec: 13-25
// iterations where FISTA needed 40-470 (G has exactly collinear columns: the Newton systems of a
// semismooth-Newton method are singular there, the line search is not affected).
// Each inner iteration = k_symv (all CUs) + one single-block update kernel that owns all
// reductions (fixed order, deterministic) and the convergence flag; no host round trip
// inside a batch of iterations.
#include "rbl_internal.h"
#include "device_math.h"
#include <cstdlib>
#include <cstdio>

namespace {

// y = alpha * G x + beta * x ; G is ld x ld row-major, one row per wave
__global__ __launch_bounds__(256) void k_symv(const double* __restrict__ G, long long ld,
                                               const double* __restrict__ x, double* __restrict__ y, double alpha,
                                               double beta, const int* __restrict__ done) {
    if (done && *done) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long long row = (long long)blockIdx.x * 4 + wave; row < ld; row += (long long)gridDim.x * 4) {
        const double2* g2 = reinterpret_cast<const double2*>(G + row * ld);
        const double2* x2 = reinterpret_cast<const double2*>(x);
        double acc = 0.0;
        for (long long j = lane; j < ld / 2; j += 64) {
            double2 a = g2[j], b = x2[j];
            acc = __builtin_fma(a.x, b.x, acc);
            acc = __builtin_fma(a.y, b.y, acc);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (lane == 0) y[row] = alpha * acc + beta * x[row];
    }
}

constexpr int UPD_THREADS = 1024;
constexpr int UPD_PER = 16;  // d <= 16384 in the single-block update kernels

struct FistaParams {
    int mode;      // 0 lasso (soft threshold), 1 smoothed l1 (Huber prox)
    double L;      // Lipschitz constant of the quadratic (lambda_max(G) * safety)
    double kappa;  // lasso: reg / (2 rho)
    double rho, reg, t;
    double tol;
};

__device__ inline double soft_thr(double b, double k) {
    // src/util/fast_lasso.py:15-19
    double a = fabs(b) - k;
    return a > 0.0 ? copysign(a, b) : 0.0;
}
__device__ inline double huber_prox(double b, double Ls, double reg, double t) {
    // argmin_u Ls/2 (u-b)^2 + h_t(u), h_t from src/util/w_LBFGS.py:11-19
    double uin = b * Ls / (Ls + reg / (2.0 * t));
    return (fabs(uin) <= t) ? uin : b - copysign(reg / (2.0 * Ls), b);
}

// scal[0] = FISTA t_k ; flags[0] = done, flags[1] = iterations performed
__global__ __launch_bounds__(UPD_THREADS) void k_fista_update(long long ld, const double* __restrict__ Gy,
                                                               const double* __restrict__ q, double* __restrict__ w,
                                                               double* __restrict__ yk, FistaParams P,
                                                               double* __restrict__ scal, int* __restrict__ flags,
                                                               int* __restrict__ publish) {
    // publish != NULL on the last update of a batch: (done, iterations) to pinned host memory, word 0 last
    if (flags[0]) {
        if (publish && threadIdx.x == 0) {
            publish[1] = flags[1];
            __threadfence_system();
            publish[0] = 1;
        }
        return;
    }
    __shared__ double smem[3 * UPD_THREADS / 64];
    double wn[UPD_PER], dw[UPD_PER];
    double acc[1] = {0.0};  // restart test: sum (y - x)(x - w)
    double mx_dw = 0.0, mx_w = 0.0;
#pragma unroll
    for (int k = 0; k < UPD_PER; ++k) {
        const long long j = (long long)k * UPD_THREADS + threadIdx.x;
        wn[k] = 0.0;
        dw[k] = 0.0;
        if (j < ld) {
            const double y = yk[j];
            const double b = y - (Gy[j] - q[j]) / P.L;
            const double x = (P.mode == 0) ? soft_thr(b, P.kappa / P.L) : huber_prox(b, P.rho * P.L, P.reg, P.t);
            const double dd = x - w[j];
            wn[k] = x;
            dw[k] = dd;
            acc[0] += (y - x) * dd;
            mx_dw = fmax(mx_dw, fabs(dd));
            mx_w = fmax(mx_w, fabs(x));
        }
    }
    rbl::block_sum<1, UPD_THREADS>(acc, smem);
    // block max of the two magnitudes
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mx_dw = fmax(mx_dw, __shfl_xor(mx_dw, off, 64));
        mx_w = fmax(mx_w, __shfl_xor(mx_w, off, 64));
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        smem[(threadIdx.x >> 6) * 2 + 0] = mx_dw;
        smem[(threadIdx.x >> 6) * 2 + 1] = mx_w;
    }
    __syncthreads();
    for (int wv = 0; wv < UPD_THREADS / 64; ++wv) {
        mx_dw = fmax(mx_dw, smem[wv * 2 + 0]);
        mx_w = fmax(mx_w, smem[wv * 2 + 1]);
    }
    const double t = scal[0];
    const bool restart = acc[0] > 0.0;  // O'Donoghue-Candes gradient restart
    const double tn = restart ? 1.0 : 0.5 * (1.0 + sqrt(1.0 + 4.0 * t * t));
    const double coef = restart ? 0.0 : (t - 1.0) / tn;
#pragma unroll
    for (int k = 0; k < UPD_PER; ++k) {
        const long long j = (long long)k * UPD_THREADS + threadIdx.x;
        if (j < ld) {
            w[j] = wn[k];
            yk[j] = wn[k] + coef * dw[k];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        scal[0] = tn;
        const int it = flags[1] + 1;
        const int done = (mx_dw <= P.tol * fmax(1.0, mx_w)) ? 1 : 0;
        flags[1] = it;
        if (done) flags[0] = 1;
        if (publish) {
            publish[1] = it;
            __threadfence_system();
            publish[0] = done;
        }
    }
}

// CG on A = rho G + reg I.  scal[1] = r'r, scal[2] = stop threshold on r'r
__global__ __launch_bounds__(UPD_THREADS) void k_cg_init(long long ld, const double* __restrict__ Aw,
                                                          const double* __restrict__ q, double rho,
                                                          double* __restrict__ r, double* __restrict__ p, double tol,
                                                          double* __restrict__ scal, int* __restrict__ flags) {
    __shared__ double smem[2 * UPD_THREADS / 64];
    double acc[2] = {0.0, 0.0};
    for (long long j = threadIdx.x; j < ld; j += UPD_THREADS) {
        const double b = rho * q[j];
        const double rr = b - Aw[j];
        r[j] = rr;
        p[j] = rr;
        acc[0] += rr * rr;
        acc[1] += b * b;
    }
    rbl::block_sum<2, UPD_THREADS>(acc, smem);
    if (threadIdx.x == 0) {
        scal[1] = acc[0];
        scal[2] = tol * tol * acc[1];
        flags[0] = (acc[0] <= scal[2]) ? 1 : 0;
        flags[1] = 0;
    }
}

__global__ __launch_bounds__(UPD_THREADS) void k_cg_update(long long ld, const double* __restrict__ Ap,
                                                            double* __restrict__ w, double* __restrict__ r,
                                                            double* __restrict__ p, double* __restrict__ scal,
                                                            int* __restrict__ flags, int* __restrict__ publish) {
    // publish != NULL on the last update of a batch: (done, iterations) go to pinned host memory
    // the host spins on, word 0 last
    if (flags[0]) {
        if (publish && threadIdx.x == 0) {
            publish[1] = flags[1];
            __threadfence_system();
            publish[0] = 1;
        }
        return;
    }
    __shared__ double smem[UPD_THREADS / 64];
    double pk[UPD_PER], rk[UPD_PER];
    double acc[1] = {0.0};
#pragma unroll
    for (int k = 0; k < UPD_PER; ++k) {
        const long long j = (long long)k * UPD_THREADS + threadIdx.x;
        pk[k] = (j < ld) ? p[j] : 0.0;
        acc[0] += (j < ld) ? pk[k] * Ap[j] : 0.0;
    }
    rbl::block_sum<1, UPD_THREADS>(acc, smem);
    const double rr = scal[1];
    const double alpha = (acc[0] > 0.0) ? rr / acc[0] : 0.0;
    double acc2[1] = {0.0};
#pragma unroll
    for (int k = 0; k < UPD_PER; ++k) {
        const long long j = (long long)k * UPD_THREADS + threadIdx.x;
        rk[k] = 0.0;
        if (j < ld) {
            w[j] += alpha * pk[k];
            rk[k] = r[j] - alpha * Ap[j];
            acc2[0] += rk[k] * rk[k];
        }
    }
    rbl::block_sum<1, UPD_THREADS>(acc2, smem);
    const double beta = (rr > 0.0) ? acc2[0] / rr : 0.0;
#pragma unroll
    for (int k = 0; k < UPD_PER; ++k) {
        const long long j = (long long)k * UPD_THREADS + threadIdx.x;
        if (j < ld) {
            r[j] = rk[k];
            p[j] = rk[k] + beta * pk[k];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        scal[1] = acc2[0];
        const int it = flags[1] + 1;
        const int done = (acc2[0] <= scal[2] || alpha == 0.0) ? 1 : 0;
        flags[1] = it;
        if (done) flags[0] = 1;
        if (publish) {
            publish[1] = it;
            __threadfence_system();
            publish[0] = done;
        }
    }
}


// ---- smoothed-l1 w-step (sADMM, src/util/w_LBFGS.py:11-28,54-62):
//   min F(w) = rho/2 w'Gw - rho q'w + sum_j h_t(w_j),  h_t(u) = reg u^2/(4t) if |u| <= t else reg/2 (|u| - t/2)
// Preconditioned nonlinear CG.  grad F = rho (G w - q) + h'(w); preconditioner M = diag(rho G_jj + h''(w_j))^-1.
struct NcgParams {
    double rho, reg, t, tol;
    int active;   // persistent kernel: try the linear solve on the incoming Huber pattern first (k_ncg_persist)
};
__device__ inline double hub_g(double u, double reg, double t) {   // h_t'(u), w_LBFGS.py:21-28
    return fabs(u) <= t ? reg * u / (2.0 * t) : copysign(0.5 * reg, u);
}
__device__ inline double hub_c(double u, double reg, double t) { return fabs(u) <= t ? reg / (2.0 * t) : 0.0; }
__device__ inline double precond_inv(double rho, double gd, double c) {
    const double den = rho * gd + c;
    return den > 0.0 ? 1.0 / den : 1.0;
}
template <int THREADS>
__device__ inline double block_max1(double v, double* smem /* THREADS / 64 */) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) smem[threadIdx.x >> 6] = v;
    __syncthreads();
    double m = smem[0];
    for (int w = 1; w < THREADS / 64; ++w) m = fmax(m, smem[w]);
    __syncthreads();
    return m;
}

// Gw = G w on entry.  scal[1] = g's, scal[2] = stop threshold on max |g_j|, scal[3] = 0 (no stalled step yet);
// flags[0] = done, flags[1] = iterations
__global__ __launch_bounds__(UPD_THREADS) void k_ncg_init(long long ld, const double* __restrict__ G,
                                                           const double* __restrict__ Gw, const double* __restrict__ q,
                                                           const double* __restrict__ w, NcgParams P,
                                                           double* __restrict__ p, double* __restrict__ s,
                                                           double* __restrict__ gdiag, double* __restrict__ scal,
                                                           int* __restrict__ flags) {
    __shared__ double smem[UPD_THREADS / 64];
    double acc[1] = {0.0};
    double gmax = 0.0, qmax = 0.0;
    for (long long j = threadIdx.x; j < ld; j += UPD_THREADS) {
        const double gd = G[j * ld + j], wj = w[j];
        gdiag[j] = gd;
        const double g = P.rho * (Gw[j] - q[j]) + hub_g(wj, P.reg, P.t);
        const double sj = precond_inv(P.rho, gd, hub_c(wj, P.reg, P.t)) * g;
        s[j] = sj;
        p[j] = -sj;
        acc[0] += g * sj;
        gmax = fmax(gmax, fabs(g));
        qmax = fmax(qmax, P.rho * fabs(q[j]));
    }
    rbl::block_sum<1, UPD_THREADS>(acc, smem);
    gmax = block_max1<UPD_THREADS>(gmax, smem);
    qmax = block_max1<UPD_THREADS>(qmax, smem);
    if (threadIdx.x == 0) {
        scal[1] = acc[0];
        scal[2] = P.tol * fmax(qmax, 0.5 * P.reg);
        scal[3] = 0.0;
        flags[0] = (gmax <= scal[2]) ? 1 : 0;
        flags[1] = 0;
    }
}

// one iteration: exact line search along p (Gp = G p on entry), update of w / Gw, new preconditioned gradient,
// Polak-Ribiere+ direction
__global__ __launch_bounds__(UPD_THREADS) void k_ncg_update(long long ld, const double* __restrict__ Gp,
                                                             const double* __restrict__ q, double* __restrict__ w,
                                                             double* __restrict__ Gw, double* __restrict__ p,
                                                             double* __restrict__ s, const double* __restrict__ gdiag,
                                                             NcgParams P, double* __restrict__ scal,
                                                             int* __restrict__ flags, int* __restrict__ publish) {
    if (flags[0]) {
        if (publish && threadIdx.x == 0) {
            publish[1] = flags[1];
            __threadfence_system();
            publish[0] = 1;
        }
        return;
    }
    __shared__ double smem[3 * UPD_THREADS / 64];
    // phi'(a) = a0 + a a1 + sum_j h'(w_j + a p_j) p_j
    double a3[3] = {0.0, 0.0, 0.0};
    for (long long j = threadIdx.x; j < ld; j += UPD_THREADS) {
        const double pj = p[j], lin = P.rho * (Gw[j] - q[j]);
        a3[0] += lin * pj;
        a3[1] += pj * Gp[j];
        a3[2] += fabs(lin * pj) + fabs(hub_g(w[j], P.reg, P.t) * pj);
    }
    rbl::block_sum<3, UPD_THREADS>(a3, smem);
    const double a0 = a3[0], a1 = P.rho * a3[1], ftol = 4e-16 * a3[2];
    double alpha = 0.0, lo = 0.0, hi = -1.0;
    for (int it = 0; it < 100; ++it) {
        double e2[2] = {0.0, 0.0};
        for (long long j = threadIdx.x; j < ld; j += UPD_THREADS) {
            const double pj = p[j], u = w[j] + alpha * pj;
            e2[0] += hub_g(u, P.reg, P.t) * pj;
            e2[1] += hub_c(u, P.reg, P.t) * pj * pj;
        }
        rbl::block_sum<2, UPD_THREADS>(e2, smem);
        const double f = a0 + alpha * a1 + e2[0], slope = a1 + e2[1];
        if (it == 0 && !(f < 0.0)) break;           // not a descent direction (rounding): alpha stays 0
        if (fabs(f) <= ftol) break;                 // the root of this linear piece, to rounding
        if (f < 0.0) lo = alpha; else hi = alpha;
        double an = (slope > 0.0) ? alpha - f / slope : -1.0;
        if (hi >= 0.0) {
            if (!(an > lo && an < hi)) an = 0.5 * (lo + hi);
        } else if (!(an > lo)) {
            an = lo > 0.0 ? 2.0 * lo : 1.0;         // flat piece (G p = 0, all coordinates outside [-t, t]): expand
        }
        if (an == alpha) break;
        alpha = an;
    }
    // update; sums: g's_new, g's_old, g'p_old
    double b3[3] = {0.0, 0.0, 0.0};
    double gmax = 0.0;
    for (long long j = threadIdx.x; j < ld; j += UPD_THREADS) {
        const double pj = p[j];
        const double wn = w[j] + alpha * pj, gwn = Gw[j] + alpha * Gp[j];
        w[j] = wn;
        Gw[j] = gwn;
        const double g = P.rho * (gwn - q[j]) + hub_g(wn, P.reg, P.t);
        const double sn = precond_inv(P.rho, gdiag[j], hub_c(wn, P.reg, P.t)) * g;
        b3[0] += g * sn;
        b3[1] += g * s[j];
        b3[2] += g * pj;
        s[j] = sn;
        gmax = fmax(gmax, fabs(g));
    }
    rbl::block_sum<3, UPD_THREADS>(b3, smem);
    gmax = block_max1<UPD_THREADS>(gmax, smem);
    const double gs_old = scal[1], stalled = scal[3];
    double beta = (gs_old > 0.0) ? fmax(0.0, (b3[0] - b3[1]) / gs_old) : 0.0;
    if (alpha == 0.0 || -b3[0] + beta * b3[2] >= 0.0) beta = 0.0;     // restart along the preconditioned gradient
    for (long long j = threadIdx.x; j < ld; j += UPD_THREADS) p[j] = -s[j] + beta * p[j];
    __syncthreads();
    if (threadIdx.x == 0) {
        scal[1] = b3[0];
        scal[3] = (alpha == 0.0) ? 1.0 : 0.0;
        const int it = flags[1] + 1;
        // converged, or two steps in a row without progress (the gradient is at rounding level)
        const int done = (gmax <= scal[2] || (alpha == 0.0 && stalled != 0.0)) ? 1 : 0;
        flags[1] = it;
        if (done) flags[0] = 1;
        if (publish) {
            publish[1] = it;
            __threadfence_system();
            publish[0] = done;
        }
    }
}

// x <- y / ||y|| ; scal[3] = ||y||
__global__ __launch_bounds__(UPD_THREADS) void k_normalize(long long ld, const double* __restrict__ y,
                                                            double* __restrict__ x, double* __restrict__ scal) {
    __shared__ double smem[UPD_THREADS / 64];
    double acc[1] = {0.0};
    for (long long j = threadIdx.x; j < ld; j += UPD_THREADS) acc[0] += y[j] * y[j];
    rbl::block_sum<1, UPD_THREADS>(acc, smem);
    const double nrm = sqrt(acc[0]);
    const double inv = nrm > 0.0 ? 1.0 / nrm : 0.0;
    for (long long j = threadIdx.x; j < ld; j += UPD_THREADS) x[j] = y[j] * inv;
    if (threadIdx.x == 0) scal[3] = nrm;
}

__global__ void k_power_init(long long ld, long long d, double* __restrict__ x) {
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < ld; j += (long long)gridDim.x * blockDim.x) {
        // fixed pseudo-random start vector (never orthogonal to the top eigenvector in practice)
        unsigned long long h = (unsigned long long)(j + 1) * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29;
        x[j] = (j < d) ? 0.5 + (double)(h & 0xffff) / 65536.0 : 0.0;
    }
}

__global__ __launch_bounds__(UPD_THREADS) void k_reg_terms(long long d, const double* __restrict__ w,
                                                            double* __restrict__ out) {
    __shared__ double smem[2 * UPD_THREADS / 64];
    double acc[2] = {0.0, 0.0};
    for (long long j = threadIdx.x; j < d; j += UPD_THREADS) {
        acc[0] += w[j] * w[j];
        acc[1] += fabs(w[j]);
    }
    rbl::block_sum<2, UPD_THREADS>(acc, smem);
    if (threadIdx.x == 0) {
        out[0] = acc[0];  // objective.py:83-84  sum w^2
        out[1] = acc[1];  // objective.py:85-86  ||w||_1
    }
}

// out[0] = ||w - w_prev||^2 (algorithms.py:136), out[1] = sum w^2, out[2] = ||w||_1 (objective.py:83-86)
__global__ __launch_bounds__(UPD_THREADS) void k_w_stats(long long d, const double* __restrict__ w,
                                                          const double* __restrict__ w_prev, double* __restrict__ out) {
    __shared__ double smem[3 * UPD_THREADS / 64];
    double acc[3] = {0.0, 0.0, 0.0};
    for (long long j = threadIdx.x; j < d; j += UPD_THREADS) {
        const double t = w[j] - w_prev[j];
        acc[0] += t * t;
        acc[1] += w[j] * w[j];
        acc[2] += fabs(w[j]);
    }
    rbl::block_sum<3, UPD_THREADS>(acc, smem);
    if (threadIdx.x == 0) {
        out[0] = acc[0];
        out[1] = acc[1];
        out[2] = acc[2];
    }
}

__global__ void k_soft_threshold(long long d, double* __restrict__ w, double t) {
    // smoothADMMmethod's final step, src/optim/algorithms.py:258
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < d; j += (long long)gridDim.x * blockDim.x) {
        const double a = fabs(w[j]) - t;
        w[j] = a > 0.0 ? copysign(a, w[j]) : 0.0;
    }
}

inline unsigned symv_grid(long long ld) {
    long long g = (ld + 3) / 4;
    if (g > 4096) g = 4096;
    return (unsigned)g;
}

}  // namespace

int launch_power_iteration(const double* G, int64_t ld, double* x, double* y, double* scal, int iters,
                           double* lambda_host, hipStream_t s) {
    hipLaunchKernelGGL(k_power_init, dim3(64), dim3(256), 0, s, (long long)ld, (long long)ld, x);
    hipLaunchKernelGGL(k_normalize, dim3(1), dim3(UPD_THREADS), 0, s, (long long)ld, x, x, scal);
    for (int it = 0; it < iters; ++it) {
        hipLaunchKernelGGL(k_symv, dim3(symv_grid(ld)), dim3(256), 0, s, G, (long long)ld, x, y, 1.0, 0.0,
                           (const int*)nullptr);
        hipLaunchKernelGGL(k_normalize, dim3(1), dim3(UPD_THREADS), 0, s, (long long)ld, y, x, scal);
    }
    RBL_HIP(hipGetLastError());
    RBL_HIP(hipMemcpyAsync(lambda_host, scal + 3, sizeof(double), hipMemcpyDeviceToHost, s));
    RBL_HIP(hipStreamSynchronize(s));
    return RBL_OK;
}

namespace {

// FISTA with gradient restart (lasso: mode 0, Huber-smoothed l1: mode 1), polled every BATCH iterations
int run_fista(int mode, const double* G, int64_t ld, const double* q, double rho, double reg, double smooth_t, double L,
              double tol, int max_inner, double* w, WstepWorkspace& ws, int* iters_host, hipStream_t s) {
    constexpr int BATCH = 8;
    const unsigned sg = symv_grid(ld);
    FistaParams P;
    P.mode = mode;
    P.L = L;
    P.kappa = reg / (2.0 * rho);
    P.rho = rho;
    P.reg = reg;
    P.t = smooth_t;
    P.tol = tol;
    const double one = 1.0;
    RBL_HIP(hipMemcpyAsync(ws.scal, &one, sizeof(double), hipMemcpyHostToDevice, s));
    RBL_HIP(hipMemsetAsync(ws.flags, 0, 2 * sizeof(int), s));
    RBL_HIP(hipMemcpyAsync(ws.yk, w, sizeof(double) * ld, hipMemcpyDeviceToDevice, s));
    // same batching as the CG: about as many iterations as last time (+25 %) per host round trip,
    // (done, iterations) published to pinned memory by the batch's last update
    int batch = ws.last_fista > 0 ? ws.last_fista + ws.last_fista / 4 + 2 : 4 * BATCH;
    if (batch < BATCH) batch = BATCH;
    if (batch > 512) batch = 512;
    int done_iters = 0, done = 0, iters = 0;
    while (done_iters < max_inner) {
        ws.pin[6] = -1;
        for (int b = 0; b < batch; ++b) {
            hipLaunchKernelGGL(k_symv, dim3(sg), dim3(256), 0, s, G, (long long)ld, ws.yk, ws.Gy, 1.0, 0.0, ws.flags);
            hipLaunchKernelGGL(k_fista_update, dim3(1), dim3(UPD_THREADS), 0, s, (long long)ld, ws.Gy, q, w, ws.yk, P,
                               ws.scal, ws.flags, b == batch - 1 ? ws.pin + 6 : (int*)nullptr);
        }
        RBL_HIP(hipGetLastError());
        done_iters += batch;
        rbl_spin_wait(ws.pin + 6, -1, s);
        const volatile int* st = ws.pin + 6;
        done = st[0];
        iters = st[1];
        if (done != 0) break;
        batch = 4 * BATCH;
    }
    if (done == 1) ws.last_fista = iters;
    if (iters_host) *iters_host = iters;
    return done == -1 ? RBL_ERR_HIP : RBL_OK;
}


struct WpPlan;
bool ncg_persist_try(const double* G, int64_t ld, const double* q, NcgParams P, int max_iter, double* w,
                     WstepWorkspace& ws, bool want_Gw, int* status, int* iters, hipStream_t s, int* rc);

// smoothed-l1 w-step by preconditioned nonlinear CG with exact line search (k_ncg_*), batched like the CG:
// about as many iterations as last time per host round trip.  Falls back to FISTA (which has no line
// search to fail) if the cap is reached.
int run_ncg(const double* G, int64_t ld, const double* q, double rho, double reg, double smooth_t, double L, double tol,
            int max_inner, double* w, WstepWorkspace& ws, int* iters_host, hipStream_t s, bool want_Gw) {
    const unsigned sg = symv_grid(ld);
    static const int ncg_active = [] {
        const char* e = getenv("RBL_NCG_ACTIVE");     // =0: the nonlinear CG alone (comparison)
        return (e && e[0] == '0') ? 0 : 1;
    }();
    NcgParams P{rho, reg, smooth_t, tol, ncg_active};
    {
        // one persistent launch (k_ncg_persist) where the row width allows it
        int status = 0, it = 0, rc = RBL_OK;
        if (ncg_persist_try(G, ld, q, P, max_inner < 600 ? max_inner : 600, w, ws, want_Gw, &status, &it, s, &rc)) {
            RBL_TRY(rc);
            if (status == 1) {
                ws.last_fista = it;
                ws.gw_valid = want_Gw;
                ws.form = 1;
                if (iters_host) *iters_host = it;
                return RBL_OK;
            }
            if (status < 0) {
                rbl_set_error("w-step: the persistent nonlinear-CG kernel did not complete (status %d)", status);
                return RBL_ERR_HIP;
            }
            int more = 0;      // iteration cap: FISTA takes over from the w reached, as on the batched path
            RBL_TRY(run_fista(1, G, ld, q, rho, reg, smooth_t, L, tol, max_inner, w, ws, &more, s));
            if (iters_host) *iters_host = it + more;
            return RBL_OK;
        }
    }
    double *p = ws.p, *sv = ws.r, *Gp = ws.Gy, *Gw = ws.wn, *gdiag = ws.yk;
    hipLaunchKernelGGL(k_symv, dim3(sg), dim3(256), 0, s, G, (long long)ld, w, Gw, 1.0, 0.0, (const int*)nullptr);
    hipLaunchKernelGGL(k_ncg_init, dim3(1), dim3(UPD_THREADS), 0, s, (long long)ld, G, Gw, q, w, P, p, sv, gdiag, ws.scal,
                       ws.flags);
    // one batch of about as many iterations as last time (+1): updates after convergence are no-op launches
    // (~3.5 us each), a batch that falls short costs one more host round trip - the count moves by +-1
    // (batch sizes +1/+2/+4 and follow-up batches of 4/8 all measured the same over 110 iterations of C2smooth)
    int batch = ws.last_fista > 0 ? ws.last_fista + 1 : 32;
    if (batch < 4) batch = 4;
    if (batch > 128) batch = 128;
    const int cap = max_inner < 600 ? max_inner : 600;
    int done_iters = 0, done = 0, iters = 0;
    while (done_iters < cap) {
        ws.pin[8] = -1;
        for (int b = 0; b < batch; ++b) {
            hipLaunchKernelGGL(k_symv, dim3(sg), dim3(256), 0, s, G, (long long)ld, p, Gp, 1.0, 0.0, ws.flags);
            hipLaunchKernelGGL(k_ncg_update, dim3(1), dim3(UPD_THREADS), 0, s, (long long)ld, Gp, q, w, Gw, p, sv, gdiag, P,
                               ws.scal, ws.flags, b == batch - 1 ? ws.pin + 8 : (int*)nullptr);
        }
        RBL_HIP(hipGetLastError());
        done_iters += batch;
        rbl_spin_wait(ws.pin + 8, -1, s);
        const volatile int* st = ws.pin + 8;
        done = st[0];
        iters = st[1];
        if (done != 0) break;
        batch = 4;
    }
    if (done == -1) return RBL_ERR_HIP;
    if (done == 1) {
        ws.last_fista = iters;
        if (iters_host) *iters_host = iters;
        return RBL_OK;
    }
    int more = 0;
    RBL_TRY(run_fista(1, G, ld, q, rho, reg, smooth_t, L, tol, max_inner, w, ws, &more, s));
    if (iters_host) *iters_host = iters + more;
    return RBL_OK;
}

}  // namespace

namespace {

// ============================================================================================
// One persistent kernel per d-space w-step (round 3).  The batched path above runs 2 + 2 k kernels of ~5 us each
// with ~1.5 us of boundary between them (CG at d = 1000: ~14 us per inner iteration, 0.10 ms per ADMM iteration; the
// nonlinear CG 0.2-0.6 ms) for a few microseconds of arithmetic.  Here the whole w-step is ONE launch:
//   * block b owns `rpb` = 16 consecutive rows of G, staged ONCE into its LDS (16 x 1000 doubles = 128 KB): per inner
//     iteration it computes its rows of G p from LDS;
//   * the 16 results leave the block as ONE 128-byte line written by ONE wave instruction of write-through (`sc1`)
//     8-byte stores, the storing wave drains its stores (s_waitcnt vmcnt(0)) and one lane arrives on a counter
//     (hierarchical: one counter per group of blocks b % 8 - blocks that share an XCD under the observed placement,
//     which only matters for speed -, the group's last arriver adds to the top counter); one lane polls the top
//     counter with `sc1` loads, a workgroup barrier, then every block reads the whole product (8 KB) with 16-byte
//     `sc1` buffer loads.  No cache is written back or invalidated: every handed-off byte is stored AND loaded `sc1`
//     (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility": the row "agent-scope
//     atomic adds, one lane of each storing workgroup / sc1 poll / workgroup barrier before every load"); the first
//     version of this kernel used __threadfence() pairs around the counter and cost as much per inner iteration as
//     the two launches it replaced (measured: C2l2 242.5 against 243.7 it/s);
//   * every block then performs the REST of the iteration - dot products, step length, vector updates, the exact line
//     search of the nonlinear CG - redundantly on vectors it keeps in registers.  Identical arithmetic on identical
//     inputs in a fixed order: every block holds the same bits, so convergence is decided identically everywhere and
//     no flag has to be exchanged; standard CG / NCG arithmetic, nothing pipelined or reordered across iterations;
//   * the exchange buffer is double-buffered by iteration parity (a fast block may be one matvec ahead of a slow
//     block that still reads the previous product);
//   * every wait is bounded (the kernel drains with status -2 instead of hanging if the blocks cannot all be
//     resident - they are: at most one block per CU is launched);
//   * block 0 writes w, every block its rows of G w (the rho prediction of the single-sweep iteration wants it), and
//     (done, iterations) go to pinned host memory: one host wait per w-step.
// rows of d <= 2048 (8 elements per thread); wider problems keep the batched path.
constexpr int WP_THREADS = 256;
constexpr int WP_RPB = 16;                         // rows of G per block
constexpr unsigned WP_SPIN_CAP = 1u << 22;         // sweeps of a collecting wave before it gives up (seconds)

typedef unsigned long long wp_u64 __attribute__((address_space(1)));
typedef unsigned wp_u32 __attribute__((address_space(1)));

// index of the k-th vector element a thread owns (pairs of neighbours)
__device__ inline int wp_idx(int k) { return 2 * ((k >> 1) * WP_THREADS + (int)threadIdx.x) + (k & 1); }

// Publishes this block's WP_RPB results (ybuf, LDS) and returns with the whole vector in v[] (the elements this thread
// owns).  The data IS the flag (MI355X_MICROARCH.md, Guideline 16's form R2): every double travels as two 8-byte
// granules {tag = this exchange's number | 32 bits of the value}, each written by ONE write-through (sc1) 8-byte store -
// the block's 32 granules by one wave instruction - and every block re-reads the granules it needs with sc1 loads until
// all of them carry this exchange's tag.  No counter, no fence, no drain: one store and one load round trip per
// exchange (the counter form - sc1 stores, drain, hierarchical arrival counter, poll, workgroup barrier, sc1 loads -
// measured 6.7 us per exchange, this one is bounded by the visibility of a store).  Tags are unique over the launches of
// a handle (launch number << 12 | exchange number), so nothing has to be cleared between launches; two buffers are used
// alternately (a block can be one exchange ahead of the slowest one, never two).  false = some block gave up waiting.
template <int PER>
__device__ inline bool wp_exchange(unsigned tag, const double* ybuf, unsigned long long* xg, int ld, double (&v)[PER],
                                   unsigned* abort_word) {
    __syncthreads();                                // ybuf complete
    if (threadIdx.x < 2 * WP_RPB) {
        const int r = (int)threadIdx.x >> 1, h = (int)threadIdx.x & 1;
        const unsigned long long bits = __builtin_bit_cast(unsigned long long, ybuf[r]);
        const unsigned long long half = h ? (bits >> 32) : (bits & 0xffffffffull);
        __hip_atomic_store((wp_u64*)xg + 2 * ((size_t)blockIdx.x * WP_RPB + r) + h, ((unsigned long long)tag << 32) | half,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    unsigned lo32[PER], hi32[PER];
    bool have[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        have[k] = wp_idx(k) >= ld;                  // elements beyond the vector: nothing to wait for
        lo32[k] = hi32[k] = 0u;
    }
    bool ok = true;
    for (unsigned spins = 0;; ++spins) {
        bool all = true;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            if (have[k]) continue;
            const int j = wp_idx(k);
            const unsigned long long g0 = __hip_atomic_load((wp_u64*)xg + 2 * (size_t)j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long g1 = __hip_atomic_load((wp_u64*)xg + 2 * (size_t)j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(g0 >> 32) == tag && (unsigned)(g1 >> 32) == tag) {
                lo32[k] = (unsigned)g0;
                hi32[k] = (unsigned)g1;
                have[k] = true;
            } else {
                all = false;
            }
        }
        if (__all(all)) break;                      // wave-uniform exit
        if (spins > WP_SPIN_CAP || __hip_atomic_load((wp_u32*)abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
            __hip_atomic_store((wp_u32*)abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = false;
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) v[k] = __builtin_bit_cast(double, ((unsigned long long)hi32[k] << 32) | lo32[k]);
    return __syncthreads_and(ok ? 1 : 0) != 0;      // every wave of the block takes the same way out
}

// rows [r0, r1) of y = alpha G x + beta x -> ybuf[row - r0] (LDS; rows beyond r1 up to r0 + WP_RPB: 0); x in LDS,
// the block's rows of G in LDS (GLDS) or in global memory
template <bool GLDS>
__device__ inline void wp_matvec(const double* __restrict__ G, const double* gs, int ld, int r0, int r1,
                                 const double* xs, double alpha, double beta, double* ybuf) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int row = r0 + wave; row < r0 + WP_RPB; row += WP_THREADS / 64) {
        double acc = 0.0;
        if (row < r1) {
            const double* g = GLDS ? gs + (size_t)(row - r0) * ld : G + (size_t)row * ld;
            for (int j = lane; j < ld; j += 64) acc = __builtin_fma(g[j], xs[j], acc);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        }
        if (lane == 0) ybuf[row - r0] = row < r1 ? alpha * acc + beta * xs[row] : 0.0;
    }
}

template <bool GLDS>
__device__ inline void wp_stage_rows(const double* __restrict__ G, double* gs, int ld, int r0, int r1) {
    if (!GLDS) return;
    // 128 KB per block: 16-byte loads, 8 in flight per thread (a plain element loop - one load, one wait, one LDS
    // store - spent ~100 us here, most of the kernel; ld is a multiple of 4, so rows keep 16-byte alignment)
    const int cnt2 = (r1 - r0) * ld / 2;
    const double2* src = reinterpret_cast<const double2*>(G + (size_t)r0 * ld);
    double2* dst = reinterpret_cast<double2*>(gs);
    for (int base = 0; base < cnt2; base += 8 * WP_THREADS) {
        double2 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * WP_THREADS + (int)threadIdx.x;
            t[u] = i < cnt2 ? src[i] : double2{0.0, 0.0};
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * WP_THREADS + (int)threadIdx.x;
            if (i < cnt2) dst[i] = t[u];
        }
    }
}

__device__ inline void wp_publish(int* pin, int done, int iters) {
    pin[1] = iters;
    __threadfence_system();
    *reinterpret_cast<volatile int*>(pin) = done;
}

// (rho G + reg I) w = rho q by CG, warm-started from w (w_LBFGS.py:31-53 solves the same system by L-BFGS-B)
template <bool GLDS, int PER>
__global__ __launch_bounds__(WP_THREADS) void k_cg_persist(const double* __restrict__ G, int ld, const double* __restrict__ q,
                                                            double rho, double reg, double tol, int max_iter,
                                                            double* __restrict__ w, unsigned long long* x0, unsigned long long* x1,
                                                            unsigned tag_base, double* __restrict__ Gw_out, unsigned* abort_word,
                                                            int* pin, long long* dbg) {
    extern __shared__ __attribute__((aligned(16))) double wp_lds[];
    if (dbg && blockIdx.x == 0 && threadIdx.x == 0) dbg[0] = (long long)wall_clock64();
    double* xs = wp_lds;                       // ld
    double* red = wp_lds + ld;                 // 16 (block sums)
    double* ybuf = red + 16;                   // WP_RPB
    double* gs = ybuf + WP_RPB + 16;           // WP_RPB * ld
    const int r0 = blockIdx.x * WP_RPB, r1 = min(ld, r0 + WP_RPB);
    unsigned xn = 0;                            // exchanges so far (the next one's tag: tag_base + xn + 1)
    wp_stage_rows<GLDS>(G, gs, ld, r0, r1);
    double wj[PER], rj[PER], pj[PER], aj[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int j = wp_idx(k);
        wj[k] = (j < ld) ? w[j] : 0.0;
        if (j < ld) xs[j] = wj[k];
    }
    __syncthreads();
    if (dbg && blockIdx.x == 0 && threadIdx.x == 0) dbg[1] = (long long)wall_clock64();
    wp_matvec<GLDS>(G, gs, ld, r0, r1, xs, rho, reg, ybuf);          // A w
    int iters = 0, done = 0, ok = 1;
    double rr = 0.0, thr = 0.0;
    ++xn;
    if (!wp_exchange<PER>(tag_base + xn, ybuf, x0, ld, aj, abort_word)) ok = 0;
    if (dbg && blockIdx.x == 0 && threadIdx.x == 0) dbg[2] = (long long)wall_clock64();
    if (ok) {
        double acc[2] = {0.0, 0.0};
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int j = wp_idx(k);
            rj[k] = pj[k] = 0.0;
            if (j < ld) {
                const double bb = rho * q[j];
                rj[k] = bb - aj[k];
                pj[k] = rj[k];
                acc[0] += rj[k] * rj[k];
                acc[1] += bb * bb;
            }
        }
        rbl::block_sum<2, WP_THREADS>(acc, red);
        rr = acc[0];
        thr = tol * tol * acc[1];
        done = rr <= thr ? 1 : 0;
    }
    while (ok && !done && iters < max_iter) {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int j = wp_idx(k);
            if (j < ld) xs[j] = pj[k];
        }
        __syncthreads();
        wp_matvec<GLDS>(G, gs, ld, r0, r1, xs, rho, reg, ybuf);      // A p
        ++xn;
        if (!wp_exchange<PER>(tag_base + xn, ybuf, (iters & 1) ? x0 : x1, ld, aj, abort_word)) {
            ok = 0;
            break;
        }
        double a1[1] = {0.0};
#pragma unroll
        for (int k = 0; k < PER; ++k) a1[0] += pj[k] * aj[k];        // (elements beyond ld: p = 0 and A p = 0)
        rbl::block_sum<1, WP_THREADS>(a1, red);
        const double alpha = (a1[0] > 0.0) ? rr / a1[0] : 0.0;
        double a2[1] = {0.0};
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            wj[k] += alpha * pj[k];
            rj[k] -= alpha * aj[k];
            a2[0] += rj[k] * rj[k];
        }
        rbl::block_sum<1, WP_THREADS>(a2, red);
        const double beta = (rr > 0.0) ? a2[0] / rr : 0.0;
#pragma unroll
        for (int k = 0; k < PER; ++k) pj[k] = rj[k] + beta * pj[k];
        rr = a2[0];
        ++iters;
        done = (rr <= thr || alpha == 0.0) ? 1 : 0;
    }
    if (dbg && blockIdx.x == 0 && threadIdx.x == 0) dbg[3] = (long long)wall_clock64();
    if (ok && Gw_out) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int j = wp_idx(k);
            if (j < ld) xs[j] = wj[k];
        }
        __syncthreads();
        wp_matvec<GLDS>(G, gs, ld, r0, r1, xs, 1.0, 0.0, ybuf);      // G w of the solution (rho prediction)
        __syncthreads();
        if (threadIdx.x < r1 - r0) Gw_out[r0 + threadIdx.x] = ybuf[threadIdx.x];
    }
    if (blockIdx.x == 0) {
        if (ok) {
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int j = wp_idx(k);
                if (j < ld) w[j] = wj[k];
            }
        }
        __syncthreads();
        if (dbg && threadIdx.x == 0) dbg[4] = (long long)wall_clock64();
        if (threadIdx.x == 0) wp_publish(pin, ok ? done : -2, iters);
        if (dbg && threadIdx.x == 0) dbg[5] = (long long)wall_clock64();
    }
}

// smoothed-l1 w-step: the preconditioned nonlinear CG of k_ncg_init / k_ncg_update in one launch
template <bool GLDS, int PER>
__global__ __launch_bounds__(WP_THREADS) void k_ncg_persist(const double* __restrict__ G, int ld, const double* __restrict__ q,
                                                             NcgParams P, int max_iter, double* __restrict__ w,
                                                             unsigned long long* x0, unsigned long long* x1, unsigned tag_base,
                                                             double* __restrict__ Gw_out, unsigned* abort_word, int* pin) {
    extern __shared__ __attribute__((aligned(16))) double wp_lds[];
    double* xs = wp_lds;
    double* red = wp_lds + ld;
    double* ybuf = red + 16;
    double* gs = ybuf + WP_RPB + 16;
    const int r0 = blockIdx.x * WP_RPB, r1 = min(ld, r0 + WP_RPB);
    unsigned xn = 0;
    wp_stage_rows<GLDS>(G, gs, ld, r0, r1);
    double wj[PER], gwj[PER], pj[PER], sj[PER], gdj[PER], qj[PER], gpj[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int j = wp_idx(k);
        wj[k] = (j < ld) ? w[j] : 0.0;
        qj[k] = (j < ld) ? q[j] : 0.0;
        gdj[k] = (j < ld) ? G[(size_t)j * ld + j] : 1.0;
        if (j < ld) xs[j] = wj[k];
    }
    __syncthreads();
    wp_matvec<GLDS>(G, gs, ld, r0, r1, xs, 1.0, 0.0, ybuf);          // G w
    int iters = 0, done = 0, ok = 1;
    double gs_old = 0.0, thr = 0.0, stalled = 0.0;
    ++xn;
    if (!wp_exchange<PER>(tag_base + xn, ybuf, x0, ld, gwj, abort_word)) ok = 0;
    if (ok) {
        double gmax = 0.0, qmax = 0.0;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int j = wp_idx(k);
            if (j < ld) {
                gmax = fmax(gmax, fabs(P.rho * (gwj[k] - qj[k]) + hub_g(wj[k], P.reg, P.t)));
                qmax = fmax(qmax, P.rho * fabs(qj[k]));
            } else {
                gwj[k] = 0.0;
            }
        }
        gmax = block_max1<WP_THREADS>(gmax, red);
        qmax = block_max1<WP_THREADS>(qmax, red);
        thr = P.tol * fmax(qmax, 0.5 * P.reg);
        done = gmax <= thr ? 1 : 0;
    }
    // Phase A: the Huber term is quadratic (|w_j| <= t) or linear (|w_j| > t) in every coordinate, so WITH THE PATTERN OF
    // THE SOLUTION KNOWN the w-step is the linear system (rho G + diag(c)) w = rho q - s, c_j = reg/(2t) on the quadratic
    // coordinates, s_j = (reg/2) sign(w_j) on the others.  Between two ADMM iterations the pattern rarely changes: take it
    // from the incoming w and run Jacobi-preconditioned CG on that system (one all-gather per iteration, as below; a linear
    // CG gains 1-1.5 digits per iteration on this spectrum where the nonlinear one needs restarts).  The result is accepted
    // only if it lies in the assumed pattern - then the linearised gradient IS the gradient and the stop rule is the
    // nonlinear CG's, on the true gradient.  With a wrong pattern the system can be singular on the linear coordinates
    // (collinear columns) and its "solution" 1e14 away, so the attempt is dropped early: when an iterate is still outside
    // the pattern at the 6th step (the first steps often cross and come back: 6M x 1000, iterations 50-58 of an sADMM
    // run - 2, 2, 1, 0, 0 coordinates outside, converged in 5-7 steps where the nonlinear CG takes 23-33), when the
    // residual grows 100-fold, or after 25 steps.  The nonlinear CG then starts from the point reached if the true
    // gradient is smaller there than at the incoming w (a wrong-pattern solution is often 1e-6 from the right one),
    // else from the incoming w as if nothing had happened.  The host stops trying for a while after a dropped attempt
    // (run_ncg_persist).
    int phase_a = -1;   // -1 not attempted, 0 attempted and dropped, 1 accepted
    if (ok && !done && P.active) {
        phase_a = 0;
        double cj[PER], rj[PER], zj[PER], sgj[PER], w0j[PER], gw0j[PER];
        double a1[1] = {0.0};
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int j = wp_idx(k);
            cj[k] = rj[k] = zj[k] = pj[k] = sgj[k] = 0.0;
            w0j[k] = wj[k];
            gw0j[k] = gwj[k];
            if (j < ld) {
                cj[k] = hub_c(wj[k], P.reg, P.t);
                sgj[k] = (cj[k] == 0.0) ? copysign(1.0, wj[k]) : 0.0;   // the sign the linear part assumes
                rj[k] = -(P.rho * (gwj[k] - qj[k]) + hub_g(wj[k], P.reg, P.t));
                zj[k] = precond_inv(P.rho, gdj[k], cj[k]) * rj[k];
                pj[k] = zj[k];
                a1[0] += rj[k] * zj[k];
            }
        }
        rbl::block_sum<1, WP_THREADS>(a1, red);
        double rz = a1[0];
        int lin = 0;
        double rmax0 = 0.0;
#pragma unroll
        for (int k = 0; k < PER; ++k) rmax0 = fmax(rmax0, fabs(rj[k]));
        rmax0 = block_max1<WP_THREADS>(rmax0, red);
        for (int itA = 0; itA < 25 && iters < max_iter; ++itA) {
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int j = wp_idx(k);
                if (j < ld) xs[j] = pj[k];
            }
            __syncthreads();
            wp_matvec<GLDS>(G, gs, ld, r0, r1, xs, 1.0, 0.0, ybuf);      // G p
            ++xn;
            if (!wp_exchange<PER>(tag_base + xn, ybuf, (xn & 1) ? x0 : x1, ld, gpj, abort_word)) {
                ok = 0;
                break;
            }
            double b1[1] = {0.0};
#pragma unroll
            for (int k = 0; k < PER; ++k) b1[0] += pj[k] * (P.rho * gpj[k] + cj[k] * pj[k]);
            rbl::block_sum<1, WP_THREADS>(b1, red);
            ++iters;
            if (!(b1[0] > 0.0)) break;
            const double alpha = rz / b1[0];
            double b2[2] = {0.0, 0.0};
            double rmax = 0.0;
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int j = wp_idx(k);
                if (j < ld) {
                    wj[k] += alpha * pj[k];
                    gwj[k] += alpha * gpj[k];
                    rj[k] -= alpha * (P.rho * gpj[k] + cj[k] * pj[k]);
                    zj[k] = precond_inv(P.rho, gdj[k], cj[k]) * rj[k];
                    b2[0] += rj[k] * zj[k];
                    rmax = fmax(rmax, fabs(rj[k]));
                    const bool quad = fabs(wj[k]) <= P.t;
                    if (quad != (cj[k] != 0.0) || (!quad && copysign(1.0, wj[k]) != sgj[k])) b2[1] += 1.0;
                }
            }
            rbl::block_sum<2, WP_THREADS>(b2, red);
            rmax = block_max1<WP_THREADS>(rmax, red);
            if ((b2[1] != 0.0 && itA >= 5) || rmax > 100.0 * rmax0) break;   // still outside the pattern / diverging
            if (rmax <= thr) {
                lin = b2[1] == 0.0 ? 1 : 0;
                break;
            }
            const double beta = (rz > 0.0) ? b2[0] / rz : 0.0;
#pragma unroll
            for (int k = 0; k < PER; ++k) pj[k] = zj[k] + beta * pj[k];
            rz = b2[0];
        }
        if (ok && lin) {
            // inside the assumed pattern the linearised gradient IS the gradient; the recurrence's residual is not trusted
            double gmax = 0.0;
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int j = wp_idx(k);
                if (j < ld) gmax = fmax(gmax, fabs(P.rho * (gwj[k] - qj[k]) + hub_g(wj[k], P.reg, P.t)));
            }
            gmax = block_max1<WP_THREADS>(gmax, red);
            if (gmax <= thr) {
                done = 1;
                phase_a = 1;
            }
        }
        if (ok && !done) {
            // not accepted: the nonlinear CG starts from the point reached if the true gradient is smaller there than at
            // the incoming w (a wrong-pattern solution is often 1e-6 from the right one), else from the incoming w
            double ga = 0.0, g0 = 0.0;
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int j = wp_idx(k);
                if (j < ld) {
                    ga = fmax(ga, fabs(P.rho * (gwj[k] - qj[k]) + hub_g(wj[k], P.reg, P.t)));
                    g0 = fmax(g0, fabs(P.rho * (gw0j[k] - qj[k]) + hub_g(w0j[k], P.reg, P.t)));
                }
            }
            ga = block_max1<WP_THREADS>(ga, red);
            g0 = block_max1<WP_THREADS>(g0, red);
            if (!(ga < g0)) {
#pragma unroll
                for (int k = 0; k < PER; ++k) {
                    wj[k] = w0j[k];
                    gwj[k] = gw0j[k];
                }
            }
        }
    }
    if (ok && !done) {
        // start of the nonlinear CG at the current point
        double acc[1] = {0.0};
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int j = wp_idx(k);
            sj[k] = pj[k] = 0.0;
            if (j < ld) {
                const double g = P.rho * (gwj[k] - qj[k]) + hub_g(wj[k], P.reg, P.t);
                sj[k] = precond_inv(P.rho, gdj[k], hub_c(wj[k], P.reg, P.t)) * g;
                pj[k] = -sj[k];
                acc[0] += g * sj[k];
            }
        }
        rbl::block_sum<1, WP_THREADS>(acc, red);
        gs_old = acc[0];
    }
    while (ok && !done && iters < max_iter) {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int j = wp_idx(k);
            if (j < ld) xs[j] = pj[k];
        }
        __syncthreads();
        wp_matvec<GLDS>(G, gs, ld, r0, r1, xs, 1.0, 0.0, ybuf);      // G p
        ++xn;
        if (!wp_exchange<PER>(tag_base + xn, ybuf, (xn & 1) ? x0 : x1, ld, gpj, abort_word)) {
            ok = 0;
            break;
        }
        // exact line search: phi'(a) = a0 + a a1 + sum_j h'(w_j + a p_j) p_j, piecewise linear and increasing
        double a3[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const double lin = P.rho * (gwj[k] - qj[k]);
            a3[0] += lin * pj[k];
            a3[1] += pj[k] * gpj[k];
            a3[2] += fabs(lin * pj[k]) + fabs(hub_g(wj[k], P.reg, P.t) * pj[k]);
        }
        rbl::block_sum<3, WP_THREADS>(a3, red);
        const double a0 = a3[0], a1 = P.rho * a3[1], ftol = 4e-16 * a3[2];
        double alpha = 0.0, lo = 0.0, hi = -1.0;
        for (int it = 0; it < 100; ++it) {
            double e2[2] = {0.0, 0.0};
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const double u = wj[k] + alpha * pj[k];
                e2[0] += hub_g(u, P.reg, P.t) * pj[k];
                e2[1] += hub_c(u, P.reg, P.t) * pj[k] * pj[k];
            }
            rbl::block_sum<2, WP_THREADS>(e2, red);
            const double f = a0 + alpha * a1 + e2[0], slope = a1 + e2[1];
            if (it == 0 && !(f < 0.0)) break;
            if (fabs(f) <= ftol) break;
            if (f < 0.0) lo = alpha; else hi = alpha;
            double an = (slope > 0.0) ? alpha - f / slope : -1.0;
            if (hi >= 0.0) {
                if (!(an > lo && an < hi)) an = 0.5 * (lo + hi);
            } else if (!(an > lo)) {
                an = lo > 0.0 ? 2.0 * lo : 1.0;
            }
            if (an == alpha) break;
            alpha = an;
        }
        double b3[3] = {0.0, 0.0, 0.0};
        double gmax = 0.0;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int j = wp_idx(k);
            if (j < ld) {
                wj[k] += alpha * pj[k];
                gwj[k] += alpha * gpj[k];
                const double g = P.rho * (gwj[k] - qj[k]) + hub_g(wj[k], P.reg, P.t);
                const double sn = precond_inv(P.rho, gdj[k], hub_c(wj[k], P.reg, P.t)) * g;
                b3[0] += g * sn;
                b3[1] += g * sj[k];
                b3[2] += g * pj[k];
                sj[k] = sn;
                gmax = fmax(gmax, fabs(g));
            }
        }
        rbl::block_sum<3, WP_THREADS>(b3, red);
        gmax = block_max1<WP_THREADS>(gmax, red);
        double beta = (gs_old > 0.0) ? fmax(0.0, (b3[0] - b3[1]) / gs_old) : 0.0;
        if (alpha == 0.0 || -b3[0] + beta * b3[2] >= 0.0) beta = 0.0;
#pragma unroll
        for (int k = 0; k < PER; ++k) pj[k] = -sj[k] + beta * pj[k];
        gs_old = b3[0];
        ++iters;
        done = (gmax <= thr || (alpha == 0.0 && stalled != 0.0)) ? 1 : 0;
        stalled = (alpha == 0.0) ? 1.0 : 0.0;
    }
    if (ok && Gw_out) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int j = wp_idx(k);
            if (j < ld) xs[j] = wj[k];
        }
        __syncthreads();
        wp_matvec<GLDS>(G, gs, ld, r0, r1, xs, 1.0, 0.0, ybuf);
        __syncthreads();
        if (threadIdx.x < r1 - r0) Gw_out[r0 + threadIdx.x] = ybuf[threadIdx.x];
    }
    if (blockIdx.x == 0) {
        if (ok) {
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int j = wp_idx(k);
                if (j < ld) w[j] = wj[k];
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            pin[2] = phase_a;
            wp_publish(pin, ok ? done : -2, iters);
        }
    }
}

struct WpPlan {
    bool ok, glds;
    int nblocks, xbytes;
    size_t lds_bytes;
};

// launch shape of the persistent w-step: WP_RPB rows per block, G rows in LDS when they fit
WpPlan wp_plan(int64_t ld) {
    WpPlan p{false, false, 0, 0, 0};
    static const int enabled = [] {
        const char* e = getenv("RBL_WSTEP_PERSIST");
        return (e && e[0] == '0') ? 0 : 1;
    }();
    if (!enabled || ld > WSTEP_PERSIST_MAX_LD || ld < 4) return p;
    {
        // several handles of this process on one device (threads as ranks, the test rigs): two persistent kernels side by
        // side can each hold some CUs and wait for blocks the other keeps out - the batched launches serve there.
        // RBL_WSTEP_PERSIST=2 keeps the persistent form regardless (lab).
        static const int force = [] {
            const char* e = getenv("RBL_WSTEP_PERSIST");
            return (e && e[0] == '2') ? 1 : 0;
        }();
        int dev = 0;
        if (!force && (hipGetDevice(&dev) != hipSuccess || rbl_live_handles(dev) > 1)) return p;
    }
    static const int cus = [] {
        int dev = 0, c = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev);
        return c < 1 ? 1 : c;
    }();
    p.nblocks = (int)((ld + WP_RPB - 1) / WP_RPB);
    if (p.nblocks > cus) return p;   // at most one block per CU: all blocks resident at once (the device-wide wait needs that)
    const size_t fixed = sizeof(double) * ((size_t)ld + 16 + WP_RPB + 16);   // p | block sums | this block's rows of the product
    const size_t rows = sizeof(double) * (size_t)WP_RPB * (size_t)ld;
    p.glds = fixed + rows <= 156 * 1024;
    p.lds_bytes = fixed + (p.glds ? rows : 0);
    p.xbytes = p.nblocks * WP_RPB * (int)sizeof(double);
    p.ok = true;
    return p;
}

template <typename K>
int wp_set_lds(K kernel, size_t bytes) {
    if (bytes > 64 * 1024)
        RBL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return RBL_OK;
}

// returns RBL_OK with *status = 1 converged / 0 iteration cap / -1 launch failure / -2 a wait gave up
int run_cg_persist(const WpPlan& pl, const double* G, int64_t ld, const double* q, double rho, double reg, double tol,
                   int max_iter, double* w, WstepWorkspace& ws, bool want_Gw, int* status, int* iters, hipStream_t s) {
    int* pin = ws.pin + 4;
    pin[0] = -1;
    ws.launch_seq = (ws.launch_seq + 1) & 0xfffff;
    if (ws.launch_seq == 0) ws.launch_seq = 1;
    const unsigned tag_base = (unsigned)ws.launch_seq << 12;          // exchange tags of this launch: tag_base + 1 ...
    if (max_iter > 4000) max_iter = 4000;                             // (12 bits of exchange number)
    double* gw = want_Gw ? ws.Gy : nullptr;
    unsigned long long *x0 = reinterpret_cast<unsigned long long*>(ws.xch),
                       *x1 = reinterpret_cast<unsigned long long*>(ws.xch) + WSTEP_XCH_GRANULES;
    // RBL_WPERSIST_DEBUG=1: block 0 leaves 100 MHz wall-clock stamps (start | rows staged | first exchange | loop done |
    // G w written | published) behind the second exchange buffer; printed to stderr after the kernel
    static const bool dbg_on = [] {
        const char* e = getenv("RBL_WPERSIST_DEBUG");
        return e && e[0] == '1';
    }();
    long long* dbg = dbg_on ? reinterpret_cast<long long*>(ws.xch) + 2 * WSTEP_XCH_GRANULES : nullptr;
#define RBL_CG_PERSIST(GL, PR)                                                                                           \
    do {                                                                                                                 \
        static size_t lds_set = 0; /* (one attribute call per instantiation and size, not per launch) */                 \
        if (lds_set < pl.lds_bytes) {                                                                                    \
            RBL_TRY(wp_set_lds(k_cg_persist<GL, PR>, pl.lds_bytes));                                                     \
            lds_set = pl.lds_bytes;                                                                                      \
        }                                                                                                                \
        hipLaunchKernelGGL((k_cg_persist<GL, PR>), dim3(pl.nblocks), dim3(WP_THREADS), pl.lds_bytes, s, G, (int)ld, q, rho, \
                           reg, tol, max_iter, w, x0, x1, tag_base, gw, ws.bar, pin, dbg);                               \
    } while (0)
    const bool narrow = ld <= 4 * WP_THREADS;
    if (pl.glds && narrow) RBL_CG_PERSIST(true, 4);
    else if (pl.glds) RBL_CG_PERSIST(true, 8);
    else if (narrow) RBL_CG_PERSIST(false, 4);
    else RBL_CG_PERSIST(false, 8);
#undef RBL_CG_PERSIST
    RBL_HIP(hipGetLastError());
    rbl_spin_wait(pin, -1, s);
    const volatile int* st = pin;
    *status = st[0];
    *iters = st[1];
    if (dbg) {
        long long t[6];
        RBL_HIP(hipMemcpy(t, dbg, sizeof(t), hipMemcpyDeviceToHost));
        fprintf(stderr, "[rbl] k_cg_persist: %d iterations; staged %.2f us, first exchange %.2f, loop %.2f, G w %.2f, publish %.2f\n",
                st[1], (t[1] - t[0]) * 0.01, (t[2] - t[1]) * 0.01, (t[3] - t[2]) * 0.01, (t[4] - t[3]) * 0.01, (t[5] - t[4]) * 0.01);
    }
    return RBL_OK;
}

int run_ncg_persist(const WpPlan& pl, const double* G, int64_t ld, const double* q, NcgParams P, int max_iter, double* w,
                    WstepWorkspace& ws, bool want_Gw, int* status, int* iters, hipStream_t s) {
    int* pin = ws.pin + 8;
    pin[0] = -1;
    pin[2] = -1;
    if (ws.ncg_skip > 0) P.active = 0;
    ws.launch_seq = (ws.launch_seq + 1) & 0xfffff;
    if (ws.launch_seq == 0) ws.launch_seq = 1;
    const unsigned tag_base = (unsigned)ws.launch_seq << 12;
    if (max_iter > 4000) max_iter = 4000;
    double* gw = want_Gw ? ws.Gy : nullptr;
    unsigned long long *x0 = reinterpret_cast<unsigned long long*>(ws.xch),
                       *x1 = reinterpret_cast<unsigned long long*>(ws.xch) + WSTEP_XCH_GRANULES;
#define RBL_NCG_PERSIST(GL, PR)                                                                                          \
    do {                                                                                                                 \
        static size_t lds_set = 0;                                                                                       \
        if (lds_set < pl.lds_bytes) {                                                                                    \
            RBL_TRY(wp_set_lds(k_ncg_persist<GL, PR>, pl.lds_bytes));                                                    \
            lds_set = pl.lds_bytes;                                                                                      \
        }                                                                                                                \
        hipLaunchKernelGGL((k_ncg_persist<GL, PR>), dim3(pl.nblocks), dim3(WP_THREADS), pl.lds_bytes, s, G, (int)ld, q, P, \
                           max_iter, w, x0, x1, tag_base, gw, ws.bar, pin);                                              \
    } while (0)
    const bool narrow = ld <= 4 * WP_THREADS;
    if (pl.glds && narrow) RBL_NCG_PERSIST(true, 4);
    else if (pl.glds) RBL_NCG_PERSIST(true, 8);
    else if (narrow) RBL_NCG_PERSIST(false, 4);
    else RBL_NCG_PERSIST(false, 8);
#undef RBL_NCG_PERSIST
    RBL_HIP(hipGetLastError());
    rbl_spin_wait(pin, -1, s);
    const volatile int* st = pin;
    *status = st[0];
    *iters = st[1];
    // the linear first phase pays where the Huber pattern is stable between ADMM iterations (6M x 1000: 252 -> 270 it/s)
    // and costs its iterations where it is not (a few thousand rows: +10-40 % inner iterations if tried every time):
    // after a dropped attempt the next 1, 2, 4, ... 16 w-steps run the nonlinear CG alone
    if (st[2] == 1) {
        ws.ncg_backoff = 0;
    } else if (st[2] == 0) {
        ws.ncg_backoff = ws.ncg_backoff ? (ws.ncg_backoff < 16 ? 2 * ws.ncg_backoff : 16) : 1;
        ws.ncg_skip = ws.ncg_backoff;
    } else if (ws.ncg_skip > 0) {
        --ws.ncg_skip;
    }
    return RBL_OK;
}

bool ncg_persist_try(const double* G, int64_t ld, const double* q, NcgParams P, int max_iter, double* w,
                     WstepWorkspace& ws, bool want_Gw, int* status, int* iters, hipStream_t s, int* rc) {
    const WpPlan pl = wp_plan(ld);
    if (!pl.ok || !ws.bar || !ws.xch) return false;
    *rc = run_ncg_persist(pl, G, ld, q, P, max_iter, w, ws, want_Gw, status, iters, s);
    return true;
}

}  // namespace

int run_wstep(int wstep, const double* G, int64_t ld, const double* q, double rho, double reg, double smooth_t,
              double L, double tol, int max_inner, double* w, WstepWorkspace& ws, int* iters_host, hipStream_t s,
              bool* fs_pending, const double* rho_dev, double* w_prev_out, bool want_Gw) {
    if (fs_pending) *fs_pending = false;
    ws.gw_valid = false;
    ws.form = 0;
    if (ld > (long long)UPD_THREADS * UPD_PER) {
        rbl_set_error("w-step: d=%lld exceeds the single-block update limit %d", (long long)ld, UPD_THREADS * UPD_PER);
        return RBL_ERR_INVALID;
    }
    if (wstep == RBL_WSTEP_L2 && ws.eig_ok) {
        // (rho G + reg I) w = rho q through the one-time eigendecomposition of G (eig.hip): two mat-vecs for any rho,
        // plus one step of iterative refinement against G itself
        if (iters_host) *iters_host = 1;
        ws.form = 3;
        return launch_ridge_eig(G, ws.eig_Vt, ws.eig_V, ws.eig_lambda, ld, q, rho, reg, w, ws.Gy, ws.r, s);
    }
    if (wstep == RBL_WSTEP_L2) {
        const WpPlan pl = wp_plan(ld);
        if (pl.ok && ws.bar && ws.xch) {
            // the whole CG in one persistent launch (k_cg_persist)
            int status = 0, it = 0;
            RBL_TRY(run_cg_persist(pl, G, ld, q, rho, reg, tol, max_inner < 20000 ? max_inner : 20000, w, ws, want_Gw, &status,
                                   &it, s));
            if (status < 0) {
                rbl_set_error("w-step: the persistent CG kernel did not complete (status %d)", status);
                return RBL_ERR_HIP;
            }
            if (status == 1) ws.last_iters = it;
            ws.gw_valid = want_Gw;
            ws.form = 1;
            if (iters_host) *iters_host = it;
            return RBL_OK;
        }
        // (rho G + reg I) w = rho q.  The warm-started CG needs about as many iterations as last
        // time: one batch of that many (+2) is enqueued, its last update publishes (done,
        // iterations) to pinned memory and the host spins on that word - one host round trip per
        // w-step and no device-to-host copy; updates after convergence are no-ops (device flag).
        const unsigned sg = symv_grid(ld);
        hipLaunchKernelGGL(k_symv, dim3(sg), dim3(256), 0, s, G, (long long)ld, w, ws.Gy, rho, reg, (const int*)nullptr);
        hipLaunchKernelGGL(k_cg_init, dim3(1), dim3(UPD_THREADS), 0, s, (long long)ld, ws.Gy, q, rho, ws.r, ws.p, tol,
                           ws.scal, ws.flags);
        int batch = ws.last_iters > 0 ? ws.last_iters + 2 : 16;
        if (batch < 4) batch = 4;
        if (batch > 64) batch = 64;
        int done_iters = 0, done = 0, iters = 0;
        while (done_iters < max_inner) {
            ws.pin[4] = -1;
            for (int b = 0; b < batch; ++b) {
                hipLaunchKernelGGL(k_symv, dim3(sg), dim3(256), 0, s, G, (long long)ld, ws.p, ws.Gy, rho, reg, ws.flags);
                hipLaunchKernelGGL(k_cg_update, dim3(1), dim3(UPD_THREADS), 0, s, (long long)ld, ws.Gy, w, ws.r, ws.p,
                                   ws.scal, ws.flags, b == batch - 1 ? ws.pin + 4 : (int*)nullptr);
            }
            RBL_HIP(hipGetLastError());
            done_iters += batch;
            rbl_spin_wait(ws.pin + 4, -1, s);
            const volatile int* st = ws.pin + 4;
            done = st[0];
            iters = st[1];
            if (done != 0) break;   // 1 = converged (-1 after the spin's stream wait: launch failure, stop too)
            batch = 8;
        }
        if (done == 1) ws.last_iters = iters;
        if (iters_host) *iters_host = iters;
        return done == -1 ? RBL_ERR_HIP : RBL_OK;
    }
    if (wstep == RBL_WSTEP_L1) {
        // exact active-set solve first (lasso_fs.hip); FISTA only when the support does not fit
        // its capacity (e.g. the dense initial w of algorithms.py:42) or it hits its cap.  The
        // kernel writes its status block straight into pinned host memory.
        ws.pin[0] = -1;   // sentinel: the kernel stores its status (>= 0) here last
        ws.form = 2;
        RBL_TRY(launch_lasso_fs(G, ld, ld, q, w, reg / (2.0 * rho), ws.pin, s, rho_dev, reg, w_prev_out,
                                want_Gw ? ws.Gy : nullptr));
        if (fs_pending) {
            *fs_pending = true;
            return RBL_OK;
        }
        bool fell_back = false;
        return finish_wstep_l1(G, ld, q, rho, reg, L, tol, max_inner, w, ws, iters_host, s, &fell_back);
    }
    static const bool smooth_fista = [] {      // RBL_SMOOTH_FISTA=1: the round-1 solver, for comparisons
        const char* e = getenv("RBL_SMOOTH_FISTA");
        return e && e[0] == '1';
    }();
    if (smooth_fista) return run_fista(1, G, ld, q, rho, reg, smooth_t, L, tol, max_inner, w, ws, iters_host, s);
    return run_ncg(G, ld, q, rho, reg, smooth_t, L, tol, max_inner, w, ws, iters_host, s, want_Gw);
}

int finish_wstep_l1(const double* G, int64_t ld, const double* q, double rho, double reg, double L, double tol,
                    int max_inner, double* w, WstepWorkspace& ws, int* iters_host, hipStream_t s, bool* fell_back) {
    rbl_spin_wait(ws.pin, -1, s);
    const volatile int* st = ws.pin;
    if (st[0] == 0) {
        if (iters_host) *iters_host = st[1];
        if (fell_back) *fell_back = false;
        return RBL_OK;
    }
    if (fell_back) *fell_back = true;
    ws.form = 0;
    return run_fista(0, G, ld, q, rho, reg, 1.0, L, tol, max_inner, w, ws, iters_host, s);
}

int launch_symv(const double* G, int64_t ld, const double* x, double* y, hipStream_t s) {
    hipLaunchKernelGGL(k_symv, dim3(symv_grid(ld)), dim3(256), 0, s, G, (long long)ld, x, y, 1.0, 0.0, (const int*)nullptr);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}

int launch_symv_ab(const double* G, int64_t ld, const double* x, double* y, double alpha, double beta, hipStream_t s) {
    hipLaunchKernelGGL(k_symv, dim3(symv_grid(ld)), dim3(256), 0, s, G, (long long)ld, x, y, alpha, beta, (const int*)nullptr);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}

int launch_reg_terms(int64_t d, const double* w, double* out2, hipStream_t s) {
    hipLaunchKernelGGL(k_reg_terms, dim3(1), dim3(UPD_THREADS), 0, s, (long long)d, w, out2);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}

int launch_w_stats(int64_t d, const double* w, const double* w_prev, double* out3, hipStream_t s) {
    hipLaunchKernelGGL(k_w_stats, dim3(1), dim3(UPD_THREADS), 0, s, (long long)d, w, w_prev, out3);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}

int launch_soft_threshold(int64_t d, double* w, double t, hipStream_t s) {
    hipLaunchKernelGGL(k_soft_threshold, dim3(16), dim3(256), 0, s, (long long)d, w, t);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}
