// lasso_fs.hip - exact active-set solver for the l1 w-step
//     min_w  1/2 w'Gw - q'w + kappa ||w||_1        (kappa = reg / (2 rho))
// (reference: src/optim/algorithms.py:190-202 -> backtracking FISTA in n-space,
// src/util/fast_lasso.py:22-69).  FISTA in d-space needs hundreds of d x d mat-vecs per
// ADMM iteration when G is ill conditioned (the synthetic data has exactly collinear
// redundant columns), while the lasso support is tiny (26 of 1000 at BASELINE config C1)
// and changes by a coordinate or two between ADMM iterations.  This kernel runs the
// feature-sign search (Lee, Battle, Raina, Ng, NIPS 2006) warm-started from the previous
// w, entirely inside ONE workgroup:
//   outer: g = G w - q from the active columns only; if the largest inactive |g_i| exceeds
//          kappa, activate i with sign -sign(g_i); else done (KKT holds).
//   inner: solve G_AA x = q_A - kappa*theta (Cholesky in LDS + iterative refinement against
//          the unfactored matrix kept in the upper triangle), exact line search along
//          w_A -> x over the points where a coefficient changes sign, drop coefficients that
//          hit zero, repeat until the active-set KKT residual vanishes.
// Finite termination at the exact minimiser.  If the active set outgrows FS_MAX or the
// iteration cap is hit, the kernel reports it and the host continues with FISTA from the
// current (feasible, improved) point - see run_wstep() in wstep.hip.
#include "rbl_internal.h"

namespace {

constexpr int FS_THREADS = 256;
constexpr int FS_MAX = 96;          // active-set capacity
constexpr int FS_S = FS_MAX + 1;    // LDS row stride (odd: no bank-aligned columns)
constexpr int FS_MAXD = 16384;

struct FsShared {
    double M[FS_MAX * FS_S];  // lower triangle: Cholesky factor; strict upper: original G_AA
    double diag[FS_MAX];      // original diagonal of G_AA
    double wA[FS_MAX], x[FS_MAX], theta[FS_MAX], rhs[FS_MAX], gq[FS_MAX], y[FS_MAX], rr[FS_MAX], tc[FS_MAX + 1],
        fc[FS_MAX + 1];
    int A[FS_MAX];
    double red_v[FS_THREADS / 64];
    int red_i[FS_THREADS / 64];
    double bcast[4];
    int ibcast[4];
    int scan[FS_THREADS];
    unsigned char active[FS_MAXD];
};

__device__ inline double sgn(double v) { return v > 0.0 ? 1.0 : (v < 0.0 ? -1.0 : 0.0); }

// block-wide (max value, smallest index) reduction; result valid in every thread
__device__ inline void block_argmax(double& v, int& idx, FsShared& S) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        double ov = __shfl_xor(v, off, 64);
        int oi = __shfl_xor(idx, off, 64);
        if (ov > v || (ov == v && oi < idx)) {
            v = ov;
            idx = oi;
        }
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        S.red_v[threadIdx.x >> 6] = v;
        S.red_i[threadIdx.x >> 6] = idx;
    }
    __syncthreads();
    v = S.red_v[0];
    idx = S.red_i[0];
    for (int k = 1; k < FS_THREADS / 64; ++k)
        if (S.red_v[k] > v || (S.red_v[k] == v && S.red_i[k] < idx)) {
            v = S.red_v[k];
            idx = S.red_i[k];
        }
    __syncthreads();
}

__device__ inline double block_max(double v, FsShared& S) {
    int dummy = 0;
    block_argmax(v, dummy, S);
    return v;
}

// x <- solution of (L L') x = b with L in the lower triangle of S.M; b is consumed
__device__ inline void chol_solve(FsShared& S, int na, double* b, double* xo) {
    const int tid = threadIdx.x;
    for (int k = 0; k < na; ++k) {  // forward: L y = b
        __syncthreads();
        const double yk = b[k] / S.M[k * FS_S + k];
        for (int i = k + 1 + tid; i < na; i += FS_THREADS) b[i] -= S.M[i * FS_S + k] * yk;
        __syncthreads();
        if (tid == 0) S.y[k] = yk;
    }
    __syncthreads();
    for (int k = na - 1; k >= 0; --k) {  // backward: L' x = y
        __syncthreads();
        const double xk = S.y[k] / S.M[k * FS_S + k];
        for (int i = tid; i < k; i += FS_THREADS) S.y[i] -= S.M[k * FS_S + i] * xk;
        __syncthreads();
        if (tid == 0) xo[k] = xk;
    }
    __syncthreads();
}

// out[0] = status (0 converged, 1 active set overflow, 2 iteration cap), out[1] = outer
// iterations, out[2] = inner (feature-sign) steps, out[3] = final support size
__global__ __launch_bounds__(FS_THREADS) void k_lasso_fs(const double* __restrict__ G, long long ld, long long d,
                                                          const double* __restrict__ q, double* __restrict__ w,
                                                          double kappa_val, const double* __restrict__ rho_dev,
                                                          double reg, double* __restrict__ w_prev_out,
                                                          double* __restrict__ Gw_out, int max_steps,
                                                          int* __restrict__ out) {
    extern __shared__ __align__(16) unsigned char fs_raw[];
    FsShared& S = *reinterpret_cast<FsShared*>(fs_raw);
    const int tid = threadIdx.x;
    const int nd = (int)d;
    // kappa = reg / (2 rho) with rho read on the device when the launch was enqueued before the
    // host knew it (the w-step of the next iteration, api.hip: rbl_phase_finish)
    const double kappa = rho_dev ? reg / (2.0 * rho_dev[0]) : kappa_val;
    if (w_prev_out)   // the warm start is the previous iterate: keep it for the dual residual
        for (long long j = tid; j < ld; j += FS_THREADS) w_prev_out[j] = w[j];

    // ---- initial active set = support of the warm start, in index order (deterministic)
    const int per = (nd + FS_THREADS - 1) / FS_THREADS;
    int cnt = 0;
    double qmax = 0.0;
    for (int j = tid * per; j < (tid + 1) * per && j < nd; ++j) {
        const bool nz = w[j] != 0.0;
        S.active[j] = nz ? 1 : 0;
        cnt += nz;
    }
    for (int j = tid; j < nd; j += FS_THREADS) qmax = fmax(qmax, fabs(q[j]));
    S.scan[tid] = cnt;
    __syncthreads();
    for (int off = 1; off < FS_THREADS; off <<= 1) {
        int add = tid >= off ? S.scan[tid - off] : 0;
        __syncthreads();
        S.scan[tid] += add;
        __syncthreads();
    }
    int na = S.scan[FS_THREADS - 1];
    const double scale = fmax(block_max(qmax, S), kappa);
    int status = 0, outer = 0, steps = 0;
    if (na > FS_MAX) {
        if (tid == 0) {
            out[1] = 0;
            out[2] = 0;
            out[3] = na;
            __threadfence_system();
            out[0] = 1;   // written last: the host polls this word (pinned memory)
        }
        return;
    }
    {
        int pos = S.scan[tid] - cnt;
        for (int j = tid * per; j < (tid + 1) * per && j < nd; ++j)
            if (S.active[j]) {
                S.A[pos] = j;
                S.wA[pos] = w[j];
                S.theta[pos] = sgn(w[j]);
                ++pos;
            }
    }
    __syncthreads();
    bool need_step = na > 0;
    const double kkt_tol = 1e-12 * scale;

    while (true) {
        ++outer;
        if (!need_step) {
            // ---- activation test: largest |g_i| over inactive coordinates
            double best = -1.0;
            int besti = 0x7fffffff;
            double bestg = 0.0;
            for (int i = tid; i < nd; i += FS_THREADS) {
                if (S.active[i]) continue;
                double g = -q[i];
                for (int a = 0; a < na; ++a) g = __builtin_fma(G[(long long)S.A[a] * ld + i], S.wA[a], g);
                if (Gw_out) Gw_out[i] = g + q[i];   // (G w)_i: final once this test lets the loop end
                const double ag = fabs(g);
                if (ag > best) {  // ascending i: keeps the smallest index among equals
                    best = ag;
                    besti = i;
                    bestg = g;
                }
            }
            double v = best;
            int idx = besti;
            block_argmax(v, idx, S);
            if (idx == besti && best == v && tid == (besti % FS_THREADS)) S.bcast[0] = bestg;
            __syncthreads();
            if (!(v > kappa * (1.0 + 1e-12) + 1e-14 * scale)) break;  // KKT holds everywhere: done
            if (na >= FS_MAX) {
                status = 1;
                break;
            }
            if (tid == 0) {
                S.A[na] = idx;
                S.wA[na] = 0.0;
                S.theta[na] = -sgn(S.bcast[0]);
                S.active[idx] = 1;
            }
            ++na;
            __syncthreads();
        }
        need_step = false;

        // ---- feature-sign steps on the current active set
        bool inner_done = false;
        while (!inner_done) {
            if (++steps > max_steps) {
                status = 2;
                break;
            }
            // gather G_AA (both triangles), its diagonal, gq = G_AA wA - q_A and the right-hand side
            for (int e = tid; e < na * na; e += FS_THREADS) {
                const int a = e / na, b = e - a * na;
                S.M[a * FS_S + b] = G[(long long)S.A[a] * ld + S.A[b]];
            }
            __syncthreads();
            double dmax = 0.0;
            for (int a = tid; a < na; a += FS_THREADS) {
                S.diag[a] = S.M[a * FS_S + a];
                double acc = -q[S.A[a]];
                for (int b = 0; b < na; ++b) acc = __builtin_fma(S.M[a * FS_S + b], S.wA[b], acc);
                S.gq[a] = acc;
                S.rhs[a] = q[S.A[a]] - kappa * S.theta[a];
                S.rr[a] = S.rhs[a];
            }
            for (int a = tid; a < na; a += FS_THREADS) dmax = fmax(dmax, S.M[a * FS_S + a]);
            dmax = block_max(dmax, S);
            // Cholesky of G_AA + jitter*I (right-looking) in the lower triangle; the strict upper
            // triangle and S.diag keep the unfactored matrix for the refinement below
            const double jitter = 1e-13 * dmax;
            for (int a = tid; a < na; a += FS_THREADS) S.M[a * FS_S + a] += jitter;
            for (int k = 0; k < na; ++k) {
                __syncthreads();
                double piv = S.M[k * FS_S + k];
                if (!(piv > 1e-14 * dmax)) piv = 1e-14 * dmax;  // numerically dependent column
                const double lkk = sqrt(piv);
                __syncthreads();  // every thread has read the pivot before it is overwritten
                for (int i = k + 1 + tid; i < na; i += FS_THREADS) S.M[i * FS_S + k] /= lkk;
                if (tid == 0) S.M[k * FS_S + k] = lkk;
                __syncthreads();
                const int mrem = na - 1 - k;
                for (int e = tid; e < mrem * mrem; e += FS_THREADS) {
                    const int ii = e / mrem, jj = e - ii * mrem;
                    if (jj <= ii) {
                        const int i = k + 1 + ii, j = k + 1 + jj;
                        S.M[i * FS_S + j] -= S.M[i * FS_S + k] * S.M[j * FS_S + k];
                    }
                }
            }
            __syncthreads();
            chol_solve(S, na, S.rr, S.x);
            // iterative refinement against the unfactored G_AA (strict upper triangle + diag)
            for (int pass = 0; pass < 2; ++pass) {
                for (int a = tid; a < na; a += FS_THREADS) {
                    double acc = S.rhs[a] - S.diag[a] * S.x[a];
                    for (int b = 0; b < na; ++b)
                        if (b != a) acc -= (b > a ? S.M[a * FS_S + b] : S.M[b * FS_S + a]) * S.x[b];
                    S.rr[a] = acc;
                }
                __syncthreads();
                chol_solve(S, na, S.rr, S.y);  // xo aliases S.y: entry k is written after its last read
                for (int a = tid; a < na; a += FS_THREADS) S.x[a] += S.y[a];
                __syncthreads();
            }
            // ---- exact line search on the segment wA -> x
            // phi(t) = 1/2 aa t^2 + bb t + kappa sum|wA + t delta|, with G_AA delta = -(kappa theta + gq)
            double aa = 0.0, bb = 0.0;
            if (tid == 0) {
                for (int a = 0; a < na; ++a) {
                    const double del = S.x[a] - S.wA[a];
                    aa += del * (-(kappa * S.theta[a] + S.gq[a]));
                    bb += S.gq[a] * del;
                }
                S.bcast[1] = aa;
                S.bcast[2] = bb;
            }
            __syncthreads();
            aa = S.bcast[1];
            bb = S.bcast[2];
            for (int c = tid; c <= na; c += FS_THREADS) {
                double t = 1.0;
                bool valid = true;
                if (c < na) {
                    const double wa = S.wA[c], xa = S.x[c];
                    valid = (wa != 0.0) && (xa * wa < 0.0);
                    t = valid ? wa / (wa - xa) : 2.0;
                }
                double f = 1e300;
                if (valid) {
                    double l1 = 0.0;
                    for (int a = 0; a < na; ++a) l1 += fabs(S.wA[a] + t * (S.x[a] - S.wA[a]));
                    f = 0.5 * aa * t * t + bb * t + kappa * l1;
                }
                S.tc[c] = t;
                S.fc[c] = f;
            }
            __syncthreads();
            if (tid == 0) {
                int bc = na;  // t = 1
                for (int c = 0; c < na; ++c)
                    if (S.fc[c] < S.fc[bc]) bc = c;
                const double t = S.tc[bc];
                int keep = 0;
                for (int a = 0; a < na; ++a) {
                    double nv = S.wA[a] + t * (S.x[a] - S.wA[a]);
                    if (a < na && S.tc[a] == t && bc != na) nv = 0.0;  // the coefficient(s) that cross at t
                    if (a == bc) nv = 0.0;
                    const int j = S.A[a];
                    if (nv != 0.0) {
                        S.A[keep] = j;
                        S.wA[keep] = nv;
                        S.theta[keep] = sgn(nv);
                        ++keep;
                    } else {
                        S.active[j] = 0;
                        w[j] = 0.0;
                    }
                }
                S.ibcast[0] = keep;
            }
            __syncthreads();
            na = S.ibcast[0];
            if (na == 0) break;
            // ---- KKT residual on the active set: |g_a + kappa sign(w_a)|
            double res = 0.0;
            for (int a = tid; a < na; a += FS_THREADS) {
                double g = -q[S.A[a]];
                for (int b = 0; b < na; ++b) g = __builtin_fma(G[(long long)S.A[a] * ld + S.A[b]], S.wA[b], g);
                res = fmax(res, fabs(g + kappa * S.theta[a]));
            }
            res = block_max(res, S);
            inner_done = res <= kkt_tol;
        }
        if (status != 0) break;
        if (outer > 4 * FS_MAX + 64) {
            status = 2;
            break;
        }
    }
    __syncthreads();
    for (int a = tid; a < na; a += FS_THREADS) w[S.A[a]] = S.wA[a];
    // G w of the solution for the caller's rho prediction (saves a d x d product): the inactive
    // coordinates were written by the last activation test, the active ones are added here
    if (Gw_out && status == 0) {
        for (int a = tid; a < na; a += FS_THREADS) {
            double g = 0.0;
            for (int b = 0; b < na; ++b) g = __builtin_fma(G[(long long)S.A[a] * ld + S.A[b]], S.wA[b], g);
            Gw_out[S.A[a]] = g;
        }
        for (long long j = nd + tid; j < ld; j += FS_THREADS) Gw_out[j] = 0.0;   // padding columns
    }
    if (tid == 0) {
        out[1] = outer;
        out[2] = steps;
        out[3] = na;
        __threadfence_system();
        out[0] = status;   // written last: the host polls this word (pinned memory)
    }
}

}  // namespace


int launch_lasso_fs(const double* G, int64_t ld, int64_t d, const double* q, double* w, double kappa, int* out_dev,
                    hipStream_t s, const double* rho_dev, double reg, double* w_prev_out, double* Gw_out) {
    if (d > FS_MAXD) {
        rbl_set_error("lasso_fs: d too large");
        return RBL_ERR_INVALID;
    }
    static bool attr_set = false;
    if (!attr_set) {
        RBL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lasso_fs),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(FsShared)));
        attr_set = true;
    }
    hipLaunchKernelGGL(k_lasso_fs, dim3(1), dim3(FS_THREADS), sizeof(FsShared), s, G, (long long)ld, (long long)d, q, w,
                       kappa, rho_dev, reg, w_prev_out, Gw_out, 6 * FS_MAX + 64, out_dev);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}
