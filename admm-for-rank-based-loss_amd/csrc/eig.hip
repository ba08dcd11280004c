// eig.hip - one-time eigendecomposition of the Gram matrix for the l2 w-step.
//
// The ridge w-step (reference: SciPy L-BFGS-B on rho/2 ||D w - b||^2 + reg/2 ||w||^2,
// src/util/w_LBFGS.py:31-53) is the linear system (rho G + reg I) w = rho q with a NEW rho every ADMM
// iteration (src/optim/algorithms.py:154-157), so a factorisation of the system matrix cannot be kept - but
// G = D^T D (algorithms.py:24) never changes, and with G = V diag(lambda) V^T
//     w = V diag(rho / (rho lambda_j + reg)) V^T q
// is two d x d mat-vecs for ANY rho (plus one step of iterative refinement against the unfactored matrix):
// 5 small launches instead of the 28 of a 14-iteration warm-started CG (152 -> 30 us per iteration at d = 1000).
// OPT-IN (RBL_RIDGE_EIG=1), not the default: measured on MI355X (round 2) it lifts C2sq from 109.2 to 112.4 it/s,
// C3 from 66.6 to 68.8 and C4shard from 100.3 to 103.5, but the one-row-pair-per-block Jacobi below needs
// ~0.8 s of setup at d = 1000 - the 0.13 ms it saves per iteration pay that back after 6000 iterations, and a
// solve runs a few hundred.  A blocked Jacobi (64 x 64 sub-problems in LDS, block rotations as small GEMMs:
// 31 rounds per sweep instead of 999) would be the way to make it the default.
//
// One-sided (Hestenes) Jacobi on B = G V, stored transposed so that the vectors being rotated are contiguous
// rows: Bt = V^T G starts as G (symmetric), Vt = V^T as I; rotating rows p, q of both until the rows of Bt are
// mutually orthogonal makes V^T G^2 V diagonal, i.e. the rows of Vt are eigenvectors of G.  V stays orthogonal
// to rounding whatever G looks like (G is singular here: the synthetic data has exactly collinear columns);
// rows of Bt that have shrunk to rounding level (the null space) are left alone.  Round-robin pair schedule:
// m/2 independent pairs per launch, m-1 launches per sweep, a handful of sweeps (quadratic convergence);
// everything is fixed-order and deterministic, so ranks that hold the same G get the same bits.
#include "rbl_internal.h"
#include "device_math.h"

#include <cstring>
#include <vector>

namespace {

constexpr int EJ_THREADS = 256;

__global__ __launch_bounds__(EJ_THREADS) void k_eig_init(const double* __restrict__ G, long long ld, double* __restrict__ Bt,
                                                          double* __restrict__ Vt) {
    const long long total = ld * ld;
    for (long long i = (long long)blockIdx.x * EJ_THREADS + threadIdx.x; i < total; i += (long long)gridDim.x * EJ_THREADS) {
        Bt[i] = G[i];
        Vt[i] = (i / ld == i % ld) ? 1.0 : 0.0;
    }
}

// pair i of round r of the circle method on m (even) players: player m-1 stays, the others rotate
__device__ inline void ej_pair(int m, int r, int i, int& p, int& q) {
    const int mm = m - 1;
    if (i == 0) {
        p = mm;
        q = r % mm;
    } else {
        p = (r + i) % mm;
        q = (r - i + mm) % mm;
    }
    if (p > q) {
        const int t = p;
        p = q;
        q = t;
    }
}

// one block per pair: rotate rows p, q of Bt (and of Vt) so that the two rows of Bt become orthogonal.
// offmax: largest |cos| between two rows met in this sweep (as the bits of a non-negative double, atomicMax)
__global__ __launch_bounds__(EJ_THREADS) void k_eig_step(double* __restrict__ Bt, double* __restrict__ Vt, long long ld, int d,
                                                          int m, int round, double tiny2, unsigned long long* __restrict__ offmax) {
    __shared__ double smem[3 * EJ_THREADS / 64];
    int p, q;
    ej_pair(m, round, blockIdx.x, p, q);
    if (q >= d) return;                       // padding rows / the dummy player of an odd d
    double* bp = Bt + (long long)p * ld;
    double* bq = Bt + (long long)q * ld;
    double acc[3] = {0.0, 0.0, 0.0};
    for (long long j = threadIdx.x; j < ld; j += EJ_THREADS) {
        const double x = bp[j], y = bq[j];
        acc[0] += x * x;
        acc[1] += y * y;
        acc[2] += x * y;
    }
    rbl::block_sum<3, EJ_THREADS>(acc, smem);
    const double a = acc[0], b = acc[1], g = acc[2];
    if (!(a > tiny2) || !(b > tiny2)) return;  // a null-space row: nothing to orthogonalise against
    const double cosv = fabs(g) / sqrt(a * b);
    if (threadIdx.x == 0) atomicMax(offmax, (unsigned long long)__double_as_longlong(cosv));
    if (!(cosv > 1e-16)) return;
    const double zeta = (b - a) / (2.0 * g);
    const double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
    const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
    double* vp = Vt + (long long)p * ld;
    double* vq = Vt + (long long)q * ld;
    for (long long j = threadIdx.x; j < ld; j += EJ_THREADS) {
        const double x = bp[j], y = bq[j];
        bp[j] = c * x - s * y;
        bq[j] = s * x + c * y;
        const double u = vp[j], v = vq[j];
        vp[j] = c * u - s * v;
        vq[j] = s * u + c * v;
    }
}

// lambda_j = v_j . (G v_j) = row j of Vt . row j of Bt (Rayleigh quotient; |v_j| = 1), and V = Vt^T
__global__ __launch_bounds__(EJ_THREADS) void k_eig_finish(const double* __restrict__ Bt, const double* __restrict__ Vt,
                                                            long long ld, int d, double* __restrict__ lambda,
                                                            double* __restrict__ V) {
    __shared__ double smem[EJ_THREADS / 64];
    const int j = blockIdx.x;
    double acc[1] = {0.0};
    for (long long k = threadIdx.x; k < ld; k += EJ_THREADS) {
        const double v = Vt[(long long)j * ld + k];
        acc[0] += v * Bt[(long long)j * ld + k];
        V[k * ld + j] = v;
    }
    rbl::block_sum<1, EJ_THREADS>(acc, smem);
    if (threadIdx.x == 0) lambda[j] = (j < d && acc[0] > 0.0) ? acc[0] : 0.0;
}

// out_i = (row i of Vt) . (a1 x1 + a2 x2) / (rho lambda_i + reg): the coefficients of the solution of
// (rho G + reg I) y = a1 x1 + a2 x2 in the eigenbasis.  One row per wave.
__global__ __launch_bounds__(256) void k_eig_coef(const double* __restrict__ Vt, long long ld, const double* __restrict__ x1,
                                                   double a1, const double* __restrict__ x2, double a2,
                                                   const double* __restrict__ lambda, double rho, double reg,
                                                   double* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long long row = (long long)blockIdx.x * 4 + wave; row < ld; row += (long long)gridDim.x * 4) {
        const double* v = Vt + row * ld;
        double acc = 0.0;
        for (long long j = lane; j < ld; j += 64) {
            double x = a1 * x1[j];
            if (x2) x += a2 * x2[j];
            acc = __builtin_fma(v[j], x, acc);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (lane == 0) out[row] = acc / (rho * lambda[row] + reg);
    }
}

// w (+)= V c
__global__ __launch_bounds__(256) void k_eig_apply(const double* __restrict__ V, long long ld, const double* __restrict__ c,
                                                    double* __restrict__ w, int accumulate) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long long row = (long long)blockIdx.x * 4 + wave; row < ld; row += (long long)gridDim.x * 4) {
        const double* v = V + row * ld;
        double acc = 0.0;
        for (long long j = lane; j < ld; j += 64) acc = __builtin_fma(v[j], c[j], acc);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (lane == 0) w[row] = accumulate ? w[row] + acc : acc;
    }
}

inline unsigned ej_rows_grid(long long ld) {
    long long g = (ld + 3) / 4;
    if (g > 4096) g = 4096;
    return (unsigned)g;
}

}  // namespace

// G (ld x ld, symmetric PSD, rows / columns >= d zero) -> Vt (rows = eigenvectors), V (its transpose), lambda.
// Bt is ld x ld scratch.  Returns the number of sweeps in *sweeps_out (<= 0: not converged - the caller keeps CG).
int launch_eig_jacobi(const double* G, int64_t ld, int64_t d, double* Bt, double* Vt, double* V, double* lambda,
                      unsigned long long* offmax_dev, hipStream_t s, int* sweeps_out) {
    if (sweeps_out) *sweeps_out = 0;
    const int m = (int)((d + 1) & ~1LL);      // even number of players; an odd d plays against a dummy (index d)
    if (m < 2) {
        // d = 1: G is its own decomposition
        hipLaunchKernelGGL(k_eig_init, dim3(1), dim3(EJ_THREADS), 0, s, G, (long long)ld, Bt, Vt);
        hipLaunchKernelGGL(k_eig_finish, dim3((unsigned)ld), dim3(EJ_THREADS), 0, s, Bt, Vt, (long long)ld, (int)d, lambda, V);
        RBL_HIP(hipGetLastError());
        if (sweeps_out) *sweeps_out = 1;
        return RBL_OK;
    }
    hipLaunchKernelGGL(k_eig_init, dim3(1024), dim3(EJ_THREADS), 0, s, G, (long long)ld, Bt, Vt);
    // rows whose squared norm is below tiny2 belong to the null space: (eps * ||G||_F)^2 bounds what rounding
    // alone leaves in a row of G V; the Frobenius norm is bounded by d * max diagonal for a PSD matrix
    std::vector<double> diag((size_t)d);
    RBL_HIP(hipMemcpy2DAsync(diag.data(), sizeof(double), G, sizeof(double) * (size_t)(ld + 1), sizeof(double), (size_t)d,
                             hipMemcpyDeviceToHost, s));
    RBL_HIP(hipStreamSynchronize(s));
    double dmax = 0.0;
    for (double x : diag) dmax = x > dmax ? x : dmax;
    const double tiny = 1e-13 * dmax * (double)d;
    const double tiny2 = tiny * tiny;
    for (int sweep = 1; sweep <= 30; ++sweep) {
        RBL_HIP(hipMemsetAsync(offmax_dev, 0, sizeof(unsigned long long), s));
        for (int r = 0; r < m - 1; ++r)
            hipLaunchKernelGGL(k_eig_step, dim3((unsigned)(m / 2)), dim3(EJ_THREADS), 0, s, Bt, Vt, (long long)ld, (int)d, m, r,
                               tiny2, offmax_dev);
        RBL_HIP(hipGetLastError());
        unsigned long long bits = 0;
        RBL_HIP(hipMemcpyAsync(&bits, offmax_dev, sizeof(bits), hipMemcpyDeviceToHost, s));
        RBL_HIP(hipStreamSynchronize(s));
        double off;
        memcpy(&off, &bits, sizeof(off));
        if (off <= 1e-14) {
            hipLaunchKernelGGL(k_eig_finish, dim3((unsigned)ld), dim3(EJ_THREADS), 0, s, Bt, Vt, (long long)ld, (int)d, lambda,
                               V);
            RBL_HIP(hipGetLastError());
            RBL_HIP(hipStreamSynchronize(s));
            if (sweeps_out) *sweeps_out = sweep;
            return RBL_OK;
        }
    }
    return RBL_OK;   // *sweeps_out stays 0: no decomposition, the caller keeps the CG w-step
}

// (rho G + reg I) w = rho q through the decomposition, one step of iterative refinement against G itself.
// tmp1, tmp2: ld doubles each.
int launch_ridge_eig(const double* G, const double* Vt, const double* V, const double* lambda, int64_t ld, const double* q,
                     double rho, double reg, double* w, double* tmp1, double* tmp2, hipStream_t s) {
    const unsigned g = ej_rows_grid(ld);
    hipLaunchKernelGGL(k_eig_coef, dim3(g), dim3(256), 0, s, Vt, (long long)ld, q, rho, (const double*)nullptr, 0.0, lambda, rho,
                       reg, tmp1);
    hipLaunchKernelGGL(k_eig_apply, dim3(g), dim3(256), 0, s, V, (long long)ld, tmp1, w, 0);
    // residual r = rho q - (rho G + reg I) w, correction through the same decomposition
    RBL_TRY(launch_symv_ab(G, ld, w, tmp2, rho, reg, s));
    hipLaunchKernelGGL(k_eig_coef, dim3(g), dim3(256), 0, s, Vt, (long long)ld, q, rho, tmp2, -1.0, lambda, rho, reg, tmp1);
    hipLaunchKernelGGL(k_eig_apply, dim3(g), dim3(256), 0, s, V, (long long)ld, tmp1, w, 1);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}
