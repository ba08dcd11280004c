// device_math.h - per-lane math shared by the element-wise, PAV and objective kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace rbl {

// exp(x)/(1+exp(x)) and its derivative without overflow
// (reference: safe_1divexp / safe_expdivexp2, src/util/individual_solver.py:44-49,72-76)
__device__ inline void sigmoid2(double x, double& s, double& ds) {
    double e = exp(-fabs(x));
    double inv = 1.0 / (1.0 + e);
    s = (x > 0.0) ? inv : e * inv;
    ds = e * inv * inv;
}
__device__ inline double sigmoid1(double x) {
    double e = exp(-fabs(x));
    double inv = 1.0 / (1.0 + e);
    return (x > 0.0) ? inv : e * inv;
}
// log(1+exp(x)) (reference: log1exp, src/util/individual_solver.py:52-57)
__device__ inline double softplus(double x) { return fmax(x, 0.0) + log1p(exp(-fabs(x))); }

// Root of g(x) = sigma*sigmoid(x) + rho*(x - m): the BCE prox of one element or one
// pooled block (mean sigma, mean m) - src/util/individual_solver.py:60-80 and
// src/util/pav.py:134-140.  g is increasing and g(m - sigma/rho) <= 0 <= g(m), so the
// root is bracketed; Newton steps are kept only while they stay inside the bracket and
// at least halve the previous step, otherwise the bracket is bisected (plain Newton
// cycles between the flat tails of the sigmoid when sigma/rho is large).
__device__ inline double prox_bce(double sigma, double rho, double m) {
    double lo = m - sigma / rho, hi = m, x = hi;
    double dxold = hi - lo, dx = dxold;
    double s, ds;
    sigmoid2(x, s, ds);
    double g = sigma * s + rho * (x - m);
    double h = sigma * ds + rho;
    for (int it = 0; it < 200; ++it) {
        if (g == 0.0) break;
        double xn;
        bool bis = (((x - hi) * h - g) * ((x - lo) * h - g) > 0.0) || (fabs(2.0 * g) > fabs(dxold * h));
        dxold = dx;
        if (bis) {
            dx = 0.5 * (hi - lo);
            xn = lo + dx;
        } else {
            dx = g / h;
            xn = x - dx;
        }
        if (xn == x) break;
        x = xn;
        sigmoid2(x, s, ds);
        g = sigma * s + rho * (x - m);
        h = sigma * ds + rho;
        if (g < 0.0) lo = x; else hi = x;
    }
    return x;
}

// Same root, started from a guess x0 (the previous ADMM iterate of the same sample: z moves
// little between iterations, so 1-3 Newton steps suffice).  The guess is clamped into the
// bracket; the safeguards are those of prox_bce, so the result is the same root to rounding.
__device__ inline double prox_bce_warm(double sigma, double rho, double m, double x0) {
    double lo = m - sigma / rho, hi = m;
    double x = fmin(fmax(x0, lo), hi);
    if (!(x == x)) x = hi;
    double dxold = hi - lo, dx = dxold;
    double s, ds;
    sigmoid2(x, s, ds);
    double g = sigma * s + rho * (x - m);
    double h = sigma * ds + rho;
    if (g < 0.0) lo = x; else hi = x;
    for (int it = 0; it < 200; ++it) {
        if (g == 0.0) break;
        double xn;
        bool bis = (((x - hi) * h - g) * ((x - lo) * h - g) > 0.0) || (fabs(2.0 * g) > fabs(dxold * h));
        dxold = dx;
        if (bis) {
            dx = 0.5 * (hi - lo);
            xn = lo + dx;
        } else {
            dx = g / h;
            xn = x - dx;
        }
        if (xn == x) break;
        x = xn;
        sigmoid2(x, s, ds);
        g = sigma * s + rho * (x - m);
        h = sigma * ds + rho;
        if (g < 0.0) lo = x; else hi = x;
    }
    return x;
}

// Exact hinge prox: the limit of the bisection of src/util/individual_solver.py:11-42.
__device__ inline double prox_hinge(double sigma, double rho, double m) {
    double a = m - sigma / rho;
    return (a >= -1.0) ? a : ((m <= -1.0) ? m : -1.0);
}

template <int LOSS>
__device__ inline double prox(double sigma, double rho, double m) {
    return LOSS == 0 ? prox_bce(sigma, rho, m) : prox_hinge(sigma, rho, m);
}

template <int LOSS>
__device__ inline double sample_loss(double v) {
    // objective.py:11-24 with D = -y*X: BCE-with-logits == softplus(v), hinge == max(1+v,0)
    return LOSS == 0 ? softplus(v) : fmax(1.0 + v, 0.0);
}

// ---- order-preserving float64 <-> uint64 key transform for the radix sort
__device__ inline unsigned long long flip_key(double x) {
    unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}
__device__ inline double unflip_key(unsigned long long k) {
    unsigned long long b = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

// ---- block-wide sum of K values (fixed order: deterministic)
template <int K, int THREADS>
__device__ inline void block_sum(double (&val)[K], double* smem /* K*THREADS/64 */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) val[k] += __shfl_xor(val[k], off, 64);
    }
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) smem[wave * K + k] = val[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double s = 0.0;
        for (int w = 0; w < THREADS / 64; ++w) s += smem[w * K + k];
        val[k] = s;
    }
}

}  // namespace rbl
