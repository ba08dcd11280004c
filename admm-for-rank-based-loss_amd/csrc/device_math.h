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
// This variant sits on the critical path of the single-sweep kernel (2 of 64 lanes active), so
// it trims the dependent chain: reciprocals by v_rcp_f64 + two Newton refinements (<= 2 ulp)
// instead of IEEE division, and it stops once an accepted Newton step is below 1e-10 relative
// (the step after that is O(step^2 * sigma/rho) ~ 1e-17: no further evaluation is needed).
__device__ inline double fast_rcp(double a) {
    double r = __builtin_amdgcn_rcp(a);
    r = __builtin_fma(__builtin_fma(-a, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-a, r, 1.0), r, r);
    return r;
}
__device__ inline void sigmoid2_fast(double x, double& s, double& ds) {
    double e = exp(-fabs(x));
    double inv = fast_rcp(1.0 + e);
    s = (x > 0.0) ? inv : e * inv;
    ds = e * inv * inv;
}
__device__ inline double prox_bce_warm(double sigma, double rho, double m, double x0) {
    double lo = m - sigma / rho, hi = m;
    double x = fmin(fmax(x0, lo), hi);
    if (!(x == x)) x = hi;
    double dxold = hi - lo, dx = dxold;
    double s, ds;
    sigmoid2_fast(x, s, ds);
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
            dx = g * fast_rcp(h);
            xn = x - dx;
            if (fabs(dx) <= 1e-10 * (fabs(x) + 1.0)) return xn;  // quadratic convergence: done
        }
        if (xn == x) break;
        x = xn;
        sigmoid2_fast(x, s, ds);
        g = sigma * s + rho * (x - m);
        h = sigma * ds + rho;
        if (g < 0.0) lo = x; else hi = x;
    }
    return x;
}

// Cold start from the expansion of the root around m.  With a = sigma/rho, s = sigmoid(m), s1 = s (1 - s),
// s2 = s1 (1 - 2 s) (first and second derivative of the sigmoid at m) the root of sigma*sigmoid(x) + rho*(x - m) is
// m + d, d = d1 - a s2 d1^2 / (2 (1 + a s1)), d1 = -a s / (1 + a s1), up to O(a^4).  a is ~1e-3 for single elements at
// the bench sizes (sigma ~ 1/n): one Newton evaluation confirms the estimate (the first-order estimate of round 2
// needed two).  sm = sigmoid(m) may be shared by callers that solve several sigma for one m.  The safeguards are
// prox_bce's: any estimate gives the same root to rounding.  (For a <= 1e-4 the estimate itself is exact to rounding -
// 1.04e-16 against an 80-bit Newton solve over m in [-30, 30] - but returning it unconfirmed bought nothing measurable:
// k_pav_bottom spends its time in the pooled blocks' solves, not in level 0.)
__device__ inline double prox_bce_est(double sigma, double rho, double m, double sm) {
    const double a = sigma / rho, ds = sm * (1.0 - sm), d2s = ds * (1.0 - 2.0 * sm);
    const double ih = 1.0 / (1.0 + a * ds);
    const double d1 = -a * sm * ih;
    return prox_bce_warm(sigma, rho, m, m + (d1 - 0.5 * a * d2s * d1 * d1 * ih));
}
// log(1 + exp(x)) for x = m + d with |d| <= 1e-3 from the expansion around m (spm = softplus(m), sm = sigmoid(m)):
// the neglected fourth-order term is <= 5e-15 of the value for every m (all derivatives decay like the value itself
// for m << 0); callers that hold exp(-|m|) already pay one log1p for spm and no exponential per evaluation.
__device__ inline double softplus_near(double x, double m, double spm, double sm) {
    const double d = x - m;
    if (fabs(d) > 1e-3) return softplus(x);
    const double ds = sm * (1.0 - sm), d2s = ds * (1.0 - 2.0 * sm);
    return spm + d * (sm + d * (0.5 * ds + d * (d2s * (1.0 / 6.0))));
}

// sum over the 64 lanes of a wave, returned to every lane: butterfly inside each 16-lane row
// with DPP (no LDS round trips), then the four row sums through scalar registers.
__device__ inline double dpp_mov(double x, const int ctrl_sel) {
    const long long b = __double_as_longlong(x);
    int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
    int rl, rh;
    switch (ctrl_sel) {
        case 0: rl = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xf, 0xf, false); rh = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xf, 0xf, false); break;   // quad_perm [1,0,3,2]
        case 1: rl = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xf, 0xf, false); rh = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xf, 0xf, false); break;   // quad_perm [2,3,0,1]
        case 2: rl = __builtin_amdgcn_update_dpp(0, lo, 0x141, 0xf, 0xf, false); rh = __builtin_amdgcn_update_dpp(0, hi, 0x141, 0xf, 0xf, false); break; // row_half_mirror
        default: rl = __builtin_amdgcn_update_dpp(0, lo, 0x140, 0xf, 0xf, false); rh = __builtin_amdgcn_update_dpp(0, hi, 0x140, 0xf, 0xf, false); break; // row_mirror
    }
    return __longlong_as_double(((long long)rh << 32) | (unsigned int)rl);
}
__device__ inline double readlane_d(double x, int lane) {
    const long long b = __double_as_longlong(x);
    int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
    int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ inline double wave_sum_all(double x) {
    x += dpp_mov(x, 0);
    x += dpp_mov(x, 1);
    x += dpp_mov(x, 2);
    x += dpp_mov(x, 3);   // every lane of a 16-lane row now holds that row's sum
    return (readlane_d(x, 0) + readlane_d(x, 16)) + (readlane_d(x, 32) + readlane_d(x, 48));
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

// element prox with the first-order estimate as the starting point (level 0 of the PAV tree, the EHRM branch test)
template <int LOSS>
__device__ inline double prox_est(double sigma, double rho, double m) {
    return LOSS == 0 ? prox_bce_est(sigma, rho, m, sigmoid1(m)) : prox_hinge(sigma, rho, m);
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
